// Weight-gradient GEMM for the MFMA-shaped layers (decoder block, prediction heads): dW[n][k] = sum_m P[m][n] Q[m][k] with both
// extents >= 512. Same contraction and transpose-read fragments as gemm_tn2.cuh (ds_read_b64_tr_b16 turns the m-major slab rows
// into k-vectors of mfma_f32_16x16x32_bf16), rebuilt around the two things PMC showed gemm_tn2 to be bound by at these shapes
// (profiles/r02/pmc_gemm_tn2_head_shape.txt: MFMA busy 12 %, 743 MB of L2 requests for 83 MB of operands, waves waiting 33 %):
//   * a 128 x 256 output tile (X = 128 columns of the narrow operand shared by the 4 waves, Y = 64 columns of the wide operand per
//     wave): every operand element is re-read 1.65x less often than with 64 x 256, and a wave issues 64 MFMAs per barrier, not 16;
//   * the slabs go global -> LDS by DMA (global_load_lds_dwordx4, no staging registers, no ds_write pass) into a ring of THREE
//     64-row stages: two stages are in flight while one is consumed, the wait is a counted vmcnt(12) (12 DMA instructions per thread
//     and stage) and the only barrier per stage is a bare s_barrier - __syncthreads() would carry a vmcnt(0) and drain the ring.
// A DMA image is lane-linear, so rows cannot be padded; the transpose reads are kept conflict-free by an XOR swizzle of the 16-byte
// chunk index with ((row & 7) << 1), applied to the per-lane SOURCE address of the DMA and again on the read: the 16 rows a wave
// touches per read then spread their 32-byte pieces over all 8 bank groups of a 256-byte LDS line (2 passes for 512 bytes: optimal).
// Needs M % 64 == 0, WX % 128 == 0, WY % 256 == 0 (the launcher falls back to gemm_tn2 otherwise). Split over rows -> slabs
// [split][Nn * Kk + Nn] folded by the same second-stage launch as gemm_tn2; bias gradient = one more MFMA against a ones fragment.
#pragma once
#include "gemm_tn2.cuh"
#include "gemm_fast.cuh"
#include <type_traits>

constexpr int TN3_BX = 128, TN3_BY = 256, TN3_SL = 64, TN3_ST = 3;
constexpr int TN3_XB = TN3_SL * TN3_BX * 2, TN3_YB = TN3_SL * TN3_BY * 2, TN3_STAGE_B = TN3_XB + TN3_YB;      // bytes
constexpr int TN3_LDS = TN3_ST * TN3_STAGE_B;

template <bool SWAP>
__global__ __launch_bounds__(256) void gemm_tn3_kernel(const WgradP w, int splits) {
  constexpr int NT = TN3_BX / 16, KT = TN3_BY / 64;            // 8 x-tiles shared by the waves, 4 y-tiles per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char tn3_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const bf16_t* X = reinterpret_cast<const bf16_t*>(SWAP ? w.Q : w.P);
  const bf16_t* Y = reinterpret_cast<const bf16_t*>(SWAP ? w.P : w.Q);
  const int ldx = SWAP ? w.ldq : w.ldp, ldy = SWAP ? w.ldp : w.ldq;
  // grid = (splits, x tiles, y tiles): the split index is the FASTEST-varying block coordinate, so with 8 splits every workgroup of one
  // row range lands on the same XCD (block b runs on XCD b % 8) and the 4 x 8..11 tiles of that range share their slabs in ONE L2
  const int x0 = blockIdx.y * TN3_BX, y0 = blockIdx.z * TN3_BY, split = blockIdx.x;
  const int mbeg = split * w.rows_per_split, mend = min(w.M, mbeg + w.rows_per_split);
  const int nsl = (mend - mbeg) / TN3_SL;

  f32x4_t acc[NT][KT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bool do_db = w.db != nullptr && (SWAP ? x0 == 0 : (y0 == 0 && wave == 0));
  constexpr int NB = SWAP ? KT : NT;
  f32x4_t accb[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) accb[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  // DMA of one 64-row stage: LDS position (row, chunk c') holds global chunk c' ^ ((row & 7) << 1) of that row
  auto dma_piece = [&](int stage, int mb, int p) {              // p = 0..3: X (64 rows x 16 chunks), 4..11: Y (64 rows x 32 chunks)
    unsigned char* xs = tn3_smem + stage * TN3_STAGE_B;
    unsigned char* ys = xs + TN3_XB;
    if (p < 4) {
      const int sl = p * 256 + tid, row = sl >> 4, ch = (sl & 15) ^ ((row & 7) << 1);
      const bf16_t* src = X + (size_t)(mb + row) * ldx + x0 + ch * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xs + (p * 256 + wave * 64) * 16), 16, 0, 0);      // wave-uniform base; lane l lands at + 16 l
    } else {
      const int i = p - 4, sl = i * 256 + tid, row = sl >> 5, ch = (sl & 31) ^ ((row & 7) << 1);
      const bf16_t* src = Y + (size_t)(mb + row) * ldy + y0 + ch * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ys + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
  };
  auto dma = [&](int stage, int mb) {
#pragma unroll
    for (int p = 0; p < 12; ++p) dma_piece(stage, mb, p);
  };

  // Transpose reads as inline assembly: hipcc knows that global_load_lds writes LDS and puts an s_waitcnt vmcnt(0) in front of every
  // LDS read that might alias it - which drains the DMA ring at the top of each stage. The reads below are invisible to that pass;
  // their own completion is a hand-placed s_waitcnt lgkmcnt(0) that takes every fragment register as an in/out operand, so that no
  // MFMA can be scheduled above it.
  // Address of this lane: row (within a 16-row half of a k-step) rl = lg*4 + (lr >> 2), columns 4 (lr & 3) .. + 3 of 16-column tile t:
  //   byte = row * ROWB + ((t ^ (rl & 7)) * 32) + (lr & 3) * 8; the k-step / half offsets are instruction immediates.
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  const int rl = lg * 4 + (lr >> 2), s3 = rl & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)tn3_smem;
  const unsigned xlane = lds0 + rl * (TN3_BX * 2) + (lr & 3) * 8, ylane = lds0 + TN3_XB + rl * (TN3_BY * 2) + (lr & 3) * 8;
  unsigned xa[NT], ya[KT];
#pragma unroll
  for (int i = 0; i < NT; ++i) xa[i] = xlane + ((i ^ s3) << 5);
#pragma unroll
  for (int j = 0; j < KT; ++j) ya[j] = ylane + (((wave * KT + j) ^ s3) << 5);
#define TN3_TR(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  auto mk = [](const u32x2_t& lo, const u32x2_t& hi) { return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y)); };

  if (nsl > 0) dma(0, mbeg);
  if (nsl > 1) dma(1, mbeg + TN3_SL);
  auto iter = [&](int s, auto dma_) {
    constexpr bool DMA = decltype(dma_)::value;                            // (a compile-time flag: a branch around the DMA requests keeps the
    if (s + 1 < nsl) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      //  scheduler from spreading them between the MFMAs)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // stage s has landed (stage s + 1 may still be in flight)
    __builtin_amdgcn_s_barrier();                                          // ... for every wave; and everyone is done with stage s - 1
    asm volatile("" ::: "memory");
    const unsigned so = (unsigned)(s % TN3_ST) * TN3_STAGE_B;
    u32x2_t xl[2][NT], xh[2][NT], yl[2][KT], yh[2][KT];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        if (ks == 0) { TN3_TR(xl[0][i], xa[i] + so, 0); TN3_TR(xh[0][i], xa[i] + so, 16 * TN3_BX * 2); }
        else { TN3_TR(xl[1][i], xa[i] + so, 32 * TN3_BX * 2); TN3_TR(xh[1][i], xa[i] + so, 48 * TN3_BX * 2); }
      }
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        if (ks == 0) { TN3_TR(yl[0][j], ya[j] + so, 0); TN3_TR(yh[0][j], ya[j] + so, 16 * TN3_BY * 2); }
        else { TN3_TR(yl[1][j], ya[j] + so, 32 * TN3_BY * 2); TN3_TR(yh[1][j], ya[j] + so, 48 * TN3_BY * 2); }
      }
    }
    // LDS returns in order and lgkmcnt is a 4-bit counter: "<= 15 outstanding" of the 48 reads covers the 24 of k-step 0
    asm volatile("s_waitcnt lgkmcnt(15)"
                 : "+v"(xl[0][0]), "+v"(xl[0][1]), "+v"(xl[0][2]), "+v"(xl[0][3]), "+v"(xl[0][4]), "+v"(xl[0][5]), "+v"(xl[0][6]), "+v"(xl[0][7]),
                   "+v"(xh[0][0]), "+v"(xh[0][1]), "+v"(xh[0][2]), "+v"(xh[0][3]), "+v"(xh[0][4]), "+v"(xh[0][5]), "+v"(xh[0][6]), "+v"(xh[0][7]),
                   "+v"(yl[0][0]), "+v"(yl[0][1]), "+v"(yl[0][2]), "+v"(yl[0][3]), "+v"(yh[0][0]), "+v"(yh[0][1]), "+v"(yh[0][2]), "+v"(yh[0][3]));
    // the next-but-one stage's 12 DMA requests (into the stage consumed in iteration s - 1): issued here and spread between the
    // MFMAs of k-step 0 by the scheduling groups below - back to back in front of the reads they cost ~60 issue cycles each with the
    // matrix pipe idle (one wave per SIMD)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 1)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xl[1][0]), "+v"(xl[1][1]), "+v"(xl[1][2]), "+v"(xl[1][3]), "+v"(xl[1][4]), "+v"(xl[1][5]), "+v"(xl[1][6]), "+v"(xl[1][7]),
                       "+v"(xh[1][0]), "+v"(xh[1][1]), "+v"(xh[1][2]), "+v"(xh[1][3]), "+v"(xh[1][4]), "+v"(xh[1][5]), "+v"(xh[1][6]), "+v"(xh[1][7]),
                       "+v"(yl[1][0]), "+v"(yl[1][1]), "+v"(yl[1][2]), "+v"(yl[1][3]), "+v"(yh[1][0]), "+v"(yh[1][1]), "+v"(yh[1][2]), "+v"(yh[1][3]));
      bf16x8_t xf[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) xf[i] = mk(xl[ks][i], xh[ks][i]);
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        const bf16x8_t yf = mk(yl[ks][j], yh[ks][j]);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, xf[i], acc[i][j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf, acc[i][j], 0, 0, 0);
          // one DMA request behind every fifth MFMA of the stage (12 requests over 64 MFMAs), pinned in place
          if (DMA && ((ks * KT + j) * NT + i) % 5 == 4 && ((ks * KT + j) * NT + i) / 5 < 12) {
            __builtin_amdgcn_sched_barrier(0);
            dma_piece((s + 2) % TN3_ST, mbeg + (s + 2) * TN3_SL, ((ks * KT + j) * NT + i) / 5);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (SWAP && do_db) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, ones, accb[j], 0, 0, 0);
      }
      if (!SWAP && do_db) {
#pragma unroll
        for (int i = 0; i < NT; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], ones, accb[i], 0, 0, 0);
      }
    }
  };
  int s = 0;
  for (; s + 2 < nsl; ++s) iter(s, std::true_type{});
  for (; s < nsl; ++s) iter(s, std::false_type{});
#undef TN3_TR

  // D layout: col = lr, row = lg*4 + r. !SWAP: row = x (n), col = y (k). SWAP: row = y (n), col = x (k). Slab layout as gemm_tn2.
  float* slab = w.ws + (size_t)split * ((size_t)w.Nn * w.Kk + w.Nn);
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int xi = x0 + i * 16, yj = y0 + (wave * KT + j) * 16;
        const int n = SWAP ? yj + lg * 4 + r : xi + lg * 4 + r;
        const int k = SWAP ? xi + lr : yj + lr;
        slab[(size_t)n * w.Kk + k] = acc[i][j][r];
      }
  if (do_db && lr == 0) {
    float* dslab = slab + (size_t)w.Nn * w.Kk;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) dslab[(SWAP ? y0 + (wave * KT + i) * 16 : x0 + i * 16) + lg * 4 + r] = accb[i][r];
  }
}
