// Dense NT GEMM for the MFMA-shaped layers (decoder block pwconv1 / pwconv2 forward and data gradient, pixel heads forward and
// data gradient; models/convnextv2.py:46-52, models/fcmae.py:126-151,249-265): C[M][N] = A[M][K] B[N][K]^T (+ bias, + R, row mask).
//
// The 128 x 128-tile kernel of gemm_fast.cuh runs these shapes at 500-700 TFLOP/s: with K = 512 every 128 x 128 tile pulls 262 KB
// through L2 for 16.8 MFLOP (64 flop / byte; the whole GEMM 411 MB), and a wave issues 32 MFMAs per barrier. This kernel:
//   * 256 x BN tile (BN = 256 for wide N, 128 for N = 512), 512 threads = 8 waves, two per SIMD; wave tile 128 x 64 (BN = 256:
//     waves 2 x 4) or 64 x 64 (BN = 128: waves 4 x 2) of mfma_f32_32x32x16_bf16 - half the L2 -> LDS bytes per flop of the
//     128 x 128 tile, 32 (16) MFMAs of 32 cycles per wave and 64-deep K step;
//   * both operand slabs go global -> LDS by DMA (global_load_lds_dwordx4) into a double buffer of 64-deep K steps (2 x 64 KB /
//     2 x 48 KB); rows are 128 bytes unpadded, the 16-byte slot of chunk kc of row r is kc ^ ((r >> 1) & 7): conflict-free under
//     the ds_read_b128 lane groups of the microarchitecture guide for the 32-row fragment pattern (rows l % 32, chunk 2 ks + l / 32)
//     AND for the permuted B rows below;
//   * the fragment reads are inline asm (hipcc parks an s_waitcnt vmcnt(0) in front of any LDS read that may alias an in-flight
//     DMA), double-buffered in registers: the reads of k-slice ks + 1 are in flight under the MFMAs of ks;
//   * the MFMA is issued TRANSPOSED (D'[n][m] = B A^T) with the B rows of a 32-row fragment permuted by
//     s(8 a + 4 b + c) = 16 b + 4 a + c, so a lane ends with 16 CONSECUTIVE output columns of one row: the epilogue is two 16-byte
//     stores per 32 x 32 tile straight from the accumulators (bias / residual / row mask lane-local, no LDS pass);
//   * XCD-aware tile order: workgroup b runs on XCD b % 8; XCD x walks the row blocks x, x + 8, ... and for each ALL column tiles
//     back to back, so the column tiles of a row block share its A slab in one L2 and B stays resident there.
// Needs K % 64 == 0, N % 8 == 0, 16-byte aligned rows. Rows / columns beyond M / N are clamped on load and not stored.
#pragma once
#include "gemm_fast.cuh"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int NT4_BM = 256, NT4_BK = 64;
template <int BN> struct Nt4Cfg {
  static constexpr int A_B = NT4_BM * NT4_BK * 2, B_B = BN * NT4_BK * 2, STAGE_B = A_B + B_B, LDS = 2 * STAGE_B;
  static constexpr int WM = BN == 256 ? 2 : 4, WN = 8 / WM;          // wave grid
  static constexpr int MI = NT4_BM / (32 * WM), NJ = BN / (32 * WN);  // 32 x 32 tiles per wave
  static constexpr int NI = (NT4_BM + BN) / 8, PER = NI / 8;          // DMA wave-instructions (8 rows each) per stage / per wave
};

template <int MI, int NJ>
__device__ __forceinline__ void nt4_epilogue(const GemmP& p, const f32x16_t (&acc)[MI][NJ], int m0, int n0, int wm, int wn, int l32, int hb) {
  // epilogue: lane = row m0 + wm * 32 MI + mi * 32 + l % 32, columns n0 + wn * 64 + nj * 32 + 16 (l / 32) + r, r = 0..15.
  // Every optional operand (row mask, bias, residual) is REQUESTED first - clamped addresses, pointer selects, opaque masks - and
  // consumed afterwards: a load under a per-lane branch is a serial round trip each ("a per-lane conditional load").
  bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C);
  const bf16_t* Rg = reinterpret_cast<const bf16_t*>(p.R);
  const unsigned act_m = opaque_mask(p.act != nullptr) & 0xffu, r_m = opaque_mask(Rg != nullptr), b_m = opaque_mask(p.bias != nullptr);
  int rowc[MI];
  uint8_t lv[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    rowc[i] = min(m0 + wm * 32 * MI + i * 32 + l32, p.M - 1);
    lv[i] = *(p.act ? p.act + rowc[i] : reinterpret_cast<const uint8_t*>(p.B));
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = n0 + wn * 64 + j * 32 + 16 * hb + 8 * h, colc = min(col, p.N - 8);
      const float* bp = p.bias ? p.bias + colc : reinterpret_cast<const float*>(p.B);
      const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
      uint4 rraw[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        rraw[i] = *reinterpret_cast<const uint4*>(Rg ? Rg + (size_t)rowc[i] * p.ldr + colc : reinterpret_cast<const bf16_t*>(p.B));
      const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = m0 + wm * 32 * MI + i * 32 + l32;
        const bool live = ((lv[i] & act_m) | (~act_m & 1u)) != 0;
        const uint4 rm = make_uint4(rraw[i].x & r_m, rraw[i].y & r_m, rraw[i].z & r_m, rraw[i].w & r_m);
        float v[8], rr[8];
        rr[0] = __uint_as_float(rm.x << 16); rr[1] = __uint_as_float(rm.x & 0xffff0000u);
        rr[2] = __uint_as_float(rm.y << 16); rr[3] = __uint_as_float(rm.y & 0xffff0000u);
        rr[4] = __uint_as_float(rm.z << 16); rr[5] = __uint_as_float(rm.z & 0xffff0000u);
        rr[6] = __uint_as_float(rm.w << 16); rr[7] = __uint_as_float(rm.w & 0xffff0000u);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = live ? acc[i][j][8 * h + e] + __uint_as_float(__float_as_uint(bv[e]) & b_m) + rr[e] : 0.f;
        if (row < p.M && col < p.N) {
#ifdef NT4_SC1_STORES
          // write-through store that does not keep the line in this XCD's L2 (MI355X_MICROARCH.md, "stores of each flavour"): the output
          // streams out once and is next read by another kernel; kept lines would evict the operand slabs the tiles of this XCD share
          typedef __attribute__((ext_vector_type(4))) unsigned u32x4s_t;
          const u32x4s_t pk = {f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7])};
          bf16_t* dst = Cg + (size_t)row * p.ldc + col;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(pk) : "memory");
#else
          st8<bf16_t>(Cg + (size_t)row * p.ldc + col, v);
#endif
        }
      }
    }
  }
}

template <int BN>
__global__ __launch_bounds__(512) void gemm_nt4_kernel(const GemmP p, int tm, int tn) {
  using Cf = Nt4Cfg<BN>;
  constexpr int MI = Cf::MI, NJ = Cf::NJ, PER = Cf::PER;
  static_assert(NJ == 2, "wave tile is 64 columns wide");
  extern __shared__ __attribute__((aligned(16))) unsigned char nt4_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % Cf::WM, wn = wave / Cf::WM;
  // XCD-aware order (see the header): b % 8 = XCD, per XCD (row block, column tile) with the column tile fastest
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
  const int mt = (idx / tn) * 8 + xcd, nt = idx % tn;
  if (mt >= tm) return;
  const int m0 = mt * NT4_BM, n0 = nt * BN;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

  // DMA plan: wave-instruction I = wave + 8 k covers rows 8 (I mod tile) .. + 7 of A (I < 32) or B; lane l -> row + l / 8, slot l % 8,
  // which holds global chunk (l % 8) ^ ((row >> 1) & 7)
  const bf16_t* dsrc[PER];
  unsigned ddst[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int I = wave + 8 * k;
    const bool isa = I < NT4_BM / 8;
    const int row = (isa ? I : I - NT4_BM / 8) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((row >> 1) & 7);
    dsrc[k] = isa ? A + (size_t)min(m0 + row, p.M - 1) * p.lda + ch * 8 : B + (size_t)min(n0 + row, p.N - 1) * p.ldb + ch * 8;
    ddst[k] = (isa ? 0u : (unsigned)Cf::A_B) + (unsigned)((isa ? I : I - NT4_BM / 8) * 1024);
  }
  auto dma = [&](int buf) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      __builtin_amdgcn_global_load_lds((gptr_t)dsrc[k], (lptr_t)(nt4_smem + buf * Cf::STAGE_B + ddst[k]), 16, 0, 0);
      dsrc[k] += NT4_BK;
    }
  };

  // fragment addresses: A rows wm * 32 MI + mi * 32 + l % 32; B rows wn * 64 + nj * 32 + s(l % 32); chunk 2 ks + l / 32
  const int l32 = lane & 31, hb = lane >> 5;
  const int sb = ((l32 >> 2) & 1) * 16 + (l32 >> 3) * 4 + (l32 & 3);        // s(8 a + 4 b + c) = 16 b + 4 a + c
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)nt4_smem;
  unsigned aaddr[2][4], baddr[2][4];
#pragma unroll
  for (int buf = 0; buf < 2; ++buf)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      aaddr[buf][ks] = lds0 + buf * Cf::STAGE_B + (wm * 32 * MI + l32) * 128 + (((2 * ks + hb) ^ ((l32 >> 1) & 7)) << 4);
      baddr[buf][ks] = lds0 + buf * Cf::STAGE_B + Cf::A_B + (wn * 64 + sb) * 128 + (((2 * ks + hb) ^ ((sb >> 1) & 7)) << 4);
    }

  f32x16_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#define NT4_RD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  auto fr = [](const u32x4_t& v) { return __builtin_bit_cast(bf16x8_t, v); };
  const int nk = p.K / NT4_BK;
  dma(0);
  auto stage = [&](auto bufc, bool more) {
    constexpr int buf = decltype(bufc)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this stage has landed (nothing younger is in flight)
    __builtin_amdgcn_s_barrier();                              // ... for every wave; and every wave is done with the other buffer
    asm volatile("" ::: "memory");
    if (more) dma(buf ^ 1);
    u32x4_t af[2][MI], bf_[2][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) NT4_RD(af[0][i], aaddr[buf][0], i * 4096);
#pragma unroll
    for (int j = 0; j < NJ; ++j) NT4_RD(bf_[0][j], baddr[buf][0], j * 4096);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < MI; ++i) NT4_RD(af[nxt][i], aaddr[buf][ks + 1], i * 4096);
#pragma unroll
        for (int j = 0; j < NJ; ++j) NT4_RD(bf_[nxt][j], baddr[buf][ks + 1], j * 4096);
      }
      // LDS returns in order: at most the MI + NJ reads of slice ks + 1 outstanding = slice ks is there
      if constexpr (MI == 4) {
        if (ks < 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af[cur][0]), "+v"(af[cur][1]), "+v"(af[cur][2]), "+v"(af[cur][3]), "+v"(bf_[cur][0]), "+v"(bf_[cur][1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[cur][0]), "+v"(af[cur][1]), "+v"(af[cur][2]), "+v"(af[cur][3]), "+v"(bf_[cur][0]), "+v"(bf_[cur][1]));
      } else {
        if (ks < 3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[cur][0]), "+v"(af[cur][1]), "+v"(bf_[cur][0]), "+v"(bf_[cur][1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[cur][0]), "+v"(af[cur][1]), "+v"(bf_[cur][0]), "+v"(bf_[cur][1]));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr(bf_[cur][j]), fr(af[cur][i]), acc[i][j], 0, 0, 0);
    }
  };
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    stage(std::integral_constant<int, 0>{}, true);
    stage(std::integral_constant<int, 1>{}, kt + 2 < nk);
  }
  if (kt < nk) stage(std::integral_constant<int, 0>{}, false);
#undef NT4_RD

  nt4_epilogue<MI, NJ>(p, acc, m0, n0, wm, wn, l32, hb);
}

// A third variant was measured and removed (round 4, profiles/r04/gemm_and_wgrad_group_probes.txt, "NT4=3"): 256 x 128 tile, 4 waves, two
// workgroups per CU, 32-deep K steps through a ring of three 24 KB stages with counted vmcnt (so that one workgroup's prologue / epilogue
// runs under the other's main loop): 54.1 us at N = 2048, K = 512 (this kernel 53.5, the 128 x 128 kernels 58.0), 45.7 vs 40.8 vs 41.1 at
// N = 512, K = 2048. Three tilings landing on the same time says the bound is none of tile shape, pipeline depth or occupancy: every variant
// fills LDS at ~10-12 bytes per cycle and CU (205-411 MB of operand slabs per GEMM in ~40 us of main loop), the rate the microarchitecture
// guide quotes for global_load(_lds)_dwordx4 streams, i.e. 64-128 flop per byte x ~12 B/cycle = 20-38 % of the MFMA peak.
