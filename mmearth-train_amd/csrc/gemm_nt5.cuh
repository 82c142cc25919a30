// Deep-K dense NT GEMM: C[M][N] = A[M][K] W[N][K]^T (+ bias, + R, row mask), bf16 in / fp32 accumulate / bf16 out, K >= 1024.
// (decoder pwconv2 and pwconv1 data gradient N = 512, K = 2048; heads data gradient N = 512, K = 2816; models/convnextv2.py:46-52,
// models/fcmae.py:126-151.)
//
// Why another NT kernel: profiles/r04/blas_yardstick.txt. On these shapes the vendor library runs at 0.8-0.9 PF/s and the 128 x 128
// kernel of gemm_fast.cuh at 0.45-0.6. Both pull operands L2 -> LDS at the same ~9-10 TB/s ("The operand-fill
// ceiling, re-derived"); the vendor's kernel for the shape (MT128x256x64, wave tile 64 x 128, stream-K) moves 1.27x fewer bytes per flop
// and keeps every CU pulling. This kernel takes the tile and the slab ring of gemm_tn3.cuh; the stream-K schedule is NOT here yet
// (196 tiles on 256 CUs for N = 512: 0.57 PF/s, which is why MPMAE_OPT_NT5 is off by default):
//   * 128 x 256 tile, 4 waves (2 x 2), wave tile 64 rows x 128 columns: 12 fragment reads feed 32 MFMAs per 32-deep k-step
//     (2.67 MFMAs per read instead of 2) and a wave issues 64 MFMAs per barrier;
//   * both operand slabs go global -> LDS by DMA (global_load_lds_dwordx4) into a ring of THREE 64-deep stages (3 x 48 KB): two
//     stages in flight while one is consumed; the wait is a counted s_waitcnt vmcnt(12) (12 DMA instructions per thread and stage),
//     the only barrier per stage a bare s_barrier;
//   * rows are 128 bytes unpadded (a DMA image is lane-linear); the 16-byte chunk c of row r sits at slot c ^ swz(r),
//     swz(r) = (r & 3) | ((r >> 3) & 1) << 2 - the swizzle of gemm_fast.cuh's direct-to-LDS path, conflict-free for the A row
//     pattern (rows lr) and for the interleaved W rows below (tools/lds_bank_model.py);
//   * fragment reads are inline asm (hipcc parks an s_waitcnt vmcnt(0) in front of any LDS read that may alias an in-flight DMA),
//     all 24 of a stage requested up front, the MFMAs of k-step 0 start behind s_waitcnt lgkmcnt(12);
//   * the MFMA is issued transposed (D[n][m] = W A^T) with the W rows of a tile PAIR interleaved (row (lr >> 2) * 8 + t * 4 + (lr & 3)),
//     so a lane ends with 8 consecutive output columns of one row: the epilogue is one 16-byte store per row and pair straight from
//     the accumulators, bias / residual / row mask lane-local;
//   * XCD-aware tile order: workgroup b runs on XCD b % 8; an XCD walks row blocks x, x + 8, ... and for each ALL column tiles back to
//     back, so the column tiles of a row block share its A slab in one L2.
// Needs K % 64 == 0, N % 8 == 0, 16-byte aligned rows. Rows / columns beyond M / N are clamped on load and not stored.
#pragma once
#include "gemm_fast.cuh"

constexpr int NT5_BM = 128, NT5_BN = 256, NT5_BK = 64, NT5_ST = 3;
constexpr int NT5_AB = NT5_BM * NT5_BK * 2, NT5_WB = NT5_BN * NT5_BK * 2, NT5_STAGE_B = NT5_AB + NT5_WB, NT5_LDS = NT5_ST * NT5_STAGE_B;

typedef __attribute__((ext_vector_type(4))) unsigned nt5_u32x4_t;

__global__ __launch_bounds__(256) void gemm_nt5_kernel(const GemmP p, int mtiles, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nt5_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int mt = (q / ntiles) * 8 + xcd, nt = q % ntiles;
  if (mt >= mtiles) return;                                   // (whole workgroup: no barrier has been reached)
  const int m0 = mt * NT5_BM, n0 = nt * NT5_BN;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* W = reinterpret_cast<const bf16_t*>(p.B);
  auto swz = [](int row) { return (row & 3) | (((row >> 3) & 1) << 2); };

  // DMA of one 64-deep stage: 4 A + 8 W instructions per thread; LDS slot (row, c') holds global chunk c' ^ swz(row)
  auto dma = [&](int stage, int k0) {
    unsigned char* as = nt5_smem + stage * NT5_STAGE_B;
    unsigned char* ws = as + NT5_AB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sl = i * 256 + tid, row = sl >> 3, ch = (sl & 7) ^ swz(row);
      const bf16_t* src = A + (size_t)min(m0 + row, p.M - 1) * p.lda + k0 + ch * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(as + (i * 256 + wave * 64) * 16), 16, 0, 0);      // wave-uniform base; lane l lands at + 16 l
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int sl = i * 256 + tid, row = sl >> 3, ch = (sl & 7) ^ swz(row);
      const bf16_t* src = W + (size_t)min(n0 + row, p.N - 1) * p.ldb + k0 + ch * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ws + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
  };

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // per-lane fragment addresses (stage 0), one per k-step: tile / pair offsets are instruction immediates
  const int browl = (lr >> 2) * 8 + (lr & 3);                   // W rows of a pair: + t * 4
  const int swa = swz(lr), swb = swz(browl);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)nt5_smem;
  unsigned aa[2], wa[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    aa[ks] = lds0 + (wm * 64 + lr) * 128 + (((ks * 4 + lg) ^ swa) << 4);
    wa[ks] = lds0 + NT5_AB + (wn * 128 + browl) * 128 + (((ks * 4 + lg) ^ swb) << 4);
  }
#define NT5_RD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  auto fr = [](const nt5_u32x4_t& v) { return __builtin_bit_cast(bf16x8_t, v); };

  const int nk = p.K / NT5_BK;
  dma(0, 0);
  if (nk > 1) dma(1, NT5_BK);
  unsigned so = 0;                                              // byte offset of the stage being consumed
  for (int s = 0; s < nk; ++s) {
    if (s + 1 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // stage s has landed (stage s + 1 may still be in flight)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                          // ... for every wave; and everyone is done with stage s - 1
    asm volatile("" ::: "memory");
    if (s + 2 < nk) dma((s + 2) % NT5_ST, (s + 2) * NT5_BK);               // into the stage consumed in iteration s - 1
    nt5_u32x4_t af[2][4], wf[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned a_ = aa[ks] + so, w_ = wa[ks] + so;
      NT5_RD(af[ks][0], a_, 0); NT5_RD(af[ks][1], a_, 2048); NT5_RD(af[ks][2], a_, 4096); NT5_RD(af[ks][3], a_, 6144);
      NT5_RD(wf[ks][0], w_, 0);     NT5_RD(wf[ks][1], w_, 512);   NT5_RD(wf[ks][2], w_, 4096);  NT5_RD(wf[ks][3], w_, 4608);
      NT5_RD(wf[ks][4], w_, 8192);  NT5_RD(wf[ks][5], w_, 8704);  NT5_RD(wf[ks][6], w_, 12288); NT5_RD(wf[ks][7], w_, 12800);
    }
    // LDS returns in order: "<= 12 outstanding" of the 24 reads = the 12 of k-step 0 are back
    asm volatile("s_waitcnt lgkmcnt(12)"
                 : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]), "+v"(wf[0][3]),
                   "+v"(wf[0][4]), "+v"(wf[0][5]), "+v"(wf[0][6]), "+v"(wf[0][7]));
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr(wf[0][j]), fr(af[0][i]), acc[i][j], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]), "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[1][2]), "+v"(wf[1][3]),
                   "+v"(wf[1][4]), "+v"(wf[1][5]), "+v"(wf[1][6]), "+v"(wf[1][7]));
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr(wf[1][j]), fr(af[1][i]), acc[i][j], 0, 0, 0);
    so = (so == (NT5_ST - 1) * NT5_STAGE_B) ? 0u : so + NT5_STAGE_B;
  }
#undef NT5_RD

  // epilogue: lane = row m0 + wm*64 + i*16 + lr, columns n0 + wn*128 + jp*32 + lg*8 + (t*4 + r). Every optional operand is REQUESTED
  // first (clamped addresses, pointer selects, opaque masks) and consumed afterwards.
  bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C);
  const bf16_t* Rg = reinterpret_cast<const bf16_t*>(p.R);
  const unsigned act_m = opaque_mask(p.act != nullptr) & 0xffu, r_m = opaque_mask(Rg != nullptr), b_m = opaque_mask(p.bias != nullptr);
  uint8_t lv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rowc = min(m0 + wm * 64 + i * 16 + lr, p.M - 1);
    lv[i] = *(p.act ? p.act + rowc : reinterpret_cast<const uint8_t*>(p.B));
  }
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) {
    const int col = n0 + wn * 128 + jp * 32 + lg * 8;
    const int colc = min(col, p.N - 8);
    const float* bp = p.bias ? p.bias + colc : reinterpret_cast<const float*>(p.B);
    const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
    uint4 rraw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rowc = min(m0 + wm * 64 + i * 16 + lr, p.M - 1);
      rraw[i] = *reinterpret_cast<const uint4*>(Rg ? Rg + (size_t)rowc * p.ldr + colc : reinterpret_cast<const bf16_t*>(p.B));
    }
    if (col >= p.N) continue;                                   // N % 8 == 0 guaranteed by the dispatcher
    const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + wm * 64 + i * 16 + lr;
      if (row >= p.M) continue;
      const bool live = ((lv[i] & act_m) | (~act_m & 1u)) != 0;
      const uint4 rm = make_uint4(rraw[i].x & r_m, rraw[i].y & r_m, rraw[i].z & r_m, rraw[i].w & r_m);
      float v[8], rr[8];
      rr[0] = __uint_as_float(rm.x << 16); rr[1] = __uint_as_float(rm.x & 0xffff0000u);
      rr[2] = __uint_as_float(rm.y << 16); rr[3] = __uint_as_float(rm.y & 0xffff0000u);
      rr[4] = __uint_as_float(rm.z << 16); rr[5] = __uint_as_float(rm.z & 0xffff0000u);
      rr[6] = __uint_as_float(rm.w << 16); rr[7] = __uint_as_float(rm.w & 0xffff0000u);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = live ? acc[i][2 * jp + (e >> 2)][e & 3] + __uint_as_float(__float_as_uint(bv[e]) & b_m) + rr[e] : 0.f;
      st8<bf16_t>(Cg + (size_t)row * p.ldc + col, v);
    }
  }
}
