// bf16 MFMA GEMMs for the compute-shaped layers (decoder block, stages 2-3, heads, proj).
//
//   gemm_nt_bf16 :  C[M,N] = A[M,K] * B[N,K]^T + bias  (+ R)      128 x BN x 64 tiles, 4 waves
//   gemm_tn_bf16 :  dW[n,k] += sum_m P[m,n] * Q[m,k]  (+ db)       128 x 128 output tiles, split-M
//
// Design (MI355X_MICROARCH / cdna_hip_programming guides): 64-wide waves each own a 64x64 (or
// 64x32) register tile of 16x16x32 bf16 MFMAs; operands are staged global -> registers -> LDS with
// 16-byte accesses, LDS rows padded to 144 B so every ds_read_b128 fragment read is
// conflict-free; the next K-slab's global loads are in flight while the current slab's MFMAs
// run (one barrier per slab, two LDS buffers); the epilogue goes back through LDS so that global
// stores are 16 B per lane along rows. The TN (weight-gradient) form transposes while loading:
// lanes read 4-byte pairs along the contiguous n/k axis for 8 consecutive m and repack them into
// m-contiguous 16-byte LDS rows.
#pragma once
#include "gemm.cuh"

// (a 160-byte row stride = 32 mod 64, which the ds_read_b128 lane-group table suggests, measured SLOWER: 5.39 vs 5.32 ms/step)
#ifndef MPMAE_FPAD
#define MPMAE_FPAD 8
#endif
constexpr int FBM = 128, FBK = 64, FPAD = MPMAE_FPAD, FLD = FBK + FPAD;   // LDS row = 72 bf16 = 144 B

__device__ __forceinline__ uint4 ldg16_guard(const bf16_t* base, int row, int nrows, int ld, int k, int K) {
  // 8 bf16 at (row, k..k+7); zero beyond the matrix. K % 8 == 0 is required by the dispatcher.
  if (row < nrows && k < K) return *reinterpret_cast<const uint4*>(base + (size_t)row * ld + k);
  return make_uint4(0u, 0u, 0u, 0u);
}

// EPI: EPI_STORE (bias, optional residual R, row mask) | EPI_GELU_SUMSQ (store h, per-block column
// partials of gelu(h)^2 -> ws[blockIdx.x][N]) | EPI_DZ_STATS (store dz, partials of dz and
// dz*gelu(R) -> ws[blockIdx.x][N], ws[gridDim.x + blockIdx.x][N]); statistics: single group only.
// GLDS: the operand slabs go global -> LDS directly (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass). The LDS
// image of such a load is lane-linear (wave-uniform base + 16 B x lane), so rows cannot be padded; bank conflicts of the
// fragment reads are removed by an XOR swizzle of the 16-byte chunk index instead, applied to the per-lane SOURCE address
// and again on the read: chunk ^ (row & 7) for 128-byte rows, chunk ^ ((row >> 1) & 3) for 64-byte rows (both conflict-free
// under the ds_read_b128 lane grouping of the microarchitecture guide). Needs K % BK == 0 (no zero fill); rows / columns
// beyond M / N are clamped (their products are never stored).
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Epilogue of the direct-to-LDS kernels: the MFMAs are issued transposed over interleaved column-tile pairs, so a lane holds 8 CONSECUTIVE output
// columns of one row and stores them as one 16-byte vector straight from the accumulators (bias, optional residual, row mask).
template <int BN>
__device__ __forceinline__ void nt_glds_epilogue(const GemmP& p, f32x4_t (&acc)[4][BN / 32], int m0, int n0, int wm, int wn, int lr, int lg) {
  constexpr int NJ = BN / 32;
  // epilogue: lane = row m0 + wm*64 + i*16 + lr, columns n0 + wn*(BN/2) + jp*32 + lg*8 + (t*4 + r)
  bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C);
  const bf16_t* Rg = reinterpret_cast<const bf16_t*>(p.R);
  if constexpr (NJ == 2) {
    // 64-wide tiles (stage 3, downsample, stem: activity bytes AND residuals): every optional operand of the tile's rows is
    // requested first - clamped addresses, pointer selects, opaque masks - and consumed afterwards; in the store loop below
    // (`live = !act || act[row]; if (R) ld8(R...)`) they are two dependent round trips per 16-row group. (At 128-wide tiles the
    // extra 32 + 16 registers cost more occupancy than the round trips: head GEMM 69 -> 80 us.)
    const unsigned act_m = opaque_mask(p.act != nullptr) & 0xffu, r_m = opaque_mask(Rg != nullptr), b_m = opaque_mask(p.bias != nullptr);
    const int colc = min(n0 + wn * (BN / 2) + lg * 8, p.N - 8);
    const float* bp = p.bias ? p.bias + colc : reinterpret_cast<const float*>(p.B);
    const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
    uint8_t lv[4];
    uint4 rraw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rowc = min(m0 + wm * 64 + i * 16 + lr, p.M - 1);
      lv[i] = *(p.act ? p.act + rowc : reinterpret_cast<const uint8_t*>(p.B));
      rraw[i] = *reinterpret_cast<const uint4*>(Rg ? Rg + (size_t)rowc * p.ldr + colc : reinterpret_cast<const bf16_t*>(p.B));
    }
    const int col = n0 + wn * (BN / 2) + lg * 8;
    if (col < p.N) {
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
        for (int e = 0; e < 8; ++e) v[e] = live ? acc[i][e >> 2][e & 3] + __uint_as_float(__float_as_uint(bv[e]) & b_m) + rr[e] : 0.f;
        st8<bf16_t>(Cg + (size_t)row * p.ldc + col, v);
      }
    }
    return;
  }
#pragma unroll
  for (int jp = 0; jp < NJ / 2; ++jp) {
    const int col = n0 + wn * (BN / 2) + jp * 32 + lg * 8;
    if (col >= p.N) continue;                          // N % 8 == 0 guaranteed by the dispatcher
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col), b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + wm * 64 + i * 16 + lr;
      if (row >= p.M) continue;
      const bool live = !p.act || p.act[row];
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * jp + (e >> 2)][e & 3] + bv[e];
      if (Rg) {
        float rr[8];
        ld8<bf16_t>(Rg + (size_t)row * p.ldr + col, rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rr[e];
      }
      if (!live) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
      st8<bf16_t>(Cg + (size_t)row * p.ldc + col, v);
    }
  }
}

// Ring form of the direct-to-LDS NT kernel (round 6): the same 128 x BN tile, fragment layout, swizzle and epilogue, with the K slabs in a ring of
// NST stages instead of two buffers. gemm_nt_bf16_kernel<.., GLDS> requests slab kt + 1 at the top of iteration kt and ends the iteration with
// __syncthreads(), whose vmcnt(0) waits for that request: a workgroup has ONE slab in flight, and an iteration of 16-32 MFMAs per wave is shorter
// than a round trip under load - it lives off the second workgroup of the CU. Here NST - 1 slabs are in flight per workgroup: the wait at the top of
// iteration kt is a COUNTED vmcnt (PER requests per thread and slab, NST - 2 slabs may stay outstanding), the only barrier a bare s_barrier that
// publishes slab kt and frees the stage slab kt + NST - 1 lands in. The fragment reads are inline assembly (hipcc puts an s_waitcnt vmcnt(0) in
// front of any LDS read it believes a global_load_lds may alias - gemm_tn3.cuh) with their own lgkmcnt waits.
template <int BN, int BK, int NST>
__global__ __launch_bounds__(256) void gemm_nt_ring_kernel(const GemmP p) {
  constexpr int CPR = BK / 8, ACH = FBM * CPR / 256, BCH = BN * CPR / 256, PER = ACH + BCH, NJ = BN / 32, KS = BK / 32;
  constexpr int A_B = FBM * BK * 2, B_B = BN * BK * 2, STAGE_B = A_B + B_B;
  static_assert((BK == 64 || BK == 32) && NJ % 2 == 0 && NST >= 3 && (NST - 2) * PER <= 63, "ring shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;
  int mt, nt;
  xcd_tile(mt, nt);
  const int m0 = mt * FBM, n0 = nt * BN;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  f32x4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto swz = [](int row) { return BK == 64 ? ((row & 3) | (((row >> 3) & 1) << 2)) : ((row & 1) | (((row >> 3) & 1) << 1)); };
  // per-thread DMA sources of slab 0 (row clamped: products of rows / columns beyond M / N are never stored); slab kt is + kt * BK elements
  const bf16_t* srcA[ACH];
  const bf16_t* srcB[BCH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int sl = i * 256 + tid, row = sl / CPR, ch = (sl % CPR) ^ swz(row);
    srcA[i] = A + (size_t)min(m0 + row, p.M - 1) * p.lda + ch * 8;
  }
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    const int sl = i * 256 + tid, row = sl / CPR, ch = (sl % CPR) ^ swz(row);
    srcB[i] = B + (size_t)min(n0 + row, p.N - 1) * p.ldb + ch * 8;
  }
  auto dma = [&](int stage, int kt) {
    unsigned char* as = smem_raw + stage * STAGE_B;
    unsigned char* bs = as + A_B;
#pragma unroll
    for (int i = 0; i < ACH; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + kt * BK), (lptr_t)(as + (i * 256 + wave * 64) * 16), 16, 0, 0);      // wave-uniform base; lane l lands at + 16 l
#pragma unroll
    for (int i = 0; i < BCH; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcB[i] + kt * BK), (lptr_t)(bs + (i * 256 + wave * 64) * 16), 16, 0, 0);
  };
  // fragment addresses (bytes from the stage base; the tile index is an instruction immediate): A rows (mod 16) = lr; B rows of a tile PAIR are
  // interleaved, row = (lr >> 2) * 8 + t * 4 + (lr & 3) (gemm_nt_bf16_kernel)
  const int swa = swz(lr), browl = (lr >> 2) * 8 + (lr & 3), swb = swz(browl);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem_raw;
  unsigned aa[KS], ba[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    aa[ks] = lds0 + ((wm * 64 + lr) * BK + (((ks * 4) + lg) ^ swa) * 8) * 2;
    ba[ks] = lds0 + A_B + ((wn * (BN / 2) + browl) * BK + (((ks * 4) + lg) ^ swb) * 8) * 2;
  }
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#define NTR_RD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  const int nk = p.K / BK;                     // K % BK == 0 (dispatcher)
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) dma(s, s);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * PER) : "memory");      // slab kt has landed, NST - 2 later ones may be in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                              // (the last NST - 2 iterations wait for everything)
    __builtin_amdgcn_s_barrier();                    // ... for every wave's share; and every wave is done with slab kt - 1, whose stage is refilled now
    asm volatile("" ::: "memory");
    if (kt + NST - 1 < nk) dma((kt + NST - 1) % NST, kt + NST - 1);
    const unsigned so = (unsigned)(kt % NST) * STAGE_B;
    u32x4_t af[KS][4], bfr[KS][NJ];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i == 0) NTR_RD(af[ks][0], aa[ks] + so, 0);
        else if (i == 1) NTR_RD(af[ks][1], aa[ks] + so, 16 * BK * 2);
        else if (i == 2) NTR_RD(af[ks][2], aa[ks] + so, 32 * BK * 2);
        else NTR_RD(af[ks][3], aa[ks] + so, 48 * BK * 2);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {      // j = 2 jp + t: rows (jp * 32 + t * 4) * BK
        if (j == 0) NTR_RD(bfr[ks][0], ba[ks] + so, 0);
        else if (j == 1) NTR_RD(bfr[ks][1], ba[ks] + so, 4 * BK * 2);
        else if (j == 2) NTR_RD(bfr[ks][j], ba[ks] + so, 32 * BK * 2);
        else NTR_RD(bfr[ks][j], ba[ks] + so, 36 * BK * 2);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // LDS returns in order: "<= (KS - 1 - ks) * (4 + NJ) outstanding" covers the reads of k-step ks; the fragment registers are in / out operands
      // of the wait so that no MFMA is scheduled above it
      if (ks + 1 < KS) {
        if constexpr (NJ == 4) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(af[ks][0]), "+v"(af[ks][1]), "+v"(af[ks][2]), "+v"(af[ks][3]), "+v"(bfr[ks][0]), "+v"(bfr[ks][1]), "+v"(bfr[ks][2]), "+v"(bfr[ks][3]));
        else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af[ks][0]), "+v"(af[ks][1]), "+v"(af[ks][2]), "+v"(af[ks][3]), "+v"(bfr[ks][0]), "+v"(bfr[ks][1]));
      } else {
        if constexpr (NJ == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[ks][0]), "+v"(af[ks][1]), "+v"(af[ks][2]), "+v"(af[ks][3]), "+v"(bfr[ks][0]), "+v"(bfr[ks][1]), "+v"(bfr[ks][2]), "+v"(bfr[ks][3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[ks][0]), "+v"(af[ks][1]), "+v"(af[ks][2]), "+v"(af[ks][3]), "+v"(bfr[ks][0]), "+v"(bfr[ks][1]));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bfr[ks][j]), __builtin_bit_cast(bf16x8_t, af[ks][i]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);      // k-step ks's MFMAs stay in front of the wait for k-step ks + 1 (hipcc had sunk 15 of 16 behind it)
    }
  }
#undef NTR_RD
  nt_glds_epilogue<BN>(p, acc, m0, n0, wm, wn, lr, lg);
}

template <int BN, int EPI, int BK = FBK, bool GLDS = false>
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(const GemmP p) {
  constexpr int LDK = GLDS ? BK : BK + FPAD, CPR = BK / 8, ACH = FBM * CPR / 256;   // LDS row, 16-byte chunks per row, A chunks per thread
  using T = bf16_t;
  constexpr int NJ = BN / 32;                 // 16-wide N tiles per wave (wave tile 64 x BN/2)
  constexpr int BCH = BN * CPR / 256;           // 16-byte chunks of the B tile per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem_raw);                 // [2][128][72]
  bf16_t* Bs = As + 2 * FBM * LDK;                                  // [2][BN][72]
  float* stage = reinterpret_cast<float*>(smem_raw);                // [128][68] fp32 (epilogue)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;
  int mt, nt;
  xcd_tile(mt, nt);
  const int m0 = mt * FBM, n0 = nt * BN;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

  f32x4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  uint4 ra[ACH], rb[BCH];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int c = tid + 256 * i;
      ra[i] = ldg16_guard(A, m0 + c / CPR, p.M, p.lda, k0 + (c % CPR) * 8, p.K);
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int c = tid + 256 * i;
      rb[i] = ldg16_guard(B, n0 + c / CPR, p.N, p.ldb, k0 + (c % CPR) * 8, p.K);
    }
  };
  auto lstore = [&](int buf) {
    bf16_t* a = As + buf * FBM * LDK;
    bf16_t* b = Bs + buf * BN * LDK;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int c = tid + 256 * i;
      *reinterpret_cast<uint4*>(a + (c / CPR) * LDK + (c % CPR) * 8) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int c = tid + 256 * i;
      *reinterpret_cast<uint4*>(b + (c / CPR) * LDK + (c % CPR) * 8) = rb[i];
    }
  };

  const int nk = (p.K + BK - 1) / BK;
  if constexpr (GLDS) {
    // Swizzle: conflict-free for BOTH fragment row patterns below (tools/lds_bank_model.py): A rows are lr (mod 16); B rows
    // of a tile PAIR are interleaved, row = (lr >> 2) * 8 + t * 4 + (lr & 3), so that with the MFMA issued transposed
    // (D[n][m] = B . A^T) a lane ends up with 8 CONSECUTIVE output columns of one row: the epilogue is one 16-byte store per
    // row and pair straight from the accumulators - no LDS staging pass, no barriers.
    static_assert((BK == 64 || BK == 32) && EPI == EPI_STORE && NJ % 2 == 0, "swizzle table / plain epilogue");
    auto swz = [](int row) { return BK == 64 ? ((row & 3) | (((row >> 3) & 1) << 2)) : ((row & 1) | (((row >> 3) & 1) << 1)); };
    auto dma = [&](int buf, int k0) {
#pragma unroll
      for (int i = 0; i < ACH; ++i) {
        const int sl = i * 256 + tid, row = sl / CPR, ch = (sl % CPR) ^ swz(row);
        const bf16_t* src = A + (size_t)min(m0 + row, p.M - 1) * p.lda + k0 + ch * 8;
        bf16_t* dst = As + buf * FBM * LDK + (i * 256 + wave * 64) * 8;            // wave-uniform; lane l lands at + 8 l
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < BCH; ++i) {
        const int sl = i * 256 + tid, row = sl / CPR, ch = (sl % CPR) ^ swz(row);
        const bf16_t* src = B + (size_t)min(n0 + row, p.N - 1) * p.ldb + k0 + ch * 8;
        bf16_t* dst = Bs + buf * BN * LDK + (i * 256 + wave * 64) * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
      }
    };
    const int swa = swz(lr);                                         // A rows: (mod 16) = lr
    const int browl = (lr >> 2) * 8 + (lr & 3);                      // B rows of a pair: + t * 4
    const int swb = swz(browl);                                      // independent of t (bit 2 is not used)
    dma(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) dma((kt + 1) & 1, (kt + 1) * BK);
      const bf16_t* a = As + (kt & 1) * FBM * LDK + (wm * 64 + lr) * LDK;
      const bf16_t* b = Bs + (kt & 1) * BN * LDK + (wn * (BN / 2) + browl) * LDK;
#pragma unroll
      for (int ks = 0; ks < BK; ks += 32) {
        const int coa = (((ks >> 3) + lg) ^ swa) * 8, cob = (((ks >> 3) + lg) ^ swb) * 8;
        bf16x8_t af[4], bfr[NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a + i * 16 * LDK + coa));
#pragma unroll
        for (int j = 0; j < NJ; ++j)      // j = 2 jp + t
          bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(b + ((j >> 1) * 32 + (j & 1) * 4) * LDK + cob));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
      __syncthreads();                      // carries the vmcnt(0) that retires the slab just requested
    }
    nt_glds_epilogue<BN>(p, acc, m0, n0, wm, wn, lr, lg);
    return;
  } else {
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const bf16_t* a = As + (kt & 1) * FBM * LDK + (wm * 64 + lr) * LDK + lg * 8;
    const bf16_t* b = Bs + (kt & 1) * BN * LDK + (wn * (BN / 2) + lr) * LDK + lg * 8;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 32) {
      bf16x8_t af[4], bfr[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a + i * 16 * LDK + ks));
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(b + j * 16 * LDK + ks));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore((kt + 1) & 1);
    __syncthreads();
  }
  }

  // ---- epilogue: registers -> LDS (fp32, 64 columns at a time) -> 16-byte global stores ----
  constexpr int SLD = 68;
  constexpr bool STATS = (EPI == EPI_GELU_SUMSQ || EPI == EPI_DZ_STATS);
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
  const bf16_t* R = reinterpret_cast<const bf16_t*>(p.R);
  constexpr int HALVES = BN / 64;
  if constexpr (STATS) {
    // Column statistics straight from the accumulator layout (lane = column lr, 16 rows per wave
    // tile column): 16 in-register adds + 2 shuffles per column instead of a 64-lane butterfly.
    constexpr int HLD = BN + 8;
    bf16_t* htile = reinterpret_cast<bf16_t*>(smem_raw + 128 * SLD * sizeof(float));       // [128][BN+8] (DZ_STATS)
    // colacc[q][stat][BN]: q = row group inside the tile (one group for the batch-global sparse GRN; up to 4 for
    // the per-sample GRN of the dense decoder, 49 rows per group against 128-row tiles)
    float* colacc = reinterpret_cast<float*>(smem_raw + 128 * SLD * sizeof(float) + 128 * HLD * sizeof(bf16_t));  // [4][2][BN]
    uint8_t* actl = reinterpret_cast<uint8_t*>(colacc + 8 * BN);                            // [128] row activity
    const bool grouped = p.rpg > 0 && p.rpg < p.M;
    const int g0 = grouped ? m0 / p.rpg : 0;
    for (int i = tid; i < 8 * BN; i += 256) colacc[i] = 0.f;
    if (tid < 128) actl[tid] = (p.act && m0 + tid < p.M) ? p.act[m0 + tid] : 1;
    if constexpr (EPI == EPI_DZ_STATS) {
      for (int c = tid; c < 128 * (BN / 8); c += 256) {
        const int row = c / (BN / 8), ch = (c - row * (BN / 8)) * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (m0 + row < p.M && n0 + ch < p.N) v = *reinterpret_cast<const uint4*>(R + (size_t)(m0 + row) * p.ldr + n0 + ch);
        *reinterpret_cast<uint4*>(htile + row * HLD + ch) = v;
      }
    }
    int qrow[4][4];                                   // group slot of this lane's 16 rows (independent of the column tile)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) qrow[i][r] = grouped ? (m0 + wm * 64 + i * 16 + lg * 4 + r) / p.rpg - g0 : 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int cl = wn * (BN / 2) + j * 16 + lr;
      const int col = n0 + cl;
      const float bias = (p.bias && col < p.N) ? p.bias[col] : 0.f;
      float cs0[4] = {0.f, 0.f, 0.f, 0.f}, cs1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = wm * 64 + i * 16 + lg * 4 + r;
          const int row = m0 + rl;
          if (row < p.M && col < p.N) {
            float v = acc[i][j][r] + bias;
            if (!actl[rl]) v = 0.f;
            v = bf2f(f2bf(v));                       // statistics on the value as stored
            float a0, a1 = 0.f;
            if constexpr (EPI == EPI_GELU_SUMSQ) { const float g = gelu_t<T>(v); a0 = g * g; }
            else { a0 = v; a1 = v * gelu_t<T>(bf2f(htile[rl * HLD + cl])); }
            if (!grouped) { cs0[0] += a0; cs1[0] += a1; }
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) { cs0[q] += (qrow[i][r] == q) ? a0 : 0.f; cs1[q] += (qrow[i][r] == q) ? a1 : 0.f; }
            }
          }
        }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q > 0 && !grouped) break;
        float a = cs0[q];
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        if (lg == 0) atomicAdd(&colacc[(q * 2 + 0) * BN + cl], a);
        if constexpr (EPI == EPI_DZ_STATS) {
          float b = cs1[q];
          b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
          if (lg == 0) atomicAdd(&colacc[(q * 2 + 1) * BN + cl], b);
        }
      }
    }
    __syncthreads();
    // partial slabs: single group: ws[stat][tile][N]; grouped: ws[stat][tile][4][N] (folded by reduce_tile_groups_kernel)
    const int nq = grouped ? 4 : 1;
    for (int c = tid; c < nq * BN; c += 256) {
      const int q = c / BN, cc = c - q * BN;
      if (n0 + cc < p.N) {
        p.ws[((size_t)mt * nq + q) * p.N + n0 + cc] = colacc[(q * 2 + 0) * BN + cc];
        if (EPI == EPI_DZ_STATS) p.ws[(((size_t)gridDim.x + mt) * nq + q) * p.N + n0 + cc] = colacc[(q * 2 + 1) * BN + cc];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int half = 0; half < HALVES; ++half) {
    const bool mine = (HALVES == 1) || (wn == half);
    if (mine) {
      const int cbase = (HALVES == 1) ? wn * 32 : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stage[(wm * 64 + i * 16 + lg * 4 + r) * SLD + cbase + j * 16 + lr] = acc[i][j][r];
    }
    __syncthreads();
    {
      const int row = tid >> 1, cseg = (tid & 1) * 32;
      const int grow = m0 + row;

      const bool live = !p.act || (grow < p.M && p.act[grow]);
      if (grow < p.M) {
#pragma unroll
        for (int v8 = 0; v8 < 4; ++v8) {
          const int col = n0 + half * 64 + cseg + v8 * 8;
          if (col < p.N) {              // N % 8 == 0 guaranteed by the dispatcher
            float v[8];
            const float4 s0 = *reinterpret_cast<const float4*>(stage + row * SLD + cseg + v8 * 8);
            const float4 s1 = *reinterpret_cast<const float4*>(stage + row * SLD + cseg + v8 * 8 + 4);
            v[0] = s0.x; v[1] = s0.y; v[2] = s0.z; v[3] = s0.w; v[4] = s1.x; v[5] = s1.y; v[6] = s1.z; v[7] = s1.w;
            if (p.bias) {
              const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
              const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
              v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
              v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            if (EPI == EPI_STORE && R) {
              float rr[8];
              ld8<bf16_t>(R + (size_t)grow * p.ldr + col, rr);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rr[e];
            }
            if (!live) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            st8<bf16_t>(C + (size_t)grow * p.ldc + col, v);
          }
        }
      }
    }
    __syncthreads();
  }
}

// =====================================================================================
// TN form (weight gradients): dW[n*sn + k*sk] += sum_m P[m,n]*Q[m,k]; db[n] += sum_m P[m,n]
// =====================================================================================
constexpr int TBM = 32, TLD = TBM + 8;     // reduction slab of 32 rows; LDS row = 40 bf16 = 80 B

__device__ __forceinline__ void tn_load_pack(const bf16_t* X, int ld, int ncols, int mbase, int mend, int col,
                                             uint4& lo, uint4& hi) {
  // reads the 2-column pair (col, col+1) for 8 consecutive rows; returns the two m-contiguous
  // 8-element vectors (column col -> lo, column col+1 -> hi)
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = mbase + i;
    w[i] = (m < mend && col < ncols) ? *reinterpret_cast<const uint32_t*>(X + (size_t)m * ld + col) : 0u;
  }
  lo.x = __byte_perm(w[0], w[1], 0x5410); hi.x = __byte_perm(w[0], w[1], 0x7632);
  lo.y = __byte_perm(w[2], w[3], 0x5410); hi.y = __byte_perm(w[2], w[3], 0x7632);
  lo.z = __byte_perm(w[4], w[5], 0x5410); hi.z = __byte_perm(w[4], w[5], 0x7632);
  lo.w = __byte_perm(w[6], w[7], 0x5410); hi.w = __byte_perm(w[6], w[7], 0x7632);
}

__device__ __forceinline__ float bf16x8_sum(const uint4& v) {
  return __uint_as_float(v.x << 16) + __uint_as_float(v.x & 0xffff0000u) + __uint_as_float(v.y << 16) +
         __uint_as_float(v.y & 0xffff0000u) + __uint_as_float(v.z << 16) + __uint_as_float(v.z & 0xffff0000u) +
         __uint_as_float(v.w << 16) + __uint_as_float(v.w & 0xffff0000u);
}

__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const WgradP w) {
  __shared__ __attribute__((aligned(16))) bf16_t Pt[2][128 * TLD];
  __shared__ __attribute__((aligned(16))) bf16_t Qt[2][128 * TLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
  const int mbeg = blockIdx.z * w.rows_per_split;
  const int mend = min(w.M, mbeg + w.rows_per_split);
  const bf16_t* P = reinterpret_cast<const bf16_t*>(w.P);
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(w.Q);
  const bool do_db = (w.db != nullptr) && (blockIdx.y == 0);

  // loader mapping: column pair cp = tid % 64 (columns 2cp, 2cp+1), row octet ro = tid / 64 (rows 8ro..8ro+7)
  const int cp = (tid & 63) * 2, ro = (tid >> 6) * 8;
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float db0 = 0.f, db1 = 0.f;
  uint4 plo, phi, qlo, qhi;

  auto gload = [&](int mb) {
    tn_load_pack(P, w.ldp, w.Nn, mb + ro, mend, n0 + cp, plo, phi);
    tn_load_pack(Q, w.ldq, w.Kk, mb + ro, mend, k0 + cp, qlo, qhi);
  };
  auto lstore = [&](int buf) {
    *reinterpret_cast<uint4*>(&Pt[buf][cp * TLD + ro]) = plo;
    *reinterpret_cast<uint4*>(&Pt[buf][(cp + 1) * TLD + ro]) = phi;
    *reinterpret_cast<uint4*>(&Qt[buf][cp * TLD + ro]) = qlo;
    *reinterpret_cast<uint4*>(&Qt[buf][(cp + 1) * TLD + ro]) = qhi;
    if (do_db) { db0 += bf16x8_sum(plo); db1 += bf16x8_sum(phi); }
  };

  const int nsl = (mend - mbeg + TBM - 1) / TBM;
  if (nsl > 0) {
    gload(mbeg);
    lstore(0);
  }
  __syncthreads();
  for (int s = 0; s < nsl; ++s) {
    if (s + 1 < nsl) gload(mbeg + (s + 1) * TBM);
    const bf16_t* a = &Pt[s & 1][(wn * 64 + lr) * TLD + lg * 8];
    const bf16_t* b = &Qt[s & 1][(wk * 64 + lr) * TLD + lg * 8];
    bf16x8_t af[4], bfr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a + i * 16 * TLD));
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(b + j * 16 * TLD));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if (s + 1 < nsl) lstore((s + 1) & 1);
    __syncthreads();
  }
  float* slab = w.ws + (size_t)blockIdx.z * w.Nn * w.Kk;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + lg * 4 + r;
        const int k = k0 + wk * 64 + j * 16 + lr;
        if (n < w.Nn && k < w.Kk) slab[(size_t)n * w.Kk + k] = acc[i][j][r];
      }
  if (do_db) {
    // the 4 row-octet groups hold partial sums for the same column pair: reduce through LDS
    float* red = reinterpret_cast<float*>(&Pt[0][0]);
    red[(tid >> 6) * 128 + cp] = db0;
    red[(tid >> 6) * 128 + cp + 1] = db1;
    __syncthreads();
    float* dslab = w.ws + (size_t)gridDim.z * w.Nn * w.Kk + (size_t)blockIdx.z * w.Nn;
    if (tid < 128 && n0 + tid < w.Nn) dslab[n0 + tid] = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
  }
}

// =====================================================================================
// Element-wise GRN application and its backward over [M, H] rows (materialised operands of
// the compute-shaped GEMMs), and column statistics.
// =====================================================================================
// z = gelu(h) * scale[g, j] + beta[j]   (inactive rows -> 0)
template <typename T>
__global__ __launch_bounds__(256) void grn_apply_kernel(const T* __restrict__ h, T* __restrict__ z,
                                                        const float* __restrict__ scale, const float* __restrict__ beta,
                                                        int M, int H, int rpg, const uint8_t* __restrict__ act) {
  const size_t nvec = (size_t)M * H / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 8;
    const int m = e / H, j = e - (size_t)m * H;
    float v[8];
    ld8<T>(h + e, v);
    const bool live = !act || act[m];
    // parameter vectors are loaded unconditionally as two float4 each: inside the `live` select
    // hipcc scalarises them into 16 dependent dword loads with a vmcnt(0) after every one
    const float* sc = scale + (size_t)(m / rpg) * H + j;
    const float4 s0 = *reinterpret_cast<const float4*>(sc), s1 = *reinterpret_cast<const float4*>(sc + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + j), b1 = *reinterpret_cast<const float4*>(beta + j + 4);
    const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float bev[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float r = gelu_t<T>(v[q]) * scv[q] + bev[q];
      v[q] = live ? r : 0.f;
    }
    st8<T>(z + e, v);
  }
}

// dh = (dz*scale[g,j] + coef[g,j]*gelu(h)) * gelu'(h), written over dz
template <typename T>
__global__ __launch_bounds__(256) void grn_bwd_apply_kernel(T* __restrict__ dz, const T* __restrict__ h,
                                                            const float* __restrict__ scale, const float* __restrict__ coef,
                                                            int M, int H, int rpg) {
  const size_t nvec = (size_t)M * H / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 8;
    const int m = e / H, j = e - (size_t)m * H;
    float d[8], hv[8];
    ld8<T>(dz + e, d);
    ld8<T>(h + e, hv);
    const size_t gi = (size_t)(m / rpg) * H + j;
    const float4 s0 = *reinterpret_cast<const float4*>(scale + gi), s1 = *reinterpret_cast<const float4*>(scale + gi + 4);
    const float4 c0 = *reinterpret_cast<const float4*>(coef + gi), c1 = *reinterpret_cast<const float4*>(coef + gi + 4);
    const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float cov[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float g, dg;
      gelu_both_t<T>(hv[q], g, dg);
      d[q] = (d[q] * scv[q] + cov[q] * g) * dg;
    }
    st8<T>(dz + e, d);
  }
}

// The two element-wise GRN passes with their FINALISATION in the prologue (round 5; single GRN group - the batch-global sparse GRN - at the
// widths whose pointwise products run on the tiled GEMMs, C = 320 / 384): every workgroup recomputes the H-vector from the column sums
// (grn_fwd_finalize_kernel / grn_bwd_finalize_kernel of rows.cuh, same summation order) into LDS, workgroup 0 publishes it (and adds the GRN
// gamma / beta gradients): one 7-us launch less per block and direction on the main lane.
template <typename T>
__global__ __launch_bounds__(256) void grn_apply_fin_kernel(const T* __restrict__ h, T* __restrict__ z, const float* __restrict__ G2,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            int M, int H, const uint8_t* __restrict__ act, float* __restrict__ Gx,
                                                            float* __restrict__ Ainv, float* __restrict__ scale) {
  extern __shared__ float gaf_smem[];             // [2][H]: scale | beta
  __shared__ float red[4];
  float* sc = gaf_smem;
  float* bt = gaf_smem + H;
  float s = 0.f;
  for (int j = threadIdx.x; j < H; j += 256) {
    const float v = sqrtf(G2[j]);
    sc[j] = v;
    bt[j] = beta[j];
    s += v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float A = (red[0] + red[1] + red[2] + red[3]) / H;
  const float ainv = 1.f / (A + eps);
  const bool pub = blockIdx.x == 0;
  if (pub && threadIdx.x == 0) Ainv[0] = ainv;
  for (int j = threadIdx.x; j < H; j += 256) {
    const float gx = sc[j], v = 1.f + gamma[j] * (gx * ainv);
    sc[j] = v;
    if (pub) { Gx[j] = gx; scale[j] = v; }
  }
  __syncthreads();
  const size_t nvec = (size_t)M * H / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 8;
    const int m = e / H, j = e - (size_t)m * H;
    float v[8];
    ld8<T>(h + e, v);
    const uint8_t lv = *(act ? act + m : reinterpret_cast<const uint8_t*>(h));
    const bool live = !act || lv;
    const float4 s0 = *reinterpret_cast<const float4*>(sc + j), s1 = *reinterpret_cast<const float4*>(sc + j + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(bt + j), b1 = *reinterpret_cast<const float4*>(bt + j + 4);
    const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float bev[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float r = gelu_t<T>(v[q]) * scv[q] + bev[q];
      v[q] = live ? r : 0.f;
    }
    st8<T>(z + e, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void grn_bwd_apply_fin_kernel(T* __restrict__ dz, const T* __restrict__ h, const float* __restrict__ scale,
                                                                const float* __restrict__ S0, const float* __restrict__ S1,
                                                                const float* __restrict__ Gx, const float* __restrict__ Ainv,
                                                                const float* __restrict__ gamma, int M, int H, float* __restrict__ coef,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
  extern __shared__ float gaf_smem[];             // [2][H]: scale | coef
  __shared__ float red[4];
  float* sc = gaf_smem;
  float* cf = gaf_smem + H;
  const float ainv = Ainv[0];
  float s = 0.f;
  for (int j = threadIdx.x; j < H; j += 256) {
    sc[j] = scale[j];
    s += gamma[j] * S1[j] * Gx[j];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float T2 = (red[0] + red[1] + red[2] + red[3]) * ainv * ainv / H;
  const bool pub = blockIdx.x == 0;
  for (int j = threadIdx.x; j < H; j += 256) {
    const float gx = Gx[j], s1 = S1[j];
    const float dGx = gamma[j] * s1 * ainv - T2;
    const float c = (gx > 0.f) ? dGx / gx : 0.f;
    cf[j] = c;
    if (pub) {
      coef[j] = c;
      atomicAdd(dgamma + j, gx * ainv * s1);
      atomicAdd(dbeta + j, S0[j]);
    }
  }
  __syncthreads();
  const size_t nvec = (size_t)M * H / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 8;
    const int m = e / H, j = e - (size_t)m * H;
    float d[8], hv[8];
    ld8<T>(dz + e, d);
    ld8<T>(h + e, hv);
    const float4 s0 = *reinterpret_cast<const float4*>(sc + j), s1 = *reinterpret_cast<const float4*>(sc + j + 4);
    const float4 c0 = *reinterpret_cast<const float4*>(cf + j), c1 = *reinterpret_cast<const float4*>(cf + j + 4);
    const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float cov[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float g, dg;
      gelu_both_t<T>(hv[q], g, dg);
      d[q] = (d[q] * scv[q] + cov[q] * g) * dg;
    }
    st8<T>(dz + e, d);
  }
}

// column statistics over row groups: mode 0: s0[g,j] += sum gelu(h)^2
//                                   mode 1: s0[g,j] += sum dz ; s1[g,j] += sum dz*gelu(h)
// grid = (ceil(H/64), row chunks); block 256 = 64 columns x 4 row lanes
template <typename T>
__global__ __launch_bounds__(256) void colstats_kernel(const T* __restrict__ h, const T* __restrict__ dz, int mode,
                                                       float* __restrict__ s0, float* __restrict__ s1, int M, int H,
                                                       int rpg, int rows_per_block, float* __restrict__ ws) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int mb = blockIdx.y * rows_per_block, me = min(M, mb + rows_per_block);
  int g = mb / rpg;            // rows_per_block never straddles a group boundary when rpg < M
  float a0 = 0.f, a1 = 0.f;
  if (c < H) {
    for (int m = mb + rl; m < me; m += 4) {
      const float gv = gelu_t<T>(ldf<T>(h + (size_t)m * H + c));
      if (mode == 0) a0 += gv * gv;
      else { const float d = ldf<T>(dz + (size_t)m * H + c); a0 += d; a1 += d * gv; }
    }
  }
  red[0][rl][threadIdx.x & 63] = a0;
  red[1][rl][threadIdx.x & 63] = a1;
  __syncthreads();
  if (rl == 0 && c < H) {
    const int l = threadIdx.x & 63;
    const float r0 = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
    const float r1 = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
    if (rpg < M) {                 // one block per (group, column chunk): sole writer
      s0[(size_t)g * H + c] += r0;
      if (mode == 1) s1[(size_t)g * H + c] += r1;
    } else {                       // single group: slab row per row-block, reduced afterwards
      ws[(size_t)blockIdx.y * H + c] = r0;
      if (mode == 1) ws[((size_t)gridDim.y + blockIdx.y) * H + c] = r1;
    }
  }
}

// second stage of the per-group column statistics of 128-row GEMM tiles (see the STATS epilogue above):
// out[g][n] += sum over the (at most 2) tiles that overlap group g of ws[tile][g - first_group(tile)][n]
__global__ __launch_bounds__(256) void reduce_tile_groups_kernel(const float* __restrict__ ws, int mtiles, int N, int rpg, int G,
                                                                 float* __restrict__ out) {
  const size_t total = (size_t)G * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = i / N, n = i - (size_t)g * N;
    const int t0 = (g * rpg) / 128, t1 = ((g + 1) * rpg - 1) / 128;
    float s = 0.f;
    for (int t = t0; t <= t1 && t < mtiles; ++t) s += ws[((size_t)t * 4 + (g - (t * 128) / rpg)) * N + n];
    out[i] += s;
  }
}
