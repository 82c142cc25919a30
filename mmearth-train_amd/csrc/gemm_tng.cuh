// GROUPED weight-gradient GEMM for the encoder's pointwise layers: all pwconv1 / pwconv2 weight gradients of one stage in ONE launch.
//   dW_p[n][k] = sum_m P_p[m][n] Q_p[m][k],  db_p[n] = sum_m P_p[m][n]      for problems p = 0..nprob-1 of identical shape
// (autograd of MinkowskiLinear in the reference's sparse Block, models/convnextv2_sparse.py:41-43, 51-53).
//
// Why: profiles/r03/wgrad_probe.txt - launched one by one, every encoder weight gradient took a flat 25-27 us whatever its shape
// (a serial chain per 32-row slab with one slab of look-ahead, 76 row splits each writing a 410 KB fp32 slab, a fold launch behind
// every kernel). The operands of a whole stage persist until the stage's data-gradient chain is done (the dz / dx rings are as
// deep as the net), so the stage's 2 x depth problems run as one grid:
//   * X = the narrow operand (C columns), Y = the wide one (4C). Everything is built from 80-COLUMN REGIONS: a region row is
//     160 bytes = FIVE 32-byte bank groups, an odd number, so the 8 rows a half-wave touches in a ds_read_b64_tr_b16 fall into 8
//     distinct groups with NO padding and NO swizzle - which is what lets the slabs go global -> LDS by DMA
//     (global_load_lds_dwordx4 writes a lane-linear image: rows cannot be padded). C = 80 / 160 / 320 and 4C are multiples of 80.
//   * workgroup tile = RX x RY regions, one 80 x 80 output block (5 x 5 MFMA tiles, 25 mfma_f32_16x16x32_bf16 per 32-row k-step,
//     100 accumulator VGPRs) per wave: (2, 2) = 160 x 160 at C >= 160, (1, 4) = 80 x 320 at C = 80;
//   * a ring of NST 32-row stages filled by DMA, counted vmcnt waits, one bare s_barrier per k-step (as gemm_tn3.cuh), <= 80 KB
//     of LDS: two workgroups per CU;
//   * row splits are the FASTEST-varying workgroup coordinate and a multiple of 8 where possible: the tiles of one row range
//     land on one XCD and share its L2 (X is re-read by the 4 / 8 y tiles of a range);
//   * every (problem, split) writes one [Nn Kk + Nn] fp32 slab; ONE fold launch (wgrad_group_fold_kernel) adds the <= 8-64
//     slabs of every problem into dW / db in a fixed order (deterministic, unlike atomics).
// Rows past M (ragged last k-step) are loaded from the clamped last row and zeroed in LDS before use.
#pragma once
#include "gemm_tn3.cuh"

constexpr int TNG_MAXP = 20;        // problems per launch
constexpr int TNG_RW = 80;          // region width (columns)
constexpr int TNG_SL = 32;          // rows per stage = one MFMA k-step
constexpr int TNG_RB = TNG_SL * TNG_RW * 2;      // bytes of one region stage (5120)

struct TngProb {
  const bf16_t* X; const bf16_t* Y;      // narrow / wide operand, row-major [M][ld]
  int ldx, ldy;
  float* slab;                           // [splits][WX WY + Nn] partials of this problem
  int swap;                              // 0: P = X (Nn = WX, Kk = WY; pwconv2) | 1: P = Y (Nn = WY, Kk = WX; pwconv1)
  int want_db;
};
struct TngP {
  int nprob, M, WX, WY, rps, splits, xt, yt;
  TngProb p[TNG_MAXP];
};

template <int RX, int RY, int NST>
struct TngCfg {
  static constexpr int NREG = RX + RY, NI = NREG * 5;                      // DMA wave-instructions (1 KB each) per stage
  static constexpr int PER = (NI + 3) / 4;                                  // ... per wave (the last ones of a ragged count load into a dummy area)
  static constexpr int STAGE_B = NREG * TNG_RB;
  static constexpr int DUMMY_B = (4 * PER - NI) * 1024;
  static constexpr int LDS = NST * STAGE_B + DUMMY_B;
};

template <int RX, int RY, int NST, bool SWAP>
__device__ __forceinline__ void tng_body(const TngP& g, const TngProb& pr, int tile, int split, unsigned char* smem) {
  using Cf = TngCfg<RX, RY, NST>;
  constexpr int PER = Cf::PER, STAGE_B = Cf::STAGE_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int wx = wave % RX, wy = wave / RX;
  const int xtile = tile % g.xt, ytile = tile / g.xt;
  const int x0 = xtile * (RX * TNG_RW), y0 = ytile * (RY * TNG_RW);
  const int mbeg = split * g.rps, mend = min(g.M, mbeg + g.rps);
  const int nsl = (mend - mbeg + TNG_SL - 1) / TNG_SL;

  f32x4_t acc[5][5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t accb[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) accb[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bool do_db = pr.want_db && (SWAP ? (xtile == 0 && wx == 0) : (ytile == 0 && wy == 0));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  // DMA plan of this wave: instruction I = wave + 4 k moves 64 consecutive 16-byte chunks of region q = I / 5 (chunk = (I % 5) * 64 + lane;
  // a region stage is 32 rows x 10 chunks), to LDS offset q * RB + (I % 5) * 1024 of the stage
  const bf16_t* dsrc[PER];
  int drow[PER], dld[PER];
  unsigned ddst[PER];
  bool dreal[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int I = wave + 4 * k;
    const bool real = I < Cf::NI;
    const int q = real ? I / 5 : 0, ii = real ? I % 5 : 0;
    const int chunk = ii * 64 + lane, row = chunk / 10, cc = chunk - row * 10;
    const bool isx = q < RX;
    dsrc[k] = (isx ? pr.X + x0 + q * TNG_RW : pr.Y + y0 + (q - RX) * TNG_RW) + cc * 8;
    dld[k] = isx ? pr.ldx : pr.ldy;
    drow[k] = row;
    dreal[k] = real;
    ddst[k] = real ? (unsigned)(q * TNG_RB + ii * 1024) : (unsigned)(NST * STAGE_B + (I - Cf::NI) * 1024);
  }
  auto dma = [&](int s) {                  // stage s of this split -> ring slot s % NST
    const int mb = mbeg + s * TNG_SL;
    const unsigned so = (unsigned)(s % NST) * STAGE_B;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int r = min(mb + drow[k], g.M - 1);                         // (rows past M: clamped, zeroed in LDS below)
      const bf16_t* src = dsrc[k] + (size_t)r * dld[k];
      unsigned char* dst = smem + (dreal[k] ? so : 0u) + ddst[k];
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    }
  };

  // transpose-read addresses: row rl = lg * 4 + (lr >> 2) of a 16-row half, columns 4 (lr & 3) .. + 3 of MFMA tile t of the wave's region
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  const int rl = lg * 4 + (lr >> 2);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned xlane = lds0 + wx * TNG_RB + rl * (TNG_RW * 2) + (lr & 3) * 8;
  const unsigned ylane = lds0 + (RX + wy) * TNG_RB + rl * (TNG_RW * 2) + (lr & 3) * 8;
#define TNG_TR(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  auto mk = [](const u32x2_t& lo, const u32x2_t& hi) { return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y)); };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nsl) dma(s);

  for (int s = 0; s < nsl; ++s) {
    // stage s has landed when at most the younger stages' requests are outstanding: min(NST - 2, nsl - 1 - s) stages of PER requests
    static_assert(NST >= 2 && NST <= 5, "ring depth");
    const int young = min(NST - 2, nsl - 1 - s);
    if (young >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
    else if (young == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (young == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // ... for every wave; and every wave is done with stage s - 1
    asm volatile("" ::: "memory");
    const unsigned so = (unsigned)(s % NST) * STAGE_B;
    const int valid = mend - (mbeg + s * TNG_SL);         // rows of this stage inside the split
    if (valid < TNG_SL) {                                 // ragged last k-step: zero the rows past the end (workgroup-uniform branch)
      const int nz = (TNG_SL - valid) * (TNG_RW * 2 / 16);      // 16-byte chunks per region
      for (int q = 0; q < Cf::NREG; ++q)
        for (int c = tid; c < nz; c += 256)
          *reinterpret_cast<uint4*>(smem + so + q * TNG_RB + valid * (TNG_RW * 2) + c * 16) = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
    }
    if (s + NST - 1 < nsl) dma(s + NST - 1);              // into the slot consumed in iteration s - 1
    u32x2_t xl[5], xh[5], yl[5], yh[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { TNG_TR(xl[i], xlane + so + i * 32, 0); TNG_TR(xh[i], xlane + so + i * 32, 16 * TNG_RW * 2); }
#pragma unroll
    for (int j = 0; j < 5; ++j) { TNG_TR(yl[j], ylane + so + j * 32, 0); TNG_TR(yh[j], ylane + so + j * 32, 16 * TNG_RW * 2); }
    // LDS returns in order: <= 8 outstanding of the 20 reads = the 10 X fragments and Y tile 0 are there
    asm volatile("s_waitcnt lgkmcnt(8)"
                 : "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[3]), "+v"(xl[4]), "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xh[4]),
                   "+v"(yl[0]), "+v"(yh[0]));
    bf16x8_t xf[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) xf[i] = mk(xl[i], xh[i]);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j == 1) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(yl[1]), "+v"(yh[1]));
      if (j == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(yl[2]), "+v"(yh[2]));
      if (j == 3) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(yl[3]), "+v"(yh[3]));
      if (j == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yl[4]), "+v"(yh[4]));
      const bf16x8_t yf = mk(yl[j], yh[j]);
#pragma unroll
      for (int i = 0; i < 5; ++i)
        acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, xf[i], acc[i][j], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf, acc[i][j], 0, 0, 0);
      if (SWAP && do_db) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, ones, accb[j], 0, 0, 0);
    }
    if (!SWAP && do_db) {
#pragma unroll
      for (int i = 0; i < 5; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], ones, accb[i], 0, 0, 0);
    }
  }
#undef TNG_TR

  // D layout: col = lr, row = lg * 4 + r. !SWAP: row = x (n), col = y (k). SWAP: row = y (n), col = x (k). Slab = [Nn Kk | Nn].
  const int Nn = SWAP ? g.WY : g.WX, Kk = SWAP ? g.WX : g.WY;
  float* slab = pr.slab + (size_t)split * ((size_t)Nn * Kk + Nn);
  const int xb = x0 + wx * TNG_RW, yb = y0 + wy * TNG_RW;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = SWAP ? yb + j * 16 + lg * 4 + r : xb + i * 16 + lg * 4 + r;
        const int k = SWAP ? xb + i * 16 + lr : yb + j * 16 + lr;
        slab[(size_t)n * Kk + k] = acc[i][j][r];
      }
  if (do_db && lr == 0) {
    float* dslab = slab + (size_t)Nn * Kk;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) dslab[(SWAP ? yb : xb) + i * 16 + lg * 4 + r] = accb[i][r];
  }
}

template <int RX, int RY, int NST>
__global__ __launch_bounds__(256, 2) void gemm_tng_kernel(const TngP g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tng_smem[];
  // workgroup -> (problem, tile, split), split fastest
  const int b = blockIdx.x;
  const int split = b % g.splits, t = b / g.splits;
  const int ntiles = g.xt * g.yt;
  const int tile = t % ntiles, prob = t / ntiles;
  // the problem record is read from the kernel-argument segment itself (scalar loads with a dynamic offset): indexing the by-value copy
  // with a runtime index makes hipcc spill the whole argument struct to scratch (ps.cuh)
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(4))) const char* kchar_p;
  typedef __attribute__((address_space(4))) const TngProb* kprob_p;
  const TngProb pr = ((kprob_p)((kchar_p)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TngP, p)))[prob];
#else
  const TngProb pr = g.p[prob];
#endif
  if (pr.swap) tng_body<RX, RY, NST, true>(g, pr, tile, split, tng_smem);
  else tng_body<RX, RY, NST, false>(g, pr, tile, split, tng_smem);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same kernel for the widths that are multiples of 96 (tiny / large: 96 .. 768, femto from stage 1): 48-COLUMN regions (a region row
// is 96 bytes = THREE 32-byte bank groups, odd again), a wave owns 2 x 2 regions = a 96 x 96 output block (6 x 6 MFMA tiles, 36 MFMAs per
// k-step, 144 accumulator VGPRs); workgroup = (2, 2) waves = 192 x 192 at C >= 192, (1, 4) waves = 96 x 384 at C = 96.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int TNG48_RW = 48, TNG48_RB = TNG_SL * TNG48_RW * 2;      // 3072 bytes per region stage = 3 DMA wave-instructions
template <int RX, int RY, int NST>
struct Tng48Cfg {
  static constexpr int NREG = 2 * (RX + RY), NI = NREG * 3, PER = (NI + 3) / 4;
  static constexpr int STAGE_B = NREG * TNG48_RB, DUMMY_B = (4 * PER - NI) * 1024, LDS = NST * STAGE_B + DUMMY_B;
};

template <int RX, int RY, int NST, bool SWAP>
__device__ __forceinline__ void tng48_body(const TngP& g, const TngProb& pr, int tile, int split, unsigned char* smem) {
  using Cf = Tng48Cfg<RX, RY, NST>;
  constexpr int PER = Cf::PER, STAGE_B = Cf::STAGE_B, RW = TNG48_RW, RB = TNG48_RB, NT = 6;      // 6 MFMA tiles per wave and side
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int wx = wave % RX, wy = wave / RX;
  const int xtile = tile % g.xt, ytile = tile / g.xt;
  const int x0 = xtile * (2 * RX * RW), y0 = ytile * (2 * RY * RW);
  const int mbeg = split * g.rps, mend = min(g.M, mbeg + g.rps);
  const int nsl = (mend - mbeg + TNG_SL - 1) / TNG_SL;

  f32x4_t acc[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t accb[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) accb[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bool do_db = pr.want_db && (SWAP ? (xtile == 0 && wx == 0) : (ytile == 0 && wy == 0));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  // DMA plan: instruction I = wave + 4 k moves chunks (I % 3) * 64 + lane of region q = I / 3 (a region stage is 32 rows x 6 chunks)
  const bf16_t* dsrc[PER];
  int drow[PER], dld[PER];
  unsigned ddst[PER];
  bool dreal[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int I = wave + 4 * k;
    const bool real = I < Cf::NI;
    const int q = real ? I / 3 : 0, ii = real ? I % 3 : 0;
    const int chunk = ii * 64 + lane, row = chunk / 6, cc = chunk - row * 6;
    const bool isx = q < 2 * RX;
    dsrc[k] = (isx ? pr.X + x0 + q * RW : pr.Y + y0 + (q - 2 * RX) * RW) + cc * 8;
    dld[k] = isx ? pr.ldx : pr.ldy;
    drow[k] = row;
    dreal[k] = real;
    ddst[k] = real ? (unsigned)(q * RB + ii * 1024) : (unsigned)(NST * STAGE_B + (I - Cf::NI) * 1024);
  }
  auto dma = [&](int s) {
    const int mb = mbeg + s * TNG_SL;
    const unsigned so = (unsigned)(s % NST) * STAGE_B;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int r = min(mb + drow[k], g.M - 1);
      const bf16_t* src = dsrc[k] + (size_t)r * dld[k];
      unsigned char* dst = smem + (dreal[k] ? so : 0u) + ddst[k];
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    }
  };

  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  const int rl = lg * 4 + (lr >> 2);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned xlane = lds0 + (2 * wx) * RB + rl * (RW * 2) + (lr & 3) * 8;
  const unsigned ylane = lds0 + (2 * RX + 2 * wy) * RB + rl * (RW * 2) + (lr & 3) * 8;
#define TNG_TR(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  auto mk = [](const u32x2_t& lo, const u32x2_t& hi) { return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y)); };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nsl) dma(s);

  for (int s = 0; s < nsl; ++s) {
    static_assert(NST == 3, "ring depth");
    if (s + 1 < nsl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned so = (unsigned)(s % NST) * STAGE_B;
    const int valid = mend - (mbeg + s * TNG_SL);
    if (valid < TNG_SL) {
      const int nz = (TNG_SL - valid) * (RW * 2 / 16);
      for (int q = 0; q < Cf::NREG; ++q)
        for (int c = tid; c < nz; c += 256)
          *reinterpret_cast<uint4*>(smem + so + q * RB + valid * (RW * 2) + c * 16) = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
    }
    if (s + NST - 1 < nsl) dma(s + NST - 1);
    // tile i of a side: region i / 3 (+ RB), MFMA tile i % 3 of it (+ 32 bytes); second half of the k-step: + 16 rows
    u32x2_t xl[NT], xh[NT], yl[NT], yh[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) { TNG_TR(xl[i], xlane + so, (i / 3) * RB + (i % 3) * 32); TNG_TR(xh[i], xlane + so, (i / 3) * RB + (i % 3) * 32 + 16 * RW * 2); }
#pragma unroll
    for (int j = 0; j < NT; ++j) { TNG_TR(yl[j], ylane + so, (j / 3) * RB + (j % 3) * 32); TNG_TR(yh[j], ylane + so, (j / 3) * RB + (j % 3) * 32 + 16 * RW * 2); }
    asm volatile("s_waitcnt lgkmcnt(10)"
                 : "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[3]), "+v"(xl[4]), "+v"(xl[5]), "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]),
                   "+v"(xh[4]), "+v"(xh[5]), "+v"(yl[0]), "+v"(yh[0]));
    bf16x8_t xf[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) xf[i] = mk(xl[i], xh[i]);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j == 1) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(yl[1]), "+v"(yh[1]));
      if (j == 2) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(yl[2]), "+v"(yh[2]));
      if (j == 3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(yl[3]), "+v"(yh[3]));
      if (j == 4) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(yl[4]), "+v"(yh[4]));
      if (j == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yl[5]), "+v"(yh[5]));
      const bf16x8_t yf = mk(yl[j], yh[j]);
#pragma unroll
      for (int i = 0; i < NT; ++i)
        acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, xf[i], acc[i][j], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf, acc[i][j], 0, 0, 0);
      if (SWAP && do_db) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, ones, accb[j], 0, 0, 0);
    }
    if (!SWAP && do_db) {
#pragma unroll
      for (int i = 0; i < NT; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], ones, accb[i], 0, 0, 0);
    }
  }
#undef TNG_TR

  const int Nn = SWAP ? g.WY : g.WX, Kk = SWAP ? g.WX : g.WY;
  float* slab = pr.slab + (size_t)split * ((size_t)Nn * Kk + Nn);
  const int xb = x0 + wx * 2 * RW, yb = y0 + wy * 2 * RW;      // tile i covers columns xb + 16 i .. (the two regions of a wave are adjacent)
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = SWAP ? yb + j * 16 + lg * 4 + r : xb + i * 16 + lg * 4 + r;
        const int k = SWAP ? xb + i * 16 + lr : yb + j * 16 + lr;
        slab[(size_t)n * Kk + k] = acc[i][j][r];
      }
  if (do_db && lr == 0) {
    float* dslab = slab + (size_t)Nn * Kk;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) dslab[(SWAP ? yb : xb) + i * 16 + lg * 4 + r] = accb[i][r];
  }
}

template <int RX, int RY, int NST>
__global__ __launch_bounds__(256) void gemm_tng48_kernel(const TngP g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tng_smem[];
  const int b = blockIdx.x;
  const int split = b % g.splits, t = b / g.splits;
  const int ntiles = g.xt * g.yt;
  const int tile = t % ntiles, prob = t / ntiles;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(4))) const char* kchar_p;
  typedef __attribute__((address_space(4))) const TngProb* kprob_p;
  const TngProb pr = ((kprob_p)((kchar_p)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TngP, p)))[prob];
#else
  const TngProb pr = g.p[prob];
#endif
  if (pr.swap) tng48_body<RX, RY, NST, true>(g, pr, tile, split, tng_smem);
  else tng48_body<RX, RY, NST, false>(g, pr, tile, split, tng_smem);
}

// Fold of a group: dW_p[n sn + k sk] += sum_s slab_p[s][n Kk + k], db_p[n] += sum_s slab_p[s][Nn Kk + n]; fixed summation order.
struct TngFoldProb { const float* slab; float* dW; float* db; int nk, per, Kk, sn, sk, pad; };
struct TngFoldP { int nprob, splits; TngFoldProb p[TNG_MAXP]; };

__global__ __launch_bounds__(256) void wgrad_group_fold_kernel(const TngFoldP f) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(4))) const char* kchar_p;
  typedef __attribute__((address_space(4))) const TngFoldProb* kprob_p;
  const TngFoldProb pr = ((kprob_p)((kchar_p)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TngFoldP, p)))[blockIdx.y];
#else
  const TngFoldProb pr = f.p[blockIdx.y];
#endif
  const int nv = pr.per >> 2;                            // per % 4 == 0 (checked by the launcher)
  for (int v = blockIdx.x * 256 + threadIdx.x; v < nv; v += gridDim.x * 256) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* src = reinterpret_cast<const float4*>(pr.slab) + v;
#pragma unroll 8
    for (int k = 0; k < f.splits; ++k) {
      const float4 a = src[(size_t)k * nv];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    const int e = v * 4;
    if (e < pr.nk) {                         // dW[n * sn + k * sk] (Kk % 4 == 0: the four elements share a row n)
      const int n = e / pr.Kk, k = e - n * pr.Kk;
      float* dst = pr.dW + (size_t)n * pr.sn + (size_t)k * pr.sk;
      dst[0] += s.x; dst[pr.sk] += s.y; dst[2 * pr.sk] += s.z; dst[3 * pr.sk] += s.w;
    } else if (pr.db) {
      float* dst = pr.db + (e - pr.nk);
      dst[0] += s.x; dst[1] += s.y; dst[2] += s.z; dst[3] += s.w;
    }
  }
}
