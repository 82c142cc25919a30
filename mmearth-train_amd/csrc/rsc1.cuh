// One-shot form of the WIDE fused pointwise kernels (rsc_wide, rsc.cuh) for the compute-shaped stages (C = 160 / 320).
//
// rsc_wide streams its slice of the weight matrix through LDS in 32-column chunks: global -> staging VGPRs -> ds_write -> barrier per
// chunk, one chunk of prefetch. At C = 320 a workgroup is a chain of 5 such round trips behind its prologue and the kernel takes 30-40 us
// for 4 GFLOP and 87 MB of L2 -> CU traffic (op_table_standalone.txt, round 4). Here the WHOLE slice ([cps][KC] bf16, 40-50 KB) goes
// global -> LDS by DMA (global_load_lds_dwordx4: no staging registers) at the top of the kernel, next to the activation rows' own loads:
// ONE memory round trip, ONE barrier, then every tile pair of the slice is computed without further synchronisation. Two or three
// workgroups per CU cover each other's round trip.
//   * The LDS image of a DMA is lane-linear (wave-uniform base + 16 B x lane): rows cannot be padded. Fragment reads are made conflict-free by
//     an XOR swizzle of the 16-byte chunk index, applied to the per-lane SOURCE address and again on the read. A lane reads chunk q = 4 s + lg
//     of weight row R = 32 jp + 8 (lr >> 2) + 4 t + (lr & 3); ds_read_b128 is served in the lane groups of the microarchitecture guide
//     ({0-3, 12-15, 20-27}, ...), i.e. per group a = lr >> 2 in {0, 3} with lg even-side and a in {1, 2} with lg odd-side, all b = lr & 3.
//       KC = 320 (40 chunks per row = 8 mod 16): slot = 8 (R & 1) + q'  ->  q' = q ^ (((R >> 1) & 1) | (a << 1))   (f < 8, 40 = 5 x 8)
//       KC = 160 (20 chunks per row = 4 mod 16): slot = 4 (R & 3) + q'  ->  q' = q ^ {0, 3, 2, 1}[a]               (f < 4, 20 = 5 x 4)
//     (16 distinct 16-byte slots mod 256 B in every lane group; checked with tools/lds_bank_model.py --rsc1.)
//   * everything else - activation rows in registers in MFMA fragment layout, transposed MFMA over interleaved tile pairs, lane-local
//     bias / GELU / statistics, per-wave statistic rows added in a fixed order - is rsc_wide's.
// MODE 0: x-hat, rstd, xn, h = LN(d) W1^T + b1, sum gelu(h)^2;  MODE 1: dz = dout W2, (sum dz, sum dz * gelu(h)).
#pragma once
#include "rsc.cuh"
typedef const __attribute__((address_space(1))) void* rsc1_gptr_t;
typedef __attribute__((address_space(3))) void* rsc1_lptr_t;

#ifdef RSC1_STAMPS      // tools/probes/rs1_stamps.hip: phase stamps (s_memtime) + wall clock (100 MHz) of every wave
__device__ unsigned long long rsc1_stamp_buf[4096 * 4 * 8];
#define RSC1_ST(k) do { if (lane == 0) { const size_t wg_ = blockIdx.x + (size_t)gridDim.x * blockIdx.y; if (wg_ < 4096) rsc1_stamp_buf[(wg_ * 4 + wave) * 8 + (k)] = ((k) >= 6) ? wall_clock64() : __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define RSC1_ST(k) do { } while (0)
#endif

//       KC = 384 (48 chunks per row = 0 mod 16): slot = q'               ->  q' = q ^ ((R & 3) | (a << 2))              (f < 16, 48 = 3 x 16)
//       (KC = 192: 24 chunks = 8 mod 16, the KC = 320 rule, 24 = 3 x 8)
template <int KC> __device__ __forceinline__ int rsc1_swz(int row) {
  constexpr int CM = (KC / 8) % 16;
  static_assert(CM == 8 || CM == 4 || CM == 0, "swizzle table");
  const int a = (row >> 3) & 3;
  if (CM == 8) return ((row >> 1) & 1) | (a << 1);
  if (CM == 0) return (row & 3) | (a << 2);
  return (4 - a) & 3;                              // {0, 3, 2, 1}
}

// grid = (GX, HN / CPS); block = 256; LDS = CPS * KC * 2 + 4 * 2 * CPS * 4 + 2 * KC * 4. Workgroup (gx, y) keeps weight slice y for its whole
// life and walks the row tiles gx, gx + GX, ... (64 RT rows each): the slice is fetched once per workgroup, not once per tile.
// Every global operand is requested in ONE burst at the top - LayerNorm gamma / beta (into LDS: 2 KC floats, one float4 per thread), the
// first tile's activation rows (registers, MFMA fragment layout) and h rows (MODE 1, all tile pairs), the weight slice (DMA) - and awaited
// once; the NEXT tile's rows are requested before the current tile's products. Phase stamps of the first version
// (tools/probes/rs1_stamps.hip, profiles/r05/rs1_stamps.txt): a workgroup lived 11 us of which 8.5 were the LayerNorm prologue - the
// gamma / beta vectors were fetched inside the per-k-step loop, one dependent L2 round trip (~1 us under load) per step and row tile - and
// MODE 1 paid one round trip per tile pair for its h operand; with one burst the wait is 3 us and the LayerNorm arithmetic itself
// (redone by every column slice: 20 x at C = 320) 3.4 us with three workgroups sharing a CU's VALUs.
// MODE 2: MODE 0 without the LayerNorm - A is xn as stored by the producer (dwln.cuh), h = xn W1^T + b1, sum gelu(h)^2.
template <int KC, int MODE, int RT, int CPS, bool PFA = (KC <= 192)>
__global__ __launch_bounds__(256, (RT == 1 && !(MODE == 0 && KC >= 320)) ? 3 : 2) void rsc_wide1_kernel(const RsP p, int ntiles) {
  using T = bf16_t;
  constexpr int HN = 4 * KC, KS = KC / 32, CPR = KC / 8, NP = CPS / 32, NINST = CPS * CPR / 64, cps = CPS;
  static_assert(KC == 160 || KC == 320 || KC == 192 || KC == 384, "swizzle table");
  static_assert(CPS % 32 == 0 && HN % CPS == 0 && NINST % 4 == 0, "slice shape");
  static_assert(2 * KC / 4 <= 256, "one float4 of gamma / beta per thread");
  constexpr bool LN = MODE == 0, DZ = MODE == 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char rsc_smem[];
  bf16_t* Wc = reinterpret_cast<bf16_t*>(rsc_smem);                                   // [CPS][KC], lane-linear, chunk-swizzled
  float* red = reinterpret_cast<float*>(rsc_smem + (size_t)CPS * KC * sizeof(bf16_t)); // [4 waves][2][CPS]
  float* vec = red + 4 * 2 * CPS;                                                      // [2][KC]: LayerNorm gamma | beta (MODE 0)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int n_begin = blockIdx.y * CPS;
  float* redw = red + (size_t)wave * 2 * CPS;
  RSC1_ST(6); RSC1_ST(0);

  float4 gb4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (LN) {
    const int j = min(tid, 2 * KC / 4 - 1);
    gb4 = *reinterpret_cast<const float4*>((j < KC / 4 ? p.v0 : p.v1 - KC) + 4 * j);
  }
  uint4 raw[RT][KS], hraw[DZ ? NP : 1][RT];
  uint8_t abl[RT];
  auto request = [&](int tile) {          // rows of tile `tile` (clamped addresses: a tile beyond the last re-reads the last rows and is never used)
    const int rb = tile * (64 * RT) + wave * (16 * RT);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int rowc = min(rb + rt * 16 + lr, p.M - 1);
      abl[rt] = *(p.act ? p.act + rowc : reinterpret_cast<const uint8_t*>(p.A));      // pointer select, not a branch
#pragma unroll
      for (int s = 0; s < KS; ++s) raw[rt][s] = *reinterpret_cast<const uint4*>(p.A + (size_t)rowc * KC + s * 32 + lg * 8);
      if (DZ) {
#pragma unroll
        for (int jp = 0; jp < NP; ++jp)
          hraw[jp][rt] = *reinterpret_cast<const uint4*>(p.R + (size_t)rowc * HN + n_begin + jp * 32 + lg * 8);
      }
    }
  };
  request(blockIdx.x);
  {
    const bf16_t* Wb = p.W + (size_t)n_begin * p.ldw;
#pragma unroll
    for (int ii = 0; ii < NINST / 4; ++ii) {
      const int i = ii * 4 + wave;
      const int sl = i * 64 + lane, row = sl / CPR, chs = sl - row * CPR;
      const bf16_t* src = Wb + (size_t)row * p.ldw + ((chs ^ rsc1_swz<KC>(row)) << 3);
      bf16_t* dst = Wc + (size_t)i * 64 * 8;                                         // wave-uniform; lane l lands at + 8 l
      __builtin_amdgcn_global_load_lds((rsc1_gptr_t)src, (rsc1_lptr_t)dst, 16, 0, 0);
    }
  }
  if (LN && tid < 2 * KC / 4) *reinterpret_cast<float4*>(vec + 4 * tid) = gb4;
  for (int i = lane; i < 2 * CPS; i += 64) redw[i] = 0.f;
  __syncthreads();                                   // (carries the s_waitcnt vmcnt(0) that retires the DMA and every load above)
  RSC1_ST(1);

  const int browl = (lr >> 2) * 8 + (lr & 3);        // weight row of this lane inside a tile pair: + 32 jp + 4 t
  const int fsw = rsc1_swz<KC>(browl);               // independent of jp and t (bits 2 and 5.. are not used)
  float csum[NP][DZ ? 2 : 1][8];                      // this lane's column partials over all tiles (folded over the 16 rows at the end)
#pragma unroll
  for (int jp = 0; jp < NP; ++jp)
#pragma unroll
    for (int q = 0; q < (DZ ? 2 : 1); ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) csum[jp][q][e] = 0.f;

#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int rbase = tile * (64 * RT) + wave * (16 * RT);
    asm volatile("" ::: "memory");                     // (no loop-invariant hoisting of the gamma / beta / bias reads: 160 VGPRs)
    // ---- activation fragments (whole K extent) of this wave's RT row tiles, from the rows requested one tile ago
    bf16x8_t af[RT][KS];
    uint4 hcur[DZ ? NP : 1][RT];
    bool live[RT], inb[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = rbase + rt * 16 + lr;
      inb[rt] = row < p.M;
      live[rt] = inb[rt] && (p.act ? abl[rt] != 0 : true);
#pragma unroll
      for (int s = 0; s < KS; ++s) raw[rt][s] = and4(raw[rt][s], inb[rt]);
      if (DZ) {
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) hcur[jp][rt] = and4(hraw[jp][rt], inb[rt]);
      }
      if (LN) {
        float s1 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          float v[8];
          unpack8(raw[rt][s], v);
#pragma unroll
          for (int e = 0; e < 8; ++e) s1 += v[e];
        }
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 / KC;
        float s2 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          float v[8];
          unpack8(raw[rt][s], v);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; s2 += d * d; }
        }
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        const float rstd = rsqrtf(s2 / KC + 1e-6f);
        const bool wr_side = inb[rt] && blockIdx.y == 0;
        if (wr_side && lg == 0) p.rstd[row] = live[rt] ? rstd : 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int k = s * 32 + lg * 8;
          if (s & 1) asm volatile("" ::: "memory");            // (the gamma / beta reads of at most two k-steps in flight)
          const float4 g0 = *reinterpret_cast<const float4*>(vec + k), g1 = *reinterpret_cast<const float4*>(vec + k + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(vec + KC + k), b1 = *reinterpret_cast<const float4*>(vec + KC + k + 4);
          const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
          const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          float v[8], xh[8], xn[8];
          unpack8(raw[rt][s], v);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            xh[e] = live[rt] ? (v[e] - mean) * rstd : 0.f;
            xh[e] = bf2f(f2bf(xh[e]));                          // consumers (and backward) see the stored value
            xn[e] = live[rt] ? xh[e] * ga[e] + be[e] : 0.f;
          }
          af[rt][s] = pack_bf16x8(xn);
          if (wr_side) {
            st8<T>(p.xhat + (size_t)row * KC + k, xh);
            if (p.xn) *reinterpret_cast<uint4*>(p.xn + (size_t)row * KC + k) = __builtin_bit_cast(uint4, af[rt][s]);
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) af[rt][s] = __builtin_bit_cast(bf16x8_t, raw[rt][s]);
      }
    }
    RSC1_ST(2);
    if (PFA && tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);       // the next tile's rows travel under this tile's products (PFA: costs a second register set)

#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      asm volatile("" ::: "memory");                   // (hipcc would hoist the fragment reads of every tile pair to the top: +90 VGPRs)
      const int nl = jp * 32 + lg * 8;                 // first of this lane's 8 columns, relative to n_begin
      const int n8 = n_begin + nl;
      float bias[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bias[e] = 0.f;
      if (!DZ && p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n8), b1 = *reinterpret_cast<const float4*>(p.bias + n8 + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
      }
      f32x4_t acc[RT][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16_t* wrow = Wc + (size_t)(jp * 32 + browl + t * 4) * KC;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wrow + (((s * 4 + lg) ^ fsw) << 3)));
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[rt][s], acc[rt][t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = rbase + rt * 16 + lr;
        float o[8];
        if (!DZ) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = live[rt] ? acc[rt][e >> 2][e & 3] + bias[e] : 0.f;
          float gl[8];                      // o is rounded to bf16 by the store; the GRN sums use the fp32 value
          gelu_n<T, 8>(o, gl);
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[jp][0][e] += gl[e] * gl[e];
        } else {
          float hv[8], gh[8];
          unpack8(hcur[jp][rt], hv);
          gelu_n<T, 8>(hv, gh);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o[e] = acc[rt][e >> 2][e & 3];
            csum[jp][0][e] += o[e];
            csum[jp][DZ ? 1 : 0][e] += o[e] * gh[e];
          }
        }
        if (inb[rt] && p.out) st8<T>(p.out + (size_t)row * HN + n8, o);
      }
    }
    RSC1_ST(3);
    if (!PFA && tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);
  }
  // ---- column statistics: 16-lane folds, one row per wave, the four rows added in a fixed order
#pragma unroll
  for (int jp = 0; jp < NP; ++jp)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = sum16(csum[jp][0][e]);
      if (lr == 0) redw[jp * 32 + lg * 8 + e] = a;
      if (DZ) {
        const float b = sum16(csum[jp][DZ ? 1 : 0][e]);
        if (lr == 0) redw[cps + jp * 32 + lg * 8 + e] = b;
      }
    }
  __syncthreads();
  const int w2 = 2 * cps;
  RSC1_ST(4);
  if (p.s0a) {      // few row blocks: straight into the zero-initialised accumulators of the step (no slab, no fold launch)
    for (int i = tid; i < cps; i += 256) {
      (void)unsafeAtomicAdd(p.s0a + n_begin + i, ((red[i] + red[w2 + i]) + red[2 * w2 + i]) + red[3 * w2 + i]);
      if (DZ) (void)unsafeAtomicAdd(p.s1a + n_begin + i, ((red[cps + i] + red[w2 + cps + i]) + red[2 * w2 + cps + i]) + red[3 * w2 + cps + i]);
    }
    return;
  }
  for (int i = tid; i < cps; i += 256) {
    const float r0 = ((red[i] + red[w2 + i]) + red[2 * w2 + i]) + red[3 * w2 + i];
    if (!DZ) p.ws[(size_t)blockIdx.x * HN + n_begin + i] = r0;
    else {
      const float r1 = ((red[cps + i] + red[w2 + cps + i]) + red[2 * w2 + cps + i]) + red[3 * w2 + cps + i];
      p.ws[(size_t)blockIdx.x * 2 * HN + n_begin + i] = r0;
      p.ws[(size_t)blockIdx.x * 2 * HN + HN + n_begin + i] = r1;
    }
  }
  RSC1_ST(5); RSC1_ST(7);
}
