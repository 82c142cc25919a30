// Persistent, burst-load forms of the fused pointwise kernels for the bandwidth-shaped stages (C = 40 / 80, H = 4C): rsp_wide (this file's
// first half) and rsp_narrow (second half) compute exactly what rsc_wide / rsc_narrow compute (rsc.cuh) with a different MEMORY SCHEDULE.
//
// Why. tools/isa_chain.py on rsc_wide<40,0,4,160> / rsc_narrow<40,1,1,32,6,4> (the stage-0 kernels, 58 / 90 us for 22 / 34 us of HBM time):
// a workgroup's life is a chain of DEPENDENT global round trips - activity byte + rows -> wait -> LayerNorm -> gamma / beta -> wait, per row
// tile (RT = 4: ~14 waits before the first MFMA); a bias / h load -> wait per tile pair; a weight chunk -> wait -> barrier per K chunk; the
// epilogue's x-hat / gamma / rstd -> wait. Under load one such trip is ~1 us (phase stamps: profiles/r05/rs1_stamps_v1_chain.txt), a
// workgroup lives 10-20 of them, and 4-5 resident workgroups per CU do not hide that.
// Here
//   * the weights of the layer (12-52 KB at these widths) are staged in LDS ONCE per workgroup and stay: no chunk loop, no per-chunk barrier;
//     small vectors (LayerNorm gamma / beta, bias, GRN scale / beta / coef) sit in LDS too;
//   * a workgroup is persistent: it walks row tiles t = blockIdx.x, + gridDim.x, ...; EVERY global operand of tile t + 1 is requested in one
//     burst BEFORE the arithmetic of tile t and consumed one tile later: no dependent round trip inside the loop at all;
//   * column statistics accumulate in registers across the tiles of a workgroup and are folded (16-lane DPP sums) once at the end.
// The arithmetic - fragment layouts, transposed MFMA over interleaved tile pairs, rounding points, GELU fits - is rsc.cuh's, so that the two
// generations agree to the last bit on everything but the order of the fp32 statistics sums.
#pragma once
#include "rsc.cuh"

// ---------------------------------------------------------------------------------------------------------------------------------
// rsp_wide: N = H outputs in slices of 160 columns (grid.y = H / 160), whole K = C per row.
//   MODE 0: x-hat, rstd, xn, h = LN(d) W1^T + b1, sum gelu(h)^2          MODE 1: dz = dout W2 (optional store), (sum dz, sum dz * gelu(h))
// grid = (GX, H / 160); block = 256 (4 waves x RT row tiles of 16 rows); LDS = 160 (KP + 8) 2 + (2 KP + 160) 4 + 4 * 2 * 160 * 4
// ---------------------------------------------------------------------------------------------------------------------------------
template <int KC, int MODE, int RT>
__global__ __launch_bounds__(256) void rsp_wide_kernel(const RsP p, int ntiles) {
  using T = bf16_t;
  constexpr int HN = 4 * KC, KS = (KC + 31) / 32, KP = KS * 32, LDW = KP + RSC_PAD, VPR = KP / 8, CPS = 160, NP = CPS / 32;
  constexpr int WV = (CPS * VPR + 255) / 256;
  constexpr bool PAD = KP != KC, LN = MODE == 0, DZ = MODE == 1;
  static_assert(KC % 8 == 0 && HN % CPS == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char rsc_smem[];
  bf16_t* Wc = reinterpret_cast<bf16_t*>(rsc_smem);                                         // [CPS][LDW]
  float* vec = reinterpret_cast<float*>(rsc_smem + (size_t)CPS * LDW * sizeof(bf16_t));     // [2][KP] LayerNorm gamma | beta (zero beyond KC)
  float* bia = vec + 2 * KP;                                                                 // [CPS] bias slice
  float* red = bia + CPS;                                                                    // [4 waves][2][CPS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int n_begin = blockIdx.y * CPS;

  // ---- requests of the first tile (consumed at the top of the loop), then the resident operands
  uint4 raw[RT][KS], hraw[DZ ? RT : 1][DZ ? NP : 1];
  uint8_t abl[RT];
  auto request = [&](int tile) {          // (clamped addresses: rows beyond M re-read the last row and are masked when consumed)
    const int rb = tile * (64 * RT) + wave * (16 * RT);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int rowc = min(rb + rt * 16 + lr, p.M - 1);
      abl[rt] = *(p.act ? p.act + rowc : reinterpret_cast<const uint8_t*>(p.A));          // pointer select, not a branch
#pragma unroll
      for (int s = 0; s < KS; ++s) raw[rt][s] = *reinterpret_cast<const uint4*>(p.A + (size_t)rowc * KC + min(s * 32 + lg * 8, KC - 8));
      if (DZ) {
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) hraw[rt][jp] = *reinterpret_cast<const uint4*>(p.R + (size_t)rowc * HN + n_begin + jp * 32 + lg * 8);
      }
    }
  };
  request(blockIdx.x);
  {
    uint4 wr[WV];
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + 256 * i, vc = min(v, CPS * VPR - 1), n = vc / VPR, k = (vc - n * VPR) * 8;
      wr[i] = and4(*reinterpret_cast<const uint4*>(p.W + (size_t)(n_begin + n) * p.ldw + min(k, KC - 8)), !PAD || k < KC);
    }
    float gb = 0.f, bs = 0.f;
    if (LN && tid < 2 * KP) { const int k = tid < KP ? tid : tid - KP; gb = *((tid < KP ? p.v0 : p.v1) + min(k, KC - 1)); gb = k < KC ? gb : 0.f; }
    if (!DZ && tid < CPS) bs = *(p.bias ? p.bias + n_begin + tid : p.v0);
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + 256 * i, n = v / VPR, k = (v - n * VPR) * 8;
      if (v < CPS * VPR) *reinterpret_cast<uint4*>(Wc + n * LDW + k) = wr[i];
    }
    if (LN && tid < 2 * KP) vec[tid] = gb;
    if (!DZ && tid < CPS) bia[tid] = p.bias ? bs : 0.f;
  }
  __syncthreads();

  float csum[NP][DZ ? 2 : 1][8];                      // this lane's column partials over every tile of the workgroup
#pragma unroll
  for (int jp = 0; jp < NP; ++jp)
#pragma unroll
    for (int q = 0; q < (DZ ? 2 : 1); ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) csum[jp][q][e] = 0.f;

#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int rbase = tile * (64 * RT) + wave * (16 * RT);
    asm volatile("" ::: "memory");                     // (no loop-invariant hoisting of the LDS vector reads)
    bf16x8_t af[RT][KS];
    uint4 hcur[DZ ? RT : 1][DZ ? NP : 1];
    bool live[RT], inb[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = rbase + rt * 16 + lr;
      inb[rt] = row < p.M;
      live[rt] = inb[rt] && (p.act ? abl[rt] != 0 : true);
#pragma unroll
      for (int s = 0; s < KS; ++s) raw[rt][s] = and4(raw[rt][s], inb[rt] && (!PAD || s * 32 + lg * 8 < KC));
      if (DZ) {
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) hcur[rt][jp] = and4(hraw[rt][jp], inb[rt]);
      }
      if (LN) {
        float v[KS][8];
        float s1 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          unpack8(raw[rt][s], v[s]);
#pragma unroll
          for (int e = 0; e < 8; ++e) s1 += v[s][e];
        }
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 / KC;
        float s2 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = (!PAD || s * 32 + lg * 8 < KC) ? v[s][e] - mean : 0.f;
            s2 += d * d;
          }
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        const float rstd = rsqrtf(s2 / KC + 1e-6f);
        const bool wr_side = inb[rt] && blockIdx.y == 0;
        if (wr_side && lg == 0) p.rstd[row] = live[rt] ? rstd : 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int k = s * 32 + lg * 8;
          const bool kin = !PAD || k < KC;
          const float4 g0 = *reinterpret_cast<const float4*>(vec + k), g1 = *reinterpret_cast<const float4*>(vec + k + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(vec + KP + k), b1 = *reinterpret_cast<const float4*>(vec + KP + k + 4);
          const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
          const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          float xh[8], xn[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            xh[e] = (live[rt] && kin) ? (v[s][e] - mean) * rstd : 0.f;
            xh[e] = bf2f(f2bf(xh[e]));                          // consumers (and backward) see the stored value
            xn[e] = (live[rt] && kin) ? xh[e] * ga[e] + be[e] : 0.f;
          }
          af[rt][s] = pack_bf16x8(xn);
          if (wr_side && kin) {
            st8<T>(p.xhat + (size_t)row * KC + k, xh);
            if (p.xn) *reinterpret_cast<uint4*>(p.xn + (size_t)row * KC + k) = __builtin_bit_cast(uint4, af[rt][s]);
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) af[rt][s] = __builtin_bit_cast(bf16x8_t, raw[rt][s]);
      }
    }
    if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);       // the next tile's operands travel under this tile's arithmetic

#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      asm volatile("" ::: "memory");                   // (fragment reads of one tile pair at a time)
      const int nl = jp * 32 + lg * 8;                 // first of this lane's 8 columns, relative to n_begin
      const int n8 = n_begin + nl;
      float bias[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bias[e] = 0.f;
      if (!DZ) {
        const float4 b0 = *reinterpret_cast<const float4*>(bia + nl), b1 = *reinterpret_cast<const float4*>(bia + nl + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
      }
      bf16x8_t wf[2][KS];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s)
          wf[t][s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(
              Wc + (jp * 32 + (lr >> 2) * 8 + t * 4 + (lr & 3)) * LDW + s * 32 + lg * 8));
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        f32x4_t acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KS; ++s) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][s], af[rt][s], acc[t], 0, 0, 0);
        }
        const int row = rbase + rt * 16 + lr;
        float o[8];
        if (!DZ) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = live[rt] ? acc[e >> 2][e & 3] + bias[e] : 0.f;
          float gl[8];                      // o is rounded to bf16 by the store; the GRN sums use the fp32 value
          gelu_n<T, 8>(o, gl);
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[jp][0][e] += gl[e] * gl[e];
        } else {
          float hv[8], gh[8];
          unpack8(hcur[rt][jp], hv);
          gelu_n<T, 8>(hv, gh);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o[e] = acc[e >> 2][e & 3];
            csum[jp][0][e] += o[e];
            csum[jp][DZ ? 1 : 0][e] += o[e] * gh[e];
          }
        }
        if (inb[rt] && p.out) st8<T>(p.out + (size_t)row * HN + n8, o);      // without `out`: statistics (and x-hat / xn) only
      }
    }
  }
  // ---- column statistics: 16-lane folds, one row per wave, the four rows added in a fixed order
  float* redw = red + (size_t)wave * 2 * CPS;
#pragma unroll
  for (int jp = 0; jp < NP; ++jp)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = sum16(csum[jp][0][e]);
      if (lr == 0) redw[jp * 32 + lg * 8 + e] = a;
      if (DZ) {
        const float b = sum16(csum[jp][DZ ? 1 : 0][e]);
        if (lr == 0) redw[CPS + jp * 32 + lg * 8 + e] = b;
      }
    }
  __syncthreads();
  constexpr int w2 = 2 * CPS;
  for (int i = tid; i < CPS; i += 256) {
    const float r0 = ((red[i] + red[w2 + i]) + red[2 * w2 + i]) + red[3 * w2 + i];
    if (!DZ) p.ws[(size_t)blockIdx.x * HN + n_begin + i] = r0;
    else {
      const float r1 = ((red[CPS + i] + red[w2 + CPS + i]) + red[2 * w2 + CPS + i]) + red[3 * w2 + CPS + i];
      p.ws[(size_t)blockIdx.x * 2 * HN + n_begin + i] = r0;
      p.ws[(size_t)blockIdx.x * 2 * HN + HN + n_begin + i] = r1;
    }
  }
}
