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
#include "gemm_tn2.cuh"

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

// ---------------------------------------------------------------------------------------------------------------------------------
// rsp_narrow: N = C outputs, K = H = 4C; single GRN group (batch-global sparse GRN). NWV waves per workgroup, one 16-row tile per wave.
//   MODE 0: z = gelu(h) * scale + beta (stored if p.xn), out = x + z W2^T + b2
//   MODE 1: dz = dout W2 RECOMPUTED per 32-column group (p.D rows as MFMA operand, W2^T resident), dh = (dz * scale + coef * gelu(h)) * gelu'(h)
//           (stored to p.A), dd = LayerNorm-backward(dh W1), dgamma / dbeta partials -> one slab row per workgroup
// Resident in LDS: W [NP][HN + 8] (W2 / W1^T), MODE 1 also W2^T [HN][KP2 + 8]; scale | beta-or-coef [2][HN]; bias / LayerNorm gamma [NP].
// Optional folded GRN finalisation (p.fin_sum), once per workgroup: see rsc_narrow.
// grid = GX; block = 64 NWV
// ---------------------------------------------------------------------------------------------------------------------------------
template <int KC, int MODE, int NWV, bool WG = false>
__global__ __launch_bounds__(64 * NWV) void rsp_narrow_kernel(const RsP p, int ntiles) {
  using T = bf16_t;
  constexpr int NTH = 64 * NWV, HN = 4 * KC, KSH = HN / 32, LDW = HN + RSC_PAD, VPR = HN / 8;
  constexpr int NT = (KC + 15) / 16, NP = NT * 16;
  constexpr bool PAD = NP != KC, BW = MODE == 1;
  static_assert(!WG || (BW && NWV == 4), "the fused weight gradient rides in the 4-wave backward kernel");
  // WG (round 6): pwconv1's weight gradient inside this kernel. dW1 = dh^T xn with xn = x-hat * gamma + beta is, by linearity,
  // gamma[c] * U[j][c] + beta[c] * db1[j] with U = dh^T x-hat and db1 = sum_rows dh - and dh and x-hat are both in this kernel's registers. The 64 rows
  // of a workgroup iteration go to LDS row-major (dh [64][LDG], x-hat [64][LDX]; odd multiples of 16 elements like gemm_tn2's slabs), the transposing
  // LDS read hands a lane 4 consecutive rows of one column, and wave w accumulates the U tiles of hidden-column tiles w, w + 4, ... for all C over ALL
  // tiles of the persistent workgroup (9 + 3 accumulator tiles at C = 40); one slab row [H * C | H] per workgroup in p.wg_ws, folded by wg_fold_kernel (rst.cuh).
  // dh is then never written to HBM (100 MB per stage-0 block), the transpose-read GEMM over dh and xn and its fold leave the weight-gradient lane,
  // and the forward need not store xn.
  constexpr int NJT = HN / 16, JU = (NJT + 3) / 4, LDG = tn2_ld(HN), LDXH = tn2_ld(NP);
  constexpr int KS2 = (KC + 31) / 32, KP2 = KS2 * 32, LDW2 = KP2 + RSC_PAD, VPR2 = KP2 / 8;
  static_assert(KC % 8 == 0 && HN % 32 == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char rsc_smem[];
  bf16_t* Wc = reinterpret_cast<bf16_t*>(rsc_smem);                                        // [NP][LDW]
  bf16_t* W2c = Wc + (size_t)NP * LDW;                                                      // [HN][LDW2] (MODE 1)
  float* vec = reinterpret_cast<float*>(rsc_smem + ((size_t)NP * LDW + (BW ? (size_t)HN * LDW2 : 0)) * sizeof(bf16_t));   // [2][HN]
  float* cv = vec + 2 * HN;                                                                  // [NP]: b2 (MODE 0) / LayerNorm gamma (MODE 1), zero beyond KC
  float* fsh = cv + NP;                                                                      // [8] block-reduction scratch
  float* red = fsh + 8;                                                                      // [NWV][2][NP] (MODE 1)
  bf16_t* Gs = reinterpret_cast<bf16_t*>(red + NWV * 2 * NP);                                // WG: dh rows [16 NWV][LDG]
  bf16_t* Xh = Gs + 16 * NWV * LDG;                                                          // WG: x-hat rows [16 NWV][LDXH] (columns KC.. stay zero)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;

  // ---- per-tile operands (registers), requested one tile ahead
  uint4 hraw[KSH], draw[BW ? KS2 : 1];
  uint2 xraw[NT];                        // MODE 0: residual x; MODE 1: x-hat (row lr, columns j * 16 + lg * 4 ..)
  float rsl = 0.f;
  uint8_t abl = 1;
  auto request = [&](int tile) {
    const int rowc = min(tile * (16 * NWV) + wave * 16 + lr, p.M - 1);
    abl = *(p.act ? p.act + rowc : reinterpret_cast<const uint8_t*>(p.W));
    const bf16_t* hp = BW ? p.A2 : p.A;
#pragma unroll
    for (int s = 0; s < KSH; ++s) hraw[s] = *reinterpret_cast<const uint4*>(hp + (size_t)rowc * HN + s * 32 + lg * 8);
    if (BW) {
#pragma unroll
      for (int s2 = 0; s2 < KS2; ++s2) draw[s2] = *reinterpret_cast<const uint4*>(p.D + (size_t)rowc * KC + min(s2 * 32 + lg * 8, KC - 8));
      rsl = p.rstd[rowc];
    }
    const bf16_t* xp = BW ? p.xhat : (p.R ? p.R : p.W);
    const size_t xo = (BW || p.R) ? (size_t)rowc * KC : 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) xraw[j] = *reinterpret_cast<const uint2*>(xp + xo + min(j * 16 + lg * 4, KC - 4));
  };
  request(blockIdx.x);

  // ---- resident operands
  for (int v = tid; v < NP * VPR; v += NTH) {
    const int n = v / VPR, k = (v - n * VPR) * 8;
    *reinterpret_cast<uint4*>(Wc + n * LDW + k) = and4(*reinterpret_cast<const uint4*>(p.W + (size_t)min(n, KC - 1) * p.ldw + k), !PAD || n < KC);
  }
  if (BW) {
    for (int v = tid; v < HN * VPR2; v += NTH) {
      const int n = v / VPR2, k = (v - n * VPR2) * 8;
      *reinterpret_cast<uint4*>(W2c + n * LDW2 + k) = and4(*reinterpret_cast<const uint4*>(p.W2 + (size_t)n * p.ldw2 + min(k, KC - 8)), k < KC);
    }
  }
  for (int i = tid; i < NP; i += NTH) {
    const float* src = BW ? p.lng : (p.bias ? p.bias : p.v1);
    const float v = src[min(i, KC - 1)];
    cv[i] = (i < KC && (BW || p.bias)) ? v : 0.f;
  }
  if (!p.fin_sum) {
    for (int i = tid; i < HN / 4; i += NTH) {
      reinterpret_cast<float4*>(vec)[i] = reinterpret_cast<const float4*>(p.v0)[i];
      reinterpret_cast<float4*>(vec + HN)[i] = reinterpret_cast<const float4*>(p.v1)[i];
    }
  } else if (!BW) {                                 // grn_fwd_finalize_kernel (rows.cuh), same summation order
    constexpr int NJ = (HN + NTH - 1) / NTH;
    float fs[NJ], fg[NJ], fb[NJ];
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
      const int jc = min(tid + NTH * u, HN - 1);
      fs[u] = p.fin_sum[jc]; fg[u] = p.fin_gamma[jc]; fb[u] = p.v1[jc];
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NJ; ++u) { fs[u] = sqrtf(fs[u]); s += (tid + NTH * u < HN) ? fs[u] : 0.f; }
    s = wave_sum(s);
    if (lane == 0) fsh[wave] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) tot += fsh[w];
    const float ainv = 1.f / (tot / HN + p.fin_eps);
    const bool pub = blockIdx.x == 0;
    if (pub && tid == 0) p.fin_ainv[0] = ainv;
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
      const int j = tid + NTH * u;
      if (j < HN) {
        const float gx = fs[u], sc = 1.f + fg[u] * (gx * ainv);
        vec[j] = sc;
        vec[HN + j] = fb[u];
        if (pub) { p.fin_gx[j] = gx; p.fin_out[j] = sc; }
      }
    }
  } else {                                          // grn_bwd_finalize_kernel
    constexpr int NJ = (HN + NTH - 1) / NTH;
    float fs[NJ], fg[NJ], fx[NJ], f0[NJ], fv[NJ];
    const float* s0p = p.fin_sum0 ? p.fin_sum0 : p.fin_sum;
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
      const int jc = min(tid + NTH * u, HN - 1);
      fs[u] = p.fin_sum[jc]; fg[u] = p.fin_gamma[jc]; fx[u] = p.fin_gx[jc]; f0[u] = s0p[jc]; fv[u] = p.v0[jc];
    }
    const float ainv = p.fin_ainv[0];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NJ; ++u) s += (tid + NTH * u < HN) ? fg[u] * fs[u] * fx[u] : 0.f;
    s = wave_sum(s);
    if (lane == 0) fsh[wave] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) tot += fsh[w];
    const float T2 = tot * ainv * ainv / HN;
    const bool pub = blockIdx.x == 0;
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
      const int j = tid + NTH * u;
      if (j < HN) {
        const float gx = fx[u], s1 = fs[u];
        const float dGx = fg[u] * s1 * ainv - T2;
        const float cf = (gx > 0.f) ? dGx / gx : 0.f;
        vec[j] = fv[u];
        vec[HN + j] = cf;
        if (pub) {
          if (p.fin_out) p.fin_out[j] = cf;
          atomicAdd(p.fin_dgamma + j, gx * ainv * s1);
          atomicAdd(p.fin_dbeta + j, f0[u]);
        }
      }
    }
  }
  __syncthreads();

  f32x4_t accu[WG ? JU : 1][WG ? NT : 1], accd[WG ? JU : 1];      // WG: U tiles (hidden tile wave + 4 u, channel tile i) and the db1 tile of each
  if (WG) {
#pragma unroll
    for (int u = 0; u < JU; ++u) {
      accd[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NT; ++i) accu[u][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    if (PAD) {
      for (int i = tid; i < 16 * NWV * (NP - KC) / 4; i += NTH) {
        const int r = i / ((NP - KC) / 4), c = KC + (i - r * ((NP - KC) / 4)) * 4;
        *reinterpret_cast<uint2*>(Xh + r * LDXH + c) = make_uint2(0u, 0u);
      }
    }
  }
  float cg[BW ? NT : 1][4][2];                     // MODE 1: this lane's dgamma / dbeta partials over every tile of the workgroup
#pragma unroll
  for (int j = 0; j < (BW ? NT : 1); ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { cg[j][r][0] = 0.f; cg[j][r][1] = 0.f; }

#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    asm volatile("" ::: "memory");                   // (no loop-invariant hoisting of the LDS vector reads)
    const int row = tile * (16 * NWV) + wave * 16 + lr;
    const bool inb = row < p.M;
    const bool live = inb && (p.act ? abl != 0 : true);
    // this tile's operands out of the request registers (the next request overwrites them)
    uint4 hc[KSH];
    bf16x8_t df[BW ? KS2 : 1];
    uint2 xc[NT];
    const float rs = (BW && inb) ? rsl : 0.f;
#pragma unroll
    for (int s = 0; s < KSH; ++s) hc[s] = and4(hraw[s], inb);
    if (BW) {
#pragma unroll
      for (int s2 = 0; s2 < KS2; ++s2) df[s2] = __builtin_bit_cast(bf16x8_t, and4(draw[s2], inb && s2 * 32 + lg * 8 < KC));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) xc[j] = and2(xraw[j], inb && (!PAD || j * 16 + lg * 4 < KC) && (BW || p.R != nullptr));
    if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);       // the next tile's operands travel under this tile's arithmetic
    if (WG) {
      __syncthreads();                               // every wave has finished the previous iteration's transposed reads
#pragma unroll
      for (int j = 0; j < NT; ++j)
        if (!PAD || j * 16 + lg * 4 < KC) *reinterpret_cast<uint2*>(Xh + (wave * 16 + lr) * LDXH + j * 16 + lg * 4) = xc[j];
    }

    f32x4_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KSH; ++s) {
      asm volatile("" ::: "memory");                 // (LDS reads of one k-step at a time)
      const int k = s * 32 + lg * 8;
      const float4 sa = *reinterpret_cast<const float4*>(vec + k), sb = *reinterpret_cast<const float4*>(vec + k + 4);
      const float4 ta = *reinterpret_cast<const float4*>(vec + HN + k), tb = *reinterpret_cast<const float4*>(vec + HN + k + 4);
      const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
      const float tc[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
      float z[8];
      if (!BW) {
        float a[8], ga[8];
        unpack8(hc[s], a);
        gelu_n<T, 8>(a, ga);
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = live ? ga[e] * sc[e] + tc[e] : 0.f;                 // GRN(gelu(h))
      } else {
        // dz [row lr][s*32 + lg*8 + e], e = t*4 + r, from the tile pair (t = 0, 1) of this 32-column group
        f32x4_t d2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          d2[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s2 = 0; s2 < KS2; ++s2) {
            const bf16x8_t wf2 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(
                W2c + (s * 32 + (lr >> 2) * 8 + t * 4 + (lr & 3)) * LDW2 + s2 * 32 + lg * 8));
            d2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2, df[s2], d2[t], 0, 0, 0);
          }
        }
        float a[8], h[8], gl[8], dg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = bf2f(f2bf(d2[e >> 2][e & 3]));                    // the value the unfused path stored
        unpack8(hc[s], h);
        gelu_both_n<T, 8>(h, gl, dg);
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (a[e] * sc[e] + tc[e] * gl[e]) * dg[e];                 // dh
      }
      const bf16x8_t af = pack_bf16x8(z);
      if (WG) {
        *reinterpret_cast<uint4*>(Gs + (wave * 16 + lr) * LDG + k) = __builtin_bit_cast(uint4, af);      // dh stays on the CU (rows beyond M are zero: h = 0)
      } else if (inb) {
        bf16_t* dst = BW ? const_cast<bf16_t*>(p.A) : p.xn;
        if (dst) *reinterpret_cast<uint4*>(dst + (size_t)row * HN + k) = __builtin_bit_cast(uint4, af);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Wc + (j * 16 + lr) * LDW + s * 32 + lg * 8));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[j], 0, 0, 0);
      }
    }

    // ---- epilogue: lane holds row m = lr, columns n = j*16 + lg*4 + r
    if (!BW) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n4 = j * 16 + lg * 4;
        const bool nin = !PAD || n4 < KC;
        const float4 b4 = *reinterpret_cast<const float4*>(cv + n4);
        float x[4], o[4];
        unpack4(xc[j], x);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = live ? acc[j][r] + bb[r] + x[r] : 0.f;
        if (inb && nin) *reinterpret_cast<uint2*>(p.out + (size_t)row * KC + n4) = pack_bf16x4(o);
      }
    } else {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n4 = j * 16 + lg * 4;
        float xh[4];
        unpack4(xc[j], xh);
        const float4 g = *reinterpret_cast<const float4*>(cv + n4);
        const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dxn = live ? bf2f(f2bf(acc[j][r])) : 0.f;      // bf16 like the unfused path
          cg[BW ? j : 0][r][0] += dxn * xh[r];
          cg[BW ? j : 0][r][1] += dxn;
          const float gq = dxn * gg[r];
          acc[j][r] = gq;
          s1 += gq;
          s2 += gq * xh[r];
        }
      }
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      s1 /= KC; s2 /= KC;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n4 = j * 16 + lg * 4;
        float xh[4], o[4];
        unpack4(xc[j], xh);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = live ? rs * (acc[j][r] - s1 - xh[r] * s2) : 0.f;
        if (inb && (!PAD || n4 < KC)) *reinterpret_cast<uint2*>(p.out + (size_t)row * KC + n4) = pack_bf16x4(o);
      }
    }
    if (WG) {
      __syncthreads();                               // the workgroup's 64 rows of dh and x-hat are in LDS
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
      const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);
      const int goff = (lg * 4 + (lr >> 2)) * LDG + 4 * (lr & 3), xoff = (lg * 4 + (lr >> 2)) * LDXH + 4 * (lr & 3);
#pragma unroll
      for (int ks = 0; ks < (16 * NWV) / 32; ++ks) {
        bf16x8_t xf[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) xf[i] = tn2_frag(Xh + ks * 32 * LDXH + xoff + i * 16, 16 * LDXH);
#pragma unroll
        for (int u = 0; u < JU; ++u) {
          const int jt = wave + 4 * u;
          if (jt < NJT) {                            // (wave-uniform)
            const bf16x8_t gf = tn2_frag(Gs + ks * 32 * LDG + goff + jt * 16, 16 * LDG);
#pragma unroll
            for (int i = 0; i < NT; ++i) accu[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, xf[i], accu[u][i], 0, 0, 0);
            accd[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, ones, accd[u], 0, 0, 0);
          }
        }
      }
    }
  }
  if (WG) {
    // one slab row per workgroup: [HN * KC | HN]; D layout: row (hidden j) = lg * 4 + r, column (channel c) = lr
    float* slab = p.wg_ws + (size_t)blockIdx.x * ((size_t)HN * KC + HN);
#pragma unroll
    for (int u = 0; u < JU; ++u) {
      const int jt = wave + 4 * u;
      if (jt < NJT) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = jt * 16 + lg * 4 + r, c = i * 16 + lr;
            if (c < KC) slab[(size_t)j * KC + c] = accu[u][i][r];
          }
        if (lr == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[(size_t)HN * KC + jt * 16 + lg * 4 + r] = accd[u][r];
        }
      }
    }
  }
  if (BW) {
    float* redw = red + (size_t)wave * 2 * NP;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ga = sum16(cg[BW ? j : 0][r][0]), gb = sum16(cg[BW ? j : 0][r][1]);
        if (lr == 0) { redw[j * 16 + lg * 4 + r] = ga; redw[NP + j * 16 + lg * 4 + r] = gb; }
      }
    __syncthreads();
    for (int i = tid; i < KC; i += NTH) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) { a += red[(size_t)w * 2 * NP + i]; b += red[(size_t)w * 2 * NP + NP + i]; }
      p.ws[((size_t)blockIdx.x * 2 + 0) * KC + i] = a;
      p.ws[((size_t)blockIdx.x * 2 + 1) * KC + i] = b;
    }
  }
}

