// Row-streaming fused pointwise kernels for the bandwidth-bound stages (C = 40 / 80 / 96, H = 4C).
//
// The generic tiled GEMMs spend their time in LDS staging, barriers and a heavy shared epilogue
// when K and N are this small; rocprof shows them latency-bound at 2 workgroups per CU. Here a
// 64-lane wave owns 16 rows at a time and keeps everything in registers:
//   * the A operand is loaded from global memory DIRECTLY in MFMA fragment layout (lane (lr, lg)
//     reads 16 B at row lr, columns ks*32 + lg*8 ..), so LayerNorm / GRN prologues are lane-local
//     plus two cross-lane adds, and no LDS round trip or barrier is needed for A;
//   * the (small) weight matrix lives in LDS once per workgroup; waves stride over row groups
//     (persistent grid), so it is staged once per ~10^3 rows;
//   * results leave through a per-wave LDS transpose tile so that global stores are 16 B per lane
//     over whole contiguous rows; column statistics accumulate in registers across row groups
//     and leave as one slab row per workgroup.
// rs_wide  : N = H outputs.  MODE 0: x-hat/rstd + h = LN(d) W1^T + b1 + sum gelu(h)^2   (LN + pw1)
//                            MODE 1: dz = dout W2 + (sum dz, sum dz*gelu(h))            (pw2 dgrad)
// rs_narrow: N = C outputs.  MODE 0: out = x + GRN(gelu(h)) W2^T + b2                  (pw2)
//                            MODE 1: dd = LNbwd( dh W1 ), dh from (dz, h)              (pw1 dgrad + LN bwd)
#pragma once
#include "gemm.cuh"

struct RsP {
  const bf16_t* A;      // wide: d (MODE 0) / dout (MODE 1) [M,KC];  narrow: h (MODE 0) / dz (MODE 1) [M,HN]
  const bf16_t* A2;     // narrow MODE 1: h [M,HN]
  const bf16_t* W;      // staged weights [N][K] bf16, row stride ldw
  int ldw;
  const float* bias;    // [N] or nullptr
  const float* v0;      // wide MODE 0: ln gamma[KC];  narrow MODE 0: scale[HN];  narrow MODE 1: scale[HN]
  const float* v1;      // wide MODE 0: ln beta[KC];   narrow MODE 0: grn beta[HN]; narrow MODE 1: coef[HN]
  bf16_t* out;          // wide: h / dz [M,HN]; narrow: out / dd [M,KC]
  bf16_t* xhat;         // wide MODE 0: x-hat out [M,KC]; narrow MODE 1: x-hat in [M,KC]
  bf16_t* xn;           // wide MODE 0: optional LN output x-hat*gamma+beta [M,KC] (operand of the pw1 weight gradient)
  float* rstd;          // wide MODE 0: out [M]; narrow MODE 1: in [M]
  const bf16_t* R;      // wide MODE 1: h [M,HN] (statistics); narrow MODE 0: residual x [M,KC]
  const float* lng;     // narrow MODE 1: ln gamma [KC]
  float* ws;            // slab rows: wide: [gridDim.x][HN] (+ second stat); narrow MODE 1: [gridDim.x][2][KC]
  const uint8_t* act;   // row activity or nullptr
  int M;
  // rsc_narrow with LDS-staged vectors: optional folded GRN finalisation (see MpmaeRsArgs.fin_*)
  const float* fin_sum; const float* fin_sum0; const float* fin_gamma;
  float* fin_gx; float* fin_ainv; float* fin_out; float* fin_dgamma; float* fin_dbeta;
  float fin_eps;
  const bf16_t* D; const bf16_t* W2; int ldw2;     // rsc_narrow with operand recomputation: dout / xn [M][C], W2^T / W1 [H][ldw2]
  const float* hb;                                 // MODE 0: pwconv1 bias [H]
};

__device__ __forceinline__ bf16x8_t pack_bf16x8(const float (&v)[8]) {
  uint4 u;
  u.x = f2bf2(v[0], v[1]);
  u.y = f2bf2(v[2], v[3]);
  u.z = f2bf2(v[4], v[5]);
  u.w = f2bf2(v[6], v[7]);
  return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ void unpack8(const uint4& a, float (&o)[8]) {
  o[0] = __uint_as_float(a.x << 16); o[1] = __uint_as_float(a.x & 0xffff0000u);
  o[2] = __uint_as_float(a.y << 16); o[3] = __uint_as_float(a.y & 0xffff0000u);
  o[4] = __uint_as_float(a.z << 16); o[5] = __uint_as_float(a.z & 0xffff0000u);
  o[6] = __uint_as_float(a.w << 16); o[7] = __uint_as_float(a.w & 0xffff0000u);
}

// stage the [N][K] weight matrix into LDS rows of LDW elements (zero-padded to KP columns)
template <int NROWS, int K, int KP, int LDW>
__device__ __forceinline__ void rs_stage_w(const bf16_t* __restrict__ W, int ldw, bf16_t* Ws) {
  constexpr int VPR = KP / 8;
  for (int i = threadIdx.x; i < NROWS * VPR; i += blockDim.x) {
    const int n = i / VPR, k = (i - n * VPR) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (k < K) v = *reinterpret_cast<const uint4*>(W + (size_t)n * ldw + k);
    *reinterpret_cast<uint4*>(Ws + n * LDW + k) = v;
  }
}

// =====================================================================================
template <int KC, int HN, int MODE>
__global__ __launch_bounds__(512) void rs_wide_kernel(const RsP p) {
  using T = bf16_t;
  constexpr int KS = (KC + 31) / 32, KP = KS * 32, LDW = KP + 8;
  constexpr int NT = HN / 16, SLD = HN + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
  bf16_t* Ws = reinterpret_cast<bf16_t*>(rs_smem);                         // [HN][LDW]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, NW = blockDim.x >> 6;
  bf16_t* stage = Ws + HN * LDW + wave * 16 * SLD;                         // per wave [16][SLD]
  float* red = reinterpret_cast<float*>(Ws + HN * LDW + NW * 16 * SLD);    // [2][HN] block statistics
  const int lr = lane & 15, lg = lane >> 4;

  rs_stage_w<HN, KC, KP, LDW>(p.W, p.ldw, Ws);
  for (int i = threadIdx.x; i < 2 * HN; i += blockDim.x) red[i] = 0.f;
  float bias[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bias[j] = (MODE == 0 && p.bias) ? p.bias[j * 16 + lr] : 0.f;
  float ga[KS][8], be[KS][8];
  if (MODE == 0) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = s * 32 + lg * 8 + e;
        ga[s][e] = (k < KC) ? p.v0[k] : 0.f;
        be[s][e] = (k < KC) ? p.v1[k] : 0.f;
      }
  }
  float cs0[NT], cs1[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { cs0[j] = 0.f; cs1[j] = 0.f; }
  __syncthreads();

  const int ngroups = p.M >> 4;
  for (int g = blockIdx.x * NW + wave; g < ngroups; g += gridDim.x * NW) {
    const int r0 = g << 4;
    const int row = r0 + lr;
    // ---- A fragments straight from global memory
    bf16x8_t af[KS];
    uint4 raw[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = s * 32 + lg * 8;
      raw[s] = (k < KC) ? *reinterpret_cast<const uint4*>(p.A + (size_t)row * KC + k) : make_uint4(0u, 0u, 0u, 0u);
    }
    uint8_t live4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) live4[r] = p.act ? p.act[r0 + lg * 4 + r] : 1;
    if (MODE == 1) {
      // h tile -> stage (coalesced 16-byte rows); statistics read it in accumulator layout below
      constexpr int CH = 16 * HN / 8;
#pragma unroll
      for (int q = lane; q < CH; q += 64) {
        const int rr = q / (HN / 8), c8 = (q - rr * (HN / 8)) * 8;
        *reinterpret_cast<uint4*>(stage + rr * SLD + c8) =
            *reinterpret_cast<const uint4*>(p.R + (size_t)(r0 + rr) * HN + c8);
      }
    }
    if (MODE == 0) {
      const bool live = !p.act || p.act[row];
      float v[KS][8];
      float s1 = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        unpack8(raw[s], v[s]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s1 += v[s][e];          // columns >= KC are zero
      }
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      const float mean = s1 / KC;
      float s2 = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = s * 32 + lg * 8 + e;
          const float d = (k < KC) ? v[s][e] - mean : 0.f;
          s2 += d * d;
        }
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      const float rstd = rsqrtf(s2 / KC + 1e-6f);
      if (lg == 0) p.rstd[row] = live ? rstd : 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = s * 32 + lg * 8;
        float xh[8], xn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[e] = (live && k + e < KC) ? (v[s][e] - mean) * rstd : 0.f;
          xh[e] = bf2f(f2bf(xh[e]));                          // consumers (and backward) see the stored value
          xn[e] = (k + e < KC) ? xh[e] * ga[s][e] + be[s][e] : 0.f;
        }
        if (!live) {
#pragma unroll
          for (int e = 0; e < 8; ++e) xn[e] = 0.f;
        }
        if (k < KC) st8<T>(p.xhat + (size_t)row * KC + k, xh);
        af[s] = pack_bf16x8(xn);
        if (p.xn && k < KC) *reinterpret_cast<uint4*>(p.xn + (size_t)row * KC + k) = __builtin_bit_cast(uint4, af[s]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) af[s] = __builtin_bit_cast(bf16x8_t, raw[s]);
    }
    // ---- MFMA over all N tiles; epilogue per tile
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8_t bfr = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Ws + (j * 16 + lr) * LDW + s * 32 + lg * 8));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], bfr, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bf16_t* sp = stage + (lg * 4 + r) * SLD + j * 16 + lr;
        if (MODE == 0) {
          float v = acc[r] + bias[j];
          if (!live4[r]) v = 0.f;
          const bf16_t hb = f2bf(v);
          const float gl = gelu_t<T>(bf2f(hb));
          cs0[j] += gl * gl;
          *sp = hb;
        } else {
          const bf16_t db = f2bf(acc[r]);
          const float dzv = bf2f(db);
          const float gl = gelu_t<T>(bf2f(*sp));
          cs0[j] += dzv;
          cs1[j] += dzv * gl;
          *sp = db;
        }
      }
    }
    // ---- transposed tile -> 16-byte coalesced stores
    constexpr int CH = 16 * HN / 8;
#pragma unroll
    for (int q = lane; q < CH; q += 64) {
      const int rr = q / (HN / 8), c8 = (q - rr * (HN / 8)) * 8;
      *reinterpret_cast<uint4*>(p.out + (size_t)(r0 + rr) * HN + c8) = *reinterpret_cast<const uint4*>(stage + rr * SLD + c8);
    }
  }
  // ---- statistics: fold the 4 lane groups, then the block's waves, one slab row per block
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float a = cs0[j];
    a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
    if (lg == 0) atomicAdd(&red[j * 16 + lr], a);
    if (MODE == 1) {
      float b = cs1[j];
      b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      if (lg == 0) atomicAdd(&red[HN + j * 16 + lr], b);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HN; i += blockDim.x) {
    p.ws[(size_t)blockIdx.x * HN + i] = red[i];
    if (MODE == 1) p.ws[((size_t)gridDim.x + blockIdx.x) * HN + i] = red[HN + i];
  }
}

// =====================================================================================
// PRO = false: A is used as stored (z or dh already materialised by grn_apply / grn_bwd_apply)
template <int KC, int HN, int MODE, bool PRO>
__global__ __launch_bounds__(512) void rs_narrow_kernel(const RsP p) {
  using T = bf16_t;
  constexpr int KS = HN / 32, LDW = HN + 8;            // reduction over the hidden dimension
  constexpr int NT = (KC + 15) / 16, NP = NT * 16, SLD = NP + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
  bf16_t* Ws = reinterpret_cast<bf16_t*>(rs_smem);                         // [NP][LDW]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, NW = blockDim.x >> 6;
  float* vs0 = reinterpret_cast<float*>(Ws + NP * LDW);                    // [HN] scale
  float* vs1 = vs0 + HN;                                                   // [HN] beta / coef
  float* red = vs1 + HN;                                                   // [2][NP] (MODE 1)
  bf16_t* stage = reinterpret_cast<bf16_t*>(red + 2 * NP) + wave * 16 * SLD;   // per wave [16][SLD]
  const int lr = lane & 15, lg = lane >> 4;

  // weights: rows >= KC are zero
  {
    constexpr int VPR = HN / 8;
    for (int i = threadIdx.x; i < NP * VPR; i += blockDim.x) {
      const int n = i / VPR, k = (i - n * VPR) * 8;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (n < KC) v = *reinterpret_cast<const uint4*>(p.W + (size_t)n * p.ldw + k);
      *reinterpret_cast<uint4*>(Ws + n * LDW + k) = v;
    }
  }
  if (PRO) { for (int i = threadIdx.x; i < HN; i += blockDim.x) { vs0[i] = p.v0[i]; vs1[i] = p.v1[i]; } }
  for (int i = threadIdx.x; i < 2 * NP; i += blockDim.x) red[i] = 0.f;
  float bias[NT], lgam[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int c = j * 16 + lr;
    bias[j] = (MODE == 0 && p.bias && c < KC) ? p.bias[c] : 0.f;
    lgam[j] = (MODE == 1 && c < KC) ? p.lng[c] : 0.f;
  }
  float ag[NT], ab[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { ag[j] = 0.f; ab[j] = 0.f; }
  __syncthreads();

  const int ngroups = p.M >> 4;
  for (int g = blockIdx.x * NW + wave; g < ngroups; g += gridDim.x * NW) {
    const int r0 = g << 4;
    const int row = r0 + lr;
    // residual x (MODE 0) / x-hat (MODE 1) tile -> stage, read back in accumulator layout
    {
      const bf16_t* src = (MODE == 0) ? p.R : p.xhat;
      constexpr int VPR = KC / 8;
      for (int q = lane; q < 16 * VPR; q += 64) {
        const int rr = q / VPR, c8 = (q - rr * VPR) * 8;
        *reinterpret_cast<uint4*>(stage + rr * SLD + c8) = *reinterpret_cast<const uint4*>(src + (size_t)(r0 + rr) * KC + c8);
      }
    }
    f32x4_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int s = 0; s < KS; ++s) {
      const int k = s * 32 + lg * 8;
      const uint4 araw = *reinterpret_cast<const uint4*>(p.A + (size_t)row * HN + k);
      bf16x8_t af;
      if constexpr (!PRO) {
        af = __builtin_bit_cast(bf16x8_t, araw);
      } else {
      float a[8];
      unpack8(araw, a);
      const float4 sa = *reinterpret_cast<const float4*>(vs0 + k), sb = *reinterpret_cast<const float4*>(vs0 + k + 4);
      const float4 ta = *reinterpret_cast<const float4*>(vs1 + k), tb = *reinterpret_cast<const float4*>(vs1 + k + 4);
      const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
      const float tc[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
      float z[8];
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = gelu_t<T>(a[e]) * sc[e] + tc[e];        // GRN(gelu(h))
      } else {
        float h[8];
        unpack8(*reinterpret_cast<const uint4*>(p.A2 + (size_t)row * HN + k), h);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float gl, dg;
          gelu_both_t<T>(h[e], gl, dg);
          z[e] = (a[e] * sc[e] + tc[e] * gl) * dg;                                  // dh
        }
      }
      af = pack_bf16x8(z);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16x8_t bfr = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Ws + (j * 16 + lr) * LDW + k));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[j], 0, 0, 0);
      }
    }
    uint8_t live4[4];
    float rs4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      live4[r] = p.act ? p.act[r0 + lg * 4 + r] : 1;
      rs4[r] = (MODE == 1) ? p.rstd[r0 + lg * 4 + r] : 0.f;
    }
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bf16_t* sp = stage + (lg * 4 + r) * SLD + j * 16 + lr;
          const float v = live4[r] ? acc[j][r] + bias[j] + bf2f(*sp) : 0.f;      // + residual x
          *sp = f2bf(v);
        }
    } else {
      // LayerNorm backward on the 16 x KC tile: per row r, sums over its KC columns (16 lanes x NT tiles)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float gq[NT], xh[NT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int c = j * 16 + lr;
          xh[j] = (c < KC) ? bf2f(stage[(lg * 4 + r) * SLD + c]) : 0.f;
          const float dxn = (c < KC && live4[r]) ? bf2f(f2bf(acc[j][r])) : 0.f;   // bf16 like the unfused path
          ag[j] += dxn * xh[j];
          ab[j] += dxn;
          gq[j] = dxn * lgam[j];
          s1 += gq[j];
          s2 += gq[j] * xh[j];
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        s1 /= KC; s2 /= KC;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int c = j * 16 + lr;
          const float v = live4[r] ? rs4[r] * (gq[j] - s1 - xh[j] * s2) : 0.f;
          if (c < KC) stage[(lg * 4 + r) * SLD + c] = f2bf(v);
        }
      }
    }
    {
      constexpr int VPR = KC / 8;
      for (int q = lane; q < 16 * VPR; q += 64) {
        const int rr = q / VPR, c8 = (q - rr * VPR) * 8;
        *reinterpret_cast<uint4*>(p.out + (size_t)(r0 + rr) * KC + c8) = *reinterpret_cast<const uint4*>(stage + rr * SLD + c8);
      }
    }
  }
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float a = ag[j], b = ab[j];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      if (lg == 0) { atomicAdd(&red[j * 16 + lr], a); atomicAdd(&red[NP + j * 16 + lr], b); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KC; i += blockDim.x) {
      p.ws[((size_t)blockIdx.x * 2 + 0) * KC + i] = red[i];
      p.ws[((size_t)blockIdx.x * 2 + 1) * KC + i] = red[NP + i];
    }
  }
}
