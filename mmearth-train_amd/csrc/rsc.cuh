// Chunked row-streaming fused pointwise kernels for the compute-shaped sparse stages (C = 160 / 320, H = 4C).
//
// rs.cuh keeps a whole weight matrix in LDS, which stops fitting at C = 160 (205 KB). Here the
// activation rows stay in REGISTERS for the whole kernel and the weights stream through LDS in
// double-buffered chunks shared by the 4 waves of a workgroup (one barrier per chunk):
//   * A operand: loaded from global memory directly in MFMA fragment layout (lane (lr, lg) reads
//     16 B of row lr at k = 32 s + 8 lg), LayerNorm / GRN / GELU' prologues are lane-local plus
//     two cross-lane adds; no LDS round trip for activations at all;
//   * the MFMA is issued TRANSPOSED, D[n][m] = W[n][:] . A[m][:], so that a lane ends up with 4
//     consecutive output columns of ONE row (col = lr -> row m, rows lg*4+r -> columns n): bias,
//     residual, GELU, LayerNorm-backward row sums are lane-local and results leave as 8-byte
//     stores / operands arrive as 8-byte loads with no transpose tile;
//   * column statistics (GRN sums, LayerNorm gamma/beta gradients) are folded over the 16 lanes
//     that share a column group and leave through LDS float atomics -> one slab row per workgroup.
// rsc_wide  (N = H, chunks over N): MODE 0: x-hat, rstd, xn, h = LN(d) W1^T + b1, sum gelu(h)^2
//                                   MODE 1: dz = dout W2, (sum dz, sum dz*gelu(h))
// rsc_narrow (N = C, chunks over K = H):
//                                   MODE 0: z = gelu(h)*scale + beta (stored), out = x + z W2^T + b2
//                                   MODE 1: dh = (dz*scale + coef*gelu(h))*gelu'(h) (stored over dz),
//                                           dd = LayerNorm-backward(dh W1), dgamma, dbeta partials
#pragma once
#include <type_traits>
#include "rs.cuh"

// weight rows in LDS: odd multiple of 16 bytes (a 32 mod 64 byte stride made no difference here)
constexpr int RSC_PAD = 8;

// Loads in these kernels are UNCONDITIONAL from a clamped address and zeroed afterwards with a mask: `cond ? *ptr : 0` on a per-lane
// condition is a branch, every branch region gets its own s_waitcnt vmcnt(0), and a prologue of N such loads costs N memory latencies
// (rsc_wide: activity byte -> wait -> 2 row vectors -> wait -> 3 row vectors -> wait, before the first MFMA).
__device__ __forceinline__ uint4 and4(const uint4& v, bool keep) {
  const unsigned m = keep ? 0xffffffffu : 0u;
  return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}
__device__ __forceinline__ uint2 and2(const uint2& v, bool keep) {
  const unsigned m = keep ? 0xffffffffu : 0u;
  return make_uint2(v.x & m, v.y & m);
}

__device__ __forceinline__ uint2 pack_bf16x4(const float (&v)[4]) {
  uint2 u;
  u.x = f2bf2(v[0], v[1]);
  u.y = f2bf2(v[2], v[3]);
  return u;
}
__device__ __forceinline__ void unpack4(const uint2& a, float (&o)[4]) {
  o[0] = __uint_as_float(a.x << 16); o[1] = __uint_as_float(a.x & 0xffff0000u);
  o[2] = __uint_as_float(a.y << 16); o[3] = __uint_as_float(a.y & 0xffff0000u);
}

// =====================================================================================
// grid = (ceil(M / (64*RT)), HN / cols_per_split); block = 256
template <int KC, int MODE, int RT, int NC>
__global__ __launch_bounds__(256) void rsc_wide_kernel(const RsP p, int HN_rt, int cols_per_split) {
  using T = bf16_t;
  constexpr int HN = 4 * KC;                    // compile-time row pitch of the hidden tensors (H = 4C)
  (void)HN_rt;
  static_assert(NC % 32 == 0, "chunks are whole tile pairs");
  constexpr int KS = (KC + 31) / 32, KP = KS * 32, LDW = KP + RSC_PAD, VPR = KP / 8, WV = (NC * VPR + 255) / 256;
  constexpr bool PAD = KP != KC;                 // K padded with zero columns (C = 40 / 80)
  static_assert(KC % 8 == 0 && ((KP / 8) & 1) == 0, "LDS rows must be an odd multiple of 16 bytes");
  extern __shared__ __attribute__((aligned(16))) unsigned char rsc_smem[];
  bf16_t* Wc = reinterpret_cast<bf16_t*>(rsc_smem);                                   // [2][NC][LDW]
  float* red = reinterpret_cast<float*>(rsc_smem + (size_t)2 * NC * LDW * sizeof(bf16_t));   // [4 waves][2][cols_per_split]: every wave
  // visits each column of the split exactly once, so its column sums are plain stores into its own row and the four rows are added in
  // a fixed order at the end (LDS float atomics from four waves made the statistics - and everything behind them - differ from run to run)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int rbase = blockIdx.x * (64 * RT) + wave * (16 * RT);
  const int n_begin = blockIdx.y * cols_per_split;
  const int nch = cols_per_split / NC;
  float* redw = red + (p.perwave ? (size_t)wave * 2 * cols_per_split : 0);
  if (!p.perwave) for (int i = tid; i < 2 * cols_per_split; i += 256) red[i] = 0.f;      // (the launcher falls back to one shared row + LDS atomics where
                                                                                         //  the per-wave rows would cost a resident workgroup per CU)

  uint4 wr[WV];
  auto wload = [&](int c) {
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + 256 * i, n = v / VPR, k = (v - n * VPR) * 8;
      const int vc = min(v, NC * VPR - 1), nc = vc / VPR, kc = min((vc - nc * VPR) * 8, KC - 8);
      wr[i] = and4(*reinterpret_cast<const uint4*>(p.W + (size_t)(n_begin + c * NC + nc) * p.ldw + kc), v < NC * VPR && (!PAD || k < KC));
    }
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + 256 * i, n = v / VPR, k = (v - n * VPR) * 8;
      if (v < NC * VPR) *reinterpret_cast<uint4*>(Wc + (size_t)buf * NC * LDW + n * LDW + k) = wr[i];
    }
  };
  wload(0);

  // ---- activation fragments (whole K extent) for this wave's RT row tiles
  bf16x8_t af[RT][KS];
  bool live[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = rbase + rt * 16 + lr;
    const bool inb = row < p.M;
    const int rowc = min(row, p.M - 1);
    const uint8_t abl = *(p.act ? p.act + rowc : reinterpret_cast<const uint8_t*>(p.A));      // pointer select, not a branch
    const uint8_t ab = p.act ? abl : (uint8_t)1;
    uint4 raw[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
      raw[s] = *reinterpret_cast<const uint4*>(p.A + (size_t)rowc * KC + min(s * 32 + lg * 8, KC - 8));
    live[rt] = inb && ab;
#pragma unroll
    for (int s = 0; s < KS; ++s) raw[s] = and4(raw[s], inb && (!PAD || s * 32 + lg * 8 < KC));
    if (MODE == 0) {
      float v[KS][8];
      float s1 = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        unpack8(raw[s], v[s]);
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
      const bool wr_side = inb && blockIdx.y == 0;
      if (wr_side && lg == 0) p.rstd[row] = live[rt] ? rstd : 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = s * 32 + lg * 8;
        const bool kin = !PAD || k < KC;
        const int kc_ = kin ? k : 0;
        const float4 g0 = *reinterpret_cast<const float4*>(p.v0 + kc_), g1 = *reinterpret_cast<const float4*>(p.v0 + kc_ + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(p.v1 + kc_), b1 = *reinterpret_cast<const float4*>(p.v1 + kc_ + 4);
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
      for (int s = 0; s < KS; ++s) af[rt][s] = __builtin_bit_cast(bf16x8_t, raw[s]);
    }
  }
  wstore(0);
  __syncthreads();

  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) wload(c + 1);
    const bf16_t* wb = Wc + (size_t)(c & 1) * NC * LDW;
    // tile PAIRS: the weight rows of two 16-wide tiles are interleaved so that a lane's 2 x 4 accumulator
    // rows are 8 CONSECUTIVE output columns (tile 0 <- columns lg*8 + r, tile 1 <- columns lg*8 + 4 + r):
    // one 16-byte store / load per row and pair instead of two 8-byte ones
#pragma unroll
    for (int jp = 0; jp < NC / 32; ++jp) {
      const int nl = c * NC + jp * 32 + lg * 8;            // first of this lane's 8 columns, relative to n_begin
      const int n8 = n_begin + nl;
      uint4 hraw[RT];
      if (MODE == 1) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = rbase + rt * 16 + lr;
          hraw[rt] = and4(*reinterpret_cast<const uint4*>(p.R + (size_t)min(row, p.M - 1) * HN + n8), row < p.M);
        }
      }
      bf16x8_t wf[2][KS];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s)
          wf[t][s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(
              wb + (jp * 32 + (lr >> 2) * 8 + t * 4 + (lr & 3)) * LDW + s * 32 + lg * 8));
      float cs0[8], cs1[8], bias[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { cs0[e] = 0.f; cs1[e] = 0.f; bias[e] = 0.f; }
      if (MODE == 0 && p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n8), b1 = *reinterpret_cast<const float4*>(p.bias + n8 + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
      }
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
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = live[rt] ? acc[e >> 2][e & 3] + bias[e] : 0.f;
          float gl[8];                      // o is rounded to bf16 by the store; the GRN sums use the fp32 value
          gelu_n<T, 8>(o, gl);              // (what the fp32 reference sums)
#pragma unroll
          for (int e = 0; e < 8; ++e) cs0[e] += gl[e] * gl[e];
        } else {
          float hv[8], gh[8];
          unpack8(hraw[rt], hv);
          gelu_n<T, 8>(hv, gh);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o[e] = acc[e >> 2][e & 3];
            cs0[e] += o[e];
            cs1[e] += o[e] * gh[e];
          }
        }
        if (row < p.M && p.out) st8<T>(p.out + (size_t)row * HN + n8, o);      // without `out`: statistics (and x-hat / xn) only
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = sum16(cs0[e]);
        if (lr == 0) { if (p.perwave) redw[nl + e] = a; else atomicAdd(&redw[nl + e], a); }
        if (MODE == 1) {
          const float b = sum16(cs1[e]);
          if (lr == 0) { if (p.perwave) redw[cols_per_split + nl + e] = b; else atomicAdd(&redw[cols_per_split + nl + e], b); }
        }
      }
    }
    if (c + 1 < nch) wstore((c + 1) & 1);
    __syncthreads();
  }
  // slab row of a workgroup: MODE 0 [HN]; MODE 1 [2][HN] (sum dz | sum dz*gelu(h)) so ONE second-stage launch folds both
  if (p.s0a) {
    // few row blocks (stages 2-3: <= a few hundred adds per column): the statistics go straight into the zero-initialised accumulators of
    // the step with hardware float atomics - the second-stage launch (5-6 us on the main lane, 10 per step) disappears
    const int w2a = p.perwave ? 2 * cols_per_split : 0;
    for (int i = tid; i < cols_per_split; i += 256) {
      (void)unsafeAtomicAdd(p.s0a + n_begin + i, p.perwave ? ((red[i] + red[w2a + i]) + red[2 * w2a + i]) + red[3 * w2a + i] : red[i]);
      if (MODE == 1) (void)unsafeAtomicAdd(p.s1a + n_begin + i, p.perwave ? ((red[cols_per_split + i] + red[w2a + cols_per_split + i]) + red[2 * w2a + cols_per_split + i]) + red[3 * w2a + cols_per_split + i] : red[cols_per_split + i]);
    }
    return;
  }
  const int w2 = p.perwave ? 2 * cols_per_split : 0;
  for (int i = tid; i < cols_per_split; i += 256) {
    const float r0 = p.perwave ? ((red[i] + red[w2 + i]) + red[2 * w2 + i]) + red[3 * w2 + i] : red[i];
    if (MODE == 0) p.ws[(size_t)blockIdx.x * HN + n_begin + i] = r0;
    else {
      const float r1 = p.perwave ? ((red[cols_per_split + i] + red[w2 + cols_per_split + i]) + red[2 * w2 + cols_per_split + i]) + red[3 * w2 + cols_per_split + i] : red[cols_per_split + i];
      p.ws[(size_t)blockIdx.x * 2 * HN + n_begin + i] = r0;
      p.ws[(size_t)blockIdx.x * 2 * HN + HN + n_begin + i] = r1;
    }
  }
}

// =====================================================================================
// grid = ceil(M / (64*RT)); block = 256. rpg = rows per GRN group (M for the batch-global sparse GRN).
// PF bit 0 (few, latency-bound workgroups: C = 160 has 1.25 workgroups per CU): the activation chunk kc+1 is requested at
// the TOP of iteration kc into a second register set (a whole prologue + MFMA phase hides the HBM latency instead of the
// MFMA phase alone). PF bit 1: the GRN scale / beta / coef vectors are staged in LDS once instead of 4 dependent L2 loads
// per chunk, and (p.fin_sum != nullptr) the GRN finalisation itself runs here: every workgroup recomputes the H-vector
// from the column sums (one block reduction) while its first operand loads are in flight, workgroup 0 publishes it -
// two launches fewer per block and direction on the main lane. Needs a single GRN group (rpg >= M).
// PF bit 2 (MODE 1, HBM-bound stages C = 40 / 80): dz is not READ but recomputed, dz = dout W2 for this chunk's columns
// (p.D rows in registers for the whole kernel, W2^T chunk [KCH][C] through LDS): with the tile-pair row interleave the
// transposed MFMA leaves 8 consecutive dz columns of row lr in exactly the lane that needs them as the next MFMA's
// operand, so pw2.dgrad (which = 1) only has to produce the GRN statistics and never stores dz: -105 MB written and
// -79 MB read per stage-0 block on kernels that run at the HBM roofline.
// NWV: waves per workgroup (4; 5 = 80-row tiles: at M = 19 456 rows 64-row tiles are 304 workgroups on 256 CUs - two rounds of a kernel whose
// per-workgroup time is the weight stream, not the rows - 80-row tiles are 244)
template <int KC, int MODE, int RT, int KCH, int PF = 0, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void rsc_narrow_kernel(const RsP p, int HN_rt, int rpg) {
  constexpr int NTH = 64 * NWV;
  using T = bf16_t;
  constexpr int HN = 4 * KC;                    // the ConvNeXt hidden width; a compile-time row pitch keeps address math out of VGPRs
  (void)HN_rt;
  constexpr int NT = (KC + 15) / 16, NP = NT * 16, KSC = KCH / 32, LDW = KCH + RSC_PAD, VPR = KCH / 8, WV = (NP * VPR + NTH - 1) / NTH;
  constexpr bool PAD = NP != KC;                 // output columns padded to whole 16-wide tiles (C = 40)
  static_assert(KC % 8 == 0 && KCH % 32 == 0 && ((KCH / 8) & 1) == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char rsc_smem[];
  bf16_t* Wc = reinterpret_cast<bf16_t*>(rsc_smem);                                   // [2][NP][LDW]
  float* red = reinterpret_cast<float*>(rsc_smem + (size_t)2 * NP * LDW * sizeof(bf16_t));   // [2][KC] (MODE 1)
  float* vec = red + 2 * KC;                                                           // [2][HN] (PF): scale | beta or coef
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int rbase = blockIdx.x * (16 * NWV * RT) + wave * (16 * RT);
  constexpr int nkc = HN / KCH;
  // RC: the wide operand of this kernel is recomputed instead of read: MODE 1 dz = dout W2, MODE 0 h = xn W1^T + b1
  constexpr bool STG = (PF & 2) != 0, EARLY = (PF & 1) != 0, RC = (PF & 4) != 0, DZR = RC && MODE == 1, HR = RC && MODE == 0;
  static_assert(!RC || (STG && !EARLY), "operand recomputation: staged vectors, no early issue");
  constexpr int KS2 = (KC + 31) / 32, KP2 = KS2 * 32, LDW2 = KP2 + RSC_PAD, VPR2 = KP2 / 8, WV2 = (KCH * VPR2 + NTH - 1) / NTH;
  bf16_t* W2c = reinterpret_cast<bf16_t*>(vec + 2 * HN + 8);                          // [2][KCH][LDW2] (DZR): W2^T rows of the chunk
  if (MODE == 1) for (int i = tid; i < 2 * KC; i += NTH) red[i] = 0.f;

  uint4 wr2[RC ? WV2 : 1];
  auto wload2 = [&](int kc) {
#pragma unroll
    for (int i = 0; i < WV2; ++i) {
      const int v = tid + NTH * i, n = v / VPR2, k = (v - n * VPR2) * 8;
      wr2[i] = (v < KCH * VPR2 && k < KC) ? *reinterpret_cast<const uint4*>(p.W2 + (size_t)(kc * KCH + n) * p.ldw2 + k)
                                          : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto wstore2 = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WV2; ++i) {
      const int v = tid + NTH * i, n = v / VPR2, k = (v - n * VPR2) * 8;
      if (v < KCH * VPR2) *reinterpret_cast<uint4*>(W2c + (size_t)buf * KCH * LDW2 + n * LDW2 + k) = wr2[i];
    }
  };

  uint4 wr[WV];
  auto wload = [&](int kc) {
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + NTH * i, n = v / VPR, k = (v - n * VPR) * 8;
      const int vc = min(v, NP * VPR - 1), nc = min(vc / VPR, KC - 1), kq = (vc - (vc / VPR) * VPR) * 8;
      wr[i] = and4(*reinterpret_cast<const uint4*>(p.W + (size_t)nc * p.ldw + kc * KCH + kq), v < NP * VPR && (!PAD || n < KC));
    }
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + NTH * i, n = v / VPR, k = (v - n * VPR) * 8;
      if (v < NP * VPR) *reinterpret_cast<uint4*>(Wc + (size_t)buf * NP * LDW + n * LDW + k) = wr[i];
    }
  };

  int rowv[RT], rowc[RT];                 // row, and the row clamped into the matrix (load address of out-of-range lanes)
  bool inb[RT], live[RT];
  uint8_t abl[RT];
  size_t goff[RT];                       // row's GRN group offset into scale / beta / coef
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    rowv[rt] = rbase + rt * 16 + lr;
    inb[rt] = rowv[rt] < p.M;
    rowc[rt] = min(rowv[rt], p.M - 1);
    abl[rt] = *(p.act ? p.act + rowc[rt] : reinterpret_cast<const uint8_t*>(p.W));      // pointer select, not a branch; turned into
    live[rt] = false;                                                                 // `live` only after the operand loads are out
    goff[rt] = (inb[rt] && !STG) ? (size_t)(rowv[rt] / rpg) * HN : 0;
  }
  constexpr int NB = EARLY ? 2 : 1;
  uint4 araw[NB][RT][KSC], hraw[NB][RT][KSC];
  auto aload = [&](auto bsel, int kc) {
    constexpr int B = decltype(bsel)::value;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int s = 0; s < KSC; ++s) {
        const size_t off = (size_t)rowc[rt] * HN + kc * KCH + s * 32 + lg * 8;
        if (!RC) araw[B][rt][s] = and4(*reinterpret_cast<const uint4*>(p.A + off), inb[rt]);
        if (MODE == 1) hraw[B][rt][s] = and4(*reinterpret_cast<const uint4*>(p.A2 + off), inb[rt]);
      }
  };

  f32x4_t acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[rt][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t df[RC ? RT : 1][RC ? KS2 : 1];          // dout / xn rows of this wave as MFMA operand fragments (whole C extent)
  if (RC) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int s2 = 0; s2 < KS2; ++s2) {
        const int k = s2 * 32 + lg * 8;
        const uint4 v = and4(*reinterpret_cast<const uint4*>(p.D + (size_t)rowc[rt] * KC + min(k, KC - 8)), inb[rt] && k < KC);
        df[rt][s2] = __builtin_bit_cast(bf16x8_t, v);
      }
  }

  auto step = [&](auto bsel, int kc) {
    constexpr int B = decltype(bsel)::value;
    if (EARLY && kc + 1 < nkc) { wload(kc + 1); aload(std::integral_constant<int, EARLY ? (B ^ 1) : 0>{}, kc + 1); }
    const bf16_t* w2b = W2c + (size_t)(kc & 1) * KCH * LDW2;
    // ---- prologue on this chunk's activation fragments (registers), results stored once
    bf16x8_t af[RT][KSC];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int s = 0; s < KSC; ++s) {
        const int k = kc * KCH + s * 32 + lg * 8;
        const float* sp = STG ? vec + k : p.v0 + goff[rt] + k;
        const float* tp = STG ? vec + HN + k : p.v1 + ((MODE == 0) ? 0 : goff[rt]) + k;       // grn beta is per channel, coef per group
        const float4 sa = *reinterpret_cast<const float4*>(sp), sb = *reinterpret_cast<const float4*>(sp + 4);
        const float4 ta = *reinterpret_cast<const float4*>(tp), tb = *reinterpret_cast<const float4*>(tp + 4);
        const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
        const float tc[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
        float a[8], z[8];
        if (RC) {        // dz / h [row lr][kc*KCH + s*32 + lg*8 + e], e = t*4 + r, from the tile pair (t = 0, 1) of this 32-column group
          f32x4_t d2[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            d2[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < KS2; ++s2) {
              const bf16x8_t wf2 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(
                  w2b + (s * 32 + (lr >> 2) * 8 + t * 4 + (lr & 3)) * LDW2 + s2 * 32 + lg * 8));
              d2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2, df[rt][s2], d2[t], 0, 0, 0);
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = d2[e >> 2][e & 3];
          if (HR) {          // h = acc + b1 on live rows, as which = 0 stored it
            const float4 ha = *reinterpret_cast<const float4*>(p.hb + k), hb4 = *reinterpret_cast<const float4*>(p.hb + k + 4);
            const float hbv[8] = {ha.x, ha.y, ha.z, ha.w, hb4.x, hb4.y, hb4.z, hb4.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = live[rt] ? a[e] + hbv[e] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = bf2f(f2bf(a[e]));                    // the value the unfused path stored
        } else {
          unpack8(araw[B][rt][s], a);
        }
        if (MODE == 0) {
          float ga[8];
          gelu_n<T, 8>(a, ga);
#pragma unroll
          for (int e = 0; e < 8; ++e) z[e] = live[rt] ? ga[e] * sc[e] + tc[e] : 0.f;                 // GRN(gelu(h))
        } else {
          float h[8], gl[8], dg[8];
          unpack8(hraw[B][rt][s], h);
          gelu_both_n<T, 8>(h, gl, dg);
#pragma unroll
          for (int e = 0; e < 8; ++e) z[e] = (a[e] * sc[e] + tc[e] * gl[e]) * dg[e];                 // dh
        }
        af[rt][s] = pack_bf16x8(z);
        if (inb[rt]) {
          bf16_t* dst = (MODE == 0) ? p.xn : const_cast<bf16_t*>(p.A);
          if (dst) *reinterpret_cast<uint4*>(dst + (size_t)rowv[rt] * HN + k) = __builtin_bit_cast(uint4, af[rt][s]);
        }
      }
    if (!EARLY && kc + 1 < nkc) { wload(kc + 1); if (RC) wload2(kc + 1); aload(std::integral_constant<int, 0>{}, kc + 1); }
    const bf16_t* wb = Wc + (size_t)(kc & 1) * NP * LDW;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      bf16x8_t wf[KSC];
#pragma unroll
      for (int s = 0; s < KSC; ++s)
        wf[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wb + (j * 16 + lr) * LDW + s * 32 + lg * 8));
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int s = 0; s < KSC; ++s)
          acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], af[rt][s], acc[rt][j], 0, 0, 0);
    }
    if (kc + 1 < nkc) { wstore((kc + 1) & 1); if (RC) wstore2((kc + 1) & 1); }
    __syncthreads();
  };

  wload(0);
  if (RC) wload2(0);
  aload(std::integral_constant<int, 0>{}, 0);
  if (STG) {
    float* fsh = vec + 2 * HN;                      // [4] block-reduction scratch
    if (!p.fin_sum) {
      for (int i = tid; i < HN / 4; i += NTH) {
        reinterpret_cast<float4*>(vec)[i] = reinterpret_cast<const float4*>(p.v0)[i];
        reinterpret_cast<float4*>(vec + HN)[i] = reinterpret_cast<const float4*>(p.v1)[i];
      }
    } else if (MODE == 0) {                         // grn_fwd_finalize_kernel (rows.cuh), same summation order
      // every vector element this thread owns is requested up front (NJ = ceil(H / 256) is a compile-time count, addresses clamped):
      // as `for (j = tid; j < H; j += 256)` loops with the loads inside, the finalisation was five serial round trips - each behind
      // an s_waitcnt vmcnt(0) that also drained the operand loads already in flight - before the first MFMA of every workgroup
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
      const float ainv = 1.f / ((fsh[0] + fsh[1] + fsh[2] + fsh[3] + (NWV > 4 ? fsh[4] : 0.f)) / HN + p.fin_eps);
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
    } else {                                        // grn_bwd_finalize_kernel
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
      const float T2 = (fsh[0] + fsh[1] + fsh[2] + fsh[3] + (NWV > 4 ? fsh[4] : 0.f)) * ainv * ainv / HN;
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
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) live[rt] = inb[rt] && (p.act ? abl[rt] != 0 : true);
  wstore(0);
  if (RC) wstore2(0);
  __syncthreads();
  if (EARLY) {
#pragma unroll 1
    for (int kc = 0; kc < nkc; kc += 2) {
      step(std::integral_constant<int, 0>{}, kc);
      if (kc + 1 < nkc) step(std::integral_constant<int, EARLY ? 1 : 0>{}, kc + 1);
    }
  } else {
    for (int kc = 0; kc < nkc; ++kc) step(std::integral_constant<int, 0>{}, kc);
  }

  // ---- epilogue: lane holds row m = lr, columns n = j*16 + lg*4 + r
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = rowv[rt];
    if (MODE == 0) {
      // residual and bias vectors of every tile first, unconditionally (pointer select for the optional operands): one memory
      // latency for the epilogue instead of one per 16-column tile
      uint2 rraw[NT];
      float4 b4[NT];
      const bf16_t* rp = p.R ? p.R + (size_t)rowc[rt] * KC : p.W;
      const float* bp = p.bias ? p.bias : p.v1;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n4c = min(j * 16 + lg * 4, KC - 4);
        rraw[j] = *reinterpret_cast<const uint2*>(rp + n4c);
        b4[j] = *reinterpret_cast<const float4*>(bp + n4c);
      }
      float ob[NT][4];                                // the row as the next reader sees it (bf16-rounded), for the fused LayerNorm below
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n4 = j * 16 + lg * 4;
        const bool nin = !PAD || n4 < KC;
        float x[4], o[4];
        unpack4(and2(rraw[j], inb[rt] && nin && p.R != nullptr), x);
        const bool hb = p.bias != nullptr && nin;
        const float bb[4] = {hb ? b4[j].x : 0.f, hb ? b4[j].y : 0.f, hb ? b4[j].z : 0.f, hb ? b4[j].w : 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = live[rt] ? acc[rt][j][r] + bb[r] + x[r] : 0.f;
        const uint2 ov = pack_bf16x4(o);
        if (inb[rt] && nin && p.out) *reinterpret_cast<uint2*>(p.out + (size_t)row * KC + n4) = ov;
        unpack4(and2(ov, nin), ob[j]);
      }
      // Fused downsample LayerNorm (round 6, p.dn_y): the LAST block of a stage feeds the LayerNorm in front of the 2x2/2 convolution
      // (convnextv2_sparse.py:131-137, 210-212). A lane holds 4 columns per 16-column tile of its row, the row sums fold over the 4 lane groups
      // (two shuffles, as in the LayerNorm backward of MODE 1): x-hat, rstd and the affine output in the convolution's grouped operand layout
      // leave from here - exactly what mpmae_ln_fwd_down computes from the stored bf16 row, one launch and one read of the stage output less.
      if (p.dn_y) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) s += ob[j][r];
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        const float mean = s / KC;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = (!PAD || j * 16 + lg * 4 < KC) ? ob[j][r] - mean : 0.f; q += d * d; }
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q / KC + 1e-6f);
        if (lg == 0 && inb[rt]) p.dn_rstd[row] = live[rt] ? rstd : 0.f;
        const size_t goff = down_group_off(min(row, p.M - 1), p.dn_S, KC);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n4 = j * 16 + lg * 4, n4c = min(n4, KC - 4);
          const float4 g4 = *reinterpret_cast<const float4*>(p.dn_g + n4c), b4n = *reinterpret_cast<const float4*>(p.dn_b + n4c);
          const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, bv[4] = {b4n.x, b4n.y, b4n.z, b4n.w};
          float xh[4], y[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) xh[r] = live[rt] ? (ob[j][r] - mean) * rstd : 0.f;
          const uint2 xv = pack_bf16x4(xh);
          unpack4(xv, xh);
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] = live[rt] ? xh[r] * gv[r] + bv[r] : 0.f;
          if (inb[rt] && (!PAD || n4 < KC)) {
            *reinterpret_cast<uint2*>(p.dn_xhat + (size_t)row * KC + n4) = xv;
            *reinterpret_cast<uint2*>(p.dn_y + goff + n4) = pack_bf16x4(y);
          }
        }
      }
    } else {
      // LayerNorm backward: row sums are lane-local over (j, r) plus the 4 lane groups
      // every x-hat vector, gamma vector and the row's rstd first, unconditionally (one memory latency for the whole epilogue
      // instead of one per 16-column tile: written as `cond ? load : 0` inside the tile loop these were NT serial round trips)
      uint2 xraw[NT];
      float4 gq4[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n4c = min(j * 16 + lg * 4, KC - 4);
        xraw[j] = *reinterpret_cast<const uint2*>(p.xhat + (size_t)rowc[rt] * KC + n4c);
        gq4[j] = *reinterpret_cast<const float4*>(p.lng + n4c);
      }
      const float rsl = p.rstd[rowc[rt]];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n4 = j * 16 + lg * 4;
        const bool nin = !PAD || n4 < KC;
        xraw[j] = and2(xraw[j], inb[rt] && nin);
        float xh[4];
        unpack4(xraw[j], xh);
        const float4 g = gq4[j];
        const float gg[4] = {nin ? g.x : 0.f, nin ? g.y : 0.f, nin ? g.z : 0.f, nin ? g.w : 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dxn = live[rt] ? bf2f(f2bf(acc[rt][j][r])) : 0.f;      // bf16 like the unfused path
          const float ga = sum16(dxn * xh[r]), gb = sum16(dxn);
          if (lr == 0 && nin) { atomicAdd(&red[n4 + r], ga); atomicAdd(&red[KC + n4 + r], gb); }
          const float gq = dxn * gg[r];
          acc[rt][j][r] = gq;
          s1 += gq;
          s2 += gq * xh[r];
        }
      }
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      s1 /= KC; s2 /= KC;
      const float rs = inb[rt] ? rsl : 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n4 = j * 16 + lg * 4;
        float xh[4], o[4];
        unpack4(xraw[j], xh);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = live[rt] ? rs * (acc[rt][j][r] - s1 - xh[r] * s2) : 0.f;
        if (inb[rt] && (!PAD || n4 < KC)) *reinterpret_cast<uint2*>(p.out + (size_t)row * KC + n4) = pack_bf16x4(o);
      }
    }
  }
  if (MODE == 1) {
    __syncthreads();
    for (int i = tid; i < KC; i += NTH) {
      p.ws[((size_t)blockIdx.x * 2 + 0) * KC + i] = red[i];
      p.ws[((size_t)blockIdx.x * 2 + 1) * KC + i] = red[KC + i];
    }
  }
}
