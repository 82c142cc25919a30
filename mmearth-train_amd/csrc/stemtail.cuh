// Fused tail of the sparse stem for patch size 8 (stem depthwise kernel k = s = 1, i.e. a per-channel
// affine): LayerNorm -> GELU -> affine -> LayerNorm in ONE row-wise pass, and its whole backward in
// one pass. The unfused path (ln_fwd(+GELU), dwstride_fwd, ln_fwd / ln_bwd, dwstride_bwd, ln_bwd)
// streams the [Mfull, C0] activations through HBM 10 + 11 times; fused it is 5 + 6 times, and GELU(y)
// is recomputed in the backward instead of being stored.
// Reference: convnextv2_sparse.py:113-127 (initial_conv = conv3x3 + LN + GELU; stem = depthwise
// k=s=patch/8 + LN). Intermediate values are rounded to the storage type exactly where the unfused
// kernels store them, so both paths produce the same numbers.
#pragma once
#include "rows2.cuh"

template <typename T> __device__ __forceinline__ float rnd_t(float v) { return sizeof(T) == 2 ? bf2f(f2bf(v)) : v; }

struct StemTailP {
  const void* x;            // fwd: conv output c1 [M,C];   bwd: dy = gradient wrt the stem output [M,C]
  void* xhat1; float* rstd1;            // LN1 statistics (fwd: out, bwd: in)
  void* xhat2; float* rstd2;            // LN2 statistics
  void* out;                // fwd: stem output [M,C];      bwd: dc1 = gradient wrt the conv output
  const float* g1; const float* b1;     // LN1 affine
  const float* w; const float* wb;      // depthwise k=1 weight / bias [C]
  const float* g2; const float* b2;     // LN2 affine
  const uint8_t* act_in; const uint8_t* act_out;
  float* ws;                // bwd: slabs [3][nwaves][2][C]: (dg2, db2), (dw, dwb), (dg1, db1)
  int M, C;
};

// rows are handled by G-lane groups (G >= C/8), 64/G rows per wave, 16 bytes per lane
template <typename T, int G>
__global__ __launch_bounds__(256) void stem_tail_fwd_kernel(const StemTailP p) {
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63, gl = lane % G, rl = lane / G;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int C = p.C, nvec = C / 8;
  const bool vok = gl < nvec;
  float g1[8], b1[8], w[8], wb[8], g2[8], b2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = vok ? gl * 8 + e : 0;
    g1[e] = p.g1[c]; b1[e] = p.b1[c]; w[e] = p.w[c]; wb[e] = p.wb[c]; g2[e] = p.g2[c]; b2[e] = p.b2[c];
  }
  const T* x = reinterpret_cast<const T*>(p.x);
  struct Ops { float v[8]; uint8_t a1, a2; };
  auto fetch = [&](int m0, Ops& o) {                      // clamped, unconditional: requested one row group ahead
    const int mc = min(m0 + rl, p.M - 1);
    ld8<T>(x + (size_t)mc * C + (vok ? gl * 8 : 0), o.v);
    o.a1 = *(p.act_in ? p.act_in + mc : reinterpret_cast<const uint8_t*>(p.g1));
    o.a2 = *(p.act_out ? p.act_out + mc : reinterpret_cast<const uint8_t*>(p.g1));
  };
  Ops cur, nxt;
  fetch(wave_global * RPW, cur);
  for (int m0 = wave_global * RPW; m0 < p.M; m0 += nwaves * RPW) {
    fetch(m0 + nwaves * RPW, nxt);
    const int m = m0 + rl;
    const bool rok = m < p.M;
    const bool live1 = rok && (p.act_in ? cur.a1 != 0 : true);
    const bool live2 = rok && (p.act_out ? cur.a2 != 0 : true);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = cur.v[e];
    cur = nxt;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = vok ? v[e] : 0.f; s += v[e]; }
    const float mean = group_sum<G>(s) / C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = vok ? v[e] - mean : 0.f; q += d * d; }
    const float rstd = rsqrtf(group_sum<G>(q) / C + 1e-6f);
    float xh[8], a[8];
    float s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xh[e] = live1 ? (v[e] - mean) * rstd : 0.f;
      const float u = live1 ? gelu_t<T>(rnd_t<T>(xh[e]) * g1[e] + b1[e]) : 0.f;       // a1 (inactive rows: 0)
      float d = wb[e] + (live1 ? rnd_t<T>(u) * w[e] : 0.f);                             // depthwise k = 1
      if (!live2) d = 0.f;
      a[e] = vok ? rnd_t<T>(d) : 0.f;                                                   // s0 as stored
      s2 += a[e];
    }
    const float mean2 = group_sum<G>(s2) / C;
    float q2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = vok ? a[e] - mean2 : 0.f; q2 += d * d; }
    const float rstd2 = rsqrtf(group_sum<G>(q2) / C + 1e-6f);
    if (gl == 0 && rok) { p.rstd1[m] = live1 ? rstd : 0.f; p.rstd2[m] = live2 ? rstd2 : 0.f; }
    if (rok && vok) {
      float xh2[8], y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh2[e] = live2 ? (a[e] - mean2) * rstd2 : 0.f;
        y[e] = live2 ? rnd_t<T>(xh2[e]) * g2[e] + b2[e] : 0.f;
      }
      st8<T>(reinterpret_cast<T*>(p.xhat1) + (size_t)m * C + gl * 8, xh);
      st8<T>(reinterpret_cast<T*>(p.xhat2) + (size_t)m * C + gl * 8, xh2);
      st8<T>(reinterpret_cast<T*>(p.out) + (size_t)m * C + gl * 8, y);
    }
  }
}

template <typename T, int G>
__global__ __launch_bounds__(256) void stem_tail_bwd_kernel(const StemTailP p) {
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63, gl = lane % G, rl = lane / G;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int C = p.C, nvec = C / 8;
  const bool vok = gl < nvec;
  float g1[8], b1[8], w[8], g2[8];
  float ag2[8], ab2[8], adw[8], adb[8], ag1[8], ab1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = vok ? gl * 8 + e : 0;
    g1[e] = p.g1[c]; b1[e] = p.b1[c]; w[e] = p.w[c]; g2[e] = p.g2[c];
    ag2[e] = ab2[e] = adw[e] = adb[e] = ag1[e] = ab1[e] = 0.f;
  }
  const T* dy = reinterpret_cast<const T*>(p.x);
  const T* xhat1 = reinterpret_cast<const T*>(p.xhat1);
  const T* xhat2 = reinterpret_cast<const T*>(p.xhat2);
  // a wave walks ~19 row groups: the operands of group i+1 (3 row vectors, 2 rstd, 2 activity bytes; clamped, unconditional) are
  // requested before group i is reduced - written as one dependent chain per iteration this kernel was 75 us for 100 MB
  struct Ops { float d[8], x2[8], x1[8]; float r2, r1; uint8_t a1, a2; };
  auto fetch = [&](int m0, Ops& o) {
    const int mc = min(m0 + rl, p.M - 1);
    const size_t off = (size_t)mc * C + (vok ? gl * 8 : 0);
    ld8<T>(dy + off, o.d);
    ld8<T>(xhat2 + off, o.x2);
    ld8<T>(xhat1 + off, o.x1);
    o.r2 = p.rstd2[mc]; o.r1 = p.rstd1[mc];
    o.a1 = *(p.act_in ? p.act_in + mc : reinterpret_cast<const uint8_t*>(p.g1));       // pointer select, not a branch
    o.a2 = *(p.act_out ? p.act_out + mc : reinterpret_cast<const uint8_t*>(p.g1));
  };
  Ops cur, nxt;
  fetch(wave_global * RPW, cur);
  for (int m0 = wave_global * RPW; m0 < p.M; m0 += nwaves * RPW) {
    fetch(m0 + nwaves * RPW, nxt);
    const int m = m0 + rl;
    const bool rok = m < p.M;
    const bool live1 = rok && (p.act_in ? cur.a1 != 0 : true);
    const bool live2 = rok && (p.act_out ? cur.a2 != 0 : true);
    float d[8], x2[8], x1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { d[e] = cur.d[e]; x2[e] = cur.x2[e]; x1[e] = cur.x1[e]; }
    const float rs2 = live2 ? cur.r2 : 0.f, rs1 = live1 ? cur.r1 : 0.f;
    cur = nxt;
    // ---- LayerNorm 2 backward
    float g[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dd = (live2 && vok) ? d[e] : 0.f;
      x2[e] = (live2 && vok) ? x2[e] : 0.f;
      ag2[e] += dd * x2[e];
      ab2[e] += dd;
      g[e] = dd * g2[e];
      s1 += g[e];
      s2 += g[e] * x2[e];
    }
    s1 = group_sum<G>(s1) / C;
    s2 = group_sum<G>(s2) / C;
    // ---- depthwise k = 1 backward, GELU backward
    float gq[8], t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float ds = (live2 && vok) ? rnd_t<T>(rs2 * (g[e] - s1 - x2[e] * s2)) : 0.f;      // grad wrt s0 (as stored)
      x1[e] = (live1 && vok) ? x1[e] : 0.f;
      float gl_, dg;
      gelu_both_t<T>(x1[e] * g1[e] + b1[e], gl_, dg);
      const float a1 = live1 ? rnd_t<T>(gl_) : 0.f;
      adb[e] += ds;
      adw[e] += live1 ? ds * a1 : 0.f;
      const float da1 = live1 ? rnd_t<T>(ds * w[e]) : 0.f;
      const float dd = (live1 && vok) ? da1 * dg : 0.f;                                         // grad wrt LN1 output
      ag1[e] += dd * x1[e];
      ab1[e] += dd;
      gq[e] = dd * g1[e];
      t1 += gq[e];
      t2 += gq[e] * x1[e];
    }
    t1 = group_sum<G>(t1) / C;
    t2 = group_sum<G>(t2) / C;
    if (rok && vok) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = live1 ? rs1 * (gq[e] - t1 - x1[e] * t2) : 0.f;
      st8<T>(reinterpret_cast<T*>(p.out) + (size_t)m * C + gl * 8, o);
    }
  }
  // lanes with equal gl hold the same channels: fold the RPW row-lanes, then lanes rl == 0 write the slab rows
  float* acc[6] = {ag2, ab2, adw, adb, ag1, ab1};
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = acc[k][e];
#pragma unroll
      for (int o = G; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
      if (rl == 0 && vok)
        p.ws[(((size_t)(k >> 1) * nwaves + wave_global) * 2 + (k & 1)) * C + gl * 8 + e] = a;
    }
}
