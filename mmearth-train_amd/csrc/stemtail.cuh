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

// ---------------------------------------------------------------------------------------------------------------
// The whole stem forward of patch size 8 in ONE kernel: masked 3x3 convolution (convnextv2_sparse.py:113-117 on
// MinkowskiOps.to_sparse input) + the fused tail above. One workgroup per visible patch: the 10 x 10 x Cin window is staged in LDS
// exactly as im2col3_kernel stages it (taps outside the image or inside masked patches are zero), wave w owns output rows
// 16w .. 16w+15 of the patch, its A fragments are gathered from the window (k = (kw*3 + kh)*Cin + cin, rounded to bf16 where the
// materialised im2col matrix is), the staged bf16 weights [C0][ldw] come straight from L2 in fragment layout, and the MFMA is
// issued transposed (D[n][m] = W A^T): a lane ends with columns nt*16 + 4*lg + r of row lr, so the LayerNorm sums fold over the
// four lanes lr + 16 lg and everything up to the three 8-byte stores per tensor is lane-local. The convolution output is never
// written (the backward only needs xhat1 / rstd1 / xhat2 / rstd2); the pixel-activity bit comes from the window's centre tap. The
// bf16 im2col matrix that the stem's weight gradient multiplies at the end of the backward is written from the same A fragments.
// Values are rounded to bf16 at the points where the unfused kernels store them. Requires 9 Cin <= 128, Cin <= 12, C0 <= 48, C0 % 4 == 0.
// ---------------------------------------------------------------------------------------------------------------
struct StemFrontP {
  const float* img; const int* vis; const int* inv;
  const bf16_t* W; int ldw; const float* Wm; const float* bias;
  void* xhat1; float* rstd1; void* xhat2; float* rstd2; void* out;
  const float* g1; const float* b1; const float* w; const float* wb; const float* g2; const float* b2;
  bf16_t* col; int ldc;     // optional: the im2col matrix [rows][ldc] of the weight gradient, written from the A fragments
  int keep, grid, H, Cin, C0, track, npatch;
  uint8_t* act_out;         // optional: the activity byte of every row
};

template <int CIN>
__global__ __launch_bounds__(256) void stem_front_kernel(const StemFrontP p) {
  constexpr int S = 8, W2 = S + 2, CP = 13, NW = 5;      // NW = ceil(100 * 12 / 256) window elements per thread
  __shared__ float win[W2 * W2 * CP];
  __shared__ __attribute__((aligned(16))) float prm[7][48];        // bias, g1, b1, w, wb, g2, b2 (staged once: the epilogue reads them as float4)
  __shared__ int nb_ok[9];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int L = p.grid * p.grid, Cin = CIN ? CIN : p.Cin, C0 = p.C0, K = 9 * Cin, H = p.H, np = p.npatch;
  // ---- once per workgroup: weight fragments (clamped addresses, zero through a mask) and the seven per-channel vectors
  bf16x8_t wf[3][4];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int row = nt * 16 + lr, k0 = ks * 32 + lg * 8;
      uint4 v;
      if (p.Wm) {          // fp32 master weight in ME layout [9 Cin][C0] (uniform branch): rounded to bf16 here exactly as the staging does,
        float f[8];        // so the kernel depends on nothing but the inputs and the mask
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = p.Wm[(size_t)min(k0 + e, K - 1) * C0 + min(row, C0 - 1)];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (row < C0 && k0 + e < K) ? f[e] : 0.f;
        v.x = f2bf2(f[0], f[1]); v.y = f2bf2(f[2], f[3]); v.z = f2bf2(f[4], f[5]); v.w = f2bf2(f[6], f[7]);
      } else {
        const bool ok = row < C0 && k0 + 8 <= p.ldw;
        v = *reinterpret_cast<const uint4*>(p.W + (size_t)min(row, C0 - 1) * p.ldw + (k0 + 8 <= p.ldw ? k0 : 0));
        const unsigned mk = opaque_mask(ok);
        v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk;
      }
      wf[nt][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  {
    const float* const src[7] = {p.bias, p.g1, p.b1, p.w, p.wb, p.g2, p.b2};
    if (tid < 7 * 32) {                                  // 7 vectors x 48 columns, clamped (columns >= C0 are masked below)
      const int q = tid >> 5, c0 = tid & 31;
      prm[q][c0] = src[q][min(c0, C0 - 1)];
      if (c0 < 16) prm[q][32 + c0] = src[q][min(32 + c0, C0 - 1)];
    }
  }
  const int total = W2 * W2 * Cin;
  // window of patch `nk` (visible-patch index `patch`): every element unconditionally from a clamped address (x fastest: contiguous
  // image reads), plus the visibility of the 3 x 3 neighbour patches (threads 0..8) - all into registers, consumed one iteration later
  auto request = [&](int nk, int patch, float (&w)[NW], int& nbv) {
    const int n = nk / p.keep, py = patch / p.grid, px = patch - py * p.grid;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i = min(tid + j * 256, total - 1);
      const int wx = i % W2, r = i / W2, wy = r % W2, cin = r / W2;
      const int gy = py * S + wy - 1, gx = px * S + wx - 1;
      w[j] = p.img[((size_t)(n * Cin + cin) * H + min(max(gy, 0), H - 1)) * H + min(max(gx, 0), H - 1)];
    }
    const int t9 = min(tid, 8), qy = py + t9 / 3 - 1, qx = px + t9 % 3 - 1;
    const bool in = qy >= 0 && qx >= 0 && qy < p.grid && qx < p.grid;
    const int iv = p.inv[n * L + min(max(qy, 0), p.grid - 1) * p.grid + min(max(qx, 0), p.grid - 1)];
    nbv = (in && iv >= 0) ? 1 : 0;
  };
  const int iy = 2 * wv + (lr >> 3), ix = lr & 7;
  const float* wb_ = win + (iy * W2 + ix) * CP;
  // persistent over patches nk = blockIdx.x, + gridDim.x, ...: the window of patch i+1 travels while patch i is computed, the patch
  // id of i+2 is requested one step earlier still (it feeds the addresses of the window request)
  int nk = blockIdx.x;
  int patch_nxt = p.vis[min(nk + (int)gridDim.x, np - 1)];
  float wreg[NW];
  int nbv;
  request(nk, p.vis[min(nk, np - 1)], wreg, nbv);
  for (; nk < np; nk += gridDim.x) {
    if (tid < 9) nb_ok[tid] = nbv;
    __syncthreads();                                     // (every wave is done with the previous patch's window)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i = tid + j * 256;
      if (i < total) {
        const int wx = i % W2, r = i / W2, wy = r % W2, cin = r / W2;
        const int by = wy == 0 ? 0 : (wy == W2 - 1 ? 2 : 1), bx = wx == 0 ? 0 : (wx == W2 - 1 ? 2 : 1);
        win[(wy * W2 + wx) * CP + cin] = nb_ok[by * 3 + bx] ? wreg[j] : 0.f;
      }
    }
    __syncthreads();
    {
      const int nk1 = min(nk + (int)gridDim.x, np - 1);
      request(nk1, patch_nxt, wreg, nbv);
      patch_nxt = p.vis[min(nk + 2 * (int)gridDim.x, np - 1)];
    }
    f32x4_t acc[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int m = nk * (S * S) + wv * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float a[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = ks * 32 + lg * 8 + e, kc = min(k, K - 1);
        const int tap = kc / Cin, cin = kc - tap * Cin, kw = tap / 3, kh = tap - kw * 3;
        const float v = wb_[(kh * W2 + kw) * CP + cin];
        a[e] = k < K ? v : 0.f;
      }
      uint4 pk;
      pk.x = f2bf2(a[0], a[1]); pk.y = f2bf2(a[2], a[3]); pk.z = f2bf2(a[4], a[5]); pk.w = f2bf2(a[6], a[7]);
      const bf16x8_t af = __builtin_bit_cast(bf16x8_t, pk);
      // the A fragment IS 8 consecutive k of row lr of the im2col matrix: one 16-byte store (k >= 9 Cin: zeros, as mpmae_im2col3 pads)
      if (p.col && ks * 32 + lg * 8 + 8 <= p.ldc)
        *reinterpret_cast<uint4*>(p.col + (size_t)m * p.ldc + ks * 32 + lg * 8) = pk;
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], af, acc[nt], 0, 0, 0);
    }
    // pixel activity of this row (ME to_sparse: sum_c |x| != 0; NaN counts as active) from the centre tap
    float asum = 0.f;
    for (int c = 0; c < Cin; ++c) asum += fabsf(wb_[(W2 + 1) * CP + c]);
    const bool live = !p.track || asum != 0.f;
    // ---- tail: LN -> GELU -> per-channel affine -> LN (stem_tail_fwd_kernel, same rounding points) on 12 columns per lane
    float v[12], g1[12], b1[12], w1[12], wb1[12], g2[12], b2[12];
    bool cok[12];
    float s = 0.f;
  #pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const int c4 = nt * 16 + lg * 4;
      const float4 vb = *reinterpret_cast<const float4*>(&prm[0][c4]);
      const float4 q1 = *reinterpret_cast<const float4*>(&prm[1][c4]), q2 = *reinterpret_cast<const float4*>(&prm[2][c4]);
      const float4 q3 = *reinterpret_cast<const float4*>(&prm[3][c4]), q4 = *reinterpret_cast<const float4*>(&prm[4][c4]);
      const float4 q5 = *reinterpret_cast<const float4*>(&prm[5][c4]), q6 = *reinterpret_cast<const float4*>(&prm[6][c4]);
      const float bb[4] = {vb.x, vb.y, vb.z, vb.w};
      const float a1[4] = {q1.x, q1.y, q1.z, q1.w}, a2_[4] = {q2.x, q2.y, q2.z, q2.w}, a3[4] = {q3.x, q3.y, q3.z, q3.w};
      const float a4[4] = {q4.x, q4.y, q4.z, q4.w}, a5[4] = {q5.x, q5.y, q5.z, q5.w}, a6[4] = {q6.x, q6.y, q6.z, q6.w};
  #pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = nt * 4 + r;
        cok[j] = c4 + r < C0;
        g1[j] = a1[r]; b1[j] = a2_[r]; w1[j] = a3[r]; wb1[j] = a4[r]; g2[j] = a5[r]; b2[j] = a6[r];
        const float x = bf2f(f2bf(acc[nt][r] + bb[r]));            // c1 as the GEMM stores it (zero row when inactive)
        v[j] = (cok[j] && live) ? x : 0.f;
        s += v[j];
      }
    }
    auto row_sum = [](float t) { t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64); return t; };
    const float invC = 1.f / (float)C0;
    const float mean = row_sum(s) * invC;
    float q = 0.f;
  #pragma unroll
    for (int j = 0; j < 12; ++j) { const float d = cok[j] ? v[j] - mean : 0.f; q += d * d; }
    const float rstd = rsqrtf(row_sum(q) * invC + 1e-6f);
    float xh[12], a2[12];
    float s2 = 0.f;
  #pragma unroll
    for (int j = 0; j < 12; ++j) {
      xh[j] = live ? (v[j] - mean) * rstd : 0.f;
      const float u = live ? gelu_t<bf16_t>(bf2f(f2bf(xh[j])) * g1[j] + b1[j]) : 0.f;
      float d = wb1[j] + (live ? bf2f(f2bf(u)) * w1[j] : 0.f);
      if (!live) d = 0.f;
      a2[j] = cok[j] ? bf2f(f2bf(d)) : 0.f;
      s2 += a2[j];
    }
    const float mean2 = row_sum(s2) * invC;
    float q2 = 0.f;
  #pragma unroll
    for (int j = 0; j < 12; ++j) { const float d = cok[j] ? a2[j] - mean2 : 0.f; q2 += d * d; }
    const float rstd2 = rsqrtf(row_sum(q2) * invC + 1e-6f);
    if (lg == 0) { p.rstd1[m] = live ? rstd : 0.f; p.rstd2[m] = live ? rstd2 : 0.f; if (p.act_out) p.act_out[m] = live ? 1 : 0; }
    bf16_t* o1 = reinterpret_cast<bf16_t*>(p.xhat1) + (size_t)m * C0;
    bf16_t* o2 = reinterpret_cast<bf16_t*>(p.xhat2) + (size_t)m * C0;
    bf16_t* oy = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * C0;
  #pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const int c = nt * 16 + lg * 4;
      if (c < C0) {
        float xh2[4], y[4];
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = nt * 4 + r;
          xh2[r] = live ? (a2[j] - mean2) * rstd2 : 0.f;
          y[r] = live ? bf2f(f2bf(xh2[r])) * g2[j] + b2[j] : 0.f;
        }
        *reinterpret_cast<uint2*>(o1 + c) = make_uint2(f2bf2(xh[nt * 4], xh[nt * 4 + 1]), f2bf2(xh[nt * 4 + 2], xh[nt * 4 + 3]));
        *reinterpret_cast<uint2*>(o2 + c) = make_uint2(f2bf2(xh2[0], xh2[1]), f2bf2(xh2[2], xh2[3]));
        *reinterpret_cast<uint2*>(oy + c) = make_uint2(f2bf2(y[0], y[1]), f2bf2(y[2], y[3]));
      }
    }
  }
}
