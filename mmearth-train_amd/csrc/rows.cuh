// Row-wise kernels: LayerNorm fwd/bwd, GRN statistics finalisation, strided depthwise stem,
// mask-token fill, global-average pooling. Rows are [M, C] channels-last in storage type T.
#pragma once
#include "common.cuh"

constexpr int LN_MAXPER = 16;   // C <= 1024 with one wave per row; LN_MAXPER_WIDE: the generic kernels' second instantiation (large / huge stage 3: C = 1536 / 2816)
constexpr int LN_MAXPER_WIDE = 48;

// ---------------------------------------------------------------------------------
// LayerNorm forward: xhat = (x - mean) * rstd (biased variance, eps inside the sqrt);
//   optional y = act(xhat*gamma + beta), act in {0: identity, 1: GELU}
//   rows with rowmask[m] == 0 produce zeros (inactive sparse sites).
// ---------------------------------------------------------------------------------
template <typename T, int MP = LN_MAXPER>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, T* __restrict__ xhat,
                                                     float* __restrict__ rstd_out, T* __restrict__ y,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int act, float eps, int M, int C,
                                                     const uint8_t* __restrict__ rowmask) {
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int m = wave_global; m < M; m += nwaves) {
    const bool live = !rowmask || rowmask[m];
    float v[MP];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MP; ++i) {
      const int c = lane + i * 64;
      v[i] = (c < C) ? ldf<T>(x + (size_t)m * C + c) : 0.f;
      s += v[i];
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MP; ++i) {
      const int c = lane + i * 64;
      const float d = (c < C) ? v[i] - mean : 0.f;
      q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
    if (lane == 0 && rstd_out) rstd_out[m] = live ? rstd : 0.f;
#pragma unroll
    for (int i = 0; i < MP; ++i) {
      const int c = lane + i * 64;
      if (c < C) {
        float xh = live ? (v[i] - mean) * rstd : 0.f;
        if (xhat) {
          stf<T>(xhat + (size_t)m * C + c, xh);
          if (sizeof(T) == 2) xh = bf2f(f2bf(xh));      // consumers (and bwd) see the rounded value
        }
        if (y) {
          float u = xh * gamma[c] + beta[c];
          if (act == 1) u = gelu_t<T>(u);
          stf<T>(y + (size_t)m * C + c, live ? u : 0.f);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// LayerNorm backward. dy is the gradient w.r.t. y = act(xhat*gamma + beta).
//   dy row index = m / dy_div, scaled by dy_scale (broadcast of a pooled gradient).
//   dx (+)= rstd * (dxh - mean(dxh) - xhat*mean(dxh*xhat)),  dxh = dy'*gamma
//   dgamma += sum_m dy'*xhat ; dbeta += sum_m dy'
// ---------------------------------------------------------------------------------
template <typename T, int MP = LN_MAXPER>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, int dy_div, float dy_scale,
                                                     const T* __restrict__ xhat, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int act, T* __restrict__ dx, int accumulate,
                                                     float* __restrict__ ws,
                                                     int M, int C, const uint8_t* __restrict__ rowmask) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  float ag[MP], ab[MP];
#pragma unroll
  for (int i = 0; i < MP; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
  for (int m = wave_global; m < M; m += nwaves) {
    const bool live = !rowmask || rowmask[m];
    float g[MP], xh[MP];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MP; ++i) {
      const int c = lane + i * 64;
      g[i] = 0.f; xh[i] = 0.f;
      if (c < C && live) {
        xh[i] = ldf<T>(xhat + (size_t)m * C + c);
        float d = ldf<T>(dy + (size_t)(m / dy_div) * C + c) * dy_scale;
        const float ga = gamma[c];
        if (act == 1) d *= gelu_grad_t<T>(xh[i] * ga + beta[c]);
        ag[i] += d * xh[i];
        ab[i] += d;
        g[i] = d * ga;
        s1 += g[i];
        s2 += g[i] * xh[i];
      }
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
    const float rs = live ? rstd[m] : 0.f;
#pragma unroll
    for (int i = 0; i < MP; ++i) {
      const int c = lane + i * 64;
      if (c < C) {
        float v = rs * (g[i] - s1 - xh[i] * s2);
        if (!live) v = 0.f;
        T* o = dx + (size_t)m * C + c;
        if (accumulate) v += ldf<T>(o);
        stf<T>(o, v);
      }
    }
  }
  // reduce dgamma / dbeta over the block's 4 waves, then one atomic per channel per block
#pragma unroll
  for (int i = 0; i < MP; ++i) {
    if (i * 64 < C) {
      red[0][wave][lane] = ag[i];
      red[1][wave][lane] = ab[i];
      __syncthreads();
      if (wave == 0) {
        const int c = lane + i * 64;
        if (c < C) {
          const float a = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
          const float b = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
          // slab ws[block][2][C]; mpmae_ln_bwd reduces the slabs into dgamma / dbeta
          ws[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
          ws[((size_t)blockIdx.x * 2 + 1) * C + c] = b;
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------
// GRN statistics finalisation. One block per group g.
//   fwd: Gx = sqrt(G2[g,:]); A = mean_j Gx; Nx = Gx/(A+eps); scale[g,j] = 1 + gamma[j]*Nx[g,j]
//   bwd: dgamma[j] += Nx*S1 ; dbeta[j] += S0 ; dNx = gamma*S1
//        dGx = dNx/(A+eps) - (1/H) * sum_k dNx[k]*Gx[k] / (A+eps)^2 ; coef = dGx/Gx (0 if Gx == 0)
// ---------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void grn_fwd_finalize_kernel(const float* __restrict__ G2, const float* __restrict__ gamma,
                                                               float eps, int H, float* __restrict__ Gx,
                                                               float* __restrict__ Ainv, float* __restrict__ scale) {
  __shared__ float red[4];
  const int g = blockIdx.x;
  float s = 0.f;
  for (int j = threadIdx.x; j < H; j += 256) {
    const float v = sqrtf(G2[(size_t)g * H + j]);
    Gx[(size_t)g * H + j] = v;
    s += v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float A = (red[0] + red[1] + red[2] + red[3]) / H;
  const float ainv = 1.f / (A + eps);
  if (threadIdx.x == 0) Ainv[g] = ainv;
  for (int j = threadIdx.x; j < H; j += 256)
    scale[(size_t)g * H + j] = 1.f + gamma[j] * (Gx[(size_t)g * H + j] * ainv);
}

// GRN backward statistics FROM the pwconv2 weight gradient (round 5). With z = g * scale + beta, g = gelu(h), and dz = dout W2:
//   S0[j] = sum_m dz[m][j]          = sum_c W2[c][j] * db2[c],          db2 = sum_m dout[m][:]   (the bias gradient)
//   S1[j] = sum_m dz[m][j] g[m][j]  = sum_c W2[c][j] * T[c][j],         T   = dout^T g           (the weight gradient before the GRN affine)
//   dW2[c][j] = scale[j] T[c][j] + beta[j] db2[c]
// so where dz is never materialised (C = 40 / 80: the fused backward kernel recomputes it) the statistics pass over dout and h - a second
// read of the block's widest tensor whose only products are these two H-vectors - is the weight-gradient product itself: mpmae_wgrad runs
// with a GELU-only operand prologue into T / db2 (zero-initialised scratch), this kernel turns them into the statistics AND the parameter
// gradients. One thread per column j; W2s = the bf16 weights the forward and the data gradient multiply with ([C][ldw], k = j).
// Launch shape (round 6): 16 columns x 16 c-lanes per workgroup, every load of a thread independent of the others (C / 16 <= 5 rows each), one LDS
// fold over the c-lanes: the first version - one thread per column walking all C rows with the dW2 read-modify-write inside the loop, ONE
// workgroup at H = 160 - was a 40-trip dependent chain on the main lane in front of the fused backward kernel.
template <typename T>
static __global__ __launch_bounds__(256) void grn_stats_from_wgrad_kernel(const float* __restrict__ Tm, const float* __restrict__ dbt,
                                                                        const T* __restrict__ W2s, int ldw, const float* __restrict__ scale,
                                                                        const float* __restrict__ beta, float* __restrict__ dW2,
                                                                        float* __restrict__ db2, float* __restrict__ S0,
                                                                        float* __restrict__ S1, int C, int H) {
  __shared__ float red[2][16][17];
  const int jl = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + jl, jc = min(j, H - 1);
  const float sc = scale[jc], bt = beta[jc];
  float s0 = 0.f, s1 = 0.f;
  for (int c0 = 0; c0 < C; c0 += 64) {
    float t[4], d[4], w[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = min(c0 + cl + 16 * u, C - 1);
      t[u] = Tm[(size_t)c * H + jc]; d[u] = dbt[c]; w[u] = ldf<T>(W2s + (size_t)c * ldw + jc); g[u] = dW2[(size_t)c * H + jc];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + cl + 16 * u;
      if (c < C && j < H) {
        s0 += w[u] * d[u];
        s1 += w[u] * t[u];
        dW2[(size_t)c * H + j] = g[u] + sc * t[u] + bt * d[u];
      }
    }
  }
  red[0][cl][jl] = s0; red[1][cl][jl] = s1;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int q = threadIdx.x >> 4;
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += red[q][r][jl];
    if (j < H) { if (q == 0) S0[j] += a; else S1[j] += a; }
  }
  if (blockIdx.x == 0) for (int c = threadIdx.x; c < C; c += 256) db2[c] += dbt[c];
}


// The statistics of this thread's columns are loaded up front (all loads in flight at once) and kept in registers: written as two
// loops over j with the float atomics inside, every iteration exposed one global-load latency (the atomics pin the loads behind
// them) and the kernel took 26 us for the decoder's 2 MB of statistics.
static __global__ __launch_bounds__(256) void grn_bwd_finalize_kernel(const float* __restrict__ S0, const float* __restrict__ S1,
                                                               const float* __restrict__ Gx, const float* __restrict__ Ainv,
                                                               const float* __restrict__ gamma, int H,
                                                               float* __restrict__ coef, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int G, int gpb) {
  __shared__ float red[2][4];
  constexpr int MAXC = 16;                       // columns per thread in registers: H <= 4096
  if (H <= 256 * MAXC) {
    // A workgroup walks `gpb` groups (the dense decoder has one group per SAMPLE: 256 of them) and keeps the gamma / beta gradient
    // partials of its columns in registers: one atomic per column and WORKGROUP instead of one per column and group - with a
    // workgroup per group the 256-way same-address atomics were most of the kernel (21 us stand-alone, 44 us in the step).
    float vg[MAXC], dg[MAXC], db[MAXC];
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int j = threadIdx.x + 256 * u;
      vg[u] = gamma[j < H ? j : 0];
      dg[u] = 0.f; db[u] = 0.f;
    }
    const int g0 = blockIdx.x * gpb, g1 = min(G, g0 + gpb);
    for (int g = g0; g < g1; ++g) {
      const float ainv = Ainv[g];
      float v0[MAXC], v1[MAXC], vx[MAXC];
#pragma unroll
      for (int u = 0; u < MAXC; ++u) {
        const int j = threadIdx.x + 256 * u, jc = j < H ? j : 0;
        const size_t i = (size_t)g * H + jc;
        v0[u] = S0[i]; v1[u] = S1[i]; vx[u] = Gx[i];
      }
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < MAXC; ++u) s += (threadIdx.x + 256 * u < H) ? vg[u] * v1[u] * vx[u] : 0.f;
      s = wave_sum(s);
      float* rd = red[(g - g0) & 1];               // two buffers: one barrier per group
      if ((threadIdx.x & 63) == 0) rd[threadIdx.x >> 6] = s;
      __syncthreads();
      const float T2 = (rd[0] + rd[1] + rd[2] + rd[3]) * ainv * ainv / H;
#pragma unroll
      for (int u = 0; u < MAXC; ++u) {
        const int j = threadIdx.x + 256 * u;
        if (j < H) {
          const float dGx = vg[u] * v1[u] * ainv - T2;
          coef[(size_t)g * H + j] = (vx[u] > 0.f) ? dGx / vx[u] : 0.f;
          dg[u] += vx[u] * ainv * v1[u];
          db[u] += v0[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int j = threadIdx.x + 256 * u;
      if (j < H) { atomicAdd(dgamma + j, dg[u]); atomicAdd(dbeta + j, db[u]); }
    }
    return;
  }
  const int g = blockIdx.x;
  const float ainv = Ainv[g];
  float s = 0.f;
  for (int j = threadIdx.x; j < H; j += 256) {
    const size_t i = (size_t)g * H + j;
    s += gamma[j] * S1[i] * Gx[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s;
  __syncthreads();
  const float T2 = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * ainv * ainv / H;
  for (int j = threadIdx.x; j < H; j += 256) {
    const size_t i = (size_t)g * H + j;
    const float gx = Gx[i];
    const float dNx = gamma[j] * S1[i];
    const float dGx = dNx * ainv - T2;
    coef[i] = (gx > 0.f) ? dGx / gx : 0.f;
    atomicAdd(dgamma + j, gx * ainv * S1[i]);
    atomicAdd(dbeta + j, S0[i]);
  }
}

// ---------------------------------------------------------------------------------
// Strided depthwise stem conv (kernel = stride = k in {1,2}); ME kernel index = kw*k + kh.
//   out[m', c] = sum_{kh,kw} in[child(m', kh, kw), c] * w[(kw*k+kh)*C + c] + b[c]
// input rows are patches of (k*S)^2 points, output rows patches of S^2 points.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dwstride_fwd_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           int Mout, int C, int S, int k,
                                                           const uint8_t* __restrict__ act_in,
                                                           const uint8_t* __restrict__ act_out) {
  const size_t total = (size_t)Mout * C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int m = idx / C, c = idx - (size_t)m * C;
    const int P = S * S, nk = m / P, q = m - nk * P, iy = q / S, ix = q - iy * S;
    float acc = b[c];
    for (int kw = 0; kw < k; ++kw)
      for (int kh = 0; kh < k; ++kh) {
        const int src = nk * (k * k * P) + (k * iy + kh) * (k * S) + (k * ix + kw);
        if (!act_in || act_in[src]) acc += ldf<T>(in + (size_t)src * C + c) * w[(kw * k + kh) * C + c];
      }
    if (act_out && !act_out[m]) acc = 0.f;
    stf<T>(out + idx, acc);
  }
}

// backward: din[src,c] = dout[m',c]*w[tap,c] ; dw[tap,c] += sum dout*in ; db[c] += sum dout
template <typename T>
__global__ __launch_bounds__(256) void dwstride_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ in,
                                                           T* __restrict__ din, const float* __restrict__ w,
                                                           float* __restrict__ ws,
                                                           int Mout, int C, int S, int k,
                                                           const uint8_t* __restrict__ act_in) {
  // thread <-> channel (threadIdx.x % C-chunk), rows strided: keeps per-thread channel fixed so the
  // dw/db partial sums live in registers.
  const int cpb = (C < 256) ? C : 256;            // channels handled per block pass
  const int rows_par = 256 / cpb > 0 ? 256 / cpb : 1;
  const int tc = threadIdx.x % cpb, tr = threadIdx.x / cpb;
  for (int c0 = 0; c0 < C; c0 += cpb) {
    const int c = c0 + tc;
    float adw[4] = {0.f, 0.f, 0.f, 0.f}, adb = 0.f;
    if (c < C && tr < rows_par) {
      for (int m = blockIdx.x * rows_par + tr; m < Mout; m += gridDim.x * rows_par) {
        const int P = S * S, nk = m / P, q = m - nk * P, iy = q / S, ix = q - iy * S;
        const float g = ldf<T>(dout + (size_t)m * C + c);
        adb += g;
        for (int kw = 0; kw < k; ++kw)
          for (int kh = 0; kh < k; ++kh) {
            const int tap = kw * k + kh;
            const int src = nk * (k * k * P) + (k * iy + kh) * (k * S) + (k * ix + kw);
            const bool live = !act_in || act_in[src];
            stf<T>(din + (size_t)src * C + c, live ? g * w[tap * C + c] : 0.f);
            if (live) adw[tap] += g * ldf<T>(in + (size_t)src * C + c);
          }
      }
      // per-(block, row-lane) partial slab ws[(blockIdx.x*rows_par + tr)][(k*k+1)][C]
      float* slab = ws + (size_t)(blockIdx.x * rows_par + tr) * (k * k + 1) * C;
      for (int t = 0; t < k * k; ++t) slab[t * C + c] = adw[t];
      slab[k * k * C + c] = adb;
    }
  }
}

// k == 1 (patch 8: the "depthwise stem" is a per-channel affine), C % 8 == 0: 16-byte accesses,
// thread = (row lane, 8-channel vector), block partials folded through LDS -> slab ws[block][2][C]
template <typename T>
__global__ __launch_bounds__(256) void dwstride1_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ in,
                                                            T* __restrict__ din, const float* __restrict__ w,
                                                            float* __restrict__ ws, int Mout, int C,
                                                            const uint8_t* __restrict__ act_in) {
  __shared__ float red[256 * 16];
  const int vpr = C / 8, rl_n = 256 / vpr;
  const int v = threadIdx.x % vpr, rl = threadIdx.x / vpr;
  float adw[8], adb[8], wv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { adw[e] = 0.f; adb[e] = 0.f; wv[e] = (rl < rl_n) ? w[v * 8 + e] : 0.f; }
  if (rl < rl_n) {
    for (int m = blockIdx.x * rl_n + rl; m < Mout; m += gridDim.x * rl_n) {
      float g[8], x[8], o[8];
      ld8<T>(dout + (size_t)m * C + v * 8, g);
      ld8<T>(in + (size_t)m * C + v * 8, x);
      const bool live = !act_in || act_in[m];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        adb[e] += g[e];
        o[e] = live ? g[e] * wv[e] : 0.f;
        adw[e] += live ? g[e] * x[e] : 0.f;
      }
      st8<T>(din + (size_t)m * C + v * 8, o);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[threadIdx.x * 16 + e] = adw[e]; red[threadIdx.x * 16 + 8 + e] = adb[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    float t = 0.f;
    for (int q = 0; q < rl_n; ++q) t += red[(q * vpr + c / 8) * 16 + which * 8 + (c & 7)];
    ws[(size_t)blockIdx.x * 2 * C + i] = t;
  }
}

// k == 2 (patch 16: the depthwise stem is a 2x2 stride-2 depthwise convolution, convnextv2_sparse.py:121-127), C % 8 == 0:
// thread = (row lane, 8-channel vector), 16-byte accesses; the four children of output point (nk, iy, ix) are input
// points (2 iy + kh, 2 ix + kw) of the same patch, tap index kw * 2 + kh (ME kernel order). Forward and backward
// (din, and the dw / db partials of the block folded through LDS -> slab ws[block][5][C]).
template <typename T>
__global__ __launch_bounds__(256) void dwstride2_fwd_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            int Mout, int C, int S, const uint8_t* __restrict__ act_in,
                                                            const uint8_t* __restrict__ act_out) {
  const int vpr = C / 8;
  const long long total = (long long)Mout * vpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i / vpr), v = (int)(i - (long long)m * vpr);
    const int P = S * S, nk = m / P, q = m - nk * P, iy = q / S, ix = q - iy * S;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = b[v * 8 + e];          // parameters are only 4-byte aligned inside the flat buffer
#pragma unroll
    for (int kw = 0; kw < 2; ++kw)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int src = nk * (4 * P) + (2 * iy + kh) * (2 * S) + (2 * ix + kw);
        float x[8], wv[8];
        ld8<T>(in + (size_t)src * C + v * 8, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = w[(kw * 2 + kh) * C + v * 8 + e];
        const bool live = !act_in || act_in[src];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += live ? x[e] * wv[e] : 0.f;
      }
    if (act_out && !act_out[m]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    }
    st8<T>(out + (size_t)m * C + v * 8, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dwstride2_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ in,
                                                            T* __restrict__ din, const float* __restrict__ w,
                                                            float* __restrict__ ws, int Mout, int C, int S,
                                                            const uint8_t* __restrict__ act_in) {
  __shared__ float red[256 * 8];
  const int vpr = C / 8, rl_n = 256 / vpr;
  const int v = threadIdx.x % vpr, rl = threadIdx.x / vpr;
  float adw[4][8], adb[8], wv[4][8];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) { adw[t][e] = 0.f; wv[t][e] = (rl < rl_n) ? w[t * C + v * 8 + e] : 0.f; }
#pragma unroll
  for (int e = 0; e < 8; ++e) adb[e] = 0.f;
  if (rl < rl_n) {
    for (int m = blockIdx.x * rl_n + rl; m < Mout; m += gridDim.x * rl_n) {
      const int P = S * S, nk = m / P, q = m - nk * P, iy = q / S, ix = q - iy * S;
      float g[8];
      ld8<T>(dout + (size_t)m * C + v * 8, g);
#pragma unroll
      for (int e = 0; e < 8; ++e) adb[e] += g[e];
#pragma unroll
      for (int kw = 0; kw < 2; ++kw)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          const int t = kw * 2 + kh;
          const int src = nk * (4 * P) + (2 * iy + kh) * (2 * S) + (2 * ix + kw);
          const bool live = !act_in || act_in[src];
          float x[8], o[8];
          ld8<T>(in + (size_t)src * C + v * 8, x);
#pragma unroll
          for (int e = 0; e < 8; ++e) { o[e] = live ? g[e] * wv[t][e] : 0.f; adw[t][e] += live ? g[e] * x[e] : 0.f; }
          st8<T>(din + (size_t)src * C + v * 8, o);
        }
    }
  }
  // fold the row lanes of the block, one statistic at a time (4 taps + bias) -> slab ws[block][5][C]
  for (int t = 0; t < 5; ++t) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = (t < 4) ? adw[t < 4 ? t : 0][e] : adb[e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float a = 0.f;
      for (int q = 0; q < rl_n; ++q) a += red[(q * vpr + c / 8) * 8 + (c & 7)];
      ws[((size_t)blockIdx.x * 5 + t) * C + c] = a;
    }
  }
}

// ---------------------------------------------------------------------------------
// decoder input: rows of masked patches take the mask token (fcmae.py:253-255); rows of
// visible patches were written by the proj GEMM (EPI_SCATTER_ROWS).
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void fill_mask_token_kernel(T* __restrict__ xdec, const float* __restrict__ token,
                                                              const int* __restrict__ inv, int rows, int D) {
  const size_t total = (size_t)rows * D;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int r = idx / D, c = idx - (size_t)r * D;
    if (inv[r] < 0) stf<T>(xdec + idx, token[c]);
  }
}

// The whole decoder input in one pass (fcmae.py:249-255): row (n, patch) = the projected encoder row of its slot when the patch is visible,
// the mask token otherwise. `vis_rows` is the COMPACT output [N*keep, D] of the proj GEMM (a plain NT GEMM then, no scatter epilogue).
// 8 elements per thread, both candidate sources loaded unconditionally (clamped slot), selected afterwards.
template <typename T>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(T* __restrict__ xdec, const float* __restrict__ token,
                                                              const int* __restrict__ inv, const T* __restrict__ vis_rows,
                                                              int rows, int D, int keep, int L) {
  const int vpr = D / 8;
  const size_t total = (size_t)rows * vpr;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(idx / vpr), v = (int)(idx - (size_t)r * vpr);
    const int s = inv[r];
    float a[8], t[8];
    ld8<T>(vis_rows + ((size_t)(r / L) * keep + max(s, 0)) * D + v * 8, a);
    ld8<float>(token + v * 8, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = s >= 0 ? a[e] : t[e];
    st8<T>(xdec + (size_t)r * D + v * 8, a);
  }
}

// d(mask_token)[c] += sum over masked rows of dxdec[r, c]
// block = 256 threads = RL row lanes x D/8 column vectors (16-byte loads); grid strides over rows
template <typename T>
__global__ __launch_bounds__(256) void mask_token_bwd_kernel(const T* __restrict__ dxdec, const int* __restrict__ inv,
                                                             float* __restrict__ dtoken, int rows, int D,
                                                             T* __restrict__ vis_out, int keepn, int L) {
  __shared__ float red[256 * 8];
  const int vpr = D / 8;                         // D % 8 == 0 and vpr <= 256 checked by the launcher
  const int rl_n = 256 / vpr;
  const int v = threadIdx.x % vpr, rl = threadIdx.x / vpr;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < rl_n) {
    // 8 rows per pass, all `inv` entries first and all row vectors next (two memory latencies per pass instead of two per row:
    // the per-row form `if (inv[r] >= 0) continue; load` was a serial chain of 16 dependent loads = 20 us for 7.8 MB)
    const int stride = gridDim.x * rl_n;
    for (int r0 = blockIdx.x * rl_n + rl; r0 < rows; r0 += 8 * stride) {
      int keep[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int r = r0 + u * stride; keep[u] = r < rows ? inv[r] : 0; }
      float x[8][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int r = min(r0 + u * stride, rows - 1); ld8<T>(dxdec + (size_t)r * D + v * 8, x[u]); }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += keep[u] < 0 ? x[u][e] : 0.f;
      if (vis_out) {      // the same pass gathers the rows of the visible patches into the compact [N*keep, D] operand of proj's gradients
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = r0 + u * stride;
          if (r < rows && keep[u] >= 0) st8<T>(vis_out + ((size_t)(r / L) * keepn + keep[u]) * D + v * 8, x[u]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = a[e];
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float t = 0.f;
    for (int q = 0; q < rl_n; ++q) t += red[(q * vpr + c / 8) * 8 + (c & 7)];
    atomicAdd(dtoken + c, t);
  }
}

// pooled[n, c] = mean over L rows ; bwd handled by ln_bwd's dy_div/dy_scale broadcast
template <typename T>
__global__ __launch_bounds__(256) void pool_rows_kernel(const T* __restrict__ x, T* __restrict__ pooled, int N, int L, int C) {
  const size_t total = (size_t)N * C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int n = idx / C, c = idx - (size_t)n * C;
    float a = 0.f;
    for (int l = 0; l < L; ++l) a += ldf<T>(x + ((size_t)n * L + l) * C + c);
    stf<T>(pooled + idx, a / L);
  }
}
