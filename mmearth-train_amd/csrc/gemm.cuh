// MFMA row-GEMM and weight-gradient kernels with fused prologues / epilogues (gfx950).
//
//   gemm_kernel :  C[M,N] = epi( pro(A)[M,K] * B[N,K]^T + bias[N] )
//   wgrad_kernel:  dW[n,k] += sum_m proP(P)[m,n] * proQ(Q)[m,k]      (+ db[n] += sum_m proP(P)[m,n])
//
// Storage type T is float (exact-f32 MFMA 16x16x4, parity mode) or bf16 (MFMA 16x16x32, fast
// mode); accumulation is always fp32. Operands are staged global -> registers (prologue applied
// in fp32) -> LDS (type T, K-contiguous rows) -> MFMA fragments.
#pragma once
#include "common.cuh"

enum Pro : int {
  PRO_NONE = 0,
  PRO_LN_AFFINE = 1,    // a = xhat*p0[k] + p1[k]
  PRO_GRN = 2,          // a = gelu(h)*p0[g,k] + p1[k]          (p0 = 1+gamma*Nx per group)
  PRO_GRN_BWD = 3,      // a = (dz*p0[g,k] + p1[g,k]*gelu(h)) * gelu'(h)   (A=dz, A2=h)
  PRO_DOWN_GATHER = 4,  // row m' gathers its 2x2 children: k = kidx*Cseg + cin; a = xhat_child*p0[cin]+p1[cin]
  PRO_ROW_GATHER = 5,   // source row = (m/keep)*L + vis[m]
  PRO_IM2COL3 = 6,      // 3x3 taps of the masked NCHW fp32 image: k = (kw*3+kh)*Cseg + cin
};

enum Epi : int {
  EPI_STORE = 0,
  EPI_GELU_SUMSQ = 1,   // store h; s0[g,n] += gelu(h)^2
  EPI_RESID = 2,        // store v + R[m,n]
  EPI_DZ_STATS = 3,     // store dz; s0[g,n] += dz ; s1[g,n] += dz*gelu(R[m,n])   (R = h)
  EPI_SCATTER_ROWS = 4, // destination row = (m/keep)*L + vis[m]
  EPI_DOWN_DGRAD = 5,   // column n = kidx*Cseg + cin is written to child row (m, kidx), column cin
};

typedef MpmaeGemmArgs GemmP;

constexpr int GBM = 128, GBN = 64, GBK = 32;
constexpr int GMAXG = 4;

template <typename T> struct LdsPad;
template <> struct LdsPad<float> { static constexpr int v = 4; };
template <> struct LdsPad<bf16_t> { static constexpr int v = 8; };

// child row of output row m (stage s+1, S = points/side of the output) for 2x2 offset kidx = kw*2+kh
__device__ __forceinline__ int down_child_row(int m, int S, int kidx) {
  const int P = S * S;
  const int nk = m / P, q = m - nk * P;
  const int iy = q / S, ix = q - iy * S;
  const int a = kidx & 1, b = kidx >> 1;     // a = kh (rows), b = kw (cols)
  return nk * (4 * P) + (2 * iy + a) * (2 * S) + (2 * ix + b);
}

template <typename T, int PRO>
__device__ __forceinline__ void load_a8(const GemmP& p, int m, int k, float (&o)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = 0.f;
  if (m >= p.M || k >= p.K) return;
  const T* A = reinterpret_cast<const T*>(p.A);
  if constexpr (PRO == PRO_IM2COL3) {
    const float* img = reinterpret_cast<const float*>(p.A);
    const int S = p.S, P = S * S;
    const int nk = m / P, q = m - nk * P;
    const int n = nk / p.keep;
    const int patch = p.vis[nk];
    const int py = patch / p.grid, px = patch - py * p.grid;
    const int y0 = py * S + q / S, x0 = px * S + q % S;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kk = k + i;
      if (kk < p.K) {
        const int tap = kk / p.Cseg, cin = kk - tap * p.Cseg;
        const int kh = tap % 3, kw = tap / 3;
        const int gy = y0 + kh - 1, gx = x0 + kw - 1;
        if (gy >= 0 && gx >= 0 && gy < p.H && gx < p.H) {
          const int pp = (gy / S) * p.grid + gx / S;
          if (p.inv[n * p.L + pp] >= 0) o[i] = img[((size_t)(n * p.Cseg + cin) * p.H + gy) * p.H + gx];
        }
      }
    }
    return;
  }
  int src = m, kc = k;
  if constexpr (PRO == PRO_ROW_GATHER) src = (m / p.keep) * p.L + p.vis[m];
  if constexpr (PRO == PRO_DOWN_GATHER) {
    const int kidx = k / p.Cseg;
    kc = k - kidx * p.Cseg;
    src = down_child_row(m, p.S, kidx);
    if (p.act_src && !p.act_src[src]) return;
  }
  const T* ap = A + (size_t)src * p.lda + kc;
  if (k + 8 <= p.K) ld8<T>(ap, o);
  else { for (int i = 0; i < 8 && k + i < p.K; ++i) o[i] = ldf<T>(ap + i); }
  if constexpr (PRO == PRO_LN_AFFINE || PRO == PRO_DOWN_GATHER) {
    if (k + 8 <= p.K) {      // unconditional 16-byte parameter loads (scalar loads under a branch serialise)
      const float4 g0 = *reinterpret_cast<const float4*>(p.p0 + kc), g1 = *reinterpret_cast<const float4*>(p.p0 + kc + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(p.p1 + kc), b1 = *reinterpret_cast<const float4*>(p.p1 + kc + 4);
      o[0] = o[0] * g0.x + b0.x; o[1] = o[1] * g0.y + b0.y; o[2] = o[2] * g0.z + b0.z; o[3] = o[3] * g0.w + b0.w;
      o[4] = o[4] * g1.x + b1.x; o[5] = o[5] * g1.y + b1.y; o[6] = o[6] * g1.z + b1.z; o[7] = o[7] * g1.w + b1.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) if (k + i < p.K) o[i] = o[i] * p.p0[kc + i] + p.p1[kc + i];
    }
  }
  if constexpr (PRO == PRO_GRN) {
    const int g = m / p.rpg;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (k + i < p.K) o[i] = gelu_t<T>(o[i]) * p.p0[(size_t)g * p.K + k + i] + p.p1[k + i];
  }
  if constexpr (PRO == PRO_GRN_BWD) {
    const int g = m / p.rpg;
    float h[8];
    const T* hp = reinterpret_cast<const T*>(p.A2) + (size_t)m * p.lda + k;
    if (k + 8 <= p.K) ld8<T>(hp, h);
    else { for (int i = 0; i < 8; ++i) h[i] = (k + i < p.K) ? ldf<T>(hp + i) : 0.f; }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (k + i < p.K) {
      const size_t gi = (size_t)g * p.K + k + i;
      float g, dg;
      gelu_both_t<T>(h[i], g, dg);
      o[i] = (o[i] * p.p0[gi] + p.p1[gi] * g) * dg;
    }
  }
}

template <typename T>
__device__ __forceinline__ void load_b8(const T* B, int N, int K, int ldb, int n, int k, float (&o)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = 0.f;
  if (n >= N || k >= K) return;
  const T* bp = B + (size_t)n * ldb + k;
  if (k + 8 <= K) ld8<T>(bp, o);
  else { for (int i = 0; i < 8 && k + i < K; ++i) o[i] = ldf<T>(bp + i); }
}

// one K-slab (GBK) of MFMAs for a wave tile of (MI x 16) x (NJ x 16); As/Bs rows are K-contiguous
template <typename T, int MI, int NJ>
__device__ __forceinline__ void mma_slab(const T* As, const T* Bs, int lda_s, int ldb_s, int arow0, int bcol0,
                                         f32x4_t (&acc)[MI][NJ]) {
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, lg = lane >> 4;
  if constexpr (sizeof(T) == 2) {
    bf16x8_t af[MI], bfr[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const uint4 v = *reinterpret_cast<const uint4*>(As + (arow0 + i * 16 + lr) * lda_s + lg * 8);
      af[i] = __builtin_bit_cast(bf16x8_t, v);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const uint4 v = *reinterpret_cast<const uint4*>(Bs + (bcol0 + j * 16 + lr) * ldb_s + lg * 8);
      bfr[j] = __builtin_bit_cast(bf16x8_t, v);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
  } else {
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 4) {
      float af[MI], bfr[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = As[(arow0 + i * 16 + lr) * lda_s + kk + lg];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[j] = Bs[(bcol0 + j * 16 + lr) * ldb_s + kk + lg];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
}

template <typename T>
__device__ __forceinline__ void lds_st8(T* p, const float (&v)[8]) { st8<T>(p, v); }

template <typename T, int PRO, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
  constexpr int LDS_A = GBK + LdsPad<T>::v, LDS_B = GBK + LdsPad<T>::v;
  __shared__ __attribute__((aligned(16))) T As[GBM * LDS_A];
  __shared__ __attribute__((aligned(16))) T Bs[GBN * LDS_B];
  __shared__ float sacc[2][GMAXG][GBN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN;
  const T* B = reinterpret_cast<const T*>(p.B);

  const int arow = tid >> 1, akb = (tid & 1) * 16;
  const int brow = tid >> 2, bkb = (tid & 3) * 8;

  f32x4_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  float ra0[8], ra1[8], rb[8];
  load_a8<T, PRO>(p, m0 + arow, akb, ra0);
  load_a8<T, PRO>(p, m0 + arow, akb + 8, ra1);
  load_b8<T>(B, p.N, p.K, p.ldb, n0 + brow, bkb, rb);

  for (int k0 = 0; k0 < p.K; k0 += GBK) {
    __syncthreads();
    lds_st8<T>(As + arow * LDS_A + akb, ra0);
    lds_st8<T>(As + arow * LDS_A + akb + 8, ra1);
    lds_st8<T>(Bs + brow * LDS_B + bkb, rb);
    __syncthreads();
    if (k0 + GBK < p.K) {
      load_a8<T, PRO>(p, m0 + arow, k0 + GBK + akb, ra0);
      load_a8<T, PRO>(p, m0 + arow, k0 + GBK + akb + 8, ra1);
      load_b8<T>(B, p.N, p.K, p.ldb, n0 + brow, k0 + GBK + bkb, rb);
    }
    mma_slab<T, 4, 2>(As, Bs, LDS_A, LDS_B, wm * 64, wn * 32, acc);
  }

  // ------------------------------ epilogue ------------------------------
  constexpr bool STATS = (EPI == EPI_GELU_SUMSQ || EPI == EPI_DZ_STATS);
  const bool grouped = STATS && (p.rpg < p.M);
  const int g0 = m0 / p.rpg;
  if constexpr (STATS) {
    for (int i = tid; i < 2 * GMAXG * GBN; i += 256) (&sacc[0][0][0])[i] = 0.f;
    __syncthreads();
  }
  T* C = reinterpret_cast<T*>(p.C);
  const T* R = reinterpret_cast<const T*>(p.R);
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 32 + j * 16 + lr;
    const bool cok = col < p.N;
    const float bias = (cok && p.bias) ? p.bias[col] : 0.f;
    float cs0 = 0.f, cs1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + i * 16 + lg * 4 + r;
        if (row < p.M && cok) {
          float v = acc[i][j][r] + bias;
          if (p.act && !p.act[row]) v = 0.f;
          if constexpr (EPI == EPI_STORE) {
            stf<T>(C + (size_t)row * p.ldc + col, v);
          } else if constexpr (EPI == EPI_RESID) {
            float rr = ldf<T>(R + (size_t)row * p.ldr + col);
            if (p.act && !p.act[row]) rr = 0.f;
            stf<T>(C + (size_t)row * p.ldc + col, v + rr);
          } else if constexpr (EPI == EPI_GELU_SUMSQ) {
            stf<T>(C + (size_t)row * p.ldc + col, v);
            // statistics are taken on the value as the next kernel will re-read it (rounded to T)
            const float hv = (sizeof(T) == 2) ? bf2f(f2bf(v)) : v;
            const float g = gelu_t<T>(hv);
            if (grouped) atomicAdd(&sacc[0][row / p.rpg - g0][col - n0], g * g);
            else cs0 += g * g;
          } else if constexpr (EPI == EPI_DZ_STATS) {
            stf<T>(C + (size_t)row * p.ldc + col, v);
            const float dz = (sizeof(T) == 2) ? bf2f(f2bf(v)) : v;
            const float g = gelu_t<T>(ldf<T>(R + (size_t)row * p.ldr + col));
            if (grouped) {
              atomicAdd(&sacc[0][row / p.rpg - g0][col - n0], dz);
              atomicAdd(&sacc[1][row / p.rpg - g0][col - n0], dz * g);
            } else { cs0 += dz; cs1 += dz * g; }
          } else if constexpr (EPI == EPI_SCATTER_ROWS) {
            const int dst = (row / p.keep) * p.L + p.vis[row];
            stf<T>(C + (size_t)dst * p.ldc + col, v);
          } else if constexpr (EPI == EPI_DOWN_DGRAD) {
            const int kidx = col / p.Cseg, cin = col - kidx * p.Cseg;
            const int dst = down_child_row(row, p.S, kidx);
            if (p.act_src && !p.act_src[dst]) v = 0.f;
            stf<T>(C + (size_t)dst * p.ldc + cin, v);
          }
        }
      }
    }
    if constexpr (STATS) {
      if (!grouped) {
        cs0 += __shfl_xor(cs0, 16, 64); cs0 += __shfl_xor(cs0, 32, 64);
        if (lg == 0 && cok) atomicAdd(&sacc[0][0][col - n0], cs0);
        if constexpr (EPI == EPI_DZ_STATS) {
          cs1 += __shfl_xor(cs1, 16, 64); cs1 += __shfl_xor(cs1, 32, 64);
          if (lg == 0 && cok) atomicAdd(&sacc[1][0][col - n0], cs1);
        }
      }
    }
  }
  if constexpr (STATS) {
    __syncthreads();
    if (!grouped) {
      // single statistics group: this block's column partials go to its own slab row
      // ws[blockIdx.x][N] (second stat: + gridDim.x*N); mpmae_gemm reduces the slabs afterwards.
      for (int c = tid; c < GBN; c += 256) {
        const int col = n0 + c;
        if (col < p.N) {
          p.ws[(size_t)blockIdx.x * p.N + col] = sacc[0][0][c];
          if constexpr (EPI == EPI_DZ_STATS) p.ws[((size_t)gridDim.x + blockIdx.x) * p.N + col] = sacc[1][0][c];
        }
      }
    } else {
      for (int i = tid; i < GMAXG * GBN; i += 256) {
        const int gl = i / GBN, c = i - gl * GBN;
        const int col = n0 + c;
        const int g = g0 + gl;
        if (col < p.N && (size_t)g * p.rpg < (size_t)p.M) {
          const float a = sacc[0][gl][c];
          if (a != 0.f) atomicAdd(p.s0 + (size_t)g * p.N + col, a);
          if constexpr (EPI == EPI_DZ_STATS) {
            const float b = sacc[1][gl][c];
            if (b != 0.f) atomicAdd(p.s1 + (size_t)g * p.N + col, b);
          }
        }
      }
    }
  }
}

// =====================================================================================
// weight gradient:  dW[n*sn + k*sk] += sum_m P'[m,n] * Q'[m,k] ; db[n] += sum_m P'[m,n]
// grid = (ceil(Nn/64), ceil(Kk/64), splits); each z-slice reduces rows [z*rows_per, ...)
// =====================================================================================
typedef MpmaeWgradArgs WgradP;

constexpr int WBN = 64, WBK = 64, WBM = 32;

template <typename T, int PRO>
__device__ __forceinline__ void wg_load8(const WgradP& w, const void* X, const void* X2, int ld, int ncols,
                                         const float* v0, const float* v1, int m, int c, float (&o)[8]) {
  GemmP p;
  p.A = X; p.A2 = X2; p.M = w.M; p.K = ncols; p.lda = ld; p.p0 = v0; p.p1 = v1; p.rpg = w.rpg;
  p.vis = w.vis; p.inv = w.inv; p.act_src = w.act_src; p.keep = w.keep; p.L = w.L; p.S = w.S;
  p.Cseg = w.Cseg; p.grid = w.grid; p.H = w.H;
  load_a8<T, PRO>(p, m, c, o);
}

template <typename T, int PPRO, int QPRO>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradP w) {
  constexpr int LDT = WBM + LdsPad<T>::v;        // transposed tiles: [col][m]
  __shared__ __attribute__((aligned(16))) T Pt[WBN * LDT];
  __shared__ __attribute__((aligned(16))) T Qt[WBK * LDT];
  __shared__ float dbs[WBN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = blockIdx.x * WBN, k0 = blockIdx.y * WBK;
  const int mbeg = blockIdx.z * w.rows_per_split;
  const int mend = min(w.M, mbeg + w.rows_per_split);
  const bool do_db = (w.db != nullptr) && (blockIdx.y == 0);

  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int lm = tid >> 3, lc = (tid & 7) * 8;   // this thread loads row lm, columns lc..lc+7 of both tiles
  float rp[8], rq[8], dbacc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) dbacc[i] = 0.f;

  auto loadPQ = [&](int mb) {
    const int m = mb + lm;
    if (m < mend) {
      wg_load8<T, PPRO>(w, w.P, w.P2, w.ldp, w.Nn, w.pp0, w.pp1, m, n0 + lc, rp);
      wg_load8<T, QPRO>(w, w.Q, nullptr, w.ldq, w.Kk, w.qp0, w.qp1, m, k0 + lc, rq);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { rp[i] = 0.f; rq[i] = 0.f; }
    }
  };

  loadPQ(mbeg);
  for (int mb = mbeg; mb < mend; mb += WBM) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      stf<T>(Pt + (lc + i) * LDT + lm, rp[i]);
      stf<T>(Qt + (lc + i) * LDT + lm, rq[i]);
      dbacc[i] += rp[i];
    }
    __syncthreads();
    if (mb + WBM < mend) loadPQ(mb + WBM);
    // dW tile [n][k]: A operand = Pt rows (n), B operand = Qt rows (k), reduction along m
    mma_slab<T, 2, 2>(Pt, Qt, LDT, LDT, wn * 32, wk * 32, acc);
  }

  // partial results of this M-split go to its own slab ws[z][Nn*Kk] (+ bias slab); mpmae_wgrad
  // runs the second-stage reduction into dW / db afterwards (no global atomics).
  const int lr = lane & 15, lg = lane >> 4;
  float* slab = w.ws + (size_t)blockIdx.z * w.Nn * w.Kk;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 32 + i * 16 + lg * 4 + r;
        const int k = k0 + wk * 32 + j * 16 + lr;
        if (n < w.Nn && k < w.Kk) slab[(size_t)n * w.Kk + k] = acc[i][j][r];
      }
  if (do_db) {
    if (tid < WBN) dbs[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&dbs[lc + i], dbacc[i]);
    __syncthreads();
    float* dslab = w.ws + (size_t)gridDim.z * w.Nn * w.Kk + (size_t)blockIdx.z * w.Nn;
    if (tid < WBN && n0 + tid < w.Nn) dslab[n0 + tid] = dbs[tid];
  }
}
