// Per-sample, LDS-resident submanifold depthwise 7x7 (forward / data grad / weight grad), v5.
//
// v4 (dwconv4.cuh) stages one (S+6)^2 halo per visible patch: at S = 8/4/2 every input row is
// dragged into LDS 3/6/16 times. v5 stages the whole feature map of ONE sample once:
//   workgroup = (sample n, chunk of CW = 64/S channels), 4 waves;
//   LDS map[(grid*S+6)^2][CW] in the storage type, zero-filled, then the keep*S*S visible rows of
//   the sample are scattered into it (each row is read from HBM exactly once per channel chunk);
//   wave w then walks visible patches w, w+4, ...: lane = (ox, cw) = column ox of the patch,
//   channel cw of the chunk, S outputs (oy) per lane — the v4 inner loop, reading the shared map.
// 49 x CW weights sit in LDS as fp32.
#pragma once
#include "dwconv.cuh"

template <typename T, int S> struct Dw5 {
  static constexpr int CW = 64 / S;
  static constexpr int EPV = 16 / sizeof(T);     // elements per 16-byte vector
  static constexpr int VPL = CW / EPV;           // vectors per row chunk
  static_assert(CW % EPV == 0, "row chunk must be whole 16-byte vectors");
};

template <typename T, int S>
__host__ __device__ inline size_t dw5_map_bytes(int grid) {
  const int MS = grid * S + 6;
  return ((size_t)MS * MS * (64 / S) * sizeof(T) + 15) & ~(size_t)15;
}

// scatter (or clear) the visible rows of sample n into the padded map
template <typename T, int S, bool CLEAR>
__device__ __forceinline__ void dw5_scatter(const Geom& g, int n, const T* __restrict__ x, T* map, int MS, int C, int c0) {
  using D = Dw5<T, S>;
  constexpr int U = 4;      // rows in flight per thread: the table lookup and the row load of U items overlap
  const int items = g.keep * S * S * D::VPL;
  for (int base = threadIdx.x; base < items; base += blockDim.x * U) {
    uint4 val[U];
    int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = base + u * blockDim.x;
      const bool ok = it < items;
      const int itc = ok ? it : 0;
      const int v = itc % D::VPL, pt = itc / D::VPL;
      const int slot = pt / (S * S), q = pt - slot * (S * S);
      const int iy = q / S, ix = q - iy * S;
      const int patch = g.vis ? g.vis[n * g.keep + slot] : slot;
      const int py = patch / g.grid, px = patch - py * g.grid;
      val[u] = make_uint4(0u, 0u, 0u, 0u);
      if (!CLEAR) val[u] = *reinterpret_cast<const uint4*>(x + ((size_t)(n * g.keep + slot) * (S * S) + q) * C + c0 + v * D::EPV);
      dst[u] = ok ? (((py * S + iy + 3) * MS + px * S + ix + 3) * D::CW + v * D::EPV) : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (dst[u] >= 0) *reinterpret_cast<uint4*>(map + dst[u]) = val[u];
  }
}

// grid = (N samples, C / CW); block = 256
template <typename T, int S>
__global__ __launch_bounds__(512) void dwconv7_v5_kernel(const DwP p) {
  using D = Dw5<T, S>;
  constexpr int CW = D::CW;
  extern __shared__ __attribute__((aligned(16))) unsigned char dw5_smem[];
  const int MS = p.g.grid * S + 6;
  T* map = reinterpret_cast<T*>(dw5_smem);
  float* wl = reinterpret_cast<float*>(dw5_smem + dw5_map_bytes<T, S>(p.g.grid));
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, NW = blockDim.x >> 6;
  const int C = p.C, n = blockIdx.x, c0 = blockIdx.y * CW;

  {
    uint4* m4 = reinterpret_cast<uint4*>(dw5_smem);
    const int nvec = (int)(dw5_map_bytes<T, S>(p.g.grid) / 16);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < nvec; i += blockDim.x) m4[i] = z;
  }
  for (int i = tid; i < 49 * CW; i += blockDim.x) {
    const int k = i / CW, cc = i - k * CW;
    int kh = k / 7, kw = k - kh * 7;
    if (p.flip) { kh = 6 - kh; kw = 6 - kw; }
    wl[i] = p.w[kh * p.s_kh + kw * p.s_kw + (c0 + cc) * p.s_c];
  }
  __syncthreads();
  dw5_scatter<T, S, false>(p.g, n, reinterpret_cast<const T*>(p.x), map, MS, C, c0);
  __syncthreads();

  const int ox = lane / CW, cw = lane - ox * CW;
  const int c = c0 + cw;
  T* out = reinterpret_cast<T*>(p.out);
  const T* add = reinterpret_cast<const T*>(p.add);
  const float b = p.bias ? p.bias[c] : 0.f;

  for (int slot = wave; slot < p.g.keep; slot += NW) {
    const int nk = n * p.g.keep + slot;
    const int patch = p.g.vis ? p.g.vis[nk] : slot;
    const int py = patch / p.g.grid, px = patch - py * p.g.grid;
    const T* tile = map + ((size_t)(py * S) * MS + px * S) * CW;      // halo origin of this patch
    const size_t r0 = (size_t)nk * (S * S) + ox;
    float av[S], acc[S];
    uint8_t live[S];
#pragma unroll
    for (int o = 0; o < S; ++o) {
      av[o] = add ? ldf<T>(add + (r0 + o * S) * C + c) : 0.f;
      live[o] = p.act ? p.act[r0 + o * S] : 1;
      acc[o] = b;
    }
#pragma unroll 1
    for (int kx = 0; kx < 7; ++kx) {
      float w7[7];
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) w7[ky] = wl[(ky * 7 + kx) * CW + cw];
      float col[S + 6];      // the whole input column first: independent LDS reads in flight (hipcc serialises them through one register otherwise)
#pragma unroll
      for (int y = 0; y < S + 6; ++y) col[y] = ldf<T>(tile + (y * MS + ox + kx) * CW + cw);
#pragma unroll
      for (int y = 0; y < S + 6; ++y) {
        const float v = col[y];
#pragma unroll
        for (int o = 0; o < S; ++o) {
          const int ky = y - o;
          if (ky >= 0 && ky < 7) acc[o] += w7[ky] * v;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < S; ++o) stf<T>(out + (r0 + o * S) * C + c, live[o] ? acc[o] + av[o] : 0.f);
  }
}

// weight / bias gradient: persistent workgroups over samples; grid = (nblocks, C/CW), block 256;
// slab ws[blockIdx.x][50][C]
// GC = compile-time patch-grid side (0 = runtime): with a constant map pitch every LDS read of the tap loop is
// base + immediate; with a runtime pitch hipcc hoisted the 98 loop-invariant offsets into VGPRs (220 VGPRs, one
// workgroup per CU)
template <typename T, int S, int GC = 0>
__global__ __launch_bounds__(512) void dwconv7_wgrad_v5_kernel(const DwWgP q, const DwWgGroupP grp) {
  using D = Dw5<T, S>;
  constexpr int CW = D::CW;
  extern __shared__ __attribute__((aligned(16))) unsigned char dw5_smem[];
  const int MS = (GC ? GC : q.g.grid) * S + 6;
  T* map = reinterpret_cast<T*>(dw5_smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, NW = blockDim.x >> 6;
  const int C = q.C, c0 = blockIdx.y * CW;
  const int ox = lane / CW, cw = lane - ox * CW;
  const int c = c0 + cw;
  const void *xv, *ddv;
  float* wsv;
  dwwg_select(q, grp, xv, ddv, wsv);
  const T* dd = reinterpret_cast<const T*>(ddv);
  const T* x = reinterpret_cast<const T*>(xv);

  {
    uint4* m4 = reinterpret_cast<uint4*>(dw5_smem);
    const int nvec = (int)(dw5_map_bytes<T, S>(q.g.grid) / 16);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < nvec; i += blockDim.x) m4[i] = z;
  }
  float adw[49], adb = 0.f;
#pragma unroll
  for (int k = 0; k < 49; ++k) adw[k] = 0.f;

  for (int n = blockIdx.x; n < q.g.N; n += gridDim.x) {
    __syncthreads();
    dw5_scatter<T, S, false>(q.g, n, x, map, MS, C, c0);
    __syncthreads();
    for (int slot = wave; slot < q.g.keep; slot += NW) {
      const int nk = n * q.g.keep + slot;
      const int patch = q.g.vis ? q.g.vis[nk] : slot;
      const int py = patch / q.g.grid, px = patch - py * q.g.grid;
      const T* tile = map + ((size_t)(py * S) * MS + px * S) * CW;
      const size_t r0 = (size_t)nk * (S * S) + ox;
      float g[S];
#pragma unroll
      for (int o = 0; o < S; ++o) {
        g[o] = ldf<T>(dd + (r0 + o * S) * C + c);
        adb += g[o];
      }
      int toff = 0;     // data dependence between kx-slabs: stops hipcc hoisting all LDS reads (found with the wave-granular kernels of rounds 1-5, dwconv3.cuh, removed in round 6)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        if (kx > 0)
          asm volatile("" : "+v"(toff) : "v"(adw[kx - 1]), "v"(adw[7 + kx - 1]), "v"(adw[14 + kx - 1]),
                       "v"(adw[21 + kx - 1]), "v"(adw[28 + kx - 1]), "v"(adw[35 + kx - 1]), "v"(adw[42 + kx - 1]) : "memory");
        float col[S + 6];      // column first: independent LDS reads in flight (hipcc serialises them through one register otherwise)
#pragma unroll
        for (int y = 0; y < S + 6; ++y) col[y] = ldf<T>(tile + toff + (y * MS + ox + kx) * CW + cw);
#pragma unroll
        for (int y = 0; y < S + 6; ++y) {
          const float v = col[y];
#pragma unroll
          for (int o = 0; o < S; ++o) {
            const int ky = y - o;
            if (ky >= 0 && ky < 7) adw[ky * 7 + kx] += g[o] * v;
          }
        }
      }
    }
    __syncthreads();
    dw5_scatter<T, S, true>(q.g, n, x, map, MS, C, c0);      // clear only what was written
  }
  // fold the S columns (lane bits above log2(CW)), then the NW waves through LDS
  __syncthreads();
  float* red = reinterpret_cast<float*>(dw5_smem);           // [NW][50][CW]
#pragma unroll
  for (int k = 0; k < 49; ++k) {
    float v = adw[k];
#pragma unroll
    for (int o = CW; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    if (ox == 0) red[(wave * 50 + k) * CW + cw] = v;
  }
#pragma unroll
  for (int o = CW; o < 64; o <<= 1) adb += __shfl_xor(adb, o, 64);
  if (ox == 0) red[(wave * 50 + 49) * CW + cw] = adb;
  __syncthreads();
  float* slab = wsv + (size_t)blockIdx.x * 50 * C;
  for (int i = tid; i < 50 * CW; i += blockDim.x) {
    float v = 0.f;
    for (int w = 0; w < NW; ++w) v += red[w * 50 * CW + i];
    const int k = i / CW, cc = i - k * CW;
    slab[k * C + c0 + cc] = v;
  }
}
