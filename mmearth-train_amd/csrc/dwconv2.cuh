// Register-blocked submanifold depthwise 7x7 (forward / data-gradient / weight-gradient), v2.
//
// Work decomposition (per workgroup): one 8x8 tile of stage points of one sample x a chunk of CC
// channels (CC = 32 or 40, 8*CC threads). The 14x14 halo is gathered from the compacted rows
// through the visible-patch tables with 16-byte loads and kept in LDS as fp32 [196][CC]; thread
// (ox, c) owns the 8 outputs of column ox for channel c and slides the 7x7 window down the
// column: 98 LDS reads + 392 FMAs per 8 outputs, LDS reads linear in the thread index (no bank
// conflicts). The weight-gradient kernel uses the same tiling with 49 register accumulators per
// thread, persistent over tiles, reduced through LDS and flushed with one atomic per (tap, c).
#pragma once
#include "dwconv.cuh"

template <typename T, int CC>
__device__ __forceinline__ void dw2_load_tile(const T* __restrict__ x, const int* rowtab, float* tile, int C, int c0) {
  constexpr int VPP = CC / 8;                       // 8-element vectors per halo point
  for (int i = threadIdx.x; i < DW_HP * VPP; i += 8 * CC) {
    const int pt = i / VPP, ch = (i - pt * VPP) * 8;
    const int r = rowtab[pt];
    float v[8];
    if (r >= 0 && c0 + ch < C) ld8<T>(x + (size_t)r * C + c0 + ch, v);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    float* d = tile + pt * CC + ch;
    *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

template <typename T, int CC>
__global__ __launch_bounds__(8 * CC) void dwconv7_v2_kernel(const DwP p) {
  __shared__ __attribute__((aligned(16))) float tile[DW_HP * CC];
  __shared__ int rowtab[DW_HP];
  const int tps = p.tiles_side * p.tiles_side;
  const int n = blockIdx.x / tps, t = blockIdx.x - n * tps;
  const int tyi = t / p.tiles_side, txi = t - tyi * p.tiles_side;
  const int TS = p.TP * p.g.S;
  const int ty0 = tyi * TS, tx0 = txi * TS;
  const int c0 = blockIdx.y * CC;
  const int C = p.C;

  dw_build_rowtab(p, n, ty0, tx0, rowtab);
  __syncthreads();
  int any = 0;
  if (threadIdx.x < 64) {
    const int oy = threadIdx.x >> 3, ox = threadIdx.x & 7;
    any = (oy < TS && ox < TS && rowtab[(oy + 3) * DW_HALO + ox + 3] >= 0);
  }
  if (!__syncthreads_or(any)) return;
  dw2_load_tile<T, CC>(reinterpret_cast<const T*>(p.x), rowtab, tile, C, c0);

  const int ox = threadIdx.x / CC, tc = threadIdx.x - ox * CC;
  const int c = c0 + tc;
  const bool cok = c < C;
  float w[49];
#pragma unroll
  for (int k = 0; k < 49; ++k) {
    int kh = k / 7, kw = k - kh * 7;
    if (p.flip) { kh = 6 - kh; kw = 6 - kw; }
    w[k] = cok ? p.w[kh * p.s_kh + kw * p.s_kw + c * p.s_c] : 0.f;
  }
  const float b = (p.bias && cok) ? p.bias[c] : 0.f;
  __syncthreads();
  if (!cok || ox >= TS) return;

  float acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = b;
#pragma unroll
  for (int kx = 0; kx < 7; ++kx) {
#pragma unroll
    for (int y = 0; y < DW_HALO; ++y) {
      const float v = tile[(y * DW_HALO + ox + kx) * CC + tc];
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int ky = y - o;
        if (ky >= 0 && ky < 7) acc[o] += w[ky * 7 + kx] * v;
      }
    }
  }
  T* out = reinterpret_cast<T*>(p.out);
  const T* add = reinterpret_cast<const T*>(p.add);
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    if (o < TS) {
      const int r = rowtab[(o + 3) * DW_HALO + ox + 3];
      if (r >= 0) {
        float v = acc[o];
        if (add) v += ldf<T>(add + (size_t)r * C + c);
        stf<T>(out + (size_t)r * C + c, v);
      } else {
        const int r2 = geom_row_of(p.g, n, ty0 + o, tx0 + ox);     // inactive site of a visible patch
        if (r2 >= 0) stf<T>(out + (size_t)r2 * C + c, 0.f);
      }
    }
  }
}

template <typename T, int CC>
__global__ __launch_bounds__(8 * CC) void dwconv7_wgrad_v2_kernel(const DwWgP q) {
  __shared__ __attribute__((aligned(16))) float tile[DW_HP * CC];
  __shared__ int rowtab[DW_HP];
  const int C = q.C;
  const int c0 = blockIdx.y * CC;
  const int ox = threadIdx.x / CC, tc = threadIdx.x - ox * CC;
  const int c = c0 + tc;
  const bool cok = c < C;
  const int TS = q.TP * q.g.S;
  const int tps = q.tiles_side * q.tiles_side;
  const T* dd = reinterpret_cast<const T*>(q.dd);
  DwP p; p.g = q.g; p.act = q.act;

  float adw[49], adb = 0.f;
#pragma unroll
  for (int k = 0; k < 49; ++k) adw[k] = 0.f;

  for (int tile_id = blockIdx.x; tile_id < q.ntiles_total; tile_id += gridDim.x) {
    const int n = tile_id / tps, t = tile_id - n * tps;
    const int tyi = t / q.tiles_side, txi = t - tyi * q.tiles_side;
    const int ty0 = tyi * TS, tx0 = txi * TS;
    __syncthreads();
    dw_build_rowtab(p, n, ty0, tx0, rowtab);
    __syncthreads();
    int any = 0;
    if (threadIdx.x < 64) {
      const int oy = threadIdx.x >> 3, oxx = threadIdx.x & 7;
      any = (oy < TS && oxx < TS && rowtab[(oy + 3) * DW_HALO + oxx + 3] >= 0);
    }
    if (!__syncthreads_or(any)) continue;
    dw2_load_tile<T, CC>(reinterpret_cast<const T*>(q.x), rowtab, tile, C, c0);
    __syncthreads();
    if (cok && ox < TS) {
      float g[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int r = (o < TS) ? rowtab[(o + 3) * DW_HALO + ox + 3] : -1;
        g[o] = (r >= 0) ? ldf<T>(dd + (size_t)r * C + c) : 0.f;
        adb += g[o];
      }
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
#pragma unroll
        for (int y = 0; y < DW_HALO; ++y) {
          const float v = tile[(y * DW_HALO + ox + kx) * CC + tc];
#pragma unroll
          for (int o = 0; o < 8; ++o) {
            const int ky = y - o;
            if (ky >= 0 && ky < 7) adw[ky * 7 + kx] += g[o] * v;
          }
        }
      }
    }
  }
  // reduce the 8 ox-threads of each channel through LDS, then one atomic per (tap, channel)
  __syncthreads();
  float* red = tile;                                 // [50][CC]
  for (int i = threadIdx.x; i < 50 * CC; i += 8 * CC) red[i] = 0.f;
  __syncthreads();
  if (cok) {
#pragma unroll
    for (int k = 0; k < 49; ++k) atomicAdd(&red[k * CC + tc], adw[k]);
    atomicAdd(&red[49 * CC + tc], adb);
  }
  __syncthreads();
  // slab ws[blockIdx.x][50][C]: taps 0..48 then the bias row; reduced by mpmae_dwconv7_wgrad
  float* slab = q.ws + (size_t)blockIdx.x * 50 * C;
  for (int i = threadIdx.x; i < 50 * CC; i += 8 * CC) {
    const int k = i / CC, cc = i - k * CC;
    if (c0 + cc < C) slab[k * C + c0 + cc] = red[i];
  }
}
