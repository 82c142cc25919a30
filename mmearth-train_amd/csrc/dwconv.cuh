// Submanifold depthwise 7x7 convolution over compacted visible-patch rows (gfx950).
//
// Replaces MinkowskiDepthwiseConvolution (reference: models/convnextv2_sparse.py:37-39) and, with
// inv == nullptr / keep == grid*grid / S == 1, the dense decoder depthwise conv
// (reference: models/convnextv2.py:27-29). No hash map: neighbours are found by arithmetic on
// (patch grid, intra-patch offset) through the inverse visible-patch table.
//
// A workgroup owns a tile of TS x TS stage points (TS = TP*S <= 8) of one sample and a channel
// chunk; it stages the 14x14 halo (zeros at masked / inactive / out-of-image sites) in LDS as
// fp32, then each thread (fixed channel, strided output points) runs the 49 taps from LDS with
// the 49 weights held in registers.
#pragma once
#include "common.cuh"

constexpr int DW_HALO = 14;                 // 8 + 2*3
constexpr int DW_HP = DW_HALO * DW_HALO;    // 196 halo points

typedef MpmaeDwArgs DwP;

__device__ __forceinline__ void dw_build_rowtab(const DwP& p, int n, int ty0, int tx0, int* rowtab) {
  for (int i = threadIdx.x; i < DW_HP; i += blockDim.x) {
    const int hy = i / DW_HALO, hx = i - hy * DW_HALO;
    int r = geom_row_of(p.g, n, ty0 - 3 + hy, tx0 - 3 + hx);
    if (r >= 0 && p.act && !p.act[r]) r = -1;
    rowtab[i] = r;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv7_fwd_kernel(const DwP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);                   // [196][CC]
  int* rowtab = reinterpret_cast<int*>(smem + (size_t)DW_HP * p.CC * sizeof(float));
  const int tps = p.tiles_side * p.tiles_side;
  const int n = blockIdx.x / tps, t = blockIdx.x - n * tps;
  const int tyi = t / p.tiles_side, txi = t - tyi * p.tiles_side;
  const int TS = p.TP * p.g.S;
  const int ty0 = tyi * TS, tx0 = txi * TS;
  const int c0 = blockIdx.y * p.CC;
  const int CC = p.CC, C = p.C;
  const T* x = reinterpret_cast<const T*>(p.x);

  dw_build_rowtab(p, n, ty0, tx0, rowtab);
  __syncthreads();
  int any = 0;
  for (int i = threadIdx.x; i < TS * TS; i += blockDim.x) {
    const int oy = i / TS, ox = i - oy * TS;
    // a row that EXISTS (visible patch), active or not: inactive rows are written as zeros below. (Until round 6 this looked at the activity-masked
    // table, so a tile whose rows were all inactive - a sample with an all-zero image - returned without writing them; nothing reached this kernel then)
    any |= (geom_row_of(p.g, n, ty0 + oy, tx0 + ox) >= 0);
  }
  if (!__syncthreads_or(any)) return;

  for (int i = threadIdx.x; i < DW_HP * CC; i += blockDim.x) {
    const int hp = i / CC, c = i - hp * CC;
    const int r = rowtab[hp];
    tile[i] = (r >= 0 && c0 + c < C) ? ldf<T>(x + (size_t)r * C + c0 + c) : 0.f;
  }
  const int tc = threadIdx.x % CC, tg = threadIdx.x / CC;
  const int ngroups = blockDim.x / CC;
  const int c = c0 + tc;
  float w[49];
#pragma unroll
  for (int k = 0; k < 49; ++k) {
    int kh = k / 7, kw = k - kh * 7;
    if (p.flip) { kh = 6 - kh; kw = 6 - kw; }
    w[k] = (c < C) ? p.w[kh * p.s_kh + kw * p.s_kw + c * p.s_c] : 0.f;
  }
  const float b = (p.bias && c < C) ? p.bias[c] : 0.f;
  __syncthreads();
  if (tg >= ngroups || c >= C) return;
  T* out = reinterpret_cast<T*>(p.out);
  const T* add = reinterpret_cast<const T*>(p.add);
  for (int i = tg; i < TS * TS; i += ngroups) {
    const int oy = i / TS, ox = i - oy * TS;
    const int r = rowtab[(oy + 3) * DW_HALO + ox + 3];
    if (r < 0) {
      // inactive site inside a visible patch: the row exists in the compacted layout -> write zeros
      const int r2 = geom_row_of(p.g, n, ty0 + oy, tx0 + ox);
      if (r2 >= 0) stf<T>(out + (size_t)r2 * C + c, 0.f);
      continue;
    }
    float acc = b;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx)
        acc += w[ky * 7 + kx] * tile[((oy + ky) * DW_HALO + ox + kx) * CC + tc];
    if (add) acc += ldf<T>(add + (size_t)r * C + c);
    stf<T>(out + (size_t)r * C + c, acc);
  }
}

// weight / bias gradient: dw[kh,kw,c] += sum_p dd[p,c] * x[p + (kh-3, kw-3), c] ; db[c] += sum_p dd[p,c]
typedef MpmaeDwWgArgs DwWgP;
// Grouped depthwise weight gradients (mpmae_dwconv7_wgrad_group): problem blockIdx.z of a launch replaces the operand / slab pointers of
// the shared argument record by its own (count == 0: a plain launch). Read from the kernel-argument segment (scalar loads with a dynamic
// offset; indexing a by-value copy at run time goes to scratch, see ps.cuh).
constexpr int DWG_MAX = 12;
struct DwWgGroupP { int count, pad; const void* x[DWG_MAX]; const void* dd[DWG_MAX]; float* ws[DWG_MAX]; };
__device__ __forceinline__ void dwwg_select(const DwWgP& q, const DwWgGroupP& grp, const void*& x, const void*& dd, float*& ws) {
  x = q.x; dd = q.dd; ws = q.ws;
#if defined(__HIP_DEVICE_COMPILE__)
  if (grp.count > 0) {
    typedef __attribute__((address_space(4))) const char* kchar_p;
    typedef __attribute__((address_space(4))) const DwWgGroupP* kgrp_p;
    const kgrp_p g = (kgrp_p)((kchar_p)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(DwWgP) + 7) / 8) * 8);
    x = g->x[blockIdx.z]; dd = g->dd[blockIdx.z]; ws = g->ws[blockIdx.z];
  }
#endif
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const DwWgP q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);                   // [196][CC]
  int* rowtab = reinterpret_cast<int*>(smem + (size_t)DW_HP * q.CC * sizeof(float));
  float* red = reinterpret_cast<float*>(rowtab + DW_HP + 4);      // [50][CC]
  const int CC = q.CC, C = q.C;
  const int c0 = blockIdx.y * CC;
  const int tc = threadIdx.x % CC, tg = threadIdx.x / CC;
  const int ngroups = blockDim.x / CC;
  const int c = c0 + tc;
  const int TS = q.TP * q.g.S;
  const int tps = q.tiles_side * q.tiles_side;
  const T* x = reinterpret_cast<const T*>(q.x);
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
    for (int i = threadIdx.x; i < TS * TS; i += blockDim.x) {
      const int oy = i / TS, ox = i - oy * TS;
      any |= (rowtab[(oy + 3) * DW_HALO + ox + 3] >= 0);
    }
    if (!__syncthreads_or(any)) continue;
    for (int i = threadIdx.x; i < DW_HP * CC; i += blockDim.x) {
      const int hp = i / CC, cc = i - hp * CC;
      const int r = rowtab[hp];
      tile[i] = (r >= 0 && c0 + cc < C) ? ldf<T>(x + (size_t)r * C + c0 + cc) : 0.f;
    }
    __syncthreads();
    if (tg < ngroups && c < C) {
      for (int i = tg; i < TS * TS; i += ngroups) {
        const int oy = i / TS, ox = i - oy * TS;
        const int r = rowtab[(oy + 3) * DW_HALO + ox + 3];
        if (r < 0) continue;
        const float g = ldf<T>(dd + (size_t)r * C + c);
        adb += g;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
          for (int kx = 0; kx < 7; ++kx)
            adw[ky * 7 + kx] += g * tile[((oy + ky) * DW_HALO + ox + kx) * CC + tc];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 50 * CC; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
  if (tg < ngroups && c < C) {
#pragma unroll
    for (int k = 0; k < 49; ++k) atomicAdd(&red[k * CC + tc], adw[k]);
    atomicAdd(&red[49 * CC + tc], adb);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 50 * CC; i += blockDim.x) {
    const int k = i / CC, cc = i - k * CC;
    if (c0 + cc >= C) continue;
    const float v = red[i];
    if (k < 49) {
      const int kh = k / 7, kw = k - kh * 7;
      atomicAdd(q.dw + kh * q.s_kh + kw * q.s_kw + (c0 + cc) * q.s_c, v);
    } else if (q.db) {
      atomicAdd(q.db + c0 + cc, v);
    }
  }
}
