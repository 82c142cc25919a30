// Per-visible-patch, wave-granular submanifold depthwise 7x7 (forward / data grad / weight grad), v4.
//
// v3 (dwconv3.cuh) walks positional 8x8 tiles; at stages 1-3 a tile covers 4-49 patches of which
// 60 % are masked, so 2.5x of the taps were spent on outputs that are never stored, and 61 % of
// the stage-0 workgroups exited immediately. v4 launches work only for visible patches:
//   one wave = one visible patch (S x S outputs, S = 8/4/2/1) x CW = 64/S channels,
//   lane = (ox, cw): column ox of the patch, channel cw of the chunk, S outputs (oy) per lane,
//   halo (S+6)^2 x CW staged in LDS in the storage type, 49 x CW weights in LDS as fp32
//   (S == 1: weights straight from global, every lane uses each tap once).
// A workgroup holds NW waves = NW channel chunks of the same patch and shares the row table.
#pragma once
#include "dwconv.cuh"

template <int S> struct Dw4 {
  static constexpr int HS = S + 6, HP = HS * HS, CW = 64 / S, VPL = CW / 8;
  static constexpr int ITEMS = HP * VPL, ROUNDS = (ITEMS + 63) / 64;
};

// row table of the (S+6)^2 halo of patch `patch` of sample n (threads of the whole block)
template <int S>
__device__ __forceinline__ void dw4_rowtab(const Geom& g, int n, int patch, int* rowtab) {
  using D = Dw4<S>;
  constexpr int SH = (S == 8) ? 3 : (S == 4) ? 2 : (S == 2) ? 1 : 0;
  const int L = g.grid * g.grid, ext = g.grid * S;
  const int py0 = patch / g.grid, px0 = patch - py0 * g.grid;
  for (int i = threadIdx.x; i < D::HP; i += blockDim.x) {
    const int hy = i / D::HS, hx = i - hy * D::HS;
    const int gy = py0 * S - 3 + hy, gx = px0 * S - 3 + hx;
    int r = -1;
    if (gy >= 0 && gx >= 0 && gy < ext && gx < ext) {
      const int py = gy >> SH, px = gx >> SH;
      const int pp = py * g.grid + px;
      const int slot = g.inv ? g.inv[n * L + pp] : pp;
      if (slot >= 0) r = (n * g.keep + slot) * (S * S) + ((gy - (py << SH)) << SH) + (gx - (px << SH));
    }
    rowtab[i] = r;
  }
}

template <typename T, int S>
__device__ __forceinline__ void dw4_load_tile(const T* __restrict__ x, const int* rowtab, T* tile, int C, int c0) {
  using D = Dw4<S>;
  const int lane = threadIdx.x & 63;
  int r[D::ROUNDS];
#pragma unroll
  for (int k = 0; k < D::ROUNDS; ++k) {
    const int it = lane + 64 * k;
    r[k] = (it < D::ITEMS) ? rowtab[it / D::VPL] : -1;
  }
  if (sizeof(T) == 2) {
    uint4 v[D::ROUNDS];
#pragma unroll
    for (int k = 0; k < D::ROUNDS; ++k) {
      const int it = lane + 64 * k, vv = (it % D::VPL) * 8;
      v[k] = *reinterpret_cast<const uint4*>(x + (size_t)(r[k] < 0 ? 0 : r[k]) * C + c0 + vv);   // clamped, unconditional
    }
#pragma unroll
    for (int k = 0; k < D::ROUNDS; ++k) {
      const int it = lane + 64 * k;
      if (it < D::ITEMS)
        *reinterpret_cast<uint4*>(tile + (it / D::VPL) * D::CW + (it % D::VPL) * 8) = (r[k] >= 0) ? v[k] : make_uint4(0u, 0u, 0u, 0u);
    }
  } else {
#pragma unroll
    for (int k = 0; k < D::ROUNDS; ++k) {
      const int it = lane + 64 * k, vv = (it % D::VPL) * 8;
      const T* src = x + (size_t)(r[k] < 0 ? 0 : r[k]) * C + c0 + vv;
      const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
      if (it < D::ITEMS) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        T* d = tile + (it / D::VPL) * D::CW + (it % D::VPL) * 8;
        *reinterpret_cast<float4*>(d) = (r[k] >= 0) ? a : z;
        *reinterpret_cast<float4*>(d + 4) = (r[k] >= 0) ? b : z;
      }
    }
  }
}

// grid = (N*keep visible patches, ceil(C/CW / NW)); block = 64*NW
template <typename T, int S>
__global__ __launch_bounds__(512) void dwconv7_v4_kernel(const DwP p) {
  using D = Dw4<S>;
  extern __shared__ __attribute__((aligned(16))) unsigned char dw4_smem[];
  const int wave = threadIdx.x >> 6, NW = blockDim.x >> 6, lane = threadIdx.x & 63;
  int* rowtab = reinterpret_cast<int*>(dw4_smem);
  constexpr int RT = (D::HP + 3) & ~3;
  T* tile = reinterpret_cast<T*>(dw4_smem + RT * sizeof(int)) + (size_t)wave * D::HP * D::CW;
  float* wl = reinterpret_cast<float*>(dw4_smem + RT * sizeof(int) + (size_t)NW * D::HP * D::CW * sizeof(T)) + wave * 49 * D::CW;
  const int C = p.C;
  const int nk = blockIdx.x, n = nk / p.g.keep;
  const int patch = p.g.vis ? p.g.vis[nk] : (nk - n * p.g.keep);
  int c0 = (blockIdx.y * NW + wave) * D::CW;
  const bool chunk_ok = c0 < C;
  if (!chunk_ok) c0 = 0;

  dw4_rowtab<S>(p.g, n, patch, rowtab);
  __syncthreads();
  dw4_load_tile<T, S>(reinterpret_cast<const T*>(p.x), rowtab, tile, C, c0);
  const int ox = lane / D::CW, cw = lane - ox * D::CW;
  const int c = c0 + cw;
  if (S > 1) {
    for (int i = lane; i < 49 * D::CW; i += 64) {
      const int k = i / D::CW, cc = i - k * D::CW;
      int kh = k / 7, kw = k - kh * 7;
      if (p.flip) { kh = 6 - kh; kw = 6 - kw; }
      wl[i] = p.w[kh * p.s_kh + kw * p.s_kw + (c0 + cc) * p.s_c];
    }
  }
  // residual / activity operands of this lane's S outputs, requested before the tap loop
  T* out = reinterpret_cast<T*>(p.out);
  const T* add = reinterpret_cast<const T*>(p.add);
  int rr[S];
  float av[S];
  uint8_t live[S];
#pragma unroll
  for (int o = 0; o < S; ++o) rr[o] = rowtab[(o + 3) * D::HS + ox + 3];     // centre points of a visible patch: >= 0
#pragma unroll
  for (int o = 0; o < S; ++o) {
    av[o] = add ? ldf<T>(add + (size_t)rr[o] * C + c) : 0.f;
    live[o] = p.act ? p.act[rr[o]] : 1;
  }
  const float b = p.bias ? p.bias[c] : 0.f;
  float acc[S];
#pragma unroll
  for (int o = 0; o < S; ++o) acc[o] = b;
  __syncthreads();
  if (S == 1) {
    // one output per lane: 49 taps, weights read once each straight from global (coalesced over lanes)
    float w49[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) {
      int kh = k / 7, kw = k - kh * 7;
      if (p.flip) { kh = 6 - kh; kw = 6 - kw; }
      w49[k] = p.w[kh * p.s_kh + kw * p.s_kw + c * p.s_c];
    }
#pragma unroll
    for (int k = 0; k < 49; ++k) acc[0] += w49[k] * ldf<T>(tile + k * D::CW + cw);
  } else {
#pragma unroll 1
    for (int kx = 0; kx < 7; ++kx) {
      float w7[7];
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) w7[ky] = wl[(ky * 7 + kx) * D::CW + cw];
#pragma unroll
      for (int y = 0; y < D::HS; ++y) {
        const float v = ldf<T>(tile + (y * D::HS + ox + kx) * D::CW + cw);
#pragma unroll
        for (int o = 0; o < S; ++o) {
          const int ky = y - o;
          if (ky >= 0 && ky < 7) acc[o] += w7[ky] * v;
        }
      }
    }
  }
  if (!chunk_ok) return;
#pragma unroll
  for (int o = 0; o < S; ++o) stf<T>(out + (size_t)rr[o] * C + c, live[o] ? acc[o] + av[o] : 0.f);
}

// weight / bias gradient: persistent single-wave workgroups over visible patches;
// grid = (nblocks, C/CW); slab ws[blockIdx.x][50][C]
template <typename T, int S>
__global__ __launch_bounds__(64, 5) void dwconv7_wgrad_v4_kernel(const DwWgP q) {
  using D = Dw4<S>;
  __shared__ __attribute__((aligned(16))) T tile[D::HP * D::CW];
  __shared__ int rowtab[(D::HP + 3) & ~3];
  const int lane = threadIdx.x;
  const int ox = lane / D::CW, cw = lane - ox * D::CW;
  const int C = q.C;
  const int c0 = blockIdx.y * D::CW;
  const int c = c0 + cw;
  const T* dd = reinterpret_cast<const T*>(q.dd);
  const int npatches = q.g.N * q.g.keep;

  float adw[49], adb = 0.f;
#pragma unroll
  for (int k = 0; k < 49; ++k) adw[k] = 0.f;

  for (int nk = blockIdx.x; nk < npatches; nk += gridDim.x) {
    const int n = nk / q.g.keep;
    const int patch = q.g.vis ? q.g.vis[nk] : (nk - n * q.g.keep);
    __syncthreads();
    dw4_rowtab<S>(q.g, n, patch, rowtab);
    __syncthreads();
    dw4_load_tile<T, S>(reinterpret_cast<const T*>(q.x), rowtab, tile, C, c0);
    float g[S];
#pragma unroll
    for (int o = 0; o < S; ++o) {
      g[o] = ldf<T>(dd + (size_t)rowtab[(o + 3) * D::HS + ox + 3] * C + c);
      adb += g[o];
    }
    __syncthreads();
    if (S == 1) {
#pragma unroll
      for (int k = 0; k < 49; ++k) adw[k] += g[0] * ldf<T>(tile + k * D::CW + cw);
    } else {
      int toff = 0;     // data dependence between kx-slabs: stops hipcc hoisting all LDS reads (see dwconv3.cuh)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        if (kx > 0)
          asm volatile("" : "+v"(toff) : "v"(adw[kx - 1]), "v"(adw[7 + kx - 1]), "v"(adw[14 + kx - 1]),
                       "v"(adw[21 + kx - 1]), "v"(adw[28 + kx - 1]), "v"(adw[35 + kx - 1]), "v"(adw[42 + kx - 1]));
#pragma unroll
        for (int y = 0; y < D::HS; ++y) {
          const float v = ldf<T>(tile + toff + (y * D::HS + ox + kx) * D::CW + cw);
#pragma unroll
          for (int o = 0; o < S; ++o) {
            const int ky = y - o;
            if (ky >= 0 && ky < 7) adw[ky * 7 + kx] += g[o] * v;
          }
        }
      }
    }
  }
  // fold the S columns (lane bits above log2(CW)); lanes ox == 0 write the slab
  float* slab = q.ws + (size_t)blockIdx.x * 50 * C;
#pragma unroll
  for (int k = 0; k < 49; ++k) {
    float v = adw[k];
#pragma unroll
    for (int o = D::CW; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    if (ox == 0) slab[k * C + c] = v;
  }
#pragma unroll
  for (int o = D::CW; o < 64; o <<= 1) adb += __shfl_xor(adb, o, 64);
  if (ox == 0) slab[49 * C + c] = adb;
}
