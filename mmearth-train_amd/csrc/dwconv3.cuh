// Wave-granular submanifold depthwise 7x7 (forward / data gradient / weight gradient), v3.
//
// rocprof showed the block-granular kernels (dwconv.cuh, dwconv2.cuh) latency-bound: every
// workgroup walks a chain of dependent global round trips (visibility table -> halo rows ->
// compute) behind workgroup barriers, with 2 workgroups resident per CU. Here the unit of work
// is ONE 64-lane wave: an 8x8 tile of stage points x 8 channels. A wave keeps its 14x14x8 halo
// (raw storage type) and its 49x8 weights in ~5 KB of LDS, needs no cross-wave barrier, and uses
// <= 64 VGPRs, so 24-32 independent waves per CU overlap each other's memory latency.
//   lane = ox*8 + c8 : column ox of the tile, channel c8 of the chunk; 8 outputs (oy) per lane.
// Inactive sites hold zeros in every row tensor (all producers write zeros there), so the halo
// gather does not consult the activity map; only the outputs are masked.
#pragma once
#include "dwconv.cuh"

constexpr int W64_MAXL = 256;   // patches per sample supported by the LDS copy of the inverse table

template <typename T>
__device__ __forceinline__ void w64_setup(const Geom& g, int n, int ty0, int tx0, int* invs, int* rowtab) {
  const int lane = threadIdx.x;                     // whole block (1..8 waves) shares the row table
  const int L = g.grid * g.grid;
  const int ext = g.grid * g.S, P = g.S * g.S;
  const int sh = (g.S == 8) ? 3 : (g.S == 4) ? 2 : (g.S == 2) ? 1 : 0;     // S is a power of two
  for (int i = lane; i < DW_HP; i += blockDim.x) {
    const int hy = i / DW_HALO, hx = i - hy * DW_HALO;
    const int gy = ty0 - 3 + hy, gx = tx0 - 3 + hx;
    int r = -1;
    if (gy >= 0 && gx >= 0 && gy < ext && gx < ext) {
      const int py = gy >> sh, px = gx >> sh;
      const int patch = py * g.grid + px;
      const int slot = g.inv ? g.inv[n * L + patch] : patch;     // single global round trip (L2-resident table)
      if (slot >= 0) r = (n * g.keep + slot) * P + ((gy - (py << sh)) << sh) + (gx - (px << sh));
    }
    rowtab[i] = r;
  }
  (void)invs;
  __syncthreads();
}

template <typename T>
__device__ __forceinline__ void w64_load_tile(const T* __restrict__ x, const int* rowtab, T* tile, int C, int c0) {
  // 8 channels of one halo point = one 16-byte (bf16) / 32-byte (fp32) vector per lane. All four
  // loads of a lane are issued unconditionally (masked rows read row 0 and are zeroed afterwards):
  // loads under a per-lane branch are serialised by hipcc with a vmcnt(0) each.
  const int lane = threadIdx.x & 63;
  int r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane + 64 * k;
    r[k] = (i < DW_HP) ? rowtab[i] : -1;
  }
  if (sizeof(T) == 2) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const uint4*>(x + (size_t)(r[k] < 0 ? 0 : r[k]) * C + c0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = lane + 64 * k;
      if (i < DW_HP) *reinterpret_cast<uint4*>(tile + i * 8) = (r[k] >= 0) ? v[k] : make_uint4(0u, 0u, 0u, 0u);
    }
  } else {
    float4 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const T* src = x + (size_t)(r[k] < 0 ? 0 : r[k]) * C + c0;
      a[k] = *reinterpret_cast<const float4*>(src);
      b[k] = *reinterpret_cast<const float4*>(src + 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = lane + 64 * k;
      if (i < DW_HP) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(tile + i * 8) = (r[k] >= 0) ? a[k] : z;
        *reinterpret_cast<float4*>(tile + i * 8 + 4) = (r[k] >= 0) ? b[k] : z;
      }
    }
  }
}

// block = NW waves (blockDim.x = 64*NW): shared visibility tables, one 8-channel chunk per wave
template <typename T>
__global__ __launch_bounds__(512) void dwconv7_w64_kernel(const DwP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char w64_smem[];
  const int wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
  int* rowtab = reinterpret_cast<int*>(w64_smem);
  int* invs = rowtab + DW_HP;
  T* tile = reinterpret_cast<T*>(w64_smem + (DW_HP + W64_MAXL) * sizeof(int) + ((size_t)wave * DW_HP * 8) * sizeof(T));
  float* wl = reinterpret_cast<float*>(w64_smem + (DW_HP + W64_MAXL) * sizeof(int) + ((size_t)NW * DW_HP * 8) * sizeof(T)) + wave * 49 * 8;
  const int lane = threadIdx.x & 63;
  const int tps = p.tiles_side * p.tiles_side;
  const int n = blockIdx.x / tps, t = blockIdx.x - n * tps;
  const int tyi = t / p.tiles_side, txi = t - tyi * p.tiles_side;
  const int TS = p.TP * p.g.S;
  const int ty0 = tyi * TS, tx0 = txi * TS;
  const int C = p.C;
  int c0 = (blockIdx.y * NW + wave) * 8;
  const bool chunk_ok = c0 < C;
  if (!chunk_ok) c0 = 0;                      // idle waves of the last block mirror chunk 0 (no stores)

  w64_setup<T>(p.g, n, ty0, tx0, invs, rowtab);
  {
    const int oy = lane >> 3, oxx = lane & 7;
    const int valid = (oy < TS && oxx < TS && rowtab[(oy + 3) * DW_HALO + oxx + 3] >= 0);
    if (!__any(valid)) return;                // same tile for every wave of the block: uniform exit
  }
  w64_load_tile<T>(reinterpret_cast<const T*>(p.x), rowtab, tile, C, c0);
  for (int i = lane; i < 49 * 8; i += 64) {
    int k = i >> 3;
    const int c = i & 7;
    int kh = k / 7, kw = k - kh * 7;
    if (p.flip) { kh = 6 - kh; kw = 6 - kw; }
    wl[i] = p.w[kh * p.s_kh + kw * p.s_kw + (c0 + c) * p.s_c];
  }
  __syncthreads();

  const int ox = lane >> 3, c8 = lane & 7;
  const int c = c0 + c8;
  float acc[8];
  const float b = p.bias ? p.bias[c] : 0.f;
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = b;
  // residual / activity operands of the 8 outputs are requested BEFORE the tap loop (clamped rows,
  // unconditional) so their latency hides behind the 392 FMAs
  T* out = reinterpret_cast<T*>(p.out);
  const T* add = reinterpret_cast<const T*>(p.add);
  int rr[8];
  float av[8];
  uint8_t live[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) rr[o] = (o < TS && ox < 8) ? rowtab[(o + 3) * DW_HALO + (ox < TS ? ox : 0) + 3] : -1;
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const size_t ro = (size_t)(rr[o] < 0 ? 0 : rr[o]);
    av[o] = add ? ldf<T>(add + ro * C + c) : 0.f;
    live[o] = p.act ? p.act[ro] : 1;
  }
#pragma unroll 1                        // rolled: one kx-slab of LDS reads live at a time (54 VGPRs, 8 waves/SIMD);
  for (int kx = 0; kx < 7; ++kx) {      // fully unrolled, hipcc hoists all 98 reads first (156 VGPRs, 3 waves/SIMD)
    float w7[7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) w7[ky] = wl[(ky * 7 + kx) * 8 + c8];
#pragma unroll
    for (int y = 0; y < DW_HALO; ++y) {
      const float v = ldf<T>(tile + (y * DW_HALO + ox + kx) * 8 + c8);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int ky = y - o;
        if (ky >= 0 && ky < 7) acc[o] += w7[ky] * v;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    if (rr[o] >= 0 && ox < TS && chunk_ok) stf<T>(out + (size_t)rr[o] * C + c, live[o] ? acc[o] + av[o] : 0.f);
  }
}

// weight / bias gradient, persistent waves: blockIdx.x strides over tiles, blockIdx.y = channel
// chunk; result slab ws[blockIdx.x][50][C] (taps 0..48, then the bias row).
template <typename T>
__global__ __launch_bounds__(64, 5) void dwconv7_wgrad_w64_kernel(const DwWgP q) {
  __shared__ __attribute__((aligned(16))) T tile[DW_HP * 8];
  __shared__ int rowtab[DW_HP];
  __shared__ int invs[W64_MAXL];
  const int lane = threadIdx.x;
  const int ox = lane >> 3, c8 = lane & 7;
  const int C = q.C;
  const int c0 = blockIdx.y * 8;
  const int c = c0 + c8;
  const int TS = q.TP * q.g.S;
  const int tps = q.tiles_side * q.tiles_side;
  const T* dd = reinterpret_cast<const T*>(q.dd);

  float adw[49], adb = 0.f;
#pragma unroll
  for (int k = 0; k < 49; ++k) adw[k] = 0.f;

  for (int tile_id = blockIdx.x; tile_id < q.ntiles_total; tile_id += gridDim.x) {
    const int n = tile_id / tps, t = tile_id - n * tps;
    const int tyi = t / q.tiles_side, txi = t - tyi * q.tiles_side;
    const int ty0 = tyi * TS, tx0 = txi * TS;
    __syncthreads();
    w64_setup<T>(q.g, n, ty0, tx0, invs, rowtab);
    {
      const int oy = lane >> 3, oxx = lane & 7;
      const int valid = (oy < TS && oxx < TS && rowtab[(oy + 3) * DW_HALO + oxx + 3] >= 0);
      if (!__any(valid)) continue;
    }
    w64_load_tile<T>(reinterpret_cast<const T*>(q.x), rowtab, tile, C, c0);
    float g[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int r = (o < TS && ox < TS) ? rowtab[(o + 3) * DW_HALO + ox + 3] : -1;
      const float gv = ldf<T>(dd + (size_t)(r < 0 ? 0 : r) * C + c);      // unconditional (clamped) load
      g[o] = (r >= 0) ? gv : 0.f;
      adb += g[o];
    }
    __syncthreads();
    int toff = 0;                        // LDS offset made data-dependent on the previous kx-slab's FMAs, so that
#pragma unroll                           // hipcc cannot hoist all 98 LDS reads above the FMAs (static indices needed
    for (int kx = 0; kx < 7; ++kx) {     // for the 49 register accumulators forbid rolling this loop)
      if (kx > 0)
        asm volatile("" : "+v"(toff) : "v"(adw[kx - 1]), "v"(adw[7 + kx - 1]), "v"(adw[14 + kx - 1]),
                     "v"(adw[21 + kx - 1]), "v"(adw[28 + kx - 1]), "v"(adw[35 + kx - 1]), "v"(adw[42 + kx - 1]));
#pragma unroll
      for (int y = 0; y < DW_HALO; ++y) {
        const float v = ldf<T>(tile + toff + (y * DW_HALO + ox + kx) * 8 + c8);
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          const int ky = y - o;
          if (ky >= 0 && ky < 7) adw[ky * 7 + kx] += g[o] * v;
        }
      }
    }
  }
  // reduce over the 8 columns (lane bits 3..5), lanes 0..7 write the slab
  float* slab = q.ws + (size_t)blockIdx.x * 50 * C;
#pragma unroll
  for (int k = 0; k < 49; ++k) {
    float v = adw[k];
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (ox == 0) slab[k * C + c] = v;
  }
  adb += __shfl_xor(adb, 8, 64); adb += __shfl_xor(adb, 16, 64); adb += __shfl_xor(adb, 32, 64);
  if (ox == 0) slab[49 * C + c] = adb;
}
