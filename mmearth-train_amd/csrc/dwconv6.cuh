// Per-sample LDS-resident depthwise 7x7, packed-math variant (bf16 storage only), v6.
//
// Same staging as v5 (dwconv5.cuh): one sample's padded map for CW = 64/S channels in LDS. The
// inner loop of v5 spends 1.6-2.6 VALU/LDS instructions per multiply-add (a 2-byte LDS read and a
// shift per input value, scalar FMAs). Here a lane owns TWO adjacent channels:
//   lane = (cp, ox, sub): channel pair cp of the chunk, column ox of a patch, and the wave works on
//   two visible patches at once (sub = lane / 32);
//   one ds_read_b32 brings both channels' bf16 values, one ds_read_b64 both channels' tap weights,
//   and every multiply-add is half of a v_pk_fma_f32 -> 0.9-1.6 instructions per multiply-add.
#pragma once
#include "dwconv5.cuh"


__device__ __forceinline__ f32x2_t bf2x2_to_f2(uint32_t raw) {
  f32x2_t v;
  v.x = __uint_as_float(raw << 16);
  v.y = __uint_as_float(raw & 0xffff0000u);
  return v;
}

// grid = (N samples, C / CW); block = 64 * NW
template <int S, int GC = 0>
__global__ __launch_bounds__(512) void dwconv7_v6_kernel(const DwP p) {
  using T = bf16_t;
  constexpr int CW = 64 / S, CP = CW / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char dw5_smem[];
  const int MS = (GC ? GC : p.g.grid) * S + 6;      // GC = compile-time patch-grid side: LDS offsets become immediates
  T* map = reinterpret_cast<T*>(dw5_smem);
  float* wl = reinterpret_cast<float*>(dw5_smem + dw5_map_bytes<T, S>(p.g.grid));
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, NW = blockDim.x >> 6;
  const int C = p.C, n = blockIdx.x, c0 = blockIdx.y * CW;
  // Prologue: every global load goes out first (the sample's visible rows, the taps); the zero fill of the map runs
  // under their latency and one barrier later both land in LDS - one exposed memory latency instead of two.
  using D = Dw5<T, S>;
  constexpr int U = 4, WMAX = (49 * CW + 255) / 256;
  const int items = p.g.keep * S * S * D::VPL;
  const T* x = reinterpret_cast<const T*>(p.x);
  // A row vector's LDS destination needs the patch index (a `vis` lookup), its source address does not: the lookup and the data
  // load are both issued by row_ld (the lookup through a pointer select, no branch) and the destination is worked out by row_dst
  // AFTER all of a thread's loads are out. Computed inside row_ld, every vector cost two dependent round trips (lookup -> wait ->
  // destination -> data load), eight in a row at the top of every workgroup of the stage-2 kernel.
  auto row_ld = [&](int it, int& patch_raw) -> uint4 {    // by value: arrays captured by reference ended up in scratch
    const int itc = it < items ? it : 0;
    const int v = itc % D::VPL, pt = itc / D::VPL;
    const int slot = pt / (S * S), q = pt - slot * (S * S);
    patch_raw = *(p.g.vis ? p.g.vis + n * p.g.keep + slot : reinterpret_cast<const int*>(p.w));
    return *reinterpret_cast<const uint4*>(x + ((size_t)(n * p.g.keep + slot) * (S * S) + q) * C + c0 + v * D::EPV);
  };
  auto row_dst = [&](int it, int patch_raw) -> int {
    if (it >= items) return -1;
    const int v = it % D::VPL, pt = it / D::VPL;
    const int slot = pt / (S * S), q = pt - slot * (S * S);
    const int iy = q / S, ix = q - iy * S;
    const int patch = p.g.vis ? patch_raw : slot;
    const int py = patch / p.g.grid, px = patch - py * p.g.grid;
    return ((py * S + iy + 3) * MS + px * S + ix + 3) * CW + v * D::EPV;
  };
  auto row_st = [&](const uint4& v, int dst) { if (dst >= 0) *reinterpret_cast<uint4*>(map + dst) = v; };
  auto tap = [&](int i) {
    const int k = i / CW, cc = i - k * CW;
    int kh = k / 7, kw = k - kh * 7;
    if (p.flip) { kh = 6 - kh; kw = 6 - kw; }
    return p.w[kh * p.s_kh + kw * p.s_kw + (c0 + cc) * p.s_c];
  };
  static_assert(U == 4, "four rows in flight per thread");
  int q0, q1, q2, q3;
  const uint4 v0 = row_ld(tid, q0), v1 = row_ld(tid + blockDim.x, q1), v2 = row_ld(tid + 2 * blockDim.x, q2),
              v3 = row_ld(tid + 3 * blockDim.x, q3);
  float wv[WMAX];
#pragma unroll
  for (int u = 0; u < WMAX; ++u) {
    const int i = tid + u * blockDim.x;
    wv[u] = tap(i < 49 * CW ? i : 0);
  }
  {
    uint4* m4 = reinterpret_cast<uint4*>(dw5_smem);
    const int nvec = (int)(dw5_map_bytes<T, S>(p.g.grid) / 16);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < nvec; i += blockDim.x) m4[i] = z;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < WMAX; ++u) {
    const int i = tid + u * blockDim.x;
    if (i < 49 * CW) wl[i] = wv[u];
  }
  for (int i = tid + WMAX * blockDim.x; i < 49 * CW; i += blockDim.x) wl[i] = tap(i);      // blocks under 256 threads
  row_st(v0, row_dst(tid, q0)); row_st(v1, row_dst(tid + blockDim.x, q1)); row_st(v2, row_dst(tid + 2 * blockDim.x, q2));
  row_st(v3, row_dst(tid + 3 * blockDim.x, q3));
  for (int it = tid + blockDim.x * U; it < items; it += blockDim.x) { int q; const uint4 v = row_ld(it, q); row_st(v, row_dst(it, q)); }
  __syncthreads();

  const int cp = lane % CP, ox = (lane / CP) % S, sub = lane / (CP * S);      // CP * S == 32
  const int c = c0 + 2 * cp;
  T* out = reinterpret_cast<T*>(p.out);
  const T* add = reinterpret_cast<const T*>(p.add);
  const unsigned add_m = opaque_mask(add != nullptr), act_m = opaque_mask(p.act != nullptr) & 0xffu;
  f32x2_t b2 = {0.f, 0.f};
  if (p.bias) { b2.x = p.bias[c]; b2.y = p.bias[c + 1]; }

  // the patch lookup, residual rows and activity bytes of the NEXT slot are requested before the taps of the current one: as
  // `lookup -> wait -> tile; add / act -> wait` at the top of every iteration they were two dependent round trips per slot
  // (tools/isa_chain.py: GL W(vm0) inside the slot loop; 2-4 slots per wave)
  int patchN = 0;
  uint32_t addN[S];
  uint8_t liveN[S];
  auto fetch = [&](int slot_) {
    const int nk_ = n * p.g.keep + min(slot_, p.g.keep - 1);
    patchN = *(p.g.vis ? p.g.vis + nk_ : reinterpret_cast<const int*>(p.w));
    const size_t r0_ = (size_t)nk_ * (S * S) + ox;
#pragma unroll
    for (int o = 0; o < S; ++o) {
      // optional operands through a pointer select (unconditional loads; a branch here parks a wait in front of the tap loop)
      const uint32_t ar = *reinterpret_cast<const uint32_t*>(add ? add + (r0_ + o * S) * C + c : reinterpret_cast<const T*>(p.w));
      const uint8_t lv = *(p.act ? p.act + r0_ + o * S : reinterpret_cast<const uint8_t*>(p.w));
      addN[o] = ar & add_m;                                   // opaque masks (see opaque_mask): a ternary on `add` is turned back
      liveN[o] = (uint8_t)((lv & act_m) | (~act_m & 1u));     // into a branch around the load by the optimizer
    }
  };
  if (wave * 2 + sub < p.g.keep) fetch(wave * 2 + sub);
  for (int slot = wave * 2 + sub; slot < p.g.keep; slot += NW * 2) {
    const int nk = n * p.g.keep + slot;
    const int patch = p.g.vis ? patchN : slot;
    const int py = patch / p.g.grid, px = patch - py * p.g.grid;
    const T* tile = map + ((size_t)(py * S) * MS + px * S + ox) * CW + 2 * cp;     // halo origin + this lane's column
    const size_t r0 = (size_t)nk * (S * S) + ox;
    f32x2_t acc[S];
    uint32_t addraw[S];
    uint8_t live[S];
#pragma unroll
    for (int o = 0; o < S; ++o) { addraw[o] = addN[o]; live[o] = liveN[o]; acc[o] = b2; }
    if (slot + NW * 2 < p.g.keep) fetch(slot + NW * 2);
#pragma unroll 1
    for (int kx = 0; kx < 7; ++kx) {
      f32x2_t w7[7];
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) w7[ky] = *reinterpret_cast<const f32x2_t*>(wl + (ky * 7 + kx) * CW + 2 * cp);
      uint32_t raw[S + 6];      // the whole input column first: S + 6 independent LDS reads in flight (hipcc serialises them through one register otherwise)
#pragma unroll
      for (int y = 0; y < S + 6; ++y) raw[y] = *reinterpret_cast<const uint32_t*>(tile + (y * MS + kx) * CW);
#pragma unroll
      for (int y = 0; y < S + 6; ++y) {
        const f32x2_t v = bf2x2_to_f2(raw[y]);
#pragma unroll
        for (int o = 0; o < S; ++o) {
          const int ky = y - o;
          if (ky >= 0 && ky < 7) acc[o] = w7[ky] * v + acc[o];
        }
      }
    }
#pragma unroll
    for (int o = 0; o < S; ++o) {
      const f32x2_t a = bf2x2_to_f2(addraw[o]);
      const f32x2_t r = acc[o] + a;
      *reinterpret_cast<uint32_t*>(out + (r0 + o * S) * C + c) = live[o] ? f2bf2(r.x, r.y) : 0u;
    }
  }
}

// (a packed weight-gradient twin of this kernel - 98 accumulator registers per lane - measured slower than dwconv7_wgrad_v5 in round 2 (62 vs 45 us at stage 1),
//  stayed selectable as MPMAE_OPT_DWW = 6 until round 6: removed. The memory access fault its bs-256 run ended in was not the kernel's: the one-by-one
//  branch of mpmae_dwconv7_wgrad_group handed a stack copy of the arguments to an entry point whose recorded launch read them at replay - capi.hip, "A")

// ---------------------------------------------------------------------------------------------
// S = 1 (stage 3: 19 of the GxG positions visible; dense decoder: all of them), G = 7:
// one wave = one sample x 16 channels; the whole (G+6)^2 padded map is 5.4 KB of LDS per wave.
// lane = (cp 0..7, ox 0..G-1): a lane produces the full output COLUMN ox (G outputs) for two
// channels, so every input value read from LDS feeds up to 7 packed FMAs; outputs of masked
// positions are simply not stored. grid = (N, C/64), block = 256 (4 independent channel chunks).
// ---------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void dwconv7_v6s1_kernel(const DwP p) {
  using T = bf16_t;
  constexpr int CW = 16, MS = G + 6, MAPB = ((MS * MS * CW * 2 + 15) / 16) * 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * (MAPB + 49 * CW * 4)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  T* map = reinterpret_cast<T*>(smem + wave * (MAPB + 49 * CW * 4));
  float* wl = reinterpret_cast<float*>(smem + wave * (MAPB + 49 * CW * 4) + MAPB);
  const int C = p.C, n = blockIdx.x, c0 = (blockIdx.y * 4 + wave) * CW;
  const bool chunk_ok = c0 < C;
  const int cc0 = chunk_ok ? c0 : 0;
  // every global load of the prologue first (taps, the sample's visible rows, the slot table of this lane's output
  // column); the zero fill runs under their latency
  float wv[13];                      // 49*16 = 784 taps: 13 loads per lane, all in flight together
#pragma unroll
  for (int u = 0; u < 13; ++u) {
    const int i = lane + 64 * u, ic = i < 49 * CW ? i : 0;
    const int k = ic / CW, cc = ic - k * CW;
    int kh = k / 7, kw = k - kh * 7;
    if (p.flip) { kh = 6 - kh; kw = 6 - kw; }
    wv[u] = p.w[kh * p.s_kh + kw * p.s_kw + (cc0 + cc) * p.s_c];
  }
  const T* x = reinterpret_cast<const T*>(p.x);
  // lookups (vis, inv) and row vectors are all REQUESTED first, through pointer selects; the LDS destinations / row numbers that
  // depend on the lookups are worked out afterwards (see dwconv7_v6_kernel)
  auto row_ld = [&](int it, int& patch_raw) -> uint4 {    // G*G*2 <= 128 items: at most 2 per lane
    const int slot = it < p.g.keep * 2 ? it >> 1 : 0;
    patch_raw = *(p.g.vis ? p.g.vis + n * p.g.keep + slot : reinterpret_cast<const int*>(p.w));
    return *reinterpret_cast<const uint4*>(x + (size_t)(n * p.g.keep + slot) * C + cc0 + (it & 1) * 8);
  };
  auto row_dst = [&](int it, int patch_raw) -> int {
    if (it >= p.g.keep * 2) return -1;
    const int patch = p.g.vis ? patch_raw : it >> 1;
    const int py = patch / G, px = patch - py * G;
    return ((py + 3) * MS + px + 3) * CW + (it & 1) * 8;
  };
  int q0, q1;
  const uint4 v0 = row_ld(lane, q0), v1 = row_ld(lane + 64, q1);
  const int cp = lane & 7, oxr = lane >> 3, ox = oxr < G ? oxr : 0;
  const int c = cc0 + 2 * cp;
  int rows[G];
#pragma unroll
  for (int o = 0; o < G; ++o) rows[o] = *(p.g.inv ? p.g.inv + n * G * G + o * G + ox : reinterpret_cast<const int*>(p.w));
  const int d0 = row_dst(lane, q0), d1 = row_dst(lane + 64, q1);
#pragma unroll
  for (int o = 0; o < G; ++o) {
    const int slot = p.g.inv ? rows[o] : o * G + ox;
    rows[o] = slot >= 0 ? n * p.g.keep + slot : -1;
  }
  {
    uint4* m4 = reinterpret_cast<uint4*>(map);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int i = lane; i < MAPB / 16; i += 64) m4[i] = z;
  }
  T* out = reinterpret_cast<T*>(p.out);
  const T* add = reinterpret_cast<const T*>(p.add);
  const unsigned add_m = opaque_mask(add != nullptr);
  f32x2_t b2 = {0.f, 0.f};
  if (p.bias) { b2.x = p.bias[c]; b2.y = p.bias[c + 1]; }
  uint32_t addraw[G];
  uint8_t actb[G];                   // activity bytes of the output rows: requested HERE with the residual rows - loaded in the store loop
  f32x2_t acc[G];                    // (`live = !act || act[row]` per output) they were G dependent round trips at the end of the kernel
#pragma unroll
  for (int o = 0; o < G; ++o) {
    const uint32_t ar = *reinterpret_cast<const uint32_t*>(add ? add + (size_t)max(rows[o], 0) * C + c : reinterpret_cast<const T*>(p.w));
    actb[o] = *(p.act ? p.act + max(rows[o], 0) : reinterpret_cast<const uint8_t*>(p.w));
    addraw[o] = ar & add_m & (rows[o] >= 0 ? 0xffffffffu : 0u);   // unconditional load (pointer select + opaque mask)
    acc[o] = b2;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 13; ++u) if (lane + 64 * u < 49 * CW) wl[lane + 64 * u] = wv[u];
  if (d0 >= 0) *reinterpret_cast<uint4*>(map + d0) = v0;
  if (d1 >= 0) *reinterpret_cast<uint4*>(map + d1) = v1;
  __syncthreads();
  if (oxr >= G || !chunk_ok) return;
  const T* tile = map + ox * CW + 2 * cp;
#pragma unroll 1
  for (int kx = 0; kx < 7; ++kx) {
    f32x2_t w7[7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) w7[ky] = *reinterpret_cast<const f32x2_t*>(wl + (ky * 7 + kx) * CW + 2 * cp);
    uint32_t raw[G + 6];      // column first: independent LDS reads in flight (hipcc serialises them through one register otherwise)
#pragma unroll
    for (int y = 0; y < G + 6; ++y) raw[y] = *reinterpret_cast<const uint32_t*>(tile + (y * MS + kx) * CW);
#pragma unroll
    for (int y = 0; y < G + 6; ++y) {
      const f32x2_t v = bf2x2_to_f2(raw[y]);
#pragma unroll
      for (int o = 0; o < G; ++o) {
        const int ky = y - o;
        if (ky >= 0 && ky < 7) acc[o] = w7[ky] * v + acc[o];
      }
    }
  }
#pragma unroll
  for (int o = 0; o < G; ++o) {
    if (rows[o] < 0) continue;
    const bool live = !p.act || actb[o] != 0;
    const f32x2_t r = acc[o] + bf2x2_to_f2(addraw[o]);
    *reinterpret_cast<uint32_t*>(out + (size_t)rows[o] * C + c) = live ? f2bf2(r.x, r.y) : 0u;
  }
}

// S = 1 weight / bias gradient, same mapping as dwconv7_v6s1_kernel: wave = 16 channels, lane = (cp, ox);
// persistent over samples (n = blockIdx.x, += gridDim.x), 49 packed tap accumulators per lane, the 7 column
// lanes of a channel pair folded through LDS at the end -> slab ws[blockIdx.x][50][C].
// grid = (nblocks, C/64), block = 256 (4 independent channel chunks).
template <int G>
__global__ __launch_bounds__(256) void dwconv7_wgrad_v6s1_kernel(const DwWgP q, const DwWgGroupP grp) {
  using T = bf16_t;
  constexpr int CW = 16, MS = G + 6, MAPB = ((MS * MS * CW * 2 + 15) / 16) * 16;
  constexpr int REDB = 8 * 10 * CW * 4;             // fold buffer per wave: [8 ox lanes][10 taps][16] floats, 5 passes
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * (MAPB > REDB ? MAPB : REDB)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  unsigned char* wbase = smem + wave * (MAPB > REDB ? MAPB : REDB);
  T* map = reinterpret_cast<T*>(wbase);
  const int C = q.C, c0 = (blockIdx.y * 4 + wave) * CW;
  const bool chunk_ok = c0 < C;
  const int cc0 = chunk_ok ? c0 : 0;
  const int cp = lane & 7, ox = lane >> 3;
  const bool col_ok = ox < G;
  const int c = cc0 + 2 * cp;
  const void *xv, *ddv;
  float* wsv;
  dwwg_select(q, grp, xv, ddv, wsv);
  const T* x = reinterpret_cast<const T*>(xv);
  const T* dd = reinterpret_cast<const T*>(ddv);
  {
    uint4* m4 = reinterpret_cast<uint4*>(map);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int i = lane; i < MAPB / 16; i += 64) m4[i] = z;
  }
  f32x2_t adw[49], adb = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 49; ++k) adw[k] = (f32x2_t){0.f, 0.f};
  const int items = q.g.keep * 2;
  for (int n = blockIdx.x; n < q.g.N; n += gridDim.x) {
    // stage this sample's visible rows (each wave its own 16 channels): wave-private LDS, no block barrier needed
    int dst[2];
    uint4 val[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = lane + 64 * u;
      const bool ok = it < items;
      const int v = it & 1, slot = ok ? it >> 1 : 0;
      const int patch = q.g.vis ? q.g.vis[n * q.g.keep + slot] : slot;
      const int py = patch / G, px = patch - py * G;
      val[u] = *reinterpret_cast<const uint4*>(x + (size_t)(n * q.g.keep + slot) * C + cc0 + v * 8);
      dst[u] = ok ? ((py + 3) * MS + px + 3) * CW + v * 8 : -1;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) if (dst[u] >= 0) *reinterpret_cast<uint4*>(map + dst[u]) = val[u];
    __builtin_amdgcn_s_waitcnt(0);                 // the wave's LDS writes are complete before its own reads
    __builtin_amdgcn_wave_barrier();
    f32x2_t g[G];
#pragma unroll
    for (int o = 0; o < G; ++o) {
      const int patch = o * G + (col_ok ? ox : 0);
      const int slot = q.g.inv ? q.g.inv[n * G * G + patch] : patch;
      const int row = slot >= 0 ? n * q.g.keep + slot : 0;
      const uint32_t raw = *reinterpret_cast<const uint32_t*>(dd + (size_t)row * C + c);
      g[o] = (slot >= 0 && col_ok) ? bf2x2_to_f2(raw) : (f32x2_t){0.f, 0.f};
      adb += g[o];
    }
    const T* tile = map + (col_ok ? ox : 0) * CW + 2 * cp;
    int toff = 0;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      if (kx > 0)
        asm volatile("" : "+v"(toff) : "v"(adw[kx - 1].x), "v"(adw[7 + kx - 1].x), "v"(adw[14 + kx - 1].x),
                     "v"(adw[21 + kx - 1].x), "v"(adw[28 + kx - 1].x), "v"(adw[35 + kx - 1].x), "v"(adw[42 + kx - 1].x));
      uint32_t raw[G + 6];      // column first: independent LDS reads in flight (hipcc serialises them through one register otherwise)
#pragma unroll
      for (int y = 0; y < G + 6; ++y) raw[y] = *reinterpret_cast<const uint32_t*>(tile + toff + (y * MS + kx) * CW);
#pragma unroll
      for (int y = 0; y < G + 6; ++y) {
        const f32x2_t v = bf2x2_to_f2(raw[y]);
#pragma unroll
        for (int o = 0; o < G; ++o) {
          const int ky = y - o;
          if (ky >= 0 && ky < 7) adw[ky * 7 + kx] = g[o] * v + adw[ky * 7 + kx];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 2; ++u) if (dst[u] >= 0) *reinterpret_cast<uint4*>(map + dst[u]) = make_uint4(0u, 0u, 0u, 0u);
  }
  // fold the G column lanes of every channel pair through the wave's LDS region
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  float* red = reinterpret_cast<float*>(wbase);           // [8][10][16]
  float* slab = wsv + (size_t)blockIdx.x * 50 * C;
#pragma unroll
  for (int kb = 0; kb < 5; ++kb) {
#pragma unroll
    for (int kk = 0; kk < 10; ++kk) {
      const int k = kb * 10 + kk;
      *reinterpret_cast<f32x2_t*>(red + (ox * 10 + kk) * CW + 2 * cp) = (k < 49) ? adw[k < 49 ? k : 0] : adb;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (chunk_ok) {
      for (int i = lane; i < 10 * CW; i += 64) {
        float v = 0.f;
#pragma unroll
        for (int o = 0; o < G; ++o) v += red[o * 10 * CW + i];
        const int kk = i / CW, cc = i - kk * CW;
        slab[(kb * 10 + kk) * C + c0 + cc] = v;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
}
