// Submanifold depthwise 7x7 (MinkowskiDepthwiseConvolution, convnextv2_sparse.py:37-39) for the wide stages
// (S = 8 / 4 points per patch side, C = 40 / 80 at atto): BAND kernels.
//
// The per-sample kernels of dwconv6.cuh hold one sample's whole padded map for a CHUNK of 64/S channels, so a sample is
// five workgroups that each read 16 bytes out of every 80-byte row: rocprofv3 SQ counters at stage 0 (profiles/r02/
// sq_counters_dw.txt) show a wave living 44k cycles of which 30k are SQ_WAIT_ANY and 8k VALU - the kernel moves 50 MB in
// 60 us. Here a workgroup is (sample, band of BR patch rows) with ALL channels: a visible patch is S*S consecutive rows of
// C channels = one contiguous block of S*S*C*2 bytes (5 KB at stage 0), so the band and its 3-point halo above / below load
// as whole lines, every load of a thread is issued before its first LDS store, and the outputs of a patch column leave as
// runs of S*C*2 contiguous bytes. The inner loop is dwconv6's: a lane owns a channel pair and a patch column, S outputs
// slide down the column, every multiply-add is half a v_pk_fma_f32; masked neighbours read the zeros of the padded map.
// grid = (N, ceil(grid / BR)); block = 512. Used with BR = 1 at S = 8, C = 40 only (see the dispatch in capi.hip for the measurements).
#pragma once
#include "dwconv6.cuh"

template <int S, int C, int BR> struct DwBand {
  static constexpr int MW = 7 * S + 6, MH = BR * S + 6;                 // padded band: columns x rows of points
  static constexpr int VPR = C / 8;                                      // 16-byte vectors per row
  static constexpr size_t MAP_B = (size_t)MH * MW * C * 2, W_B = (size_t)49 * C * 4;
  static constexpr int MAXP = (BR + 2) * 7;                              // patch candidates (band + one patch row above / below)
    static constexpr size_t LDS = MAP_B + W_B + MAXP * 16 + 64;
};

template <int S, int C, int BR>
__global__ __launch_bounds__(512) void dwconv7_band_kernel(const DwP p) {
  using D = DwBand<S, C, BR>;
  using T = bf16_t;
  constexpr int MW = D::MW, MH = D::MH, VPR = D::VPR, CP = C / 2, G = 7;
  extern __shared__ __attribute__((aligned(16))) unsigned char dwb_smem[];
  T* map = reinterpret_cast<T*>(dwb_smem);
  float* wl = reinterpret_cast<float*>(dwb_smem + D::MAP_B);
  int4* plist = reinterpret_cast<int4*>(dwb_smem + D::MAP_B + D::W_B);           // {first global row, first map point, points, vec prefix}
  int* cnt = reinterpret_cast<int*>(plist + D::MAXP);                                // [0] patches listed, [1] vectors, [2] central patches
  const int tid = threadIdx.x, n = blockIdx.x, band = blockIdx.y;
  const int pr0 = band * BR, pr1 = min(G, pr0 + BR);                                 // central patch rows [pr0, pr1)
  const T* x = reinterpret_cast<const T*>(p.x);

  // ---- list of the visible patches that touch the band (wave 0; a prefix over <= 28 candidates)
  if (tid < 64) {
    const int cand = tid, prr = pr0 - 1 + cand / G, px = cand - (cand / G) * G;
    bool ok = cand < (BR + 2) * G && prr >= 0 && prr < G && prr <= pr1;
    int slot = -1;
    if (ok) { slot = p.g.inv ? p.g.inv[n * G * G + prr * G + px] : prr * G + px; ok = slot >= 0; }
    int iy0 = 0, niy = S;
    if (prr < pr0) { iy0 = S - 3; niy = 3; } else if (prr >= pr1) { iy0 = 0; niy = 3; }
    const int npts = ok ? niy * S : 0;
    // exclusive prefix of npts over the wave
    int pre = npts;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pre, o, 64); if (tid >= o) pre += t; }
    const unsigned long long m = __ballot(ok);
    const int idx = __popcll(m & ((1ull << tid) - 1ull));
    if (ok) {
      // map point of the patch's first loaded row: padded row (prr - pr0) * S + iy0 + 3, padded column px * S + 3
      plist[idx] = make_int4((n * p.g.keep + slot) * S * S + iy0 * S, ((prr - pr0) * S + iy0 + 3) * MW + px * S + 3, npts, (pre - npts) * VPR);
    }
    const unsigned long long mc = __ballot(ok && prr >= pr0 && prr < pr1);          // central (fully loaded, convolved) patches
    if (tid == 63) {
      cnt[0] = __popcll(m); cnt[1] = pre * VPR; cnt[2] = __popcll(mc);
      cnt[3] = mc ? __popcll(m & ((1ull << (__ffsll((long long)mc) - 1)) - 1ull)) : 0;
    }
  }
  // taps of all channels (flip for the data gradient), and the zero fill of the map under the list's latency
  for (int i = tid; i < 49 * C; i += 512) {
    const int k = i / C, c = i - k * C;
    int kh = k / 7, kw = k - kh * 7;
    if (p.flip) { kh = 6 - kh; kw = 6 - kw; }
    wl[i] = p.w[kh * p.s_kh + kw * p.s_kw + c * p.s_c];
  }
  {
    uint4* m4 = reinterpret_cast<uint4*>(map);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < (int)(D::MAP_B / 16); i += 512) m4[i] = z;
  }
  __syncthreads();
  // ---- every vector of the band: wave w copies patches w, w + 8, ...; all of a wave's loads are issued before its first
  // LDS store (one exposed memory latency), and the list entry of a patch is wave-uniform (no per-vector search)
  const int npl = cnt[0];
  const int wave = tid >> 6, lane = tid & 63;
  constexpr int PPW = (D::MAXP + 7) / 8, VPP = (S * S * VPR + 63) / 64;
  uint4 val[PPW][VPP];
  int dst[PPW][VPP];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int j = wave + 8 * k;
    const int4 e = plist[j < npl ? j : 0];
    const int nv = j < npl ? e.z * VPR : 0;
#pragma unroll
    for (int i = 0; i < VPP; ++i) {
      const int v = lane + 64 * i;
      dst[k][i] = -1;
      val[k][i] = make_uint4(0u, 0u, 0u, 0u);
      if (v < nv) {
        const int r = v / VPR, kk = v - r * VPR;                       // point within the patch's loaded rows, vector within the row
        const int iy = r / S, ix = r - iy * S;
        dst[k][i] = (e.y + iy * MW + ix) * C + kk * 8;
        val[k][i] = *reinterpret_cast<const uint4*>(x + (size_t)(e.x + r) * C + kk * 8);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < PPW; ++k)
#pragma unroll
    for (int i = 0; i < VPP; ++i)
      if (dst[k][i] >= 0) *reinterpret_cast<uint4*>(map + dst[k][i]) = val[k][i];
  __syncthreads();

  // ---- convolution: item = (central visible patch, patch column ox, channel pair cp); consecutive lanes = consecutive cp
  T* out = reinterpret_cast<T*>(p.out);
  const T* add = reinterpret_cast<const T*>(p.add);
  const unsigned add_m = opaque_mask(add != nullptr), act_m = opaque_mask(p.act != nullptr) & 0xffu;
  const int ncen = cnt[2], first_cen = cnt[3];                              // central patches are a contiguous run of the list
  const int items = ncen * S * CP;
  for (int it = tid; it < items; it += 512) {
    const int j = it / (S * CP), rem = it - j * (S * CP), ox = rem / CP, cp = rem - ox * CP;
    const int4 e = plist[first_cen + j];
    const T* tile = map + (size_t)(e.y - 3 * MW - 3 + ox) * C + 2 * cp;     // halo origin of the patch + this lane's column
    const size_t r0 = (size_t)e.x + ox;
    f32x2_t acc[S];
    f32x2_t b2 = {0.f, 0.f};
    if (p.bias) { b2.x = p.bias[2 * cp]; b2.y = p.bias[2 * cp + 1]; }
    uint32_t addraw[S];
    uint8_t live[S];
#pragma unroll
    for (int o = 0; o < S; ++o) {
      // optional operands through a pointer select (unconditional loads; a branch here parks a wait in front of the tap loop)
      const uint32_t ar = *reinterpret_cast<const uint32_t*>(add ? add + (r0 + o * S) * C + 2 * cp : reinterpret_cast<const T*>(p.w));
      const uint8_t lv = *(p.act ? p.act + r0 + o * S : reinterpret_cast<const uint8_t*>(p.w));
      addraw[o] = ar & add_m;                                   // opaque masks (see opaque_mask): a ternary on `add` is turned back
      live[o] = (uint8_t)((lv & act_m) | (~act_m & 1u));        // into a branch around the load by the optimizer
      acc[o] = b2;
    }
#pragma unroll 1
    for (int kx = 0; kx < 7; ++kx) {
      f32x2_t w7[7];
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) w7[ky] = *reinterpret_cast<const f32x2_t*>(wl + (ky * 7 + kx) * C + 2 * cp);
      // the whole input column of this kx FIRST (S + 6 independent LDS reads in flight), then the multiply-adds: written as one loop
      // hipcc reuses ONE destination register for the reads and parks an s_waitcnt lgkmcnt(0) behind every one of them - the column
      // became S + 6 serial LDS round trips per kx (round 4, found in the ISA; the same shape in every depthwise kernel)
      uint32_t raw[S + 6];
#pragma unroll
      for (int y = 0; y < S + 6; ++y) raw[y] = *reinterpret_cast<const uint32_t*>(tile + (size_t)(y * MW + kx) * C);
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
      *reinterpret_cast<uint32_t*>(out + (r0 + o * S) * C + 2 * cp) = live[o] ? f2bf2(r.x, r.y) : 0u;
    }
  }
}
