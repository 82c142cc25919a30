// Weight / bias gradient of the submanifold depthwise 7x7 on the MATRIX cores, S = 8 (round 4; forward / data gradient: dwmfma.cuh).
//
//   dW_c[ky][kx] = sum over patches p, rows y, columns x of  X_c[p; y + ky - 3][x + kx - 3] * D_c[p; y][x]
//
// (X the block input with its 3-point halo from the neighbour patches, D the gradient at the depthwise output.) Per channel and kx
// this is a matrix product with the CONTRACTION over (patch, column):
//      G_kx[v][y] = sum_{p, x}  A[v][(p, x)] * B[(p, x)][y],   A = X_c[p; window row v - 3][x + kx - 3],  B = D_c[p; y][x]
// and the taps are the diagonals dW[ky][kx] = sum_y G_kx[y + ky][y]. One MFMA 16x16x32 = 4 patches x 8 columns of contraction,
// 16 window rows (14 used) x 8 rows of D (the other 8 columns of the tile idle): 7 MFMAs per channel and 4 patches - the count of
// the forward kernel. Both operands are read in channel-PLANAR form from LDS ([channel][row][slot][8 columns], written by the same
// register transpose as dwmfma.cuh): a lane's B fragment is one 16-byte read, its A fragments for ALL 7 kx come from ONE 16-column
// window row (left.1 | centre | right.0 pieces = 8 dwords) by register selection and v_alignbit for the odd shifts. The bias gradient
// rides in the idle window row 15 of the kx = 0 product (A row = ones).
//
// A workgroup = (sample(s), chunk of CCH channels); the planes of X stay resident (102 KB at C = 40), D goes through in parts of 8
// patches (41 KB). A wave owns a channel quad: 4 x 7 accumulators (112 registers) live for
// the whole kernel. At the end the accumulator tiles are parked in LDS (the planes are dead), a thread per (tap, channel) sums its
// diagonal, and the result leaves as one slab
// ws[workgroup][50][C] - the layout of the VALU kernels (dwconv5.cuh), folded by the same second stage.
// grid = (N, C / CCH, problems of a group), block = 64 * CCH / 4.
#pragma once
#include "dwmfma.cuh"

template <int CCH> struct DwMfmaWg {
  static constexpr int NQ = CCH / 4, NT = 64 * NQ, NV = CCH / 8, DPS = 8;
  static size_t lds(int keep) {
    const size_t planes = (size_t)CCH * 8 * (keep + 1) * 16 + 16 * NV + (size_t)CCH * 8 * DPS * 16 + 16 * NV + (size_t)(keep + 1) * 9 * 4 + 64 * 4 + 64 * 4;
    const size_t tiles = (size_t)CCH * 7 * 129 * 4;          // the accumulator tiles parked at the end (few visible patches: larger than the planes)
    return planes > tiles ? planes : tiles;
  }
};

template <int CCH>
__global__ __launch_bounds__(64 * (CCH / 4)) void dwconv7_wgrad_mfma_kernel(const DwWgP qa, const DwWgGroupP grp) {
  using D = DwMfmaWg<CCH>;
  constexpr int S = 8, NT = D::NT, NV = D::NV, DPS = D::DPS;
  constexpr int DROWB = DPS * 16, DPLB = 8 * DROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char dww_smem[];
  const int keep = qa.g.keep, SL = keep + 1, G = qa.g.grid, L = G * G;
  const int ROWB = SL * 16, PLB = 8 * ROWB;
  const int XOFF = 0, DOFF = CCH * PLB + 16 * NV;
  unsigned char* xpl = dww_smem + XOFF;
  unsigned char* dpl = dww_smem + DOFF;
  int* nbt = reinterpret_cast<int*>(dpl + CCH * DPLB + 16 * NV);
  int* invl = nbt + (keep + 1) * 9;
  int* visl = invl + 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c0 = blockIdx.y * CCH, C = qa.C, q = wave;
  const void *xv, *ddv;
  float* wsv;
  dwwg_select(qa, grp, xv, ddv, wsv);
  const bf16_t* x = reinterpret_cast<const bf16_t*>(xv);
  const bf16_t* dd = reinterpret_cast<const bf16_t*>(ddv);

  for (int i = tid; i < CCH * 16; i += NT) {                       // zero granules of the X planes (slot = keep), once
    const int xp = i & 1, y = (i >> 1) & 7, c = i >> 4;
    *reinterpret_cast<uint2*>(xpl + c * PLB + 16 * (c >> 3) + y * ROWB + keep * 16 + xp * 8) = make_uint2(0u, 0u);
  }
  const int nparts = (keep + DPS - 1) / DPS;
  const int xtasks = keep * 16 * NV;
  constexpr int U = 3;
  auto transpose_store = [&](unsigned char* base, int plb, const uint4& a0, const uint4& a1, const uint4& a2, const uint4& a3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
      const uint32_t lo = __builtin_amdgcn_perm(dwm_dw(a1, j >> 1), dwm_dw(a0, j >> 1), sel);
      const uint32_t hi = __builtin_amdgcn_perm(dwm_dw(a3, j >> 1), dwm_dw(a2, j >> 1), sel);
      *reinterpret_cast<uint2*>(base + j * plb) = make_uint2(lo, hi);
    }
  };
  // rows of D part `part` of sample n: one task (4 points x 8 channels) per thread
  auto d_src = [&](int n, int part) -> const bf16_t* {
    const int o = tid % NV, r1 = tid / NV, xp = r1 & 1, y = (r1 >> 1) & 7, ls = r1 >> 4;
    const int sl = min(part * DPS + ls, keep - 1);
    return dd + ((size_t)(n * keep + sl) * 64 + y * 8 + 4 * xp) * C + c0 + 8 * o;
  };

  const int n = blockIdx.x;                                        // ONE sample per workgroup (grid.x = N): the 112 accumulator registers
  f32x4_t acc[4][7];                                               // and the 48 of the X rows in flight are never live together
  {
    __syncthreads();
    DWM_STAMP(0);
    // ---- every global load first: geometry, the sample's rows of X, the first part of D
    const int inv_v = qa.g.inv[n * L + (tid < L ? tid : 0)];
    const int vis_v = qa.g.vis[n * keep + (tid < keep ? tid : 0)];
    uint4 v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tk = u * NT + tid, tc = tk < xtasks ? tk : 0;
      const int o = tc % NV, r1 = tc / NV, xp = r1 & 1, r2 = r1 >> 1, y = r2 & 7, slot = r2 >> 3;
      const bf16_t* src = x + ((size_t)(n * keep + slot) * 64 + y * 8 + 4 * xp) * C + c0 + 8 * o;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[u][r] = *reinterpret_cast<const uint4*>(src + (size_t)r * C);
    }
    uint4 dv0, dv1, dv2, dv3;
    {
      const bf16_t* src = d_src(n, 0);
      dv0 = *reinterpret_cast<const uint4*>(src); dv1 = *reinterpret_cast<const uint4*>(src + C);
      dv2 = *reinterpret_cast<const uint4*>(src + 2 * (size_t)C); dv3 = *reinterpret_cast<const uint4*>(src + 3 * (size_t)C);
    }
    DWM_STAMP(1);
    if (tid < L) invl[tid] = inv_v;
    if (tid < keep) visl[tid] = vis_v;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // invl / visl: written and read by wave 0 only
    if (tid <= keep) {
      const int patch = visl[tid < keep ? tid : 0];
      const int py = patch / G, px = patch - py * G;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int yy = py + k / 3 - 1, xx = px + k % 3 - 1;
        const bool in = yy >= 0 && yy < G && xx >= 0 && xx < G;
        const int sl = invl[in ? yy * G + xx : patch];
        nbt[tid * 9 + k] = (tid < keep && in && sl >= 0) ? sl : keep;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tk = u * NT + tid;
      const int o = tk % NV, r1 = tk / NV, xp = r1 & 1, r2 = r1 >> 1, y = r2 & 7, slot = r2 >> 3;
      if (tk < xtasks) transpose_store(xpl + (8 * o) * PLB + 16 * o + y * ROWB + slot * 16 + xp * 8, PLB, v[u][0], v[u][1], v[u][2], v[u][3]);
    }
    for (int tk = U * NT + tid; tk < xtasks; tk += NT) {           // (more than 3 tasks per thread: rare)
      const int o = tk % NV, r1 = tk / NV, xp = r1 & 1, r2 = r1 >> 1, y = r2 & 7, slot = r2 >> 3;
      const bf16_t* src = x + ((size_t)(n * keep + slot) * 64 + y * 8 + 4 * xp) * C + c0 + 8 * o;
      const uint4 w0 = *reinterpret_cast<const uint4*>(src), w1 = *reinterpret_cast<const uint4*>(src + C);
      const uint4 w2 = *reinterpret_cast<const uint4*>(src + 2 * (size_t)C), w3 = *reinterpret_cast<const uint4*>(src + 3 * (size_t)C);
      transpose_store(xpl + (8 * o) * PLB + 16 * o + y * ROWB + slot * 16 + xp * 8, PLB, w0, w1, w2, w3);
    }

    DWM_STAMP(2);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) acc[cc][kx] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int part = 0; part < nparts; ++part) {
      if (part > 0) {                                              // (requested here, not one part ahead: under the MFMAs the request only moves its 2 000-3 000 cycles of issue back-pressure into that phase - 24.4 vs 24.1 us)
        const bf16_t* src = d_src(n, part);
        dv0 = *reinterpret_cast<const uint4*>(src); dv1 = *reinterpret_cast<const uint4*>(src + C);
        dv2 = *reinterpret_cast<const uint4*>(src + 2 * (size_t)C); dv3 = *reinterpret_cast<const uint4*>(src + 3 * (size_t)C);
      }
      {   // D part planes from the registers
        const int o = tid % NV, r1 = tid / NV, xp = r1 & 1, y = (r1 >> 1) & 7, ls = r1 >> 4;
        if (ls < DPS) transpose_store(dpl + (8 * o) * DPLB + 16 * o + y * DROWB + ls * 16 + xp * 8, DPLB, dv0, dv1, dv2, dv3);
      }
      DWM_STAMP(3 + 3 * part);
      dwm_lds_barrier();
      DWM_STAMP(4 + 3 * part);
      const int nks = min(2, (keep - part * DPS + 3) >> 2);
#pragma unroll 1
      for (int ks = 0; ks < nks; ++ks) {
        const int lg = lane >> 4, vrow = lane & 15;
        const int ls = 4 * ks + lg, sl = part * DPS + ls;
        const bool pv = sl < keep;
        const int r = vrow - 3;
        const int dy = r < 0 ? 0 : (r > 7 ? 2 : 1);
        const bool rowv = vrow < 14 && pv;
        const int nbrow = pv ? sl : keep;
        const int nbL = rowv ? nbt[nbrow * 9 + dy * 3 + 0] : keep, nbC = rowv ? nbt[nbrow * 9 + dy * 3 + 1] : keep, nbR = rowv ? nbt[nbrow * 9 + dy * 3 + 2] : keep;
        const int yyb = (r & 7) * ROWB;
        const int aL = yyb + nbL * 16 + 8, aC = yyb + nbC * 16, aR = yyb + nbR * 16;
        // B: row y = lane & 7 of D (patch g); an idle patch reads the zero granule of the X planes
        const int offB = pv ? DOFF + (lane & 7) * DROWB + ls * 16 : XOFF + keep * 16;
        const int strB = pv ? DPLB : PLB;
        const unsigned ones_m = (vrow == 15) ? 0xffffffffu : 0u;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          __builtin_amdgcn_sched_barrier(0);
          const int c = 4 * q + cc;
          const unsigned char* xp_ = xpl + c * PLB + 16 * (c >> 3);
          const uint4 bd = *reinterpret_cast<const uint4*>(dww_smem + offB + c * strB + 16 * (c >> 3));
          const uint2 wl = *reinterpret_cast<const uint2*>(xp_ + aL);
          const uint4 wc = *reinterpret_cast<const uint4*>(xp_ + aC);
          const uint2 wr = *reinterpret_cast<const uint2*>(xp_ + aR);
          const uint32_t w[8] = {wl.x, wl.y, wc.x, wc.y, wc.z, wc.w, wr.x, wr.y};        // window columns -4 .. 11
          const bf16x8_t B = __builtin_bit_cast(bf16x8_t, bd);
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) {
            const int e0 = kx + 1, j0 = e0 >> 1;                    // columns kx - 3 .. kx + 4 = window elements e0 .. e0 + 7
            uint32_t f[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) f[t] = (e0 & 1) ? __builtin_amdgcn_alignbit(w[j0 + t + 1], w[j0 + t], 16) : w[j0 + t];
            if (kx == 0) {
#pragma unroll
              for (int t = 0; t < 4; ++t) f[t] = (f[t] & ~ones_m) | (0x3f803f80u & ones_m);      // window row 15: ones -> the bias gradient
            }
            acc[cc][kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, make_uint4(f[0], f[1], f[2], f[3])), B, acc[cc][kx], 0, 0, 0);
          }
        }
      }
      DWM_STAMP(5 + 3 * part);
      dwm_lds_barrier();                                           // the part's planes are free for the next one
    }
  }

  DWM_STAMP(12);
  // ---- diagonals. The planes are dead: every wave parks its 28 accumulator tiles in LDS as G[channel][kx][window row v][y] (fp32,
  // 3.5 KB per channel), then a thread per (tap, channel) sums its diagonal v = y + ky (bias: row 15 of kx = 0) and writes the slab.
  // (A first version folded with LDS float atomics straight from the accumulators: 112 per lane, 4-way colliding - 49 000 of the
  // kernel's 89 000 cycles.)
  __syncthreads();
  float* Gs = reinterpret_cast<float*>(dww_smem);
  constexpr int GT = 129;                          // tile pitch in floats: odd, so the summing threads (consecutive channels: 7 * 129 apart) hit distinct banks
  {
    const int y = lane & 15, g4 = (lane >> 4) * 4;
    if (y < 8) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
          for (int r = 0; r < 4; ++r) Gs[((4 * q + cc) * 7 + kx) * GT + (g4 + r) * 8 + y] = acc[cc][kx][r];
    }
  }
  DWM_STAMP(13);
  __syncthreads();
  DWM_STAMP(14);
  float* slab = wsv + (size_t)blockIdx.x * 50 * C;
  for (int i = tid; i < 50 * CCH; i += NT) {
    const int k = i / CCH, cc = i - k * CCH;
    const int ky = k < 49 ? k / 7 : 15, kx = k < 49 ? k - (k / 7) * 7 : 0;
    const float* gp = Gs + (cc * 7 + kx) * GT;
    float sum = 0.f;
#pragma unroll
    for (int y = 0; y < 8; ++y) sum += gp[((k < 49 ? y : 0) + ky) * 8 + y];
    slab[k * C + c0 + cc] = sum;
  }
  DWM_STAMP(15);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// S = 4 (stage 1: 4 x 4 patches). Same product, other tiling: an MFMA's contraction is 8 patches x 4 columns (a lane's 8 k values =
// 2 patches), 10 of the 16 window rows and 4 of the 16 D rows are real - a quarter of the tile, 21 MFMAs per channel and sample, still
// only ~3 us of matrix time for the whole stage. X and D are both resident (25 + 24 KB at 40 channels): no parts, two barriers.
// A lane reads the 12-column window row (left | centre | right pieces) of ITS two patches and shifts each by kx.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int CCH> struct DwMfmaWg4 {
  static constexpr int NQ = CCH / 4, NT = 64 * NQ, NV = CCH / 8, GT = 65;       // GT: tile pitch in floats (16 rows x 4 + 1)
  static size_t lds(int keep) {
    const size_t planes = (size_t)CCH * 4 * (keep + 1) * 8 + 16 * NV + (size_t)CCH * 4 * keep * 8 + 16 * NV + (size_t)(keep + 1) * 9 * 4 + 64 * 4 + 64 * 4;
    const size_t tiles = (size_t)CCH * 7 * GT * 4;
    return planes > tiles ? planes : tiles;
  }
};

template <int CCH>
__global__ __launch_bounds__(64 * (CCH / 4)) void dwconv7_wgrad_mfma4_kernel(const DwWgP qa, const DwWgGroupP grp) {
  using D = DwMfmaWg4<CCH>;
  constexpr int NT = D::NT, NV = D::NV, GT = D::GT;
  extern __shared__ __attribute__((aligned(16))) unsigned char dww_smem[];
  const int keep = qa.g.keep, SL = keep + 1, G = qa.g.grid, L = G * G;
  const int ROWB = SL * 8, PLB = 4 * ROWB;                          // X planes: [channel][4 rows][slot + zero granule][4 columns]
  const int DROWB = keep * 8, DPLB = 4 * DROWB;                     // D planes: [channel][4 rows][slot][4 columns]
  const int DOFF = CCH * PLB + 16 * NV;
  unsigned char* xpl = dww_smem;
  unsigned char* dpl = dww_smem + DOFF;
  int* nbt = reinterpret_cast<int*>(dpl + CCH * DPLB + 16 * NV);
  int* invl = nbt + (keep + 1) * 9;
  int* visl = invl + 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c0 = blockIdx.y * CCH, C = qa.C, q = wave, n = blockIdx.x;
  const void *xv, *ddv;
  float* wsv;
  dwwg_select(qa, grp, xv, ddv, wsv);
  const bf16_t* x = reinterpret_cast<const bf16_t*>(xv);
  const bf16_t* dd = reinterpret_cast<const bf16_t*>(ddv);

  for (int i = tid; i < CCH * 4; i += NT) {                         // zero granules of the X planes (slot = keep)
    const int y = i & 3, c = i >> 2;
    *reinterpret_cast<uint2*>(xpl + c * PLB + 16 * (c >> 3) + y * ROWB + keep * 8) = make_uint2(0u, 0u);
  }
  auto transpose_store = [&](unsigned char* base, int plb, const uint4& a0, const uint4& a1, const uint4& a2, const uint4& a3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
      const uint32_t lo = __builtin_amdgcn_perm(dwm_dw(a1, j >> 1), dwm_dw(a0, j >> 1), sel);
      const uint32_t hi = __builtin_amdgcn_perm(dwm_dw(a3, j >> 1), dwm_dw(a2, j >> 1), sel);
      *reinterpret_cast<uint2*>(base + j * plb) = make_uint2(lo, hi);
    }
  };
  // ---- every global load first: geometry, one task (a patch row: 4 points x 8 channels) of X and of D per thread and round
  const int tasks = keep * 4 * NV;
  const int inv_v = qa.g.inv[n * L + (tid < L ? tid : 0)];
  const int vis_v = qa.g.vis[n * keep + (tid < keep ? tid : 0)];
  {
    for (int t0 = 0; t0 < tasks; t0 += NT) {
      const int tk = t0 + tid, tc = tk < tasks ? tk : 0;
      const int o = tc % NV, r1 = tc / NV, y = r1 & 3, slot = r1 >> 2;
      const size_t row = ((size_t)(n * keep + slot) * 16 + y * 4) * C + c0 + 8 * o;
      const uint4 x0 = *reinterpret_cast<const uint4*>(x + row), x1 = *reinterpret_cast<const uint4*>(x + row + C);
      const uint4 x2 = *reinterpret_cast<const uint4*>(x + row + 2 * (size_t)C), x3 = *reinterpret_cast<const uint4*>(x + row + 3 * (size_t)C);
      const uint4 d0 = *reinterpret_cast<const uint4*>(dd + row), d1 = *reinterpret_cast<const uint4*>(dd + row + C);
      const uint4 d2 = *reinterpret_cast<const uint4*>(dd + row + 2 * (size_t)C), d3 = *reinterpret_cast<const uint4*>(dd + row + 3 * (size_t)C);
      if (t0 == 0) {
        if (tid < L) invl[tid] = inv_v;
        if (tid < keep) visl[tid] = vis_v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // invl / visl: written and read by wave 0 only
        if (tid <= keep) {
          const int patch = visl[tid < keep ? tid : 0];
          const int py = patch / G, px = patch - py * G;
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const int yy = py + k / 3 - 1, xx = px + k % 3 - 1;
            const bool in = yy >= 0 && yy < G && xx >= 0 && xx < G;
            const int sl = invl[in ? yy * G + xx : patch];
            nbt[tid * 9 + k] = (tid < keep && in && sl >= 0) ? sl : keep;
          }
        }
      }
      if (tk < tasks) {
        transpose_store(xpl + (8 * o) * PLB + 16 * o + y * ROWB + slot * 8, PLB, x0, x1, x2, x3);
        transpose_store(dpl + (8 * o) * DPLB + 16 * o + y * DROWB + slot * 8, DPLB, d0, d1, d2, d3);
      }
    }
  }
  dwm_lds_barrier();

  f32x4_t acc[4][7];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc)
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) acc[cc][kx] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nks = (keep + 7) >> 3;
#pragma unroll 1
  for (int ks = 0; ks < nks; ++ks) {
    const int lg = lane >> 4, vrow = lane & 15;
    const int r = vrow - 3;
    const int dy = r < 0 ? 0 : (r > 3 ? 2 : 1);
    const int yyb = (r & 3) * ROWB;
    int aL[2], aC[2], aR[2], offB[2], strB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                   // the lane's two patches
      const int sl = 8 * ks + 2 * lg + h;
      const bool pv = sl < keep, rowv = vrow < 10 && pv;
      const int nbrow = pv ? sl : keep;
      const int nbL = rowv ? nbt[nbrow * 9 + dy * 3 + 0] : keep, nbC = rowv ? nbt[nbrow * 9 + dy * 3 + 1] : keep, nbR = rowv ? nbt[nbrow * 9 + dy * 3 + 2] : keep;
      aL[h] = yyb + nbL * 8; aC[h] = yyb + nbC * 8; aR[h] = yyb + nbR * 8;
      offB[h] = pv ? DOFF + (lane & 3) * DROWB + sl * 8 : keep * 8;  // an idle patch reads the zero granule of the X planes
      strB[h] = pv ? DPLB : PLB;
    }
    const unsigned ones_m = (vrow == 15) ? 0xffffffffu : 0u;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      __builtin_amdgcn_sched_barrier(0);
      const int c = 4 * q + cc;
      const unsigned char* xp_ = xpl + c * PLB + 16 * (c >> 3);
      uint32_t w[2][6];
      uint2 bd[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bd[h] = *reinterpret_cast<const uint2*>(dww_smem + offB[h] + c * strB[h] + 16 * (c >> 3));
        const uint2 wl = *reinterpret_cast<const uint2*>(xp_ + aL[h]), wc = *reinterpret_cast<const uint2*>(xp_ + aC[h]), wr = *reinterpret_cast<const uint2*>(xp_ + aR[h]);
        w[h][0] = wl.x; w[h][1] = wl.y; w[h][2] = wc.x; w[h][3] = wc.y; w[h][4] = wr.x; w[h][5] = wr.y;      // window columns -4 .. 7
      }
      const bf16x8_t B = __builtin_bit_cast(bf16x8_t, make_uint4(bd[0].x, bd[0].y, bd[1].x, bd[1].y));
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int e0 = kx + 1, j0 = e0 >> 1;                          // columns kx - 3 .. kx = window elements e0 .. e0 + 3
        uint32_t f[4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int t = 0; t < 2; ++t) f[2 * h + t] = (e0 & 1) ? __builtin_amdgcn_alignbit(w[h][j0 + t + 1], w[h][j0 + t], 16) : w[h][j0 + t];
        if (kx == 0) {
#pragma unroll
          for (int t = 0; t < 4; ++t) f[t] = (f[t] & ~ones_m) | (0x3f803f80u & ones_m);        // window row 15: ones -> the bias gradient
        }
        acc[cc][kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, make_uint4(f[0], f[1], f[2], f[3])), B, acc[cc][kx], 0, 0, 0);
      }
    }
  }
  // ---- diagonals: tiles G[channel][kx][v][y] (16 x 4 floats, pitch 65), a thread per (tap, channel)
  __syncthreads();
  float* Gs = reinterpret_cast<float*>(dww_smem);
  {
    const int y = lane & 15, g4 = (lane >> 4) * 4;
    if (y < 4) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
          for (int r = 0; r < 4; ++r) Gs[((4 * q + cc) * 7 + kx) * GT + (g4 + r) * 4 + y] = acc[cc][kx][r];
    }
  }
  __syncthreads();
  float* slab = wsv + (size_t)blockIdx.x * 50 * C;
  for (int i = tid; i < 50 * CCH; i += NT) {
    const int k = i / CCH, cc = i - k * CCH;
    const int ky = k < 49 ? k / 7 : 15, kx = k < 49 ? k - (k / 7) * 7 : 0;
    const float* gp = Gs + (cc * 7 + kx) * GT;
    float sum = 0.f;
#pragma unroll
    for (int y = 0; y < 4; ++y) sum += gp[((k < 49 ? y : 0) + ky) * 4 + y];
    slab[k * C + c0 + cc] = sum;
  }
}
