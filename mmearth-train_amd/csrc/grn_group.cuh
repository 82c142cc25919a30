// GRN of the DENSE decoder block (norm_layers.py:25-48 with one statistics group per SAMPLE: `rpg` = L = 49 rows of H = 2048
// hidden columns, bf16) as ONE kernel per direction.
//
// The row-group path was three launches per direction (column statistics -> finalisation -> element-wise application: 51 us
// forward, 80 us backward stand-alone at the headline shape) that read h (and dz) twice from HBM. Here a 1024-thread workgroup
// owns a sample: thread = (column vector of 8, row quarter), its <= MAXR rows of h (dz in the backward) stay in REGISTERS as raw
// 16-byte vectors between the statistics pass and the application pass - a sample is 200 KB of bf16 and the register file of
// a CU holds 512 KB - so h is read once in the forward; the backward keeps dz and re-reads h (L2 / MALL: 400 KB per sample would
// not fit beside the accumulators). The four row quarters meet in LDS ([4][H] floats per statistic), the column mean over H
// columns is a wave sum + 4 LDS words.
//   forward : G2[j] = sum_rows gelu(h)^2 ; Gx = sqrt(G2) ; Ainv = 1 / (mean_j Gx + eps) ; scale = 1 + gamma Gx Ainv ;
//             z = gelu(h) scale + beta                                            (Gx, Ainv, scale kept for the backward)
//   backward: S0 = sum_rows dz ; S1 = sum_rows dz gelu(h) ; T2 = Ainv^2 / H sum_j gamma S1 Gx ;
//             coef = (gamma S1 Ainv - T2) / Gx (0 where Gx == 0) ; dh = (dz scale + coef gelu(h)) gelu'(h) over dz ;
//             the sample's gamma / beta gradient rows (Gx Ainv S1, S0) go to slab[g][2H] and are folded by a deferred
//             mpmae_fold_group record on the side lane (256-way same-address atomics were most of the old finalisation).
// grid = groups, block = 1024, H == 8 * 256, rpg <= 4 * MAXR.
#pragma once
#include "common.cuh"

template <int MAXR> struct GrnGroupCfg {
  static constexpr int H = 2048, THREADS = 1024, MAXROWS = 4 * MAXR;
  // rows r < RREG of a thread live in registers, the rest in the thread's own 16-byte LDS slots (lane-linear: conflict-free):
  // 13 raw rows + the statistics + the parameter vectors of 8 columns do not fit the 128 VGPRs a 1024-thread workgroup leaves a lane
  static constexpr int RREG_F = MAXR < 11 ? MAXR : 11, RREG_B = MAXR < 6 ? MAXR : 6;     // forward / backward (which also holds h rows in flight)
  static constexpr int PART_B = 4 * H * 4;                      // [4][H] floats (forward: 4 quarters; backward: [2][2][H])
  static constexpr int LDS_F = PART_B + (MAXR - RREG_F) * THREADS * 16 + 16, LDS_B = PART_B + (MAXR - RREG_B) * THREADS * 16 + 16;
};

// The rows a thread keeps between the two passes are made opaque here: otherwise the compiler carries the UNPACKED floats and the
// GELU values of the first pass (8 + 8 registers per row) to the second instead of the 4 raw registers, and spills.
__device__ __forceinline__ void grn_opaque(uint4& r) { asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w)); }

// Pins the accumulators of a statistics pass after every row: without it the compiler reassociates the unrolled row loop into
// "column pair by column pair over all rows", with every row's temporaries live at once (44 - 130 spilled registers).
__device__ __forceinline__ void grn_pin8(float (&a)[8]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
}

__device__ __forceinline__ void grn_unpack8(const uint4& r, float (&o)[8]) {
  o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
  o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
  o[4] = __uint_as_float(r.z << 16); o[5] = __uint_as_float(r.z & 0xffff0000u);
  o[6] = __uint_as_float(r.w << 16); o[7] = __uint_as_float(r.w & 0xffff0000u);
}

template <int MAXR>
__global__ __launch_bounds__(1024) void grn_group_fwd_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ z,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, int rpg, float* __restrict__ Gx,
                                                             float* __restrict__ Ainv, float* __restrict__ scale) {
  using Cf = GrnGroupCfg<MAXR>;
  constexpr int H = Cf::H, RREG = Cf::RREG_F;
  extern __shared__ __attribute__((aligned(16))) unsigned char grn_smem[];
  float (*part)[H] = reinterpret_cast<float (*)[H]>(grn_smem);
  uint4* stash = reinterpret_cast<uint4*>(grn_smem + Cf::PART_B);
  float* red = reinterpret_cast<float*>(grn_smem + Cf::LDS_F - 16);
  const int tid = threadIdx.x, ct = tid & 255, g = blockIdx.x;
  // the row quarter is wave-uniform: as an SGPR the (clamped) row addresses are scalar and a load is base SGPR pair + one lane offset
  // (as lane arithmetic the compiler kept a 64-bit address per row and operand live across both passes and spilled)
  const int rq = __builtin_amdgcn_readfirstlane(tid >> 8);
  const size_t base = (size_t)g * rpg * H;
  const unsigned lo = ct * 8;
  uint4 raw[MAXR];
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const int row = min(rq + 4 * r, rpg - 1);                         // clamped, masked below: no branch around the loads
    raw[r] = *reinterpret_cast<const uint4*>(h + base + (size_t)row * H + lo);
  }
  float a[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = 0.f;
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const float keep = rq + 4 * r < rpg ? 1.f : 0.f;
    float hv[8], gv[8];
    grn_unpack8(raw[r], hv);
    if (r >= RREG) stash[(r - RREG) * 1024 + tid] = raw[r];
    gelu_n<bf16_t, 8>(hv, gv);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += keep * gv[e] * gv[e];
    grn_pin8(a);
  }
  *reinterpret_cast<float4*>(&part[rq][ct * 8]) = make_float4(a[0], a[1], a[2], a[3]);
  *reinterpret_cast<float4*>(&part[rq][ct * 8 + 4]) = make_float4(a[4], a[5], a[6], a[7]);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RREG; ++r) grn_opaque(raw[r]);
  float gx[8], s = 0.f;
  {
    float4 lo = *reinterpret_cast<const float4*>(&part[0][ct * 8]), hi = *reinterpret_cast<const float4*>(&part[0][ct * 8 + 4]);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      const float4 l2 = *reinterpret_cast<const float4*>(&part[q][ct * 8]), h2 = *reinterpret_cast<const float4*>(&part[q][ct * 8 + 4]);
      lo.x += l2.x; lo.y += l2.y; lo.z += l2.z; lo.w += l2.w; hi.x += h2.x; hi.y += h2.y; hi.z += h2.z; hi.w += h2.w;
    }
    gx[0] = sqrtf(lo.x); gx[1] = sqrtf(lo.y); gx[2] = sqrtf(lo.z); gx[3] = sqrtf(lo.w);
    gx[4] = sqrtf(hi.x); gx[5] = sqrtf(hi.y); gx[6] = sqrtf(hi.z); gx[7] = sqrtf(hi.w);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) s += gx[e];
  s = wave_sum(s);
  if (rq == 0 && (tid & 63) == 0) red[tid >> 6] = s;                 // the first row quarter's four waves cover all H columns
  __syncthreads();
  const float ainv = 1.f / ((red[0] + red[1] + red[2] + red[3]) / H + eps);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + ct * 8), gb = *reinterpret_cast<const float4*>(gamma + ct * 8 + 4);
  const float4 ba = *reinterpret_cast<const float4*>(beta + ct * 8), bb = *reinterpret_cast<const float4*>(beta + ct * 8 + 4);
  const float gam[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
  const float bet[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
  float sc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sc[e] = 1.f + gam[e] * (gx[e] * ainv);
  if (rq == 0) {
    float* gxp = Gx + (size_t)g * H + ct * 8;
    float* scp = scale + (size_t)g * H + ct * 8;
    *reinterpret_cast<float4*>(gxp) = make_float4(gx[0], gx[1], gx[2], gx[3]);
    *reinterpret_cast<float4*>(gxp + 4) = make_float4(gx[4], gx[5], gx[6], gx[7]);
    *reinterpret_cast<float4*>(scp) = make_float4(sc[0], sc[1], sc[2], sc[3]);
    *reinterpret_cast<float4*>(scp + 4) = make_float4(sc[4], sc[5], sc[6], sc[7]);
    if (tid == 0) Ainv[g] = ainv;
  }
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const int row = rq + 4 * r;
    float hv[8], gv[8];
    grn_unpack8(r >= RREG ? stash[(r - RREG) * 1024 + tid] : raw[r], hv);
    gelu_n<bf16_t, 8>(hv, gv);
#pragma unroll
    for (int e = 0; e < 8; ++e) gv[e] = gv[e] * sc[e] + bet[e];
    if (row < rpg) st8<bf16_t>(z + base + (size_t)row * H + lo, gv);
    asm volatile("" ::: "memory");
  }
}

template <int MAXR>
__global__ __launch_bounds__(1024) void grn_group_bwd_kernel(bf16_t* __restrict__ dz, const bf16_t* __restrict__ h,
                                                             const float* __restrict__ scale, const float* __restrict__ Gx,
                                                             const float* __restrict__ Ainv, const float* __restrict__ gamma,
                                                             int rpg, float* __restrict__ slab) {
  using Cf = GrnGroupCfg<MAXR>;
  constexpr int H = Cf::H, RREG = Cf::RREG_B;
  extern __shared__ __attribute__((aligned(16))) unsigned char grn_smem[];
  float (*part)[2][H] = reinterpret_cast<float (*)[2][H]>(grn_smem);
  uint4* stash = reinterpret_cast<uint4*>(grn_smem + Cf::PART_B);
  float* red = reinterpret_cast<float*>(grn_smem + Cf::LDS_B - 16);
  const int tid = threadIdx.x, ct = tid & 255, g = blockIdx.x;
  // the row quarter is wave-uniform: as an SGPR the (clamped) row addresses are scalar and a load is base SGPR pair + one lane offset
  // (as lane arithmetic the compiler kept a 64-bit address per row and operand live across both passes and spilled)
  const int rq = __builtin_amdgcn_readfirstlane(tid >> 8);
  const size_t base = (size_t)g * rpg * H;
  const unsigned lo = ct * 8;
  uint4 raw[MAXR];
  float a0[8], a1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
  // chunks of CH rows: the chunk's dz and h loads are issued together, then consumed; scheduling fences between the chunks keep
  // the compiler from hoisting every load of the pass (2 x MAXR x 4 registers) above the first use
  constexpr int CH = 4;
#pragma unroll
  for (int c0 = 0; c0 < MAXR; c0 += CH) {
    uint4 hr[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int r = c0 + i;
      if (r < MAXR) {
        const int row = min(rq + 4 * r, rpg - 1);
        raw[r] = *reinterpret_cast<const uint4*>(dz + base + (size_t)row * H + lo);
        hr[i] = *reinterpret_cast<const uint4*>(h + base + (size_t)row * H + lo);
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int r = c0 + i;
      if (r < MAXR) {
        const float keep = rq + 4 * r < rpg ? 1.f : 0.f;
        float hv[8], gv[8], d[8];
        grn_unpack8(hr[i], hv);
        grn_unpack8(raw[r], d);
        if (r >= RREG) stash[(r - RREG) * 1024 + tid] = raw[r];
        gelu_n<bf16_t, 8>(hv, gv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { a0[e] += keep * d[e]; a1[e] += keep * d[e] * gv[e]; }
        grn_pin8(a0);
        grn_pin8(a1);
      }
    }
  }
  // the four row quarters meet pairwise (quarters 2, 3 park their sums, quarters 0, 1 add their own on top): [2][2][H] floats
  // = 32 KB instead of 64 KB + the reduction words, which is past the static LDS limit
  if (rq >= 2) {
    *reinterpret_cast<float4*>(&part[0][rq - 2][ct * 8]) = make_float4(a0[0], a0[1], a0[2], a0[3]);
    *reinterpret_cast<float4*>(&part[0][rq - 2][ct * 8 + 4]) = make_float4(a0[4], a0[5], a0[6], a0[7]);
    *reinterpret_cast<float4*>(&part[1][rq - 2][ct * 8]) = make_float4(a1[0], a1[1], a1[2], a1[3]);
    *reinterpret_cast<float4*>(&part[1][rq - 2][ct * 8 + 4]) = make_float4(a1[4], a1[5], a1[6], a1[7]);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RREG; ++r) grn_opaque(raw[r]);
  if (rq < 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[0][rq][ct * 8 + e] += a0[e]; part[1][rq][ct * 8 + e] += a1[e]; }
  }
  __syncthreads();
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = ct * 8 + e;
    s0[e] = part[0][0][c] + part[0][1][c];
    s1[e] = part[1][0][c] + part[1][1][c];
  }
  const float* gp = Gx + (size_t)g * H + ct * 8;
  const float* sp = scale + (size_t)g * H + ct * 8;
  const float4 xa = *reinterpret_cast<const float4*>(gp), xb = *reinterpret_cast<const float4*>(gp + 4);
  const float4 sa = *reinterpret_cast<const float4*>(sp), sb = *reinterpret_cast<const float4*>(sp + 4);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + ct * 8), gb = *reinterpret_cast<const float4*>(gamma + ct * 8 + 4);
  const float ainv = Ainv[g];
  const float gx[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
  const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
  const float gam[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
  float t = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) t += gam[e] * s1[e] * gx[e];
  t = wave_sum(t);
  if (rq == 0 && (tid & 63) == 0) red[tid >> 6] = t;
  __syncthreads();
  const float T2 = (red[0] + red[1] + red[2] + red[3]) * ainv * ainv / H;
  float coef[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) coef[e] = gx[e] > 0.f ? (gam[e] * s1[e] * ainv - T2) / gx[e] : 0.f;
  if (rq == 0) {
    float* o = slab + (size_t)g * 2 * H + ct * 8;
    *reinterpret_cast<float4*>(o) = make_float4(gx[0] * ainv * s1[0], gx[1] * ainv * s1[1], gx[2] * ainv * s1[2], gx[3] * ainv * s1[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(gx[4] * ainv * s1[4], gx[5] * ainv * s1[5], gx[6] * ainv * s1[6], gx[7] * ainv * s1[7]);
    *reinterpret_cast<float4*>(o + H) = make_float4(s0[0], s0[1], s0[2], s0[3]);
    *reinterpret_cast<float4*>(o + H + 4) = make_float4(s0[4], s0[5], s0[6], s0[7]);
  }
#pragma unroll
  for (int c0 = 0; c0 < MAXR; c0 += CH) {
    uint4 hr[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int r = c0 + i;
      if (r < MAXR) hr[i] = *reinterpret_cast<const uint4*>(h + base + (size_t)min(rq + 4 * r, rpg - 1) * H + lo);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int r = c0 + i;
      if (r < MAXR) {
        const int row = rq + 4 * r;
        float hv[8], gv[8], dg[8], d[8];
        grn_unpack8(hr[i], hv);
        grn_unpack8(r >= RREG ? stash[(r - RREG) * 1024 + tid] : raw[r], d);
        gelu_both_n<bf16_t, 8>(hv, gv, dg);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = (d[e] * sc[e] + coef[e] * gv[e]) * dg[e];
        if (row < rpg) st8<bf16_t>(dz + base + (size_t)row * H + lo, d);
        asm volatile("" ::: "memory");
      }
    }
  }
}
