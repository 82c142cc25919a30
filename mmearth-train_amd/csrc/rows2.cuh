// Vectorised LayerNorm forward / backward (v2) and column statistics (v2).
//
// A row of C channels is handled by G = 8/16/32/64 lanes, each lane owning PER 8-channel vectors
// (16 B in bf16), so a 64-lane wave processes 64/G rows at once with 16-byte global accesses and
// log2(G) shuffle steps; waves stride over row groups with no workgroup barrier.
#pragma once
#include "rows.cuh"

// sum over the G lanes of a row group (G = 8..64, groups are aligned): the first 8 / 16 lanes fold with DPP
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: one v_add_f32_dpp each), only the steps that cross a
// 16-lane row go through ds_bpermute
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_mov<0xB1>(v);                       // xor 1
  v += dpp_mov<0x4E>(v);                       // xor 2
  v += dpp_mov<0x141>(v);                      // row_half_mirror: the other quad of the 8-lane half
  if (G >= 16) v += dpp_mov<0x140>(v);         // row_mirror: the other half of the 16-lane row
#pragma unroll
  for (int o = 16; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// (down_group_off: common.cuh)

template <typename T, int G, int PER>
__global__ __launch_bounds__(256) void ln_fwd_v2_kernel(const T* __restrict__ x, T* __restrict__ xhat,
                                                        float* __restrict__ rstd_out, T* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int act, float eps, int M, int C,
                                                        const uint8_t* __restrict__ rowmask, int down_S) {
  constexpr int RPW = 64 / G;                      // rows per wave
  const int lane = threadIdx.x & 63;
  const int gl = lane % G, rl = lane / G;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nvec = C / 8;
  float ga[PER][8], be[PER][8];
  if (y) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int vi = gl + p * G;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // unconditional, clamped (a load under a per-lane condition is a branch + wait of its own: 16 of them made this
        // kernel 28 us at C = 512 where a plain copy of the same rows takes 4 us)
        const float gv = gamma[min(vi, nvec - 1) * 8 + e], bv = beta[min(vi, nvec - 1) * 8 + e];
        ga[p][e] = (vi < nvec) ? gv : 0.f;
        be[p][e] = (vi < nvec) ? bv : 0.f;
      }
    }
  }
  for (int m0 = wave_global * RPW; m0 < M; m0 += nwaves * RPW) {
    const int m = m0 + rl;
    const bool rok = m < M;
    const bool live = rok && (!rowmask || rowmask[m]);
    float v[PER][8];
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int vi = gl + p * G;
      ld8<T>(x + (size_t)min(m, M - 1) * C + min(vi, nvec - 1) * 8, v[p]);          // unconditional, clamped; zeroed below
      if (!(rok && vi < nvec)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[p][e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[p][e];
    }
    const float mean = group_sum<G>(s) / C;
    float q = 0.f;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int vi = gl + p * G;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[p][e] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(group_sum<G>(q) / C + eps);
    if (gl == 0 && rok && rstd_out) rstd_out[m] = live ? rstd : 0.f;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int vi = gl + p * G;
      if (rok && vi < nvec) {
        float xh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xh[e] = live ? (v[p][e] - mean) * rstd : 0.f;
        if (xhat) {
          st8<T>(xhat + (size_t)m * C + vi * 8, xh);
          if (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) xh[e] = bf2f(f2bf(xh[e]));
          }
        }
        if (y) {
          float u[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            u[e] = xh[e] * ga[p][e] + be[p][e];
            if (act == 1) u[e] = gelu_t<T>(u[e]);
            if (!live) u[e] = 0.f;
          }
          st8<T>(y + (down_S ? down_group_off(m, down_S, C) : (size_t)m * C) + vi * 8, u);
        }
      }
    }
  }
}

// backward; per-wave dgamma/dbeta partials -> slab ws[wave_global][2][C]
template <typename T, int G, int PER>
__global__ __launch_bounds__(256) void ln_bwd_v2_kernel(const T* __restrict__ dy, int dy_div, float dy_scale,
                                                        const T* __restrict__ xhat, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int act, T* __restrict__ dx, int accumulate,
                                                        float* __restrict__ ws, int M, int C,
                                                        const uint8_t* __restrict__ rowmask, int down_S) {
  constexpr int RPW = 64 / G;
  __shared__ float ln2_comb[4 * 2 * 1024];       // [wave][dgamma | dbeta][C], C <= 1024 (checked by the launcher)
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int gl = lane % G, rl = lane / G;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nvec = C / 8;
  float ga[PER][8], be[PER][8], ag[PER][8], ab[PER][8];
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int vi = gl + p * G;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gv = gamma[min(vi, nvec - 1) * 8 + e], bv = beta ? beta[min(vi, nvec - 1) * 8 + e] : 0.f;      // unconditional, clamped
      ga[p][e] = (vi < nvec) ? gv : 0.f;
      be[p][e] = (vi < nvec && act == 1) ? bv : 0.f;
      ag[p][e] = 0.f; ab[p][e] = 0.f;
    }
  }
  for (int m0 = wave_global * RPW; m0 < M; m0 += nwaves * RPW) {
    const int m = m0 + rl;
    const bool rok = m < M;
    const bool live = rok && (!rowmask || rowmask[m]);
    float g[PER][8], xh[PER][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int vi = gl + p * G;
#pragma unroll
      for (int e = 0; e < 8; ++e) { g[p][e] = 0.f; xh[p][e] = 0.f; }
      float d[8];
      {
        const int mc = min(m, M - 1), vc = min(vi, nvec - 1);                        // unconditional, clamped loads
        ld8<T>(xhat + (size_t)mc * C + vc * 8, xh[p]);
        ld8<T>(dy + (down_S ? down_group_off(mc, down_S, C) : (size_t)(mc / dy_div) * C) + vc * 8, d);
      }
      if (!(live && vi < nvec)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xh[p][e] = 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float dd = d[e] * dy_scale;
          if (act == 1) dd *= gelu_grad_t<T>(xh[p][e] * ga[p][e] + be[p][e]);
          ag[p][e] += dd * xh[p][e];
          ab[p][e] += dd;
          g[p][e] = dd * ga[p][e];
          s1 += g[p][e];
          s2 += g[p][e] * xh[p][e];
        }
      }
    }
    s1 = group_sum<G>(s1) / C;
    s2 = group_sum<G>(s2) / C;
    const float rsl = rstd[min(m, M - 1)];
    const float rs = live ? rsl : 0.f;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int vi = gl + p * G;
      if (rok && vi < nvec) {
        float o[8];
        T* dst = dx + (size_t)m * C + vi * 8;
        if (accumulate) ld8<T>(dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float vv = live ? rs * (g[p][e] - s1 - xh[p][e] * s2) : 0.f;
          o[e] = accumulate ? o[e] + vv : vv;
        }
        st8<T>(dst, o);
      }
    }
  }
  // lanes with equal gl hold the same channels: fold the RPW row-lanes, then lanes rl == 0 write
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int vi = gl + p * G;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = ag[p][e], b = ab[p][e];
#pragma unroll
      for (int o = G; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (rl == 0 && vi < nvec) { ln2_comb[(wib * 2 + 0) * C + vi * 8 + e] = a; ln2_comb[(wib * 2 + 1) * C + vi * 8 + e] = b; }
    }
  }
  // the 4 waves of a workgroup fold their partials in LDS: ONE slab row per workgroup (the per-wave rows were 16 MB of fp32 slab
  // written and re-read by the second stage for 38 MB of activations at the decoder shape)
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256)
    ws[(size_t)blockIdx.x * 2 * C + i] = ln2_comb[i] + ln2_comb[2 * C + i] + ln2_comb[4 * C + i] + ln2_comb[6 * C + i];
}


// column statistics v3: one 256-thread block per slab of `rows_per_block` rows (= one statistics
// group in grouped mode); wave w takes rows w, w+4, ...; lanes own 16-byte column vectors
// (VPL per lane), the four waves are combined through LDS and the block writes one [H] row
// (two in mode 1) to out0/out1 at row blockIdx.x. gridDim.y > 1 splits the columns: block (x, y) owns the vectors
// [y * 64 * VPL, (y + 1) * 64 * VPL) - the per-sample statistics of the dense decoder (256 groups of 49 rows) would otherwise
// run as 256 workgroups, one wave per SIMD.
template <typename T, int VPL>
__global__ __launch_bounds__(256) void colstats_v3_kernel(const T* __restrict__ h, const T* __restrict__ dz, int mode,
                                                          float* __restrict__ out0, float* __restrict__ out1,
                                                          int M, int H, int rows_per_block) {
  extern __shared__ float red[];                       // [4 waves][(mode+1)][H]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mb = blockIdx.x * rows_per_block, me = min(M, mb + rows_per_block);
  const int nvec = H / 8, vbase = blockIdx.y * (64 * VPL);
  float a0[VPL][8], a1[VPL][8];
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { a0[i][e] = 0.f; a1[i][e] = 0.f; }
  // unconditional loads from clamped column vectors (the second operand through a pointer select), lanes past the last vector
  // masked afterwards: `if (v < nvec) { load ... }` in the row loop was a branch region with its own wait per row
  const T* dzp = mode == 1 ? dz : h;
#pragma unroll 4
  for (int m = mb + wave; m < me; m += 4) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int v = vbase + lane + 64 * i, vc = min(v, nvec - 1);
      const float keep = v < nvec ? 1.f : 0.f;
      float hv[8], d[8];
      ld8<T>(h + (size_t)m * H + vc * 8, hv);
      ld8<T>(dzp + (size_t)m * H + vc * 8, d);
      if (mode == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float g = gelu_t<T>(hv[e]); a0[i][e] += keep * g * g; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { a0[i][e] += keep * d[e]; a1[i][e] += keep * d[e] * gelu_t<T>(hv[e]); }
      }
    }
  }
  const int nst = mode + 1;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int v = vbase + lane + 64 * i;
    if (v < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * nst + 0) * H + v * 8 + e] = a0[i][e];
        if (mode == 1) red[(wave * nst + 1) * H + v * 8 + e] = a1[i][e];
      }
    }
  }
  __syncthreads();
  const int jb = vbase * 8, je = min(H, jb + 64 * VPL * 8);
  for (int j = jb + threadIdx.x; j < je; j += 256) {
    out0[(size_t)blockIdx.x * H + j] = red[(0 * nst) * H + j] + red[(1 * nst) * H + j] + red[(2 * nst) * H + j] + red[(3 * nst) * H + j];
    if (mode == 1)
      out1[(size_t)blockIdx.x * H + j] = red[(0 * nst + 1) * H + j] + red[(1 * nst + 1) * H + j] + red[(2 * nst + 1) * H + j] + red[(3 * nst + 1) * H + j];
  }
}
