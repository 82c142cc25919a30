// Per-modality reconstruction losses + uncertainty weighting, forward and backward, with no
// host synchronisation (reference: models/fcmae.py:267-412, custom_loss.py:19-30).
// Predictions are channels-last rows: pixel heads [N*L, ld] (modality slice at column `coff`,
// column j = (ph*p+pw)*C + c), image heads [N, ld].
#pragma once
#include "common.cuh"

__device__ __forceinline__ float block_sum256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ float nan_to_num0(float t) { return (isnan(t) || isinf(t)) ? 0.f : t; }

typedef MpmaePixContArgs PixContP;

// Forward: one block per SAMPLE looping over its L patches, per-sample partial {sum, count}
// written to acc[2n], acc[2n+1] (no atomics; loss_finalize sums over samples).
// Backward: one block per patch.
template <typename T, bool BWD>
__device__ __forceinline__ void loss_pix_cont_patch(const PixContP& q, int b, float* sh, float& acc_s, float& acc_c) {
  const int n = b / q.L, l = b - n * q.L;
  const int py = l / q.grid, px = l - py * q.grid;
  const int p = q.p, C = q.C, J = p * p * C;
  const T* pred = reinterpret_cast<const T*>(q.pred) + (size_t)b * q.ld + q.coff;
  const bool masked = q.mask[b] != 0.f;
  if constexpr (BWD) {
    T* dp = reinterpret_cast<T*>(q.dpred) + (size_t)b * q.ld + q.coff;
    const float pl = q.patch_l[b];
    const bool counted = masked && pl != 0.f && !isnan(pl);
    if (!counted) {
      for (int j = threadIdx.x; j < J; j += blockDim.x) stf<T>(dp + j, 0.f);
      return;
    }
    const float k = q.coef[0] * q.mask[b] * 2.f / q.patch_cnt[b];
    const float mean = q.patch_mean[b], rstd = q.patch_rstd[b];
    for (int i = threadIdx.x; i < J; i += blockDim.x) {
      const int c = i / (p * p), r = i - c * p * p, ph = r / p, pw = r - ph * p;
      float t = nan_to_num0(q.target[((size_t)(n * C + c) * q.H + py * p + ph) * q.H + px * p + pw]);
      t = (t - mean) * rstd;
      const int j = (ph * p + pw) * C + c;
      const float d = ldf<T>(pred + j) - t;
      stf<T>(dp + j, isnan(d * d) ? 0.f : k * d);
    }
    return;
  } else {
    if (!masked) {
      if (threadIdx.x == 0) { q.patch_l[b] = 0.f; q.patch_cnt[b] = 0.f; q.patch_mean[b] = 0.f; q.patch_rstd[b] = 1.f; }
      return;
    }
    float mean = 0.f, rstd = 1.f;
    if (q.norm_pix) {
      float s = 0.f;
      for (int i = threadIdx.x; i < J; i += blockDim.x) {
        const int c = i / (p * p), r = i - c * p * p, ph = r / p, pw = r - ph * p;
        s += nan_to_num0(q.target[((size_t)(n * C + c) * q.H + py * p + ph) * q.H + px * p + pw]);
      }
      mean = block_sum256(s, sh) / J;
      float v = 0.f;
      for (int i = threadIdx.x; i < J; i += blockDim.x) {
        const int c = i / (p * p), r = i - c * p * p, ph = r / p, pw = r - ph * p;
        const float d = nan_to_num0(q.target[((size_t)(n * C + c) * q.H + py * p + ph) * q.H + px * p + pw]) - mean;
        v += d * d;
      }
      const float var = block_sum256(v, sh) / (J - 1);          // unbiased (torch .var default)
      rstd = 1.f / sqrtf(var + 1.0e-6f);
    }
    float se = 0.f, cnt = 0.f;
    for (int i = threadIdx.x; i < J; i += blockDim.x) {
      const int c = i / (p * p), r = i - c * p * p, ph = r / p, pw = r - ph * p;
      float t = nan_to_num0(q.target[((size_t)(n * C + c) * q.H + py * p + ph) * q.H + px * p + pw]);
      t = (t - mean) * rstd;
      const float d = ldf<T>(pred + (ph * p + pw) * C + c) - t;
      const float e = d * d;
      if (!isnan(e)) { se += e; cnt += 1.f; }
    }
    se = block_sum256(se, sh);
    cnt = block_sum256(cnt, sh);
    if (threadIdx.x == 0) {
      const float lp = se / cnt;                                  // 0/0 -> NaN -> dropped below
      const float qv = lp * q.mask[b];
      const bool counted = !isnan(qv) && qv != 0.f;
      q.patch_l[b] = counted ? lp : 0.f;
      q.patch_cnt[b] = cnt; q.patch_mean[b] = mean; q.patch_rstd[b] = rstd;
      if (counted) { acc_s += qv; acc_c += 1.f; }
    }
  }
}

// register-cached patch (J <= 64*NCACHE elements): returns through acc_s / acc_c
template <typename T, int NCACHE>
__device__ __noinline__ void loss_pix_cont_patch_cached(const PixContP& q, int b, const T* pred, const float* tg,
                                                           int lane, int p, int C, int PP, int J, float& acc_s, float& acc_c) {
    // register-cached form: every target / prediction element of the patch is loaded exactly once and
    // all loads of the patch are in flight together (one memory latency per patch instead of one per pass)
    float tv[NCACHE], pv[NCACHE];
    // loads go out in groups of 4 (x2 arrays): enough in flight to hide the latency, and the 64-bit
    // addresses of one group are dead before the next is formed (16 at once spilled 141 VGPRs)
#pragma unroll
    for (int ub = 0; ub < NCACHE; ub += 4) {
#pragma unroll
      for (int uu = 0; uu < 4 && ub + uu < NCACHE; ++uu) {
        const int u = ub + uu;
        const int i = lane + 64 * u;
        const int ic = i < J ? i : 0;
        const int c = ic / PP, r = ic - c * PP, ph = r / p, pw = r - ph * p;
        tv[u] = tg[((size_t)c * q.H + ph) * q.H + pw];          // clamped index, unconditional: a load under a
        pv[u] = ldf<T>(pred + r * C + c);                      // per-lane branch is serialised with a wait each
      }
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int u = 0; u < NCACHE; ++u) {
      const bool okk = lane + 64 * u < J;
      tv[u] = okk ? nan_to_num0(tv[u]) : 0.f;
      pv[u] = okk ? pv[u] : 0.f;
    }
    float mean = 0.f, rstd = 1.f;
    if (q.norm_pix) {
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < NCACHE; ++u) s += tv[u];
      mean = wave_sum(s) / J;
      float v = 0.f;
#pragma unroll
      for (int u = 0; u < NCACHE; ++u) { const float d = (lane + 64 * u < J) ? tv[u] - mean : 0.f; v += d * d; }
      v = wave_sum(v);
      rstd = 1.f / sqrtf(v / (J - 1) + 1.0e-6f);
    }
    float se = 0.f, cnt = 0.f;
#pragma unroll
    for (int u = 0; u < NCACHE; ++u) {
      const float d = pv[u] - (tv[u] - mean) * rstd;
      const float e = d * d;
      if (lane + 64 * u < J && !isnan(e)) { se += e; cnt += 1.f; }
    }
    se = wave_sum(se); cnt = wave_sum(cnt);
    const float lp = se / cnt;
    const float qv = lp * q.mask[b];
    const bool counted = !isnan(qv) && qv != 0.f;
    if (lane == 0) {
      q.patch_l[b] = counted ? lp : 0.f;
      q.patch_cnt[b] = cnt; q.patch_mean[b] = mean; q.patch_rstd[b] = rstd;
    }
    if (counted) { acc_s += qv; acc_c += 1.f; }

}

// forward, wave-granular: each of the 4 waves of a sample's block walks patches l = wave, wave+4, ...
// with wave-level reductions only; the block combines the four partials once at the end.
template <typename T>
__device__ __forceinline__ void loss_pix_cont_patch_wave(const PixContP& q, int b, float& acc_s, float& acc_c) {
  const int lane = threadIdx.x & 63;
  const int n = b / q.L, l = b - n * q.L;
  const int py = l / q.grid, px = l - py * q.grid;
  const int p = q.p, C = q.C, PP = p * p, J = PP * C;
  const T* pred = reinterpret_cast<const T*>(q.pred) + (size_t)b * q.ld + q.coff;
  if (q.mask[b] == 0.f) {
    if (lane == 0) { q.patch_l[b] = 0.f; q.patch_cnt[b] = 0.f; q.patch_mean[b] = 0.f; q.patch_rstd[b] = 1.f; }
    return;
  }
  const float* tg = q.target + ((size_t)n * C * q.H + py * p) * q.H + px * p;
  if (J <= 64 * 16) {
    if (J <= 64 * 2) loss_pix_cont_patch_cached<T, 2>(q, b, pred, tg, lane, p, C, PP, J, acc_s, acc_c);
    else if (J <= 64 * 8) loss_pix_cont_patch_cached<T, 8>(q, b, pred, tg, lane, p, C, PP, J, acc_s, acc_c);
    else loss_pix_cont_patch_cached<T, 16>(q, b, pred, tg, lane, p, C, PP, J, acc_s, acc_c);
    return;
  }
  float mean = 0.f, rstd = 1.f;
  if (q.norm_pix) {
    float s = 0.f, s2 = 0.f;
    for (int i = lane; i < J; i += 64) {
      const int c = i / PP, r = i - c * PP, ph = r / p, pw = r - ph * p;
      const float t = nan_to_num0(tg[((size_t)c * q.H + ph) * q.H + pw]);
      s += t; s2 += t * t;
    }
    s = wave_sum(s); s2 = wave_sum(s2);
    mean = s / J;
    // unbiased variance; two-pass form for accuracy
    float v = 0.f;
    for (int i = lane; i < J; i += 64) {
      const int c = i / PP, r = i - c * PP, ph = r / p, pw = r - ph * p;
      const float d = nan_to_num0(tg[((size_t)c * q.H + ph) * q.H + pw]) - mean;
      v += d * d;
    }
    v = wave_sum(v);
    rstd = 1.f / sqrtf(v / (J - 1) + 1.0e-6f);
  }
  float se = 0.f, cnt = 0.f;
  for (int i = lane; i < J; i += 64) {
    const int c = i / PP, r = i - c * PP, ph = r / p, pw = r - ph * p;
    float t = nan_to_num0(tg[((size_t)c * q.H + ph) * q.H + pw]);
    t = (t - mean) * rstd;
    const float d = ldf<T>(pred + (ph * p + pw) * C + c) - t;
    const float e = d * d;
    if (!isnan(e)) { se += e; cnt += 1.f; }
  }
  se = wave_sum(se); cnt = wave_sum(cnt);
  const float lp = se / cnt;
  const float qv = lp * q.mask[b];
  const bool counted = !isnan(qv) && qv != 0.f;
  if (lane == 0) {
    q.patch_l[b] = counted ? lp : 0.f;
    q.patch_cnt[b] = cnt; q.patch_mean[b] = mean; q.patch_rstd[b] = rstd;
  }
  if (counted) { acc_s += qv; acc_c += 1.f; }
}

template <typename T, bool BWD>
__device__ __forceinline__ void loss_pix_cont_body(const PixContP& q, const int bx) {
  __shared__ float sh[4];
  __shared__ float part[16][2];
  float as = 0.f, ac = 0.f;
  if constexpr (BWD) {
    loss_pix_cont_patch<T, true>(q, bx, sh, as, ac);
  } else {
    const int n = bx, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;      // NW <= 16
    for (int l = wave; l < q.L; l += NW) loss_pix_cont_patch_wave<T>(q, n * q.L + l, as, ac);
    if ((threadIdx.x & 63) == 0) { part[wave][0] = as; part[wave][1] = ac; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float ts = 0.f, tc = 0.f;
      for (int w = 0; w < NW; ++w) { ts += part[w][0]; tc += part[w][1]; }     // fixed order: deterministic
      q.acc[2 * n] = ts;
      q.acc[2 * n + 1] = tc;
    }
  }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(BWD ? 256 : 512) void loss_pix_cont_kernel(const PixContP q) {
  loss_pix_cont_body<T, BWD>(q, blockIdx.x);
}

// several modalities in one launch: blockIdx.y selects the argument record of a device-resident table
template <typename T, bool BWD>
__global__ __launch_bounds__(BWD ? 256 : 512) void loss_pix_cont_multi_kernel(const PixContP* __restrict__ tab) {
  const PixContP q = tab[blockIdx.y];
  loss_pix_cont_body<T, BWD>(q, blockIdx.x);
}

typedef MpmaePixCatArgs PixCatP;

template <typename T, bool BWD>
__device__ __forceinline__ void loss_pix_cat_patch(const PixCatP& q, int b, float& se, float& cnt) {
  const int n = b / q.L, l = b - n * q.L;
  const int py = l / q.grid, px = l - py * q.grid;
  const int p = q.p, K = q.K, PP = p * p;
  const T* pred = reinterpret_cast<const T*>(q.pred) + (size_t)b * q.ld + q.coff;
  const bool masked = q.mask[b] == 1.f;
  T* dp = BWD ? reinterpret_cast<T*>(q.dpred) + (size_t)b * q.ld + q.coff : nullptr;
  if (!masked) {
    if constexpr (BWD) for (int j = threadIdx.x; j < PP * K; j += blockDim.x) stf<T>(dp + j, 0.f);
    return;
  }
  const float k = BWD ? q.coef[0] : 0.f;
  for (int pix = threadIdx.x; pix < PP; pix += blockDim.x) {
    const int ph = pix / p, pw = pix - ph * p;
    const long long t = q.target[((size_t)n * q.H + py * p + ph) * q.H + px * p + pw];
    float z[16];
    float mx = -INFINITY;
    for (int c = 0; c < K; ++c) { z[c] = ldf<T>(pred + pix * K + c); mx = fmaxf(mx, z[c]); }
    float s = 0.f;
    for (int c = 0; c < K; ++c) s += __expf(z[c] - mx);
    const float lse = mx + __logf(s);
    if (t != -1) {
      if constexpr (BWD) {
        for (int c = 0; c < K; ++c) stf<T>(dp + pix * K + c, k * (__expf(z[c] - lse) - (c == (int)t ? 1.f : 0.f)));
      } else {
        se += lse - z[(int)t];
        cnt += 1.f;
      }
    } else if constexpr (BWD) {
      for (int c = 0; c < K; ++c) stf<T>(dp + pix * K + c, 0.f);
    }
  }
}

template <typename T, bool BWD>
__device__ __forceinline__ void loss_pix_cat_body(const PixCatP& q, const int bx) {
  __shared__ float sh[4];
  float se = 0.f, cnt = 0.f;
  if constexpr (BWD) {
    loss_pix_cat_patch<T, true>(q, bx, se, cnt);
  } else {
    const int n = bx;
    const int p = q.p, K = q.K, PP = p * p;
    for (int idx = threadIdx.x; idx < q.L * PP; idx += blockDim.x) {      // all pixels of all patches
      const int l = idx / PP, pix = idx - l * PP;
      const int b = n * q.L + l;
      if (q.mask[b] != 1.f) continue;
      const int py = l / q.grid, px = l - py * q.grid, ph = pix / p, pw = pix - ph * p;
      const long long t = q.target[((size_t)n * q.H + py * p + ph) * q.H + px * p + pw];
      if (t == -1) continue;
      const T* pred = reinterpret_cast<const T*>(q.pred) + (size_t)b * q.ld + q.coff + pix * K;
      float z[16], mx = -INFINITY;
      for (int c = 0; c < K; ++c) { z[c] = ldf<T>(pred + c); mx = fmaxf(mx, z[c]); }
      float sm = 0.f;
      for (int c = 0; c < K; ++c) sm += __expf(z[c] - mx);
      se += mx + __logf(sm) - z[(int)t];
      cnt += 1.f;
    }
    // any block size up to 16 waves: wave sums, then a fixed-order fold
    __shared__ float part[16][2];
    se = wave_sum(se); cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = se; part[threadIdx.x >> 6][1] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float ts = 0.f, tc = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ts += part[w][0]; tc += part[w][1]; }
      q.acc[2 * n] = ts; q.acc[2 * n + 1] = tc;
    }
  }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(BWD ? 256 : 1024) void loss_pix_cat_kernel(const PixCatP q) {
  loss_pix_cat_body<T, BWD>(q, blockIdx.x);
}

// several modalities in one launch: blockIdx.y selects the argument record of a device-resident table
template <typename T, bool BWD>
__global__ __launch_bounds__(BWD ? 256 : 1024) void loss_pix_cat_multi_kernel(const PixCatP* __restrict__ tab) {
  const PixCatP q = tab[blockIdx.y];
  loss_pix_cat_body<T, BWD>(q, blockIdx.x);
}

typedef MpmaeImgArgs ImgP;

template <typename T, bool BWD>
__device__ __forceinline__ void loss_img_body(const ImgP& q, const int bx) {
  __shared__ float sh[4];
  __shared__ int shi[4];
  const int n = bx, K = q.K;
  const T* pred = reinterpret_cast<const T*>(q.pred) + (size_t)n * q.ld + q.coff;
  T* dp = BWD ? reinterpret_cast<T*>(q.dpred) + (size_t)n * q.ld + q.coff : nullptr;
  if (q.kind == 1) {
    const float* tg = reinterpret_cast<const float*>(q.target) + (size_t)n * K;
    float se = 0.f, cnt = 0.f;
    const float k = BWD ? q.coef[0] : 0.f;
    for (int c = threadIdx.x; c < K; c += blockDim.x) {
      const float t = tg[c];
      const float d = ldf<T>(pred + c) - t;
      if (!isnan(t)) {
        if constexpr (BWD) stf<T>(dp + c, 2.f * k * d); else { se += d * d; cnt += 1.f; }
      } else if constexpr (BWD) stf<T>(dp + c, 0.f);
    }
    if constexpr (!BWD) {
      se = block_sum256(se, sh); cnt = block_sum256(cnt, sh);
      if (threadIdx.x == 0) { q.acc[2 * n] = se; q.acc[2 * n + 1] = cnt; }
    }
    return;
  }
  // cross entropy against argmax of the one-hot row (first maximum)
  const long long* oh = reinterpret_cast<const long long*>(q.target) + (size_t)n * K;
  long long bestv = LLONG_MIN; int besti = 0x7fffffff;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < K; c += blockDim.x) {
    const long long v = oh[c];
    if (v > bestv) { bestv = v; besti = c; }
    mx = fmaxf(mx, ldf<T>(pred + c));
  }
  for (int o = 32; o > 0; o >>= 1) {
    const long long ov = __shfl_xor(bestv, o, 64); const int oi = __shfl_xor(besti, o, 64);
    if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  __shared__ long long shv[4];
  if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = bestv; shi[threadIdx.x >> 6] = besti; sh[threadIdx.x >> 6] = mx; }
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (shv[w] > bestv || (shv[w] == bestv && shi[w] < besti)) { bestv = shv[w]; besti = shi[w]; }
    mx = fmaxf(mx, sh[w]);
  }
  float s = 0.f;
  for (int c = threadIdx.x; c < K; c += blockDim.x) s += __expf(ldf<T>(pred + c) - mx);
  s = block_sum256(s, sh);
  const float lse = mx + __logf(s);
  if constexpr (BWD) {
    const float k = q.coef[0];
    for (int c = threadIdx.x; c < K; c += blockDim.x)
      stf<T>(dp + c, k * (__expf(ldf<T>(pred + c) - lse) - (c == besti ? 1.f : 0.f)));
  } else if (threadIdx.x == 0) {
    q.acc[2 * n] = lse - ldf<T>(pred + besti);
    q.acc[2 * n + 1] = 1.f;
  }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void loss_img_kernel(const ImgP q) {
  loss_img_body<T, BWD>(q, blockIdx.x);
}

// several modalities in one launch: blockIdx.y selects the argument record of a device-resident table
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void loss_img_multi_kernel(const ImgP* __restrict__ tab) {
  const ImgP q = tab[blockIdx.y];
  loss_img_body<T, BWD>(q, blockIdx.x);
}

// L_i = sum_i / count_i ; uncertainty: w_i = (exp(-s_i) L_i + s_i) [L_i != 0] (custom_loss.py:19-30)
// acc layout: [T][N][2] per-sample partial {sum, count}
__global__ void loss_finalize_kernel(const float* __restrict__ acc, int Nn, const float* __restrict__ log_vars, int Tn,
                                     float loss_scale, float* __restrict__ losses, float* __restrict__ weighted,
                                     float* __restrict__ total, float* __restrict__ coef, float* __restrict__ dlog_vars) {
  // one wave per modality: its 64 lanes fold the per-sample {sum, count} partials (fixed order -> deterministic)
  __shared__ float w[64];
  const int i = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float ssum = 0.f, scnt = 0.f;
  for (int n = lane; n < Nn; n += 64) {
    const float2 a = *reinterpret_cast<const float2*>(acc + ((size_t)i * Nn + n) * 2);
    ssum += a.x; scnt += a.y;
  }
  ssum = wave_sum(ssum); scnt = wave_sum(scnt);
  if (lane == 0) {
    float wi = 0.f;
    const float Li = ssum / scnt;
    losses[i] = Li;
    float dLi = 1.f;
    if (log_vars) {
      const float s = log_vars[i], e = __expf(-s);
      const float nz = (Li != 0.f) ? 1.f : 0.f;
      wi = (e * Li + s) * nz;
      dLi = e * nz;
      if (dlog_vars) dlog_vars[i] += loss_scale * (1.f - e * Li) * nz;
    } else {
      wi = Li;
    }
    weighted[i] = wi;
    coef[i] = loss_scale * dLi / scnt;
    w[i] = wi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int j = 0; j < Tn; ++j) t += w[j];
    total[0] = t;
  }
}
