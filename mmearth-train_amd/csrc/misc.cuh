// Mask generation, activity maps, weight staging, optimizer (gfx950).
#pragma once
#include "common.cuh"

// ---------------------------------------------------------------------------------
// Random mask from explicit noise (reference: models/fcmae.py:214-231):
//   mask[n,l] = 1 iff rank(noise[n,l]) >= len_keep  (rank by (value, index): a stable argsort)
// also emits the visible-patch table vis[n, slot] (ascending patch index) and inv[n, patch].
// One block per sample.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_gen_kernel(const float* __restrict__ noise, int L, int keep,
                                                       float* __restrict__ mask, int* __restrict__ vis,
                                                       int* __restrict__ inv) {
  extern __shared__ float sh[];            // noise row [L], then flags [L]
  int* flag = reinterpret_cast<int*>(sh + L);
  const int n = blockIdx.x;
  for (int l = threadIdx.x; l < L; l += blockDim.x) sh[l] = noise[(size_t)n * L + l];
  __syncthreads();
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    const float v = sh[l];
    int rank = 0;
    for (int j = 0; j < L; ++j) rank += (sh[j] < v) || (sh[j] == v && j < l);
    const int masked = rank >= keep;
    flag[l] = !masked;
    mask[(size_t)n * L + l] = masked ? 1.f : 0.f;
  }
  __syncthreads();
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    if (flag[l]) {
      int slot = 0;
      for (int j = 0; j < l; ++j) slot += flag[j];
      vis[(size_t)n * keep + slot] = l;
      inv[(size_t)n * L + l] = slot;
    } else {
      inv[(size_t)n * L + l] = -1;
    }
  }
}

// The same mask for the DENSE encoder (FCMAE(sparse=False), models/convnextv2.py:183-191): every patch is computed, so the encoder's
// rows are all N * L patches in patch order and the table only says which of them the mask removed: inv[n, l] = l for a kept patch
// (its row within the sample), -1 for a masked one (zeroed input pixels, mask token in the decoder input, fcmae.py:253-255).
__global__ __launch_bounds__(256) void mask_gen_dense_kernel(const float* __restrict__ noise, int L, int keep,
                                                             float* __restrict__ mask, int* __restrict__ inv) {
  extern __shared__ float sh[];            // noise row [L]
  const int n = blockIdx.x;
  for (int l = threadIdx.x; l < L; l += blockDim.x) sh[l] = noise[(size_t)n * L + l];
  __syncthreads();
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    const float v = sh[l];
    int rank = 0;
    for (int j = 0; j < L; ++j) rank += (sh[j] < v) || (sh[j] == v && j < l);
    const int masked = rank >= keep;
    mask[(size_t)n * L + l] = masked ? 1.f : 0.f;
    inv[(size_t)n * L + l] = masked ? -1 : l;
  }
}

// act0[row] = (sum_c |img[n,c,y,x]| != 0) for every pixel of every visible patch (ME to_sparse)
__global__ __launch_bounds__(256) void activity_kernel(const float* __restrict__ img, const int* __restrict__ vis,
                                                       uint8_t* __restrict__ act, int N, int Cin, int H, int keep,
                                                       int grid, int S) {
  const int total = N * keep * S * S;
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < total; m += gridDim.x * blockDim.x) {
    const int P = S * S, nk = m / P, q = m - nk * P, n = nk / keep;
    const int patch = vis[nk], py = patch / grid, px = patch - py * grid;
    const int y = py * S + q / S, x = px * S + q % S;
    float s = 0.f;
    for (int c = 0; c < Cin; ++c) s += fabsf(img[((size_t)(n * Cin + c) * H + y) * H + x]);
    act[m] = (s != 0.f) ? 1 : 0;      // NaN != 0 is true: NaN pixels are active, as in the reference
  }
}

// act_out[m'] = OR over the k x k children (stride-k pooling of the activity map)
__global__ __launch_bounds__(256) void activity_pool_kernel(const uint8_t* __restrict__ act_in, uint8_t* __restrict__ act_out,
                                                            int Mout, int S, int k) {
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < Mout; m += gridDim.x * blockDim.x) {
    const int P = S * S, nk = m / P, q = m - nk * P, iy = q / S, ix = q - iy * S;
    int a = 0;
    for (int kh = 0; kh < k; ++kh)
      for (int kw = 0; kw < k; ++kw) a |= act_in[nk * (k * k * P) + (k * iy + kh) * (k * S) + (k * ix + kw)];
    act_out[m] = a ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------
// Weight staging: dst[r*dst_ld + c] = T(src[r*sr + c*sc]) for a table of 2-D views
// (casts fp32 master weights to the compute type in the [N][K] layouts the GEMMs read).
// ---------------------------------------------------------------------------------
typedef MpmaePrepDesc PrepDesc;


// Tiled form: a workgroup moves one 64x64 tile through LDS so that the fp32 reads run along
// whichever source dimension is contiguous (plain copies: columns; transposed copies W^T: rows) and
// the compute-type writes always run along destination rows. grid = (max tiles, ndesc).
template <typename T>
__global__ __launch_bounds__(256) void prep_tiled_kernel(const PrepDesc* __restrict__ table) {
  __shared__ float tile[64][65];
  const PrepDesc d = table[blockIdx.y];
  const int tc = (d.cols + 63) / 64, tr = (d.rows + 63) / 64;
  if ((int)blockIdx.x >= tc * tr) return;
  const int r0 = (blockIdx.x / tc) * 64, c0 = (blockIdx.x % tc) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  if (d.sc == 1 || d.sr != 1) {                  // source contiguous along columns (or fully strided)
#pragma unroll 4
    for (int rr = ty; rr < 64; rr += 4) {
      const int r = r0 + rr, c = c0 + tx;
      tile[rr][tx] = (r < d.rows && c < d.cols) ? d.src[(size_t)r * d.sr + (size_t)c * d.sc] : 0.f;
    }
  } else {                                       // source contiguous along rows: read transposed
#pragma unroll 4
    for (int cc = ty; cc < 64; cc += 4) {
      const int r = r0 + tx, c = c0 + cc;
      tile[tx][cc] = (r < d.rows && c < d.cols) ? d.src[(size_t)r + (size_t)c * d.sc] : 0.f;
    }
  }
  __syncthreads();
  T* dst = reinterpret_cast<T*>(d.dst);
#pragma unroll 4
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    if (r < d.rows && c < d.cols) stf<T>(dst + (size_t)r * d.dst_ld + c, tile[rr][tx]);
  }
}

// ---------------------------------------------------------------------------------
// AdamW over the flat fp32 parameter buffer (reference: main_pretrain.py:312-320, betas
// (0.9, 0.95), timm weight-decay grouping: decay[i] != 0 marks decayed elements).
// hp = {lr, 1/(1-beta1^t), 1/sqrt(1-beta2^t), grad_scale}
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const float* __restrict__ hp, float beta1, float beta2,
                                                    float eps, float wd, size_t n,
                                                    const uint8_t* __restrict__ decay, float* __restrict__ gnorm2,
                                                    int slot0, int nslots) {
  // (slot0, nslots): this launch's place among the per-workgroup partials of the step's gradient norm - one launch over the whole
  // buffer: (0, gridDim.x) (a per-gradient-bucket form, mpmae_adamw_part, was built in round 5, did not move the step and was removed in round 6)
  if (hp[4] != 0.f) return;      // non-finite loss this step (hp_fetch): the update is skipped, p / m / v stay intact
  const float lr = hp[0], ibc1 = hp[1], isbc2 = hp[2], gs = hp[3];
  float gsq = 0.f;               // sum g^2 (unscaled): the global gradient norm rides in the pass that reads every gradient anyway
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    gsq += g[i] * g[i];
    const float gi = g[i] * gs;
    float pi = p[i];
    if (decay[i]) pi *= (1.f - lr * wd);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * isbc2 + eps;
    p[i] = pi - lr * ibc1 * mi / denom;
  }
  if (gnorm2) {                  // one partial per workgroup (plain store: 16 k same-address atomics cost 0.2 ms); hp_fetch_kernel folds them
    __shared__ float gred[4];
    gsq = wave_sum(gsq);
    if ((threadIdx.x & 63) == 0) gred[threadIdx.x >> 6] = gsq;
    __syncthreads();
    if (threadIdx.x == 0) gnorm2[1 + slot0 + blockIdx.x] = gred[0] + gred[1] + gred[2] + gred[3];
    if (blockIdx.x == 0 && threadIdx.x == 0 && slot0 == 0) gnorm2[0] = (float)nslots;
  }
}

// ---------------------------------------------------------------------------------
// im2col of the masked image for the sparse 3x3 stem convolution (convnextv2_sparse.py:113-117 on
// MinkowskiOps.to_sparse input): row m = (n, slot, iy, ix) of a visible patch gets the 9 x Cseg taps
// k = (kw*3 + kh)*Cseg + cin of its 3x3 neighbourhood; taps outside the image or inside masked
// patches are absent (zero). One workgroup per visible patch: the (S+2)^2 x Cseg window is staged in
// LDS with row-contiguous reads of the fp32 NCHW image, rows leave as 16-byte vectors. The matrix is
// written once per step and feeds both the forward GEMM and the weight-gradient GEMM.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col3_kernel(const float* __restrict__ img, const int* __restrict__ vis,
                                                      const int* __restrict__ inv, T* __restrict__ out, int ldo,
                                                      int keep, int grid, int S, int Cseg, int H) {
  extern __shared__ float win[];                  // [(S+2)*(S+2)][CP], CP = Cseg | 1 (odd pitch: the 8 taps x channels a lane reads
  __shared__ int nb_ok[9];                        //  per output vector spread over the banks; pitch 12 was 75 % bank conflicts)
  const int CP = Cseg | 1;
  const int nk = blockIdx.x, n = nk / keep;
  const int patch = vis[nk];
  const int py = patch / grid, px = patch - py * grid;
  const int W2 = S + 2, L = grid * grid;
  // visibility of the 3 x 3 neighbour patches first (9 lookups), then every window element UNCONDITIONALLY from a clamped address:
  // `if (inside) if (inv[..] >= 0) v = img[..]` was two dependent round trips per element and loop iteration
  if (threadIdx.x < 9) {
    const int qy = py + (int)threadIdx.x / 3 - 1, qx = px + (int)threadIdx.x % 3 - 1;
    const bool in = qy >= 0 && qx >= 0 && qy < grid && qx < grid;
    nb_ok[threadIdx.x] = in ? (inv[n * L + min(max(qy, 0), grid - 1) * grid + min(max(qx, 0), grid - 1)] >= 0) : 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < W2 * W2 * Cseg; i += blockDim.x) {
    const int wx = i % W2, r = i / W2, wy = r % W2, cin = r / W2;       // x fastest: contiguous image reads
    const int gy = py * S + wy - 1, gx = px * S + wx - 1;
    const int by = wy == 0 ? 0 : (wy == W2 - 1 ? 2 : 1), bx = wx == 0 ? 0 : (wx == W2 - 1 ? 2 : 1);
    const float v = img[((size_t)(n * Cseg + cin) * H + min(max(gy, 0), H - 1)) * H + min(max(gx, 0), H - 1)];
    win[(wy * W2 + wx) * CP + cin] = nb_ok[by * 3 + bx] ? v : 0.f;
  }
  __syncthreads();
  constexpr int EPV = 16 / sizeof(T);
  const int K = 9 * Cseg, vpr = ldo / EPV;
  for (int i = threadIdx.x; i < S * S * vpr; i += blockDim.x) {
    const int q = i / vpr, v = i - q * vpr;
    const int iy = q / S, ix = q - iy * S;
    float o[8];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const int k = v * EPV + e;
      float x = 0.f;
      if (k < K) {
        const int tap = k / Cseg, cin = k - tap * Cseg;
        const int kh = tap % 3, kw = tap / 3;
        x = win[((iy + kh) * W2 + ix + kw) * CP + cin];
      }
      o[e] = x;
    }
    T* dst = out + ((size_t)nk * S * S + q) * ldo + v * EPV;
    if (sizeof(T) == 2) st8<T>(dst, o);
    else *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// Operand matrix of the ORIGINAL ConvNeXtV2 stem (use_orig_stem: convnextv2_sparse.py:99-110, MinkowskiConvolution k = stride = patch / 8):
// out[(n*keep + slot)*64 + iy*8 + ix][(kw*k + kh)*Cseg + cin] = img[n][cin][py*p + iy*k + kh][px*p + ix*k + kw] for the k x k pixels
// under stage-0 point (iy, ix) of visible patch `slot` (ME's kernel-offset order: first spatial coordinate fastest, helpers.py:676-683);
// columns >= k*k*Cseg are written as zeros; inv != NULL (dense encoder, every patch a row): patches with inv < 0 read as zeros. Only pixels of visible patches are read (the masked ones never reach the encoder), all-zero
// pixels contribute zeros by value - their ACTIVITY is a row mask of the GEMM that follows (mpmae_activity / mpmae_activity_pool).
template <typename T>
__global__ __launch_bounds__(256) void gather_kxk_kernel(const float* __restrict__ img, const int* __restrict__ vis, const int* __restrict__ inv,
                                                         T* __restrict__ out, int ldo, int keep, int grid, int p, int k, int Cseg, int H, int rows) {
  const int K = k * k * Cseg;
  const long long total = (long long)rows * ldo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % ldo), row = (int)(i / ldo);
    float v = 0.f;
    if (col < K) {
      const int tap = col / Cseg, cin = col - tap * Cseg, kh = tap % k, kw = tap / k;
      const int nk = row >> 6, q = row & 63, iy = q >> 3, ix = q & 7, n = nk / keep;
      const int patch = vis ? vis[nk] : nk - n * keep;
      const int py = patch / grid, px = patch - py * grid;
      v = img[((size_t)(n * Cseg + cin) * H + py * p + iy * k + kh) * H + px * p + ix * k + kw];
      if (inv && inv[n * grid * grid + patch] < 0) v = 0.f;      // dense encoder: a masked patch is a row whose pixels were zeroed (x *= 1 - mask)
    }
    if (sizeof(T) == 2) reinterpret_cast<bf16_t*>(out)[i] = f2bf(v);
    else reinterpret_cast<float*>(out)[i] = v;
  }
}

// dst[r*dsr + c*dsc] += src[r*sld + c]   (fold a padded, contiguous gradient into a strided parameter layout)
__global__ __launch_bounds__(256) void strided_add_kernel(float* __restrict__ dst, const float* __restrict__ src, int rows,
                                                          int cols, int sld, int dsr, int dsc) {
  const int total = rows * cols;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / cols, c = i - r * cols;
    dst[(size_t)r * dsr + (size_t)c * dsc] += src[(size_t)r * sld + c];
  }
}

// Hyper-parameter hand-over for replayed steps: the host writes record t into slot t % R of a pinned
// ring; this kernel (one per optimizer step, in stream order) copies the slot its device-side
// counter points at into `hp` and advances the counter. A plain async H2D copy would read the
// pinned record when the copy EXECUTES, by which time a host running ahead has overwritten it.
// Meters (optional, `mt.ring != nullptr`): the device-resident form of the reference's MetricLogger / SmoothedValue(window 20)
// (/root/reference/helpers.py:49-109,111-206, engine_pretrain.py:71-113): record t of a ring [window][2T + 2] = {T losses, T weighted
// losses, total loss, gradient norm} plus running sums [2T + 2] and a count, written here at no extra launch; the gradient norm of a
// step is accumulated by adamw_kernel (sum g^2 of the exchanged gradients) and lands in ITS record at the next fetch. The host reads
// ring + sums with one copy per print, ranks fold their sums with one all-reduce per epoch.
struct MeterP { const float* losses; const float* weighted; int T; float* ring; int window; float* sums; float* gnorm2;
                unsigned* err_words; int n_err; int err_stride; };
__global__ __launch_bounds__(1024) void hp_fetch_kernel(const float* __restrict__ ring, int R, int* __restrict__ counter, float* __restrict__ hp,
                                                       const float* __restrict__ total, const MeterP mt) {
  // 1024 threads, every global read issued up front from a clamped address (a runtime-length loop of dependent loads was 64 serial
  // round trips = 30 us at the head of every step); the partial count only masks what was loaded
  __shared__ float red[16];
  const int c = *counter;
  const float gs_prev = hp[3];
  const float tot = total ? *total : 0.f;
  // `total` is the guard loss: in a data-parallel run the all-reduced SUM of the ranks' losses (every rank takes the same skip
  // decision). The meter records the MEAN over ranks like the reference's all_reduce_mean (engine_pretrain.py:104): this step's
  // grad_scale (= 1 / world, 1 on a single rank) is the averaging factor of the same exchange.
  const float gs_new = ring[(size_t)(c % R) * 4 + 3];
  if (mt.ring) {
    const int W = 2 * mt.T + 2;
    const float cntf = mt.sums[W], nbf = mt.gnorm2[0];
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = mt.gnorm2[1 + threadIdx.x + j * 1024];           // capacity 4096 partials (engine.gnorm2)
    const int i = min((int)threadIdx.x, W - 2);
    const float lv = i < mt.T ? mt.losses[i] : i < 2 * mt.T ? (mt.weighted ? mt.weighted[i - mt.T] : 0.f) : tot * gs_new;
    const float sv = mt.sums[i], sg = mt.sums[W - 1];
    const int cnt = (int)cntf, nb = (int)nbf;
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) t += ((int)threadIdx.x + j * 1024 < nb) ? g[j] : 0.f;
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0 && cnt > 0 && nb > 0) {             // gradient norm of the PREVIOUS update (helpers.get_grad_norm_, :509-526); nb == 0:
                                                             // that update was skipped (non-finite loss / barrier timeout) and leaves NO record
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) a += red[j];
      const float gn = sqrtf(a) * gs_prev;
      mt.ring[(size_t)((cnt - 1) % mt.window) * W + W - 1] = gn;
      mt.sums[W - 1] = sg + gn;
    }
    if ((int)threadIdx.x < W - 1) {
      mt.ring[(size_t)(cnt % mt.window) * W + i] = lv;
      mt.sums[i] = sv + lv;
    }
    if (threadIdx.x == 0) { mt.gnorm2[0] = 0.f; mt.sums[W] = (float)(cnt + 1); }      // (partial count 0: a skipped update leaves no norm)
  }
  float h = 0.f;
  if (threadIdx.x < 4) h = ring[(size_t)(c % R) * 4 + threadIdx.x];
  __syncthreads();                                           // hp[3] (previous grad scale) was read above by every thread
  if (threadIdx.x < 4) hp[threadIdx.x] = h;
  if (threadIdx.x == 0) {
    *counter = c + 1;
    // engine_pretrain.py:83-85 stops on a non-finite loss BEFORE the optimizer step; here the check stays on the
    // device: hp[4] makes this step's AdamW a no-op, hp[5] counts skipped steps (the host polls it lazily)
    // ... and a persistent stage kernel whose grid barrier timed out trained this step on partial GRN statistics (ps.cuh
    // grid_barrier): same treatment, counted separately in hp[6] so that the host can tell the two apart
    // (the timed-out rank's loss_finalize has already poisoned ITS loss with +inf - mpmae_loss_finalize_guarded - so in a data-parallel
    // run the all-reduced guard loss is non-finite on EVERY rank and all of them skip; the word is cleared once it has been counted)
    unsigned perr = 0;
    for (int i = 0; i < mt.n_err; ++i) {
      unsigned* w = mt.err_words + (size_t)i * mt.err_stride + 2;
      const unsigned e = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      perr |= e;
      if (e) __hip_atomic_store(w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool bad = !(fabsf(tot) <= 3.0e38f) || perr != 0;
    hp[4] = bad ? 1.f : 0.f;
    if (bad) hp[5] += 1.f;
    if (perr) hp[6] += 1.f;
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// ---------------------------------------------------------------------------------
// Second stage of every large reduction (weight gradients, column statistics, LayerNorm
// parameter gradients): out[addr(e)] += sum_{p < P} part[p*W + e].
// Global float atomics on MI355X serialise at ~12 ns per same-address update (MI355X_MICROARCH
// price list, "fanin"); per-block partial slabs + this kernel replace them. With gridDim.y == 1
// the result is a plain read-modify-write (deterministic); otherwise <= 16-way atomics.
//   MODE 0: addr = e
//   MODE 1: e = n*a + k        -> out[n*b + k*c], only e < d when d > 0   (strided weight layouts)
//   MODE 3: e < a ? out[e] : out2[e - a]; a null out / out2 skips its part (weights then bias)
//   MODE 2: e = tap*a + ch     -> tap < 49 ? out[kh*b + kw*c + ch*d] : out2[ch]   (depthwise 7x7)
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int P, int W,
                                                              float* __restrict__ out, float* __restrict__ out2,
                                                              int a, int b, int c, int d) {
  // block = 64 columns x 4 row lanes; grid = (ceil(W/64), R row chunks)
  __shared__ float red[4][64];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + col;
  const int chunk = (P + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * chunk, p1 = min(P, p0 + chunk);
  float s = 0.f;
  if (e < W) {
#pragma unroll 4
    for (int p = p0 + rl; p < p1; p += 4) s += part[(size_t)p * W + e];
  }
  red[rl][col] = s;
  __syncthreads();
  if (rl != 0 || e >= W) return;
  s = red[0][col] + red[1][col] + red[2][col] + red[3][col];
  float* dst;
  if (MODE == 0) dst = out + e;
  else if (MODE == 1) { if (d > 0 && e >= d) return; const int n = e / a, k = e - n * a; dst = out + (size_t)n * b + (size_t)k * c; }
  else if (MODE == 3) { if (e < a) { if (!out) return; dst = out + e; } else { if (!out2) return; dst = out2 + (e - a); } }
  else {
    const int tap = e / a, ch = e - tap * a;
    if (tap < 49) { const int kh = tap / 7, kw = tap - kh * 7; dst = out + kh * b + kw * c + ch * d; }
    else { if (!out2) return; dst = out2 + ch; }
  }
  if (gridDim.y == 1) *dst += s;
  else atomicAdd(dst, s);
}

// MODE 3 with a per-output-row scale (one-pass losses, loss.cuh): e < nk -> out[e] += rs[e / Kk] * sum, else out2[e - nk] += rs[e - nk] * sum
__global__ __launch_bounds__(256) void reduce_partials_rowscale_kernel(const float* __restrict__ part, int P, int W, float* __restrict__ out,
                                                                       float* __restrict__ out2, int nk, int Kk, const float* __restrict__ rs) {
  __shared__ float red[4][64];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + col;
  const int chunk = (P + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * chunk, p1 = min(P, p0 + chunk);
  float s = 0.f;
  if (e < W) {
#pragma unroll 4
    for (int p = p0 + rl; p < p1; p += 4) s += part[(size_t)p * W + e];
  }
  red[rl][col] = s;
  __syncthreads();
  if (rl != 0 || e >= W) return;
  s = red[0][col] + red[1][col] + red[2][col] + red[3][col];
  float* dst;
  if (e < nk) { if (!out) return; dst = out + e; s *= rs[e / Kk]; }
  else { if (!out2) return; dst = out2 + (e - nk); s *= rs[e - nk]; }
  if (gridDim.y == 1) *dst += s;
  else atomicAdd(dst, s);
}

// One-pass losses: the per-modality gradient scalars coef[t] (mpmae_loss_finalize) applied where the unscaled loss gradient is consumed.
// B [D][ldb] is the staged, transposed head weight (rows = decoder channels, columns = prediction columns): column k is multiplied by
// coef[col_mod[k]] in place (the data-gradient GEMM then computes sum_t coef_t dpred_t W_t); rs[k] = the same scalar per prediction
// column = per ROW of the heads' weight gradient (MpmaeWgradArgs.rowscale).
template <typename T>
__global__ __launch_bounds__(256) void head_scale_kernel(const T* B, T* Bout, int ldb, int D, int W, const uint8_t* __restrict__ col_mod,
                                                         const float* __restrict__ coef, float* __restrict__ rs) {
  const int total = D * W;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / W, k = i - r * W;
    const float c = coef[col_mod[k]];
    stf<T>(Bout + (size_t)r * ldb + k, ldf<T>(B + (size_t)r * ldb + k) * c);
    if (r == 0) rs[k] = c;
  }
}

// Mode-2 second stage (depthwise taps + bias) for a GROUP of problems of one shape: blockIdx.z = problem, pointers from the table.
struct ReduceGroupP { int count, pad; const float* part[12]; float* out[12]; float* out2[12]; };
__global__ __launch_bounds__(256) void reduce_partials_group2_kernel(const ReduceGroupP r, int P, int W, int a, int b, int c, int d) {
  __shared__ float red[4][64];
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(4))) const ReduceGroupP* kgrp_p;
  const kgrp_p g = (kgrp_p)__builtin_amdgcn_kernarg_segment_ptr();
  const float* part = g->part[blockIdx.z];
  float* out = g->out[blockIdx.z];
  float* out2 = g->out2[blockIdx.z];
#else
  const float* part = r.part[0]; float* out = r.out[0]; float* out2 = r.out2[0];
#endif
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + col;
  const int chunk = (P + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * chunk, p1 = min(P, p0 + chunk);
  float s = 0.f;
  if (e < W) {
#pragma unroll 4
    for (int p = p0 + rl; p < p1; p += 4) s += part[(size_t)p * W + e];
  }
  red[rl][col] = s;
  __syncthreads();
  if (rl != 0 || e >= W) return;
  s = red[0][col] + red[1][col] + red[2][col] + red[3][col];
  float* dst;
  const int tap = e / a, ch = e - tap * a;
  if (tap < 49) { const int kh = tap / 7, kw = tap - kh * 7; dst = out + kh * b + kw * c + ch * d; }
  else { if (!out2) return; dst = out2 + ch; }
  if (gridDim.y == 1) *dst += s;
  else atomicAdd(dst, s);
}

// Mode-1 second stage for a GROUP of records in one launch (mpmae_fold_group: the deferred LayerNorm gamma / beta gradient folds of a
// gradient-bucket segment - 20 launches of 5-9 us per step on the weight-gradient lane before): blockIdx.z = record (own P / W / layout,
// read from the kernarg segment), blockIdx.x = 64-column block (blocks beyond a record's width leave at once), blockIdx.y = row chunk.
#define FOLDG_MAX 16
struct FoldGroupP { int count, pad; const float* part[FOLDG_MAX]; float* out[FOLDG_MAX]; int P[FOLDG_MAX], W[FOLDG_MAX], a[FOLDG_MAX], b[FOLDG_MAX], c[FOLDG_MAX]; };
__global__ __launch_bounds__(256) void reduce_partials_group1_kernel(const FoldGroupP r) {
  __shared__ float red[4][64];
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(4))) const FoldGroupP* kgrp_p;
  const kgrp_p g = (kgrp_p)__builtin_amdgcn_kernarg_segment_ptr();
  const float* part = g->part[blockIdx.z];
  float* out = g->out[blockIdx.z];
  const int P = g->P[blockIdx.z], W = g->W[blockIdx.z], a = g->a[blockIdx.z], b = g->b[blockIdx.z], c = g->c[blockIdx.z];
#else
  const float* part = r.part[0]; float* out = r.out[0];
  const int P = r.P[0], W = r.W[0], a = r.a[0], b = r.b[0], c = r.c[0];
#endif
  if ((int)blockIdx.x * 64 >= W) return;
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + col;
  const int chunk = (P + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * chunk, p1 = min(P, p0 + chunk);
  float s = 0.f;
  if (e < W) {
#pragma unroll 4
    for (int p = p0 + rl; p < p1; p += 4) s += part[(size_t)p * W + e];
  }
  red[rl][col] = s;
  __syncthreads();
  if (rl != 0 || e >= W) return;
  s = red[0][col] + red[1][col] + red[2][col] + red[3][col];
  const int n = e / a, k = e - n * a;
  float* dst = out + (size_t)n * b + (size_t)k * c;
  if (gridDim.y == 1) *dst += s;
  else atomicAdd(dst, s);
}

// ---------------------------------------------------------------------------------
// Aligned random crop of the input stage (kornia RandomCrop in FCMAE.forward, models/fcmae.py:419-434): every pixel-wise
// modality of sample n is cut at the SAME window (ty[n], tx[n]) - fp32 bands and int64 class maps alike (the reference
// round-trips the latter through float). dst[n, c, y, x] = src[n, c, ty[n] + y, tx[n] + x]; EB = element bytes (4 / 8).
// One thread per element, consecutive threads consecutive x: reads are contiguous runs of S elements at an arbitrary
// (element-aligned) offset, writes are fully coalesced; the destination is the engine's static input buffer, so the crop
// replaces the device-to-device copy of the input stage instead of adding a pass.
// ---------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void crop_kernel(const E* __restrict__ src, E* __restrict__ dst, int N, int C, int H, int S,
                                                   const int* __restrict__ ty, const int* __restrict__ tx) {
  const long long total = (long long)N * C * S * S;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % S), y = (int)((i / S) % S);
    const long long nc = i / ((long long)S * S);
    const int n = (int)(nc / C);
    dst[i] = src[(nc * H + ty[n] + y) * H + tx[n] + x];
  }
}


// Input stage from RAW tiles (the per-sample work of MMEarthDataset.__getitem__, /root/reference/mmearth_dataset.py:100-142, fused into
// the crop): no-data value -> NaN, per-band z-score, fp32 out:  dst[n,c,y,x] = src == nodata ? NaN : (src - mean[c]) / stdv[c].
// SrcT = the storage type of the raw tile (uint16 Sentinel-2 digital numbers, uint8 canopy height, fp32 Sentinel-1 / ASTER);
// a NaN `nodata` disables the no-data test (the float modalities use -inf, which equality handles).
template <typename SrcT>
__global__ __launch_bounds__(256) void crop_norm_kernel(const SrcT* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int S,
                                                        const int* __restrict__ ty, const int* __restrict__ tx,
                                                        const float* __restrict__ mean, const float* __restrict__ stdv, float nodata) {
  const long long total = (long long)N * C * S * S;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % S), y = (int)((i / S) % S);
    const long long nc = i / ((long long)S * S);
    const int n = (int)(nc / C), c = (int)(nc - (long long)n * C);
    const int oy = ty ? ty[n] : 0, ox = tx ? tx[n] : 0;
    const float v = (float)src[(nc * H + oy + y) * H + ox + x];
    dst[i] = (v == nodata) ? __int_as_float(0x7fc00000) : (v - mean[c]) / stdv[c];
  }
}
// class maps: dst[n,0,y,x] = lut[src] (int64; the table holds the label remap of the modality and -1 for no-data / unknown codes)
__global__ __launch_bounds__(256) void crop_lut_kernel(const uint8_t* __restrict__ src, long long* __restrict__ dst, int N, int H, int S,
                                                       const int* __restrict__ ty, const int* __restrict__ tx, const int* __restrict__ lut) {
  const long long total = (long long)N * S * S;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % S), y = (int)((i / S) % S);
    const int n = (int)(i / ((long long)S * S));
    const int oy = ty ? ty[n] : 0, ox = tx ? tx[n] : 0;
    dst[i] = (long long)lut[src[((long long)n * H + oy + y) * H + ox + x]];
  }
}
