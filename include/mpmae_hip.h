/*
 * mpmae_hip.h — C ABI of libmpmae_hip.so: the MI355X (gfx950) operator library behind the
 * MP-MAE pretraining hot path (FCMAE forward + loss + backward + AdamW).
 *
 * What it replaces. The reference (vishalned/MMEarth-train) is pure Python; its only native
 * boundary on this path is the MinkowskiEngine pybind layer called from
 *   /root/reference/models/convnextv2_sparse.py:37-45,113-129,143-150,199,218
 *   /root/reference/models/sparse_norm_layers.py:24-33,61-77
 * plus the stock torch ops of the dense decoder block (models/convnextv2.py:42-55), the heads
 * (models/fcmae.py:249-265), the losses (models/fcmae.py:267-412, custom_loss.py:19-30) and the
 * optimizer step (main_pretrain.py:312-320). Each entry point below names the reference
 * operation(s) it stands in for.
 *
 * Reductions. Large reductions (weight gradients, column statistics, LayerNorm parameter
 * gradients) are two-stage: per-block partial slabs in caller-provided scratch `ws`, then an
 * in-library second-stage kernel; global float atomics are avoided (they serialise on MI355X).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked "host";
 *   - `dt` selects the activation storage type: 0 = fp32 (exact-f32 MFMA, parity mode),
 *     1 = bf16 (bf16 MFMA, fp32 accumulate). Parameters, gradients and statistics are fp32;
 *   - the compute entry points allocate nothing and never synchronise the host: the caller owns all memory and passes
 *     workspace; work is enqueued on `stream` (a hipStream_t passed as void*). The library's only state is explicit:
 *     the process-wide option table (mpmae_set_option: developer A/B switches, defaults are the measured-best kernels), the
 *     launch-program handles a caller creates (mpmae_program_*: they own HIP events and side streams until destroyed), a
 *     cached device-property query (CU count / resident workgroups of the persistent kernels). The library links NO vendor BLAS
 *     (round 6: the optional hipBLASLt route of rounds 4-5 is removed; `nm -u` shows the HIP runtime and libstdc++ only);
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch.
 *
 * Row layout. A sparse stage holds only the visible patches: row = (n*keep + slot)*S*S + iy*S + ix
 * with S points per patch side; slot <-> patch through vis[n*keep+slot] / inv[n*L+patch] (-1 =
 * masked). Dense decoder rows are n*L + patch. Features are channels-last [rows, C].
 */
#ifndef MPMAE_HIP_H
#define MPMAE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mpmae_stream_t; /* hipStream_t */

typedef struct MpmaeGeom {
  const int* vis;  /* [N*keep] patch index of each visible slot, ascending per sample */
  const int* inv;  /* [N*L] slot of each patch or -1; NULL = every patch visible (dense decoder) */
  int N, keep, grid, S;
} MpmaeGeom;

/* prologue / epilogue selectors of mpmae_gemm / mpmae_wgrad */
enum {
  MPMAE_PRO_NONE = 0, MPMAE_PRO_LN_AFFINE = 1, MPMAE_PRO_GRN = 2, MPMAE_PRO_GRN_BWD = 3,
  MPMAE_PRO_DOWN_GATHER = 4, MPMAE_PRO_ROW_GATHER = 5, MPMAE_PRO_IM2COL3 = 6
};
enum {
  MPMAE_EPI_STORE = 0, MPMAE_EPI_GELU_SUMSQ = 1, MPMAE_EPI_RESID = 2, MPMAE_EPI_DZ_STATS = 3,
  MPMAE_EPI_SCATTER_ROWS = 4, MPMAE_EPI_DOWN_DGRAD = 5
};

/* C[M,N] = epi( pro(A)[M,K] * B[N,K]^T + bias[N] ) */
typedef struct MpmaeGemmArgs {
  const void* A; const void* A2; const void* B; const float* bias;
  void* C; const void* R;
  int M, N, K, lda, ldb, ldc, ldr;
  const float* p0; const float* p1;  /* prologue vectors (see enum comments in csrc/gemm.cuh) */
  int rpg;                           /* rows per statistics group (>= M: one group) */
  float* s0; float* s1;              /* [G][N] fp32 statistics accumulators (atomically added) */
  const int* vis; const int* inv; const uint8_t* act; const uint8_t* act_src;
  int keep, L, S, Cseg, grid;
  int H;
  float* ws; size_t ws_floats;       /* scratch for per-block statistic slabs (stats epilogues) */
} MpmaeGemmArgs;

/* dW[n*sn + k*sk] += sum_m proP(P)[m,n]*proQ(Q)[m,k] ; db[n] += sum_m proP(P)[m,n] */
typedef struct MpmaeWgradArgs {
  const void* P; const void* P2; const void* Q;
  int M, Nn, Kk, ldp, ldq;
  float* dW; int sn, sk; float* db;
  const float* pp0; const float* pp1; const float* qp0; const float* qp1;
  int rpg; int rows_per_split;
  const int* vis; const int* inv; const uint8_t* act_src;
  int keep, L, S, Cseg, grid, H;
  float* ws; size_t ws_floats;       /* scratch for the per-split partial slabs */
  const float* rowscale;             /* optional (NULL): [Nn] per-output-row factors applied by the second-stage fold, dW[n][:] += rowscale[n] * (...),
                                        db[n] += rowscale[n] * (...): the per-modality scalars of the one-pass losses (mpmae_head_scale). bf16,
                                        contiguous dW (sn == Kk, sk == 1), no prologues - otherwise the call fails */
} MpmaeWgradArgs;

typedef struct MpmaeDwArgs {
  const void* x; void* out; const void* add;
  const float* w; const float* bias;
  int s_kh, s_kw, s_c; int flip;
  MpmaeGeom g;
  int C, CC, TP, tiles_side;
  const uint8_t* act;
} MpmaeDwArgs;

typedef struct MpmaeDwWgArgs {
  const void* x; const void* dd;
  float* dw; float* db;
  int s_kh, s_kw, s_c;
  MpmaeGeom g;
  int C, CC, TP, tiles_side, ntiles_total;
  const uint8_t* act;
  float* ws; size_t ws_floats;       /* scratch for the per-block partial slabs */
} MpmaeDwWgArgs;

typedef struct MpmaePrepDesc {
  const float* src; void* dst;
  int rows, cols, sr, sc, dst_ld, pad;
} MpmaePrepDesc;

typedef struct MpmaePixContArgs {
  const void* pred; void* dpred; int ld, coff;
  const float* target; const float* mask;
  int C, p, grid, H, L; int norm_pix;
  float* acc; float* patch_l; float* patch_cnt; float* patch_mean; float* patch_rstd;
  const float* coef;
} MpmaePixContArgs;

typedef struct MpmaePixCatArgs {
  const void* pred; void* dpred; int ld, coff;
  const long long* target; const float* mask;
  int K, p, grid, H, L;
  float* acc; const float* coef;
} MpmaePixCatArgs;

typedef struct MpmaeImgArgs {
  const void* pred; void* dpred; int ld, coff;
  const void* target;
  int K, N, kind;           /* kind 0: CE(argmax one-hot int64), 1: MSE over non-NaN fp32 targets */
  float* acc; const float* coef;
} MpmaeImgArgs;

/* Row-streaming fused pointwise kernels of a sparse ConvNeXtV2 block (bf16 only, H = 4C):
 *   which = 0: x-hat, rstd, xn, h = LN(d) W1^T + b1 and sum gelu(h)^2          (LN + pwconv1)
 *   which = 1: dz = dout W2 and (sum dz, sum dz*gelu(h))                        (pwconv2 data grad)
 *   which = 4: z = gelu(h)*scale + beta (stored to xn), out = x + z W2^T + b2    (GRN apply + pwconv2)
 *   which = 5: dh = (dz*scale + coef*gelu(h))*gelu'(h) stored OVER dz (= A), then dd = LayerNorm-backward(dh W1), dgamma, dbeta
 *   which = 6 (round 6; C = 40 / 80): T[C][H] += dout^T gelu(h), db2[C] += sum_rows dout - the pwconv2 weight gradient BEFORE the GRN affine, in
 *              one read of A = dout [M,C] and R = h [M,H] (csrc/rst.cuh) into one fp32 slab row [C*H | C] per persistent workgroup in ws. Every
 *              consumer is linear in T. With W = staged pwconv2 weights [C][ldw] each workgroup also emits its share of the GRN backward statistics
 *              S0[j] = sum_c W[c][j] db2[c], S1[j] = sum_c W[c][j] T[c][j] as a small row [2][H] behind the big slabs; the call folds ONLY those into
 *              s0 / s1 (added to) - what the fused backward kernel (which = 5) waits for - and writes the number of slab rows to *wg_rows (host); the
 *              parameter gradients dW2[c][j] += scale[j] T[c][j] + beta[j] db2[c], db2 come from mpmae_rs_wgrad_fold(C, H, ws, rows, scale, beta, ..)
 *              whenever the caller likes (the weight-gradient lane): ws must stay untouched until then, ws_floats >= workgroups * (C*H + C + 2H).
 *              The statistics pass (which = 1, out = NULL) AND the separate weight-gradient product over the same two tensors both go away.
 *              With W == NULL: s0 = T, s1 = db2 raw (added to), mpmae_grn_stats_from_wgrad as a separate second step
 *   (which = 2 / 3 - pwconv2 / pwconv1 data gradient on MATERIALISED z / dh, the resident-weights kernels of rounds 1-3 - were
 *   removed in round 4: invalid value)
 * C in {40,80,96,160,192,320,384}: weights streamed through LDS in chunks (csrc/rsc.cuh), any M;
 * scale / coef are [M / rpg][H] (rpg = rows per GRN group; 0 = one group).
 * Field use per kernel is documented in those files. Replaces convnextv2_sparse.py:40-43,55 and
 * their autograd. */
/* out[(e / a) * b + (e % a) * c] += sum_{p < P} part[p * W + e], e < W  (second stage of a two-stage reduction) */
typedef struct MpmaeFoldDesc { const float* part; int P, W; float* out; int a, b, c; } MpmaeFoldDesc;

typedef struct MpmaeRsArgs {
  const void* A; const void* A2; const void* W; int ldw;
  const float* bias; const float* v0; const float* v1;
  void* out; void* xhat; void* xn; float* rstd; const void* R; const float* lng;
  float* ws; const uint8_t* act; int M;
  int C, H;                        /* layer shape */
  float* s0; float* s1;            /* statistics / parameter-gradient outputs (accumulated) */
  size_t ws_floats;
  int rpg;                         /* rows per GRN group (which = 4, 5); 0 = all rows */
  /* Optional (which = 4, 5; single GRN group; C = 40 / 80 / 160): mpmae_grn_fwd_finalize / mpmae_grn_bwd_finalize
   * folded into the kernel's prologue (every workgroup recomputes the H-vector, workgroup 0 publishes it);
   * fin_sum == NULL keeps v0 / v1 as inputs.
   *   which 4: fin_sum = G2[H] -> Gx, Ainv, scale = 1 + gamma Gx Ainv written to fin_gx, fin_ainv, fin_out; v1 = GRN beta
   *   which 5: fin_sum = S1[H], fin_sum0 = S0[H], fin_gx / fin_ainv read -> coef (also to fin_out if set), fin_dgamma /
   *            fin_dbeta accumulated; v0 = scale */
  const float* fin_sum; const float* fin_sum0; const float* fin_gamma;
  float* fin_gx; float* fin_ainv; float* fin_out; float* fin_dgamma; float* fin_dbeta;
  float fin_eps;
  /* Optional (C = 40 / 80, single GRN group): the wide tensors never materialised. which = 0 / 1 with out == NULL only
   * produce the statistics (which 0 still writes x-hat, xn, rstd). which = 5 with dz_dout / dz_w2t set recomputes
   * dz = dout W2 chunk by chunk instead of reading A (A is then only the destination of dh): dz_dout = dout [M,C],
   * dz_w2t = staged W2^T [H][dz_ldw2]. which = 4 with the same fields recomputes h = xn W1^T + b1 instead of reading A:
   * dz_dout = xn [M,C], dz_w2t = staged W1 [H][dz_ldw2], dz_bias = b1 [H]. */
  const void* dz_dout; const void* dz_w2t; int dz_ldw2; const float* dz_bias;
  /* Optional (which = 5): the fold of the LayerNorm gamma / beta gradient partials (one slab row per workgroup in `ws`) is NOT
   * launched; its description is written to *defer_fold (HOST memory) instead, for a later mpmae_fold_group call - nothing on the
   * data-gradient chain reads those gradients, so a training step folds the slabs of a whole stage on its weight-gradient lane.
   * `ws` must then stay untouched until that call has run. */
  MpmaeFoldDesc* defer_fold;
  /* Optional (which = 0, C = 160 / 320): A already IS the LayerNorm output xn (stored by mpmae_dwln_fwd) - no LayerNorm here, x-hat / rstd / xn
   * are not written: h = A W1^T + b1 and the GELU^2 column sums only. */
  int ln_done;
  /* Optional (which = 5 at C = 40 with dz_dout set; round 6): pwconv1's weight gradient INSIDE the fused backward kernel. By linearity
   * dW1 = dh^T (x-hat * gamma + beta) = gamma[c] * (dh^T x-hat)[j][c] + beta[c] * db1[j]; dh and x-hat are both in the kernel's registers, so every
   * persistent workgroup accumulates U = dh^T x-hat and db1 = sum_rows dh over its row tiles (transposing LDS reads, MFMA) and writes ONE fp32 slab
   * row [H * C | H] to wg_ws; *wg_rows (HOST memory) receives the number of rows written. dh is then NOT stored (A is untouched), and neither
   * pwconv1's transpose-read product over dh and xn nor the forward's xn store is needed. mpmae_rs_wgrad_fold folds the slab rows into the
   * parameter gradients. wg_ws_floats >= workgroups * (H * C + H) (2 workgroups per CU). which = 6 writes ITS slab-row count to *wg_rows too. */
  float* wg_ws; size_t wg_ws_floats; int* wg_rows;
  /* Optional (which = 4, C = 40 / 80 / 96, the LAST block of a stage; round 6): the LayerNorm in front of the 2x2/2 downsample convolution
   * (convnextv2_sparse.py:131-137) fused into this kernel's epilogue - exactly mpmae_ln_fwd_down on the row it has just computed: dn_xhat [M,C],
   * dn_rstd [M], dn_y = affine output in the grouped [M/4][4C] operand layout of the convolution, dn_gamma / dn_beta [C], dn_S = points per patch
   * side of THIS stage (even). `out` may then be NULL (nothing else reads the stage output). */
  void* dn_xhat; float* dn_rstd; void* dn_y; const float* dn_gamma; const float* dn_beta; int dn_S;
} MpmaeRsArgs;
int mpmae_rs(int which, const MpmaeRsArgs* args, mpmae_stream_t stream);
/* Second stage of the weight gradients accumulated inside main-lane kernels (MpmaeRsArgs.wg_ws; mpmae_rs which = 6): `rows` fp32 slab rows [A*B | A] ->
 *   dW[a][b] += v0[b] * sum_p X_p[a][b] + v1[b] * sum_p d_p[a];   db[a] += sum_p d_p[a]
 * pwconv1 (convnextv2_sparse.py:41 under autograd): (A, B) = (H, C), X = dh^T x-hat, v0 / v1 = LayerNorm gamma / beta;
 * pwconv2 (:43): (A, B) = (C, H), X = dout^T gelu(h), v0 = GRN scale (1 + gamma Nx), v1 = GRN beta. B >= 32. */
int mpmae_rs_wgrad_fold(int A, int B, const float* slabs, int rows, const float* v0, const float* v1, float* dW, float* db, mpmae_stream_t stream);
int mpmae_fold_group(const MpmaeFoldDesc* descs, int count, mpmae_stream_t stream);

/* ---- persistent per-sample stage kernels (csrc/ps.cuh) ----------------------------------------
 * ONE launch runs every Block of a sparse stage (convnextv2_sparse.py:47-56 x depth, with MinkowskiLayerNorm and the
 * batch-global MinkowskiGRN of sparse_norm_layers.py:24-33,61-77): depthwise 7x7 -> LayerNorm -> pwconv1 -> GELU -> GRN ->
 * pwconv2 -> + residual, block after block, one workgroup per sample with the sample's rows resident in LDS and ONE grid
 * barrier per block for the GRN statistics (device-scope float atomics into G2 + a monotonic arrival counter).
 * bf16 activations only. Supported shapes: (C = 160, S = 2, keep <= 20) and (C = 320, S = 1, keep <= 32), N <= number of
 * compute units of the device (every workgroup must be resident), nblk <= MPMAE_PS_MAXBLK; anything else returns
 * hipErrorInvalidValue and the caller uses mpmae_dwconv7_fwd / mpmae_rs. `sync` = `sync_words` zero-initialised device words, 4
 * {arrivals, departures, error, -} for the flat arrival counter or MPMAE_PS_SYNC_WORDS (640: + a top counter, 8 group counters and 8 group
 * flags on 128-byte lines of their own) for the XCD-hierarchical barrier of round 6 (8 groups of workgroups arrive on their own counter, the
 * groups' last arrivers on the top counter; MI355X_MICROARCH.md "barrier-xcd"): the kernel leaves every counter at zero again; error != 0 after a launch means a
 * workgroup never became resident within the spin bound (e.g. another persistent kernel shares the GPU) and the results
 * are invalid. The forward writes, per block, exactly what the backward and the weight gradients read: x-hat, rstd, xn
 * (LayerNorm output), h (pwconv1 output), z (GRN output), out, and Gx / Ainv / scale of the GRN. */
#define MPMAE_PS_MAXBLK 9
#define MPMAE_PS_SYNC_WORDS 640
typedef struct MpmaePsBlock {
  const float* dw_w; const float* dw_b;        /* ME depthwise kernel (49, C), index (kw*7 + kh)*C + c; bias (C) */
  const float* ln_g; const float* ln_b;
  const void* W1; const float* b1;             /* staged bf16 [4C][ldw1] (mpmae_prep_weights), bias (4C) */
  const float* grn_g; const float* grn_b;
  const void* W2; const float* b2;             /* staged bf16 [C][ldw2], bias (C) */
  int ldw1, ldw2;
  void* dhat; float* rstd; void* xn; void* h; void* z; void* out;   /* saved activations [M, C] / [M] / [M, 4C] */
  float* G2; float* Gx; float* Ainv; float* scale;                  /* GRN: G2 = [ng][4C] partial sums of gelu(h)^2 (zeroed by the caller,
                                                                       workgroup n accumulates into copy n % ng); results (4C each) */
} MpmaePsBlock;
typedef struct MpmaePsArgs {
  const void* x_in;                            /* stage input rows [M, C] bf16 */
  MpmaeGeom g; const uint8_t* act;
  int C, nblk; float eps;
  int ng;                                      /* accumulator copies per statistics vector (1 .. 16): measured best 4 at 256 workgroups */
  unsigned* sync;
  int sync_words, pad_;                        /* words behind `sync`: >= MPMAE_PS_SYNC_WORDS selects the XCD-hierarchical grid barrier, 4 .. the flat counter */
  MpmaePsBlock blk[MPMAE_PS_MAXBLK];
} MpmaePsArgs;
int mpmae_ps_fwd(const MpmaePsArgs* args, mpmae_stream_t stream);

/* im2col of the masked fp32 NCHW image for the sparse 3x3 stem convolution (MinkowskiConvolution
 * 3x3 of convnextv2_sparse.py:113-117): out[(n*keep+slot)*S*S + iy*S + ix][k], k = (kw*3+kh)*Cseg + cin
 * (taps outside the image / in masked patches are zero), row stride ldo >= 9*Cseg (padding columns
 * are written as zeros). The plain GEMM / weight-gradient entry points then run on `out`. */
int mpmae_im2col3(int dt, const float* img, const int* vis, const int* inv, void* out, int ldo, int N,
                  int keep, int grid, int S, int Cseg, int H, mpmae_stream_t stream);
/* Operand matrix of the ORIGINAL ConvNeXtV2 patchify stem (use_orig_stem=True: convnextv2_sparse.py:99-110,202-203 - MinkowskiConvolution
 * k = stride = patch / 8 on the to_sparse input; convnextv2.py:97-106 for the dense encoder): out[(n*keep+slot)*64 + iy*8 + ix][(kw*k+kh)*Cseg + cin]
 * = the k x k pixels under stage-0 point (iy, ix) of visible patch `slot` (vis = NULL: every patch, slot = patch; inv != NULL: a patch with inv[n*L + patch] < 0 - masked, dense encoder - reads as zeros), row stride ldo >= k*k*Cseg
 * (padding columns are zeros). The convolution is then a plain mpmae_gemm on `out` (row mask = the pooled activity map), its weight
 * gradient a plain mpmae_wgrad. Needs p == 8 k, H == grid p. */
int mpmae_gather_kxk(int dt, const float* img, const int* vis, const int* inv, void* out, int ldo, int N, int keep, int grid,
                     int p, int k, int Cseg, int H, mpmae_stream_t stream);
/* dst[r*dst_sr + c*dst_sc] += src[r*src_ld + c]: folds a padded contiguous gradient into a strided
 * parameter layout (e.g. ME's (9, Cin, Cout) convolution kernel). */
int mpmae_strided_add(float* dst, const float* src, int rows, int cols, int src_ld, int dst_sr,
                      int dst_sc, mpmae_stream_t stream);

/* Fused tail of the sparse stem when the stem depthwise kernel is 1x1 (patch size 8):
 * out = LN2(act_out * (w * GELU(LN1(x)) + wb)) with both LayerNorms' x-hat / rstd saved, and the
 * whole backward (dx, and d{g1,b1,w,wb,g2,b2} accumulated) in one pass. Replaces
 * convnextv2_sparse.py:113-127 minus the 3x3 convolution, and their autograd. */
typedef struct MpmaeStemTailArgs {
  const void* x;            /* fwd: conv output [M,C]; bwd: gradient wrt the stem output [M,C] */
  void* xhat1; float* rstd1; void* xhat2; float* rstd2;
  void* out;                /* fwd: stem output; bwd: gradient wrt the conv output */
  const float* g1; const float* b1; const float* w; const float* wb; const float* g2; const float* b2;
  const uint8_t* act_in; const uint8_t* act_out;
  float* dg1; float* db1; float* dw; float* dwb; float* dg2; float* db2;   /* bwd only */
  float* ws; size_t ws_floats;
  int M, C;
} MpmaeStemTailArgs;
int mpmae_stem_tail(int dt, int bwd, const MpmaeStemTailArgs* args, mpmae_stream_t stream);

/* The whole sparse stem forward of patch size 8 in one launch (bf16 storage): masked 3x3 convolution of the fp32 image
 * (convnextv2_sparse.py:113-117 on MinkowskiOps.to_sparse input; taps outside the image or inside masked patches are absent) with the
 * bf16-rounded weights (staged W [C0][ldw] or the fp32 master W_master; k = (kw*3 + kh)*Cin + cin, the order of mpmae_im2col3) + bias, followed by mpmae_stem_tail's forward.
 * Replaces mpmae_im2col3 + the stem GEMM + mpmae_stem_tail(fwd); the convolution output is not materialised
 * (the backward reads xhat1 / rstd1 / xhat2 / rstd2 only). track_activity != 0: rows whose pixel is all-zero are inactive (zeros
 * everywhere), as the activity map of mpmae_activity says. Requires S = 8, Cin <= 12, C0 <= 48, C0 % 4 == 0, ldw % 8 == 0. */
typedef struct MpmaeStemFrontArgs {
  const float* img; const int* vis; const int* inv;       /* [N,Cin,H,H], mask tables of mpmae_mask_gen */
  const void* W; int ldw;   /* staged bf16 weights [C0][ldw], or NULL when W_master is given */
  const float* W_master;    /* fp32 parameter in ME layout [9*Cin][C0] (k = (kw*3 + kh)*Cin + cin): rounded to bf16 in the kernel's prologue,
                               the launch then depends on no weight staging */
  const float* bias;
  void* xhat1; float* rstd1; void* xhat2; float* rstd2; void* out;      /* [N*keep*64, C0] / [N*keep*64] */
  const float* g1; const float* b1; const float* w; const float* wb; const float* g2; const float* b2;
  void* col; int ldc;       /* optional (NULL): also write the bf16 im2col matrix [N*keep*64][ldc] (ldc % 8 == 0, ldc >= 9 Cin, the layout
                               of mpmae_im2col3) that the stem's weight gradient reads - from the kernel's own MFMA operand fragments */
  int N, keep, grid, H, Cin, C0, track_activity;
  uint8_t* act_out;         /* optional (NULL): the pixel-activity byte of every row [N*keep*64] (what mpmae_activity writes), from the window's centre tap */
} MpmaeStemFrontArgs;
int mpmae_stem_front(const MpmaeStemFrontArgs* args, mpmae_stream_t stream);

/* ---- input stage ---------------------------------------------------------------------------- */
/* Aligned random crop of FCMAE.forward (kornia RandomCrop, models/fcmae.py:419-434): dst[n,c,y,x] = src[n,c,ty[n]+y,tx[n]+x]
 * for one pixel-wise modality ([N,C,H,H] -> [N,C,S,S]; elem_bytes 4 = fp32 bands, 8 = int64 class maps); the same per-sample
 * windows ty / tx (device int32 [N]) are used for every modality. */
int mpmae_crop(const void* src, void* dst, int elem_bytes, int N, int C, int H, int S, const int* ty, const int* tx,
               mpmae_stream_t stream);

/* The same crop fused with the per-sample preparation of RAW tiles that the reference does on the host in
 * MMEarthDataset.__getitem__ (mmearth_dataset.py:100-142): continuous modalities dst = (src == nodata ? NaN : (src - mean[c]) / std[c])
 * as fp32 (src_type 0 = fp32, 1 = uint16, 2 = uint8 raw storage; nodata = MODALITIES.NO_DATA_VAL, NaN disables the test);
 * class maps dst = lut256[src] as int64 (the table carries the label remap of :95-107 and -1 for no-data / unknown codes).
 * ty / tx may both be NULL (no crop: H == S). */
int mpmae_crop_norm(const void* src, int src_type, float* dst, int N, int C, int H, int S, const int* ty, const int* tx,
                    const float* mean, const float* stdv, float nodata, mpmae_stream_t stream);
int mpmae_crop_lut(const uint8_t* src, long long* dst, int N, int H, int S, const int* ty, const int* tx, const int* lut256,
                   mpmae_stream_t stream);

/* ---- masks / activity --------------------------------------------------------------------- */
/* FCMAE.gen_random_mask (models/fcmae.py:214-231) on explicit noise [N,L]: mask f32 [N,L]
 * (1 = removed), vis [N,keep], inv [N,L]. */
int mpmae_mask_gen(const float* noise, int N, int L, int keep, float* mask, int* vis, int* inv,
                   mpmae_stream_t stream);
/* The same mask for FCMAE(sparse=False) (models/convnextv2.py:183-191: the dense encoder computes every patch and only zeroes the
 * masked input pixels): mask f32 [N,L] as above; inv [N,L] = l for a kept patch (its row within the sample, all L patches being
 * rows), -1 for a masked one. */
int mpmae_mask_gen_dense(const float* noise, int N, int L, int keep, float* mask, int* inv, mpmae_stream_t stream);
/* MinkowskiOps.to_sparse activity rule (convnextv2_sparse.py:199): act[row] = sum_c|x| != 0. */
int mpmae_activity(const float* img, const int* vis, uint8_t* act, int N, int Cin, int H, int keep,
                   int grid, int S, mpmae_stream_t stream);
/* ME strided-conv output-coordinate rule: parent active iff any of its k x k children is. */
int mpmae_activity_pool(const uint8_t* act_in, uint8_t* act_out, int Mout, int S, int k,
                        mpmae_stream_t stream);

/* ---- weight staging ------------------------------------------------------------------------ */
/* casts / transposes fp32 master weights into the [N][K] compute-type layouts the GEMMs read;
 * `table` is a device array of ndesc descriptors; max_tiles = max over the views of
 * ceil(rows/64)*ceil(cols/64) (each view is moved in 64x64 tiles through LDS). */
int mpmae_prep_weights(int dt, const MpmaePrepDesc* table, int ndesc, int max_tiles,
                       mpmae_stream_t stream);

/* ---- MFMA GEMMs ---------------------------------------------------------------------------- */
/* MinkowskiLinear / nn.Linear / 1x1 nn.Conv2d / MinkowskiConvolution (3x3 s1 via im2col, 2x2 s2
 * via child gather) forward and data-gradient, with LayerNorm-affine, GELU, GRN, residual and
 * statistics fused (convnextv2_sparse.py:41-43,55,113-116,143-150; convnextv2.py:46-52;
 * fcmae.py:251,264). `args` is a HOST pointer (copied into the kernel argument buffer). */
int mpmae_gemm(int dt, int pro, int epi, const MpmaeGemmArgs* args, mpmae_stream_t stream);
/* weight + bias gradients of the same layers (autograd of the reference ops). */
int mpmae_wgrad(int dt, int ppro, int qpro, const MpmaeWgradArgs* args, int splits,
                mpmae_stream_t stream);
/* The same gradients for `count` layers of IDENTICAL shape in one launch + one fold: all pwconv1 / pwconv2 weight gradients of
 * an encoder stage (autograd of MinkowskiLinear, models/convnextv2_sparse.py:41-43,51-53), whose operands persist until the stage's
 * data-gradient chain is done. probs[i] as for mpmae_wgrad with no prologues (its ws / ws_floats are ignored); ws: fp32 scratch
 * for the per-split partial slabs of ALL problems. bf16, widths that are multiples of 80 (C = 80: 4C % 320 == 0; else C % 160 == 0),
 * any dW strides: the grouped DMA-ring kernel (gemm_tng.cuh); anything else: count calls of mpmae_wgrad (same results). */
int mpmae_wgrad_group(int dt, const MpmaeWgradArgs* probs, int count, float* ws, size_t ws_floats,
                      mpmae_stream_t stream);

/* ---- MX-fp8 pointwise path (BASELINE configs[4]: "fp8 MFMA pointwise path") ------------------
 * The nn.Linear / 1x1-conv layers with K % 128 == 0 (decoder Block pwconv1/2 forward and data gradient,
 * convnextv2.py:46-52) on v_mfma_scale_f32_16x16x128_f8f6f4: operands are OCP e4m3 bytes with one E8M0 scale per 32
 * consecutive k (OCP MX v1.0 block format), quantised from bf16 by mpmae_quant_mx:
 *   q[rows][K] bytes; scales[K/128][lds] dwords (byte b of dword [slab][row] = scale of block 4*slab + b), lds >= rows.
 * mpmae_gemm_mx: C[M,N] (bf16) = deq(A)[M,K] deq(B)[N,K]^T + bias (+ R); args->A / B are the e4m3 matrices (lda / ldb in
 * BYTES = elements), epi = EPI_STORE (0) or EPI_RESID (2); fp32 accumulation, bf16 output. */
int mpmae_quant_mx(const void* x_bf16, int ld, int rows, int K, void* q, uint32_t* scales, int lds, mpmae_stream_t stream);
int mpmae_gemm_mx(int epi, const MpmaeGemmArgs* args, const uint32_t* scales_a, int lsa, const uint32_t* scales_b, int lsb,
                  mpmae_stream_t stream);

/* ---- library options ------------------------------------------------------------------------
 * Process-wide kernel-selection / launch-shape switches for A/B measurements (set them before recording launch programs:
 * a recorded launch keeps the choice it was recorded with). The library reads NO environment variables; this table and
 * recorded programs are the only state it keeps - every other entry point is a pure function of its arguments. Defaults
 * are the measured-best choices on MI355X. */
enum MpmaeOption {
  MPMAE_OPT_DW = 0,   /* default 8: depthwise forward / data-gradient kernels: 8 = matrix-core kernels at S = 8 / 4 (dwmfma.cuh, bf16-rounded taps) + 6 elsewhere; 7 = (the band kernel of rounds 2-3, removed in round 4: same as 6); 6 = packed per-sample kernels; 5 = per-sample LDS-map kernels (fp32 mode); < 5 = the generic positional-tile kernels of dwconv.cuh (S = 1 in fp32 mode, bf16 at S = 1 on grids other than 7 x 7, odd shapes; the wave-granular 8 x 8-tile kernels of dwconv3.cuh that stood in front of them were removed in round 6). The per-patch (v4) and block-granular (v2) generations were removed in round 5 */
  MPMAE_OPT_DWW,   /* default 7: depthwise weight gradient: 7 = matrix-core kernels at S = 8 / 4 (dwmfma_wg.cuh) + 5 elsewhere; 5 = per-sample LDS-map kernels everywhere (6, a packed kernel for S >= 2 that measured slower, was removed in round 6: the value now behaves like 5) */
  MPMAE_OPT_NT_GLDS64,   /* default 1: direct-to-LDS NT GEMM also for 64-wide N tiles */
  MPMAE_OPT_NT_BK32,   /* default 1: 32-deep K slabs for K <= 512 */
  MPMAE_OPT_NT_GLDS,   /* default 1: direct-to-LDS operand slabs in the NT GEMM (2 = always 32-deep) */
  MPMAE_OPT_TN,   /* default 2: weight-gradient kernel: 2 = transpose-read (ds_read_b64_tr_b16), 1 = register-transposing */
  MPMAE_OPT_CS_SPLIT,   /* default 1: column-statistics kernel: split rows over workgroups */
  MPMAE_OPT_RSC_PF,   /* default 1: LDS-staged GRN vectors / early operand issue in the narrow row-streaming kernels */
  MPMAE_OPT_RSC_N40,   /* default 2: narrow-kernel variant at C = 40: 1 = two row tiles per wave, otherwise one */
  MPMAE_OPT_RSC_N80,   /* default 1: narrow-kernel variant at C = 80: 0 = two row tiles per wave, otherwise one */
  MPMAE_OPT_TN3_BLOCKS,   /* default 128: target workgroup count of the DMA-ring weight-gradient kernel for the decoder / head shapes (gemm_tn3.cuh; 0 = use gemm_tn2). Re-swept after the matrix-core depthwise kernels rebalanced the lanes: 128 (half the CUs, half the slabs) 3.87-3.88 vs 256 3.91 ms in three interleaved pairs - the weight-gradient lane's kernels leave CUs to the main lane's */
  MPMAE_OPT_TNG_BLOCKS,   /* default 256 (round 6, with the lighter weight-gradient lane: 3.428 / 3.435 vs 3.448 / 3.451 ms at 512, profiles/r06/option_sweep.txt; half the slab rows for its fold): target workgroup count of the GROUPED weight-gradient kernel (gemm_tng.cuh; 0 = one mpmae_wgrad per problem) */
  MPMAE_OPT_FOLD_GROUP,   /* default 0: 1 = mpmae_fold_group folds up to 16 records per launch (blockIdx.z = record) instead of one launch per record - measured SLOWER in the step (4.035-4.04 vs 4.005-4.026 ms, three interleaved pairs): the small launches slot in between the weight-gradient lane's kernels, the grouped one waits for all its producers. Round 5: the three folds behind the stem-tail backward kernel (ONE producer, the exposed tail of the step) are one grouped launch unless the value is < 0 */
  MPMAE_OPT_RSC_W5,   /* default 1: 80-row (5-wave) tiles in the narrow fused pointwise kernels at C = 160 when 64-row tiles need more than one round of workgroups and 80-row tiles do not */
  MPMAE_OPT_RSC_ATOMIC,   /* default 0: largest row-block count of a WIDE fused pointwise launch (mpmae_rs which = 0 / 1) whose GRN column statistics are added straight into s0 / s1 with hardware float atomics instead of slab rows + a second-stage fold launch (0 = never) */
  MPMAE_OPT_DET,   /* default 0: 1 = reproducible statistics: every second-stage fold runs as ONE row group per column block (fixed summation order, no atomics between row groups). With the engine option det = 1 (which also keeps the persistent stage kernel and its float atomics out of the program) two forwards of the same weights and inputs are bit-identical; -1 = the pre-round-4 behaviour of the wide pointwise kernels everywhere (one shared LDS statistics row, float atomics between the waves) for A/B: by default a row per wave is used wherever it does not cost a resident workgroup per CU */
  MPMAE_OPT_RSC1,   /* default 1: the WIDE fused pointwise kernels (mpmae_rs which = 0 / 1) at C = 160 / 320 in their one-shot form (csrc/rsc1.cuh: the workgroup's whole weight slice global -> LDS by DMA at the top, one barrier, no chunk loop); value = row tiles per wave (1 or 2); 0 = the chunk-streaming kernels of rsc.cuh */
  MPMAE_OPT_RSC1_ATOMIC,   /* default 100: the one-shot wide kernels add their column statistics straight into s0 / s1 with float atomics (no slab rows, no fold launch) when they run at most this many workgroup rows (0 = never; MPMAE_OPT_DET > 0 = never) */
  MPMAE_OPT_RSP,   /* default 1: the fused pointwise kernels at C = 40 / 80 in their persistent burst-load form (csrc/rsp.cuh: weights resident in LDS, a workgroup walks row tiles, every operand of the next tile requested under the arithmetic of the current one); value = row tiles of 16 rows per wave (1 or 2); 0 = the chunk-streaming kernels of rsc.cuh */
  MPMAE_OPT_RSP_NWV,   /* default 0 = automatic (8 for which = 5 at C = 80, else 4): waves per workgroup of the persistent NARROW kernels (one 16-row tile per wave) */
  MPMAE_OPT_RSP_NARROW,   /* default 2: which NARROW launches take the persistent burst-load kernels: bit 0 = which 4 at C = 40, bit 1 = which 5 at C = 40, bit 2 / 3 = the same at C = 80 (measured: only which 5 at C = 40 gains, 88.6 -> 70 us) */
  MPMAE_OPT_RSN3,   /* default 5: the fused pointwise BACKWARD kernel at C = 160 (mpmae_rs which = 5, single GRN group, dz materialised) in its ring-pipelined form (csrc/rsn3.cuh: weight slabs by DMA into a three-slot LDS ring, rows two chunks ahead, one bare barrier per chunk); value = waves per workgroup (4 or 5); 0 = rsc_narrow */
  MPMAE_OPT_EVX,   /* default 1: in a launch program an op's cross-lane signal is the completion event of its last kernel launch (hipExtLaunchKernelGGL stopEvent) instead of a hipEventRecord - a barrier packet of its own - behind it: 1.65 us less per signal on the signalling lane (tools/probes/ext_event_probe.hip); 0 = hipEventRecord */
  MPMAE_OPT_RST_NW,   /* default 16: waves per workgroup of that pass (16: 1024 threads, 256-row tiles, one workgroup per CU - a quarter of the slab rows; 4: 256 threads, 64-row tiles, 2-3 per CU) */
  MPMAE_OPT_NT_RING,   /* default 1: ring form of the direct-to-LDS NT GEMM (gemm_nt_ring_kernel, NST stages of BK-deep slabs): 1 = by shape (<= one tile per CU: 3 x 64; K <= 512: 3 x 32; else the two-buffer kernel), 0 = off, NST * 100 + BK (332, 432, 632, 364, 464) = that ring for every eligible product (probes) */
  MPMAE_OPT_COUNT_
};
int mpmae_set_option(int option, int value);
int mpmae_get_option(int option);

/* ---- row-wise ops -------------------------------------------------------------------------- */
/* MinkowskiLayerNorm / LayerNorm (sparse_norm_layers.py:61-77; norm_layers.py:23-31):
 * xhat = (x-mean)*rstd (biased var), optional y = act(xhat*gamma+beta), act 0 = id, 1 = GELU. */
int mpmae_ln_fwd(int dt, const void* x, void* xhat, float* rstd, void* y, const float* gamma,
                 const float* beta, int act, float eps, int M, int C, const uint8_t* rowmask,
                 mpmae_stream_t stream);
int mpmae_ln_bwd(int dt, const void* dy, int dy_div, float dy_scale, const void* xhat,
                 const float* rstd, const float* gamma, const float* beta, int act, void* dx,
                 int accumulate, float* dgamma, float* dbeta, int M, int C, const uint8_t* rowmask,
                 float* ws, size_t ws_floats, mpmae_stream_t stream);
/* LayerNorm in front of a 2x2-stride-2 downsample convolution (convnextv2_sparse.py:131-137): the
 * affine output of child row (patch, cy, cx) of a stage with S points per patch side is written to
 * y_grouped[parent (patch, cy/2, cx/2)][(cx&1)*2 + (cy&1)][C], i.e. directly in the [M/4][4C] operand
 * layout of the convolution-as-GEMM (k = kidx*C + c, ME kernel order); inactive rows write zeros.
 * The backward reads dy from the same grouped layout. */
int mpmae_ln_fwd_down(int dt, const void* x, void* xhat, float* rstd, void* y_grouped, const float* gamma,
                      const float* beta, float eps, int M, int C, int S, const uint8_t* rowmask,
                      mpmae_stream_t stream);
int mpmae_ln_bwd_down(int dt, const void* dy_grouped, const void* xhat, const float* rstd,
                      const float* gamma, void* dx, float* dgamma, float* dbeta, int M, int C, int S,
                      const uint8_t* rowmask, float* ws, size_t ws_floats, mpmae_stream_t stream);
/* The same backward with the fold of the gamma / beta gradient partials left to a later mpmae_fold_group call (see
 * MpmaeRsArgs.defer_fold): *defer_fold (host memory) receives its description, `ws` must stay untouched until then. */
int mpmae_ln_bwd_defer(int dt, const void* dy, int dy_div, float dy_scale, const void* xhat,
                       const float* rstd, const float* gamma, const float* beta, int act, void* dx,
                       int accumulate, float* dgamma, float* dbeta, int M, int C, const uint8_t* rowmask,
                       float* ws, size_t ws_floats, MpmaeFoldDesc* defer_fold, mpmae_stream_t stream);
int mpmae_ln_bwd_down_defer(int dt, const void* dy_grouped, const void* xhat, const float* rstd,
                            const float* gamma, void* dx, float* dgamma, float* dbeta, int M, int C, int S,
                            const uint8_t* rowmask, float* ws, size_t ws_floats, MpmaeFoldDesc* defer_fold,
                            mpmae_stream_t stream);
/* MinkowskiGRN (batch-global, eps 1e-6; sparse_norm_layers.py:24-33) and GRN (per-sample,
 * eps 1e-4; norm_layers.py:41-44): statistics finalisation for G groups of H channels. */
int mpmae_grn_fwd_finalize(const float* G2, const float* gamma, float eps, int G, int H, float* Gx,
                           float* Ainv, float* scale, mpmae_stream_t stream);
int mpmae_grn_bwd_finalize(const float* S0, const float* S1, const float* Gx, const float* Ainv,
                           const float* gamma, int G, int H, float* coef, float* dgamma,
                           float* dbeta, mpmae_stream_t stream);
/* mpmae_grn_fwd_finalize + mpmae_grn_apply, and mpmae_grn_bwd_finalize + mpmae_grn_bwd_apply, as ONE launch each (round 5; single GRN group):
 * every workgroup recomputes the H-vector from the column sums in its prologue, workgroup 0 publishes Gx / Ainv / scale (coef, and adds the
 * GRN gamma / beta gradients). Same outputs as the two-launch forms. H % 8 == 0 and H <= 8160 (two H-vectors in the default 64 KiB of LDS beside
 * the reduction scratch), else hipErrorInvalidValue. */
int mpmae_grn_apply_fin(int dt, const void* h, void* z, const float* G2, const float* gamma, const float* beta, float eps, int M, int H,
                        const uint8_t* act, float* Gx, float* Ainv, float* scale, mpmae_stream_t stream);
int mpmae_grn_bwd_apply_fin(int dt, void* dz, const void* h, const float* scale, const float* S0, const float* S1, const float* Gx,
                            const float* Ainv, const float* gamma, int M, int H, float* coef, float* dgamma, float* dbeta,
                            mpmae_stream_t stream);
/* GRN backward statistics FROM the pwconv2 weight gradient (round 5; sparse_norm_layers.py:24-33 differentiated, convnextv2_sparse.py:52-54):
 * with T = dout^T gelu(h) [C][H] and dbt = sum_rows dout [C] (both fp32, produced by mpmae_wgrad with a GELU-only operand prologue into
 * zero-initialised scratch) and W2s the staged pwconv2 weights [C][ldw] (storage type dt):
 *   S0[j] += sum_c W2s[c][j] dbt[c];   S1[j] += sum_c W2s[c][j] T[c][j];   dW2[c][j] += scale[j] T[c][j] + beta[j] dbt[c];   db2[c] += dbt[c]
 * - the statistics that mpmae_rs(which = 1, out = NULL) computes with a second pass over dout and h. */
int mpmae_grn_stats_from_wgrad(int dt, const float* T, const float* dbt, const void* W2s, int ldw, const float* scale, const float* beta,
                               float* dW2, float* db2, float* S0, float* S1, int C, int H, mpmae_stream_t stream);
/* element-wise GRN application z = gelu(h)*(1+gamma*Nx) + beta and its backward
 * dh = (dz*(1+gamma*Nx) + coef*gelu(h)) * gelu'(h) (in place over dz), and the column statistics
 * they need (mode 0: s0 += sum gelu(h)^2; mode 1: s0 += sum dz, s1 += sum dz*gelu(h)), per group
 * of rpg rows (sparse_norm_layers.py:24-33; norm_layers.py:41-44 and their autograd). */
int mpmae_grn_apply(int dt, const void* h, void* z, const float* scale, const float* beta, int M,
                    int H, int rpg, const uint8_t* act, mpmae_stream_t stream);
int mpmae_grn_bwd_apply(int dt, void* dz, const void* h, const float* scale, const float* coef,
                        int M, int H, int rpg, mpmae_stream_t stream);
/* the same GRN for the DENSE decoder block (one statistics group per sample, norm_layers.py:25-48) as one launch per
 * direction: statistics + finalisation + application with the group's rows held in registers in between. bf16, H == 2048,
 * rpg <= 52, M % rpg == 0 (mpmae_grn_group_ok says whether a shape qualifies; other shapes use the three calls above).
 * forward writes z, Gx[G][H], Ainv[G], scale[G][H]; backward writes dh over dz and the per-group gamma / beta gradient
 * rows to slab[G][2H] (summed into the gradients by mpmae_fold_group{slab, G, 2H, dgamma, H, dbeta - dgamma, 1}). */
int mpmae_grn_group_ok(int dt, int M, int H, int rpg);
int mpmae_grn_group_fwd(int dt, const void* h, void* z, const float* gamma, const float* beta, float eps,
                        int M, int H, int rpg, float* Gx, float* Ainv, float* scale, mpmae_stream_t stream);
int mpmae_grn_group_bwd(int dt, void* dz, const void* h, const float* scale, const float* Gx,
                        const float* Ainv, const float* gamma, int M, int H, int rpg, float* slab,
                        mpmae_stream_t stream);
int mpmae_colstats(int dt, const void* h, const void* dz, int mode, float* s0, float* s1, int M,
                   int H, int rpg, float* ws, size_t ws_floats, mpmae_stream_t stream);
/* MinkowskiDepthwiseConvolution 7x7 (convnextv2_sparse.py:37-39) / dense depthwise 7x7 pad 3
 * (convnextv2.py:27-29): forward, data gradient (flip = 1, add = upstream residual gradient),
 * weight + bias gradient. `args` are HOST pointers. */
int mpmae_dwconv7_fwd(int dt, const MpmaeDwArgs* args, mpmae_stream_t stream);
int mpmae_dwconv7_wgrad(int dt, const MpmaeDwWgArgs* args, int nblocks, mpmae_stream_t stream);
/* The same gradients for `count` depthwise layers of identical geometry in one launch + one fold (all blocks of an encoder stage: their
 * x / dd operands persist until the stage's data-gradient chain is done). probs[i].ws / ws_floats are ignored; ws: scratch for all slabs. */
int mpmae_dwconv7_wgrad_group(int dt, const MpmaeDwWgArgs* probs, int count, float* ws, size_t ws_floats,
                              mpmae_stream_t stream);
/* depthwise stem, kernel = stride = patch/8 (convnextv2_sparse.py:121-127). */
int mpmae_dwstride_fwd(int dt, const void* in, void* out, const float* w, const float* b, int Mout,
                       int C, int S, int k, const uint8_t* act_in, const uint8_t* act_out,
                       mpmae_stream_t stream);
int mpmae_dwstride_bwd(int dt, const void* dout, const void* in, void* din, const float* w,
                       float* dw, float* db, int Mout, int C, int S, int k, const uint8_t* act_in,
                       float* ws, size_t ws_floats, mpmae_stream_t stream);
/* mask-token blend of forward_decoder (fcmae.py:249-255) and its parameter gradient. With `vis_rows` (compact [N*keep, D] output of the
 * proj layer as a PLAIN GEMM) the forward writes the whole decoder input [rows = N*L, D] in one pass - the slot's row at visible
 * patches, the token elsewhere - instead of relying on a scatter epilogue; with `vis_rows_out` the backward's pass over dxdec also
 * gathers the rows of the visible patches into a compact [N*keep, D] matrix, the operand of proj's data and weight gradients. */
int mpmae_fill_mask_token(int dt, void* xdec, const float* token, const int* inv, int rows, int D, const void* vis_rows, int keep,
                          int L, mpmae_stream_t stream);
int mpmae_mask_token_bwd(int dt, const void* dxdec, const int* inv, float* dtoken, int rows, int D, void* vis_rows_out, int keep,
                         int L, mpmae_stream_t stream);
/* global average pool over the L positions of each sample (fcmae.py:262). */
int mpmae_pool_rows(int dt, const void* x, void* pooled, int N, int L, int C, mpmae_stream_t stream);

/* ---- losses (fcmae.py:267-412; custom_loss.py:19-30) --------------------------------------- */
int mpmae_loss_pix_cont(int dt, int bwd, const MpmaePixContArgs* args, int npatches,
                        mpmae_stream_t stream);
int mpmae_loss_pix_cat(int dt, int bwd, const MpmaePixCatArgs* args, int npatches,
                       mpmae_stream_t stream);
int mpmae_loss_img(int dt, int bwd, const MpmaeImgArgs* args, mpmae_stream_t stream);
/* The same three kernels over SEVERAL modalities in one launch: dev_args is a device-resident array of
 * `count` argument records of the kind (0: MpmaePixContArgs, 1: MpmaePixCatArgs, 2: MpmaeImgArgs), gridx
 * the x extent the single-modality entry point would use (forward: samples; backward: patches for the
 * pixel kinds, samples for the image kind). All records of a call must share p / L / K-limits. */
int mpmae_loss_multi(int dt, int bwd, int kind, const void* dev_args, int count, int gridx,
                     mpmae_stream_t stream);
/* Forward of the continuous pixel losses in row-band form: one workgroup per sample walks its patch rows with the target band in
 * LDS (loss.cuh). Same records, outputs and partial layout as mpmae_loss_multi(kind 0, forward). maxC = largest C of the records;
 * needs H % 4 == 0, (p*p) % 4 == 0 and maxC * (p*H + 4) * 4 <= 150 KB of LDS, else hipErrorInvalidValue (use mpmae_loss_multi). */
int mpmae_loss_pix_cont_rows(int dt, const void* dev_args, int count, int N, int maxC, int p, int H,
                             mpmae_stream_t stream);
/* Gradient twin (same band walk, no statistics): d pred of every record, zeros at patches that were not counted. */
int mpmae_loss_pix_cont_rows_bwd(int dt, const void* dev_args, int count, int N, int maxC, int p, int H,
                                 mpmae_stream_t stream);
/* One-pass form (round 5): the forward of mpmae_loss_pix_cont_rows that ALSO writes args->dpred WITHOUT the per-modality scalar coef
 * (mask * 2 / count * (pred - normalised target) at counted patches, zero elsewhere; fcmae.py:366-403 differentiated). The scalar -
 * known only after mpmae_loss_finalize - is applied by the consumers: mpmae_head_scale scales the heads' staged transposed weights per
 * modality segment (data gradient) and fills MpmaeWgradArgs.rowscale (weight / bias gradient). mpmae_loss_pix_cat_waves(bwd = 2) is the
 * categorical twin (softmax - onehot at masked, labelled pixels). */
int mpmae_loss_pix_cont_rows_fused(int dt, const void* dev_args, int count, int N, int maxC, int p, int H,
                                   mpmae_stream_t stream);
/* B [D][ldb] (staged transposed head weights, storage type dt): Bout[:, k] = B[:, k] * coef[col_mod[k]]; rowscale[k] = coef[col_mod[k]].
 * Bout == B is allowed (in place) - the engine passes a buffer of its own so that a backward can be repeated behind one forward. */
int mpmae_head_scale(int dt, const void* B, void* Bout, int ldb, int D, int W, const uint8_t* col_mod, const float* coef, float* rowscale,
                     mpmae_stream_t stream);
/* Categorical pixel losses, wave-per-patch form (forward bwd = 0 / gradient bwd = 1): same records, outputs and partial layout as
 * mpmae_loss_multi(kind 1); bwd = 2: forward + UNSCALED gradient in one pass (mpmae_loss_pix_cont_rows_fused). max_pk = largest p*p*K of the records (a multiple of 4, K <= 16); every record needs ld % 4 == 0 and
 * coff % 4 == 0 (vector accesses); 16 * max_pk elements of LDS. */
int mpmae_loss_pix_cat_waves(int dt, int bwd, const void* dev_args, int count, int N, int max_pk,
                             mpmae_stream_t stream);
/* acc: per-sample partial {sum, count} pairs laid out [T][N][2] (written by the loss kernels:
 * args->acc points at modality t's [N][2] block). */
int mpmae_loss_finalize(const float* acc, int N, const float* log_vars, int T, float loss_scale,
                        float* losses, float* weighted, float* total, float* coef,
                        float* dlog_vars, mpmae_stream_t stream);
/* The same, guarded: err_words / n_err / err_stride as in MpmaeMeters (the grid-barrier error words of the persistent stage kernels).
 * A non-zero word makes total[0] = +inf, so that the non-finite guard of mpmae_hp_fetch skips the update - in a data-parallel run on
 * EVERY rank, because the guard there reads the all-reduced loss (the reference's equivalent is the collective sys.exit of
 * engine_pretrain.py:83-85 on a non-finite all-reduced loss). n_err = 0: identical to mpmae_loss_finalize. */
int mpmae_loss_finalize_guarded(const float* acc, int N, const float* log_vars, int T, float loss_scale,
                                float* losses, float* weighted, float* total, float* coef, float* dlog_vars,
                                const unsigned* err_words, int n_err, int err_stride, mpmae_stream_t stream);

/* ---- optimizer (main_pretrain.py:312-320; helpers.py:509-526) ------------------------------ */
/* hp (device, 8 floats) = {lr, 1/(1-beta1^t), 1/sqrt(1-beta2^t), grad_scale, skip, skipped_steps, -, -}:
 * when skip != 0 mpmae_adamw leaves p / m / v untouched (non-finite loss, engine_pretrain.py:83-85). */
/* gnorm2 (may be NULL; 1 + 4096 floats): {number of partials, one partial sum of g^2 per workgroup} of the gradients this launch read
 * (helpers.get_grad_norm_, :509-526, without a pass of its own and without atomics); folded by the next mpmae_hp_fetch. */
int mpmae_adamw(float* p, const float* g, float* m, float* v, const float* hp, float beta1,
                float beta2, float eps, float wd, size_t n, const uint8_t* decay_mask, float* gnorm2,
                mpmae_stream_t stream);
int mpmae_sumsq(const float* x, size_t n, float* out, mpmae_stream_t stream);
/* hyper-parameter hand-over for replayed steps: copies record (*counter % slots) of a pinned,
 * device-visible ring of {lr, 1/(1-beta1^t), 1/sqrt(1-beta2^t), grad_scale} records into hp and
 * increments *counter, in stream order (the host fills slot t % slots before enqueueing step t and
 * must not run more than `slots` steps ahead). `total` (may be NULL) is the step's loss on the device:
 * a non-finite value sets hp[4] (skip this update) and increments hp[5]. */
/* meters (host pointer, may be NULL / ring == NULL): device-resident MetricLogger state updated by the same launch - record number
 * `count` of ring [window][2T + 2] = {T losses, T weighted losses, total, gradient norm}, running sums [2T + 2] and the count in
 * sums[2T + 2] (helpers.SmoothedValue, :49-109; engine_pretrain.py:71-113). The gradient norm of an update is sqrt(sum of the
 * partials mpmae_adamw left in gnorm2) x grad_scale; it is written into that update's record by the NEXT fetch, which clears gnorm2[0]. */
typedef struct MpmaeMeters {
  const float* losses; const float* weighted; int T;
  float* ring; int window; float* sums; float* gnorm2;
  /* error words of the persistent stage kernels' grid barriers (mpmae_ps_fwd / _bwd: args->sync[2] != 0 after a spin timeout, i.e. a
   * workgroup that never became resident): n_err rows of err_stride unsigneds, word 2 of each. Any non-zero word makes hp_fetch skip
   * this update like a non-finite loss (hp[4] = 1, hp[5] += 1), counts it in hp[6] and CLEARS the word (one timeout = one skipped
   * update); NULL / 0: not checked. */
  unsigned* err_words; int n_err; int err_stride;
} MpmaeMeters;
int mpmae_hp_fetch(const float* ring_pinned, int slots, int* counter, float* hp, const float* total,
                   const MpmaeMeters* meters, mpmae_stream_t stream);

/* ---- launch programs ----------------------------------------------------------------------
 * The reference drives its step from Python (engine_pretrain.py:46-118, one autograd graph per
 * iteration). Here a step is a fixed list of launches; a program records it once and replays it
 * from C. While an op is open (begin_op .. next begin_op / end) every entry point of this header
 * called from the same thread appends its fully-resolved kernel launches to the program instead of
 * issuing them (the stream argument is ignored). run() issues ops [first, first+count) in order:
 * lane 0 on `main`, lane k on the program's k-th side stream (forked from / joined to `main`
 * around the call); an op first waits for the events named in `waits` that were signalled earlier
 * in the same run() call and records its own `signal` event (ids > 0; 0 = none) when enqueued.
 * Argument structs (Mpmae*Args, problem arrays) are consumed AT THE CALL, recorded or not: the caller's copy may live on the stack and be
 * reused or freed before run() (enforced since round 6 - the depthwise / per-modality loss entry points used to read the caller's struct at
 * replay; tests/test_hip_kernels_bf16.py overwrites every host struct between recording and replay). Device buffers, and the pinned host
 * source of mpmae_memcpy_h2d_async, are read at replay and must outlive the program. */
typedef struct MpmaeProgram MpmaeProgram;
MpmaeProgram* mpmae_program_create(void);
void mpmae_program_destroy(MpmaeProgram* p);
int mpmae_program_begin_op(MpmaeProgram* p, int lane, const int* waits, int nwaits, int signal);
int mpmae_program_end(MpmaeProgram* p);
int mpmae_program_num_ops(const MpmaeProgram* p);
int mpmae_program_run(MpmaeProgram* p, int first, int count, mpmae_stream_t main_stream);
/* Hardware queues. ROCm maps HIP streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, balancing by stream count; streams that
 * share a queue are serialised. A side lane that lands on the main stream's queue costs the whole overlap of the backward (measured
 * 5.4 instead of 4.4 ms per step, and which streams collide depends on how many streams torch / RCCL created earlier). run() therefore
 * PROBES its side lanes at the first replay against a given main stream (two 30 us spin kernels started together: ~50 us when the
 * streams are concurrent, ~80 us on a shared queue) and replaces a colliding lane by a freshly created stream - one host
 * synchronisation, once per (program, main stream). mpmae_program_stream_overlaps() applies the same probe to a stream OUTSIDE the
 * program (gradient-exchange stream, input-stage stream) against the main stream and every side lane: 1 = concurrent with all of them,
 * 0 = shares a queue with one (the caller creates another stream and asks again), < 0 = -(hipError). p == NULL: against the main stream
 * only (the Python-loop driver's side streams). */
int mpmae_program_stream_overlaps(MpmaeProgram* p, mpmae_stream_t main_stream, mpmae_stream_t other);
/* Hand an op's `signal` event to a stream outside the program (the gradient exchange waits for "bucket ready" points of a backward that
 * is replayed as ONE run() call): export_signal keeps the event recorded although no op of the program waits for it (call after
 * mpmae_program_end); stream_wait makes `stream` wait for the event as recorded by the most recent run() (no-op if that run did not
 * reach the op). */
int mpmae_program_export_signal(MpmaeProgram* p, int signal);
int mpmae_program_stream_wait(MpmaeProgram* p, int signal, mpmae_stream_t stream);
/* recordable fills / host->device copies (gradient and statistics clears, the AdamW hyper-parameter record) */
int mpmae_memset_async(void* ptr, int value, size_t bytes, mpmae_stream_t stream);
int mpmae_memcpy_h2d_async(void* dst, const void* src_pinned, size_t bytes, mpmae_stream_t stream);

/* library identification: returns the gfx arch the kernels were built for (950). */
int mpmae_arch(void);
#ifdef __cplusplus
}
#endif
#endif /* MPMAE_HIP_H */
