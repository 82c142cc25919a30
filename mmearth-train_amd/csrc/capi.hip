// extern "C" surface of libmpmae_hip.so (see include/mpmae_hip.h), core unit: options, launch programs, masks / activity / staging,
// LayerNorm, depthwise 7x7, stem, losses, optimizer. The GEMM family lives in capi_gemm.hip, the fused pointwise / persistent stage
// kernels in capi_rs.hip (three units so that the build runs them in parallel). gfx950 only.
#include "capi_common.h"
#include "gemm.cuh"
#include "rows.cuh"
#include "dwconv.cuh"
#include "misc.cuh"
#include "loss.cuh"
#include "rows2.cuh"
#include "dwconv6.cuh"
#include "stemtail.cuh"
#include "dwmfma.cuh"
#include "dwmfma_wg.cuh"

// second stage of the two-stage reductions (see reduce_partials_kernel in misc.cuh)
void launch_reduce(int mode, const float* part, int P, int W, float* out, float* out2, int a, int b, int c, int d,
                          hipStream_t st) {
  int R = P / 16;                 // >= 4 rows per thread (4 row lanes per block)
  if (R < 1) R = 1;
  if (R > 32) R = 32;
  if (g_opt[MPMAE_OPT_DET] > 0) R = 1;      // one row group per column block: plain `+=` in a fixed order instead of atomics between row groups
  dim3 g(cdiv(W, 64), R);
  if (mode == 0) LAUNCH(reduce_partials_kernel<0>, g, dim3(256), 0, st, part, P, W, out, out2, a, b, c, d);
  else if (mode == 1) LAUNCH(reduce_partials_kernel<1>, g, dim3(256), 0, st, part, P, W, out, out2, a, b, c, d);
  else if (mode == 3) LAUNCH(reduce_partials_kernel<3>, g, dim3(256), 0, st, part, P, W, out, out2, a, b, c, d);
  else LAUNCH(reduce_partials_kernel<2>, g, dim3(256), 0, st, part, P, W, out, out2, a, b, c, d);
}

void launch_reduce_rowscale(const float* part, int P, int W, float* out, float* out2, int nk, int Kk, const float* rs, hipStream_t st) {
  int R = P / 16;
  if (R < 1) R = 1;
  if (R > 32) R = 32;
  if (g_opt[MPMAE_OPT_DET] > 0) R = 1;
  LAUNCH(reduce_partials_rowscale_kernel, dim3(cdiv(W, 64), R), dim3(256), 0, st, part, P, W, out, out2, nk, Kk, rs);
}

// ------------------------------------------------------------------------------------------
// Library options (mpmae_set_option): explicit, process-wide A/B switches of kernel selection. They replace environment
// variables read inside the library; defaults are the measured-best choices.
// ------------------------------------------------------------------------------------------
thread_local MpmaeProgram* g_rec = nullptr;
thread_local hipEvent_t g_stop_ev = nullptr;
thread_local int g_launch_err = 0;

int g_opt[MPMAE_OPT_COUNT_] = {
    /* MPMAE_OPT_DW */ 8,
    /* MPMAE_OPT_DWW */ 7,
    /* MPMAE_OPT_NT_GLDS64 */ 1,
    /* MPMAE_OPT_NT_BK32 */ 1,
    /* MPMAE_OPT_NT_GLDS */ 1,
    /* MPMAE_OPT_TN */ 2,
    /* MPMAE_OPT_CS_SPLIT */ 1,
    /* MPMAE_OPT_RSC_PF */ 1,
    /* MPMAE_OPT_RSC_N40 */ 2,
    /* MPMAE_OPT_RSC_N80 */ 1,
    /* MPMAE_OPT_TN3_BLOCKS */ 128,
    /* MPMAE_OPT_TNG_BLOCKS */ 256,
    /* MPMAE_OPT_FOLD_GROUP */ 0,
    /* MPMAE_OPT_RSC_W5 */ 1,
    /* MPMAE_OPT_RSC_ATOMIC */ 0,
    /* MPMAE_OPT_DET */ 0,
    /* MPMAE_OPT_RSC1 */ 1,
    /* MPMAE_OPT_RSC1_ATOMIC */ 100,
    /* MPMAE_OPT_RSP */ 1,
    /* MPMAE_OPT_RSP_NWV */ 0,
    /* MPMAE_OPT_RSP_NARROW */ 2,
    /* MPMAE_OPT_RSN3 */ 5,
    /* MPMAE_OPT_EVX */ 1,
    /* MPMAE_OPT_RST_NW */ 16,
    /* MPMAE_OPT_NT_RING */ 1,
};

int mpmae_set_option(int option, int value) {
  if (option < 0 || option >= MPMAE_OPT_COUNT_) return (int)hipErrorInvalidValue;
  g_opt[option] = value;
  return 0;
}
int mpmae_get_option(int option) { return (option < 0 || option >= MPMAE_OPT_COUNT_) ? -1 : g_opt[option]; }

// C linkage comes from the declarations in include/mpmae_hip.h

int mpmae_arch(void) { return 950; }

int mpmae_mask_gen(const float* noise, int N, int L, int keep, float* mask, int* vis, int* inv, mpmae_stream_t s) {
  LAUNCH(mask_gen_kernel, dim3(N), dim3(256), 2 * L * sizeof(float), S_(s), noise, L, keep, mask, vis, inv);
  RET();
}

int mpmae_mask_gen_dense(const float* noise, int N, int L, int keep, float* mask, int* inv, mpmae_stream_t s) {
  if (!noise || !mask || !inv || N < 1 || L < 1 || keep < 0 || keep > L) return (int)hipErrorInvalidValue;
  LAUNCH(mask_gen_dense_kernel, dim3(N), dim3(256), (size_t)L * sizeof(float), S_(s), noise, L, keep, mask, inv);
  RET();
}

int mpmae_activity(const float* img, const int* vis, uint8_t* act, int N, int Cin, int H, int keep, int grid, int S,
                   mpmae_stream_t s) {
  const int total = N * keep * S * S;
  LAUNCH(activity_kernel, dim3(grid1d(total)), dim3(256), 0, S_(s), img, vis, act, N, Cin, H, keep, grid, S);
  RET();
}

int mpmae_activity_pool(const uint8_t* in, uint8_t* out, int Mout, int S, int k, mpmae_stream_t s) {
  LAUNCH(activity_pool_kernel, dim3(grid1d(Mout)), dim3(256), 0, S_(s), in, out, Mout, S, k);
  RET();
}

int mpmae_prep_weights(int dt, const MpmaePrepDesc* table, int ndesc, int max_tiles, mpmae_stream_t s) {
  if (!table || ndesc < 1 || max_tiles < 1 || max_tiles > 65535) return (int)hipErrorInvalidValue;
  const int tiles = max_tiles;
  dim3 g((unsigned)tiles, ndesc);
  if (dt == 0) LAUNCH(prep_tiled_kernel<float>, g, dim3(256), 0, S_(s), table);
  else LAUNCH(prep_tiled_kernel<bf16_t>, g, dim3(256), 0, S_(s), table);
  RET();
}

// ------------------------------------------------------------------------------------------
template <typename T>
static void launch_ln_fwd(const void* x, void* xhat, float* rstd, void* y, const float* gamma, const float* beta,
                          int act, float eps, int M, int C, const uint8_t* rowmask, hipStream_t st) {
  const int blocks = grid1d((long long)M * 64, 256, 8192);
  if (C <= 64 * LN_MAXPER)
    LAUNCH(ln_fwd_kernel<T>, dim3(blocks), dim3(256), 0, st, (const T*)x, (T*)xhat, rstd, (T*)y, gamma, beta,
                     act, eps, M, C, rowmask);
  else      // large / huge stage 3 (C = 1536 / 2816)
    LAUNCH((ln_fwd_kernel<T, LN_MAXPER_WIDE>), dim3(blocks), dim3(256), 0, st, (const T*)x, (T*)xhat, rstd, (T*)y, gamma, beta,
                     act, eps, M, C, rowmask);
}

static int ln_fwd_impl(int dt, const void* x, void* xhat, float* rstd, void* y, const float* gamma, const float* beta, int act,
                       float eps, int M, int C, const uint8_t* rowmask, int down_S, mpmae_stream_t s) {
  if (C > 64 * LN_MAXPER_WIDE) return (int)hipErrorInvalidValue;
  if (down_S && ((C & 7) || C > 1024 || (down_S & 1))) return (int)hipErrorInvalidValue;
  if ((C & 7) == 0 && C <= 1024) {
    const int nvec = C / 8;
    const int G = nvec <= 8 ? 8 : nvec <= 16 ? 16 : nvec <= 32 ? 32 : 64;
    const int per = cdiv(nvec, G);
    const int rpw = 64 / G;
    const int blocks = grid1d((long long)cdiv(M, rpw) * 64, 256, 4096);
#define LNF(TT, GG, PP) LAUNCH((ln_fwd_v2_kernel<TT, GG, PP>), dim3(blocks), dim3(256), 0, S_(s), (const TT*)x, (TT*)xhat, rstd, (TT*)y, gamma, beta, act, eps, M, C, rowmask, down_S)
#define LNF_T(TT) do { if (G == 8) LNF(TT, 8, 1); else if (G == 16) LNF(TT, 16, 1); else if (G == 32) LNF(TT, 32, 1); else if (per == 1) LNF(TT, 64, 1); else LNF(TT, 64, 2); } while (0)
    if (dt == 0) LNF_T(float); else LNF_T(bf16_t);
#undef LNF_T
#undef LNF
    RET();
  }
  if (dt == 0) launch_ln_fwd<float>(x, xhat, rstd, y, gamma, beta, act, eps, M, C, rowmask, S_(s));
  else launch_ln_fwd<bf16_t>(x, xhat, rstd, y, gamma, beta, act, eps, M, C, rowmask, S_(s));
  RET();
}

int mpmae_ln_fwd(int dt, const void* x, void* xhat, float* rstd, void* y, const float* gamma, const float* beta, int act,
                 float eps, int M, int C, const uint8_t* rowmask, mpmae_stream_t s) {
  return ln_fwd_impl(dt, x, xhat, rstd, y, gamma, beta, act, eps, M, C, rowmask, 0, s);
}

int mpmae_ln_fwd_down(int dt, const void* x, void* xhat, float* rstd, void* y_grouped, const float* gamma, const float* beta,
                      float eps, int M, int C, int S, const uint8_t* rowmask, mpmae_stream_t s) {
  if (S < 2 || !y_grouped) return (int)hipErrorInvalidValue;
  return ln_fwd_impl(dt, x, xhat, rstd, y_grouped, gamma, beta, 0, eps, M, C, rowmask, S, s);
}

static int ln_bwd_impl(int dt, const void* dy, int dy_div, float dy_scale, const void* xhat, const float* rstd,
                       const float* gamma, const float* beta, int act, void* dx, int accumulate, float* dgamma, float* dbeta,
                       int M, int C, const uint8_t* rowmask, float* ws, size_t ws_floats, int down_S, mpmae_stream_t s,
                       MpmaeFoldDesc* defer = nullptr) {
  if (C > 64 * LN_MAXPER_WIDE) return (int)hipErrorInvalidValue;
  if (down_S && ((C & 7) || C > 1024 || (down_S & 1))) return (int)hipErrorInvalidValue;
  int blocks = grid1d((long long)M * 64, 256, 1024);
  if (!ws || ws_floats < (size_t)2 * C) return (int)hipErrorInvalidValue;
  if ((size_t)blocks * 2 * C > ws_floats) blocks = (int)(ws_floats / ((size_t)2 * C));
  if ((C & 7) == 0 && C <= 1024) {
    const int nvec = C / 8;
    const int G = nvec <= 8 ? 8 : nvec <= 16 ? 16 : nvec <= 32 ? 32 : 64;
    const int per = cdiv(nvec, G);
    const int rpw = 64 / G;
    int lncap;
    lncap = 512 /* LNB_BLOCKS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;   // measured: 512 -> 50 us, 1024 -> 36 us, 2048 -> 40 us (slab reduce grows)
    // one fp32 slab row (2C floats) per workgroup, folded from its 4 waves in LDS. Measured (M = 12 544, C = 512): 18 / 21 / 28 / 37 us
    // at 512 / 1024 / 2048 / 4096 workgroups - the per-wave prologue (gamma / beta vectors) and the fold are the fixed costs, so few
    // long-running waves win; narrow rows (C < 256: many rows per wave iteration) keep twice the cap
    const int capc = C >= 256 ? lncap : 2 * lncap;
    int b2 = grid1d((long long)cdiv(M, rpw) * 64, 256, capc);
    while ((size_t)b2 * 2 * C > ws_floats && b2 > 1) b2 /= 2;
#define LNB(TT, GG, PP) LAUNCH((ln_bwd_v2_kernel<TT, GG, PP>), dim3(b2), dim3(256), 0, S_(s), (const TT*)dy, dy_div, dy_scale, (const TT*)xhat, rstd, gamma, beta, act, (TT*)dx, accumulate, ws, M, C, rowmask, down_S)
#define LNB_T(TT) do { if (G == 8) LNB(TT, 8, 1); else if (G == 16) LNB(TT, 16, 1); else if (G == 32) LNB(TT, 32, 1); else if (per == 1) LNB(TT, 64, 1); else LNB(TT, 64, 2); } while (0)
    if (dt == 0) LNB_T(float); else LNB_T(bf16_t);
#undef LNB_T
#undef LNB
    blocks = b2;                                                      // slab rows = workgroups (the 4 waves fold in LDS)
  } else if (C > 64 * LN_MAXPER) {      // large / huge stage 3 (C = 1536 / 2816): the generic kernel's wide instantiation
    if (dt == 0)
      LAUNCH((ln_bwd_kernel<float, LN_MAXPER_WIDE>), dim3(blocks), dim3(256), 0, S_(s), (const float*)dy, dy_div, dy_scale,
                       (const float*)xhat, rstd, gamma, beta, act, (float*)dx, accumulate, ws, M, C, rowmask);
    else
      LAUNCH((ln_bwd_kernel<bf16_t, LN_MAXPER_WIDE>), dim3(blocks), dim3(256), 0, S_(s), (const bf16_t*)dy, dy_div, dy_scale,
                       (const bf16_t*)xhat, rstd, gamma, beta, act, (bf16_t*)dx, accumulate, ws, M, C, rowmask);
  } else if (dt == 0)
    LAUNCH(ln_bwd_kernel<float>, dim3(blocks), dim3(256), 0, S_(s), (const float*)dy, dy_div, dy_scale,
                       (const float*)xhat, rstd, gamma, beta, act, (float*)dx, accumulate, ws, M, C, rowmask);
  else
    LAUNCH(ln_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, S_(s), (const bf16_t*)dy, dy_div, dy_scale,
                       (const bf16_t*)xhat, rstd, gamma, beta, act, (bf16_t*)dx, accumulate, ws, M, C, rowmask);
  // slabs are [block][2][C]: e = n*C + k with n = 0 -> dgamma[k], n = 1 -> dbeta[k] (MODE 1, b = dbeta - dgamma)
  if (dgamma && dbeta) {
    const long long delta = dbeta - dgamma;
    if (delta > 2147483647LL || delta < -2147483647LL) return (int)hipErrorInvalidValue;
    if (defer) *defer = MpmaeFoldDesc{ws, blocks, 2 * C, dgamma, C, (int)delta, 1};
    else launch_reduce(1, ws, blocks, 2 * C, dgamma, nullptr, C, (int)delta, 1, 0, S_(s));
  } else if (dgamma || dbeta) {
    return (int)hipErrorInvalidValue;     // both or neither
  }
  RET();
}

int mpmae_ln_bwd(int dt, const void* dy, int dy_div, float dy_scale, const void* xhat, const float* rstd,
                 const float* gamma, const float* beta, int act, void* dx, int accumulate, float* dgamma, float* dbeta,
                 int M, int C, const uint8_t* rowmask, float* ws, size_t ws_floats, mpmae_stream_t s) {
  return ln_bwd_impl(dt, dy, dy_div, dy_scale, xhat, rstd, gamma, beta, act, dx, accumulate, dgamma, dbeta, M, C, rowmask,
                     ws, ws_floats, 0, s);
}

int mpmae_ln_bwd_down(int dt, const void* dy_grouped, const void* xhat, const float* rstd, const float* gamma, void* dx,
                      float* dgamma, float* dbeta, int M, int C, int S, const uint8_t* rowmask, float* ws, size_t ws_floats,
                      mpmae_stream_t s) {
  if (S < 2 || !dy_grouped) return (int)hipErrorInvalidValue;
  return ln_bwd_impl(dt, dy_grouped, 1, 1.0f, xhat, rstd, gamma, nullptr, 0, dx, 0, dgamma, dbeta, M, C, rowmask, ws,
                     ws_floats, S, s);
}

int mpmae_ln_bwd_defer(int dt, const void* dy, int dy_div, float dy_scale, const void* xhat, const float* rstd,
                       const float* gamma, const float* beta, int act, void* dx, int accumulate, float* dgamma, float* dbeta,
                       int M, int C, const uint8_t* rowmask, float* ws, size_t ws_floats, MpmaeFoldDesc* defer_fold, mpmae_stream_t s) {
  if (!defer_fold || !dgamma || !dbeta) return (int)hipErrorInvalidValue;
  return ln_bwd_impl(dt, dy, dy_div, dy_scale, xhat, rstd, gamma, beta, act, dx, accumulate, dgamma, dbeta, M, C, rowmask,
                     ws, ws_floats, 0, s, defer_fold);
}

int mpmae_ln_bwd_down_defer(int dt, const void* dy_grouped, const void* xhat, const float* rstd, const float* gamma, void* dx,
                            float* dgamma, float* dbeta, int M, int C, int S, const uint8_t* rowmask, float* ws, size_t ws_floats,
                            MpmaeFoldDesc* defer_fold, mpmae_stream_t s) {
  if (S < 2 || !dy_grouped || !defer_fold || !dgamma || !dbeta) return (int)hipErrorInvalidValue;
  return ln_bwd_impl(dt, dy_grouped, 1, 1.0f, xhat, rstd, gamma, nullptr, 0, dx, 0, dgamma, dbeta, M, C, rowmask, ws,
                     ws_floats, S, s, defer_fold);
}

int mpmae_grn_fwd_finalize(const float* G2, const float* gamma, float eps, int G, int H, float* Gx, float* Ainv,
                           float* scale, mpmae_stream_t s) {
  LAUNCH(grn_fwd_finalize_kernel, dim3(G), dim3(256), 0, S_(s), G2, gamma, eps, H, Gx, Ainv, scale);
  RET();
}

int mpmae_grn_bwd_finalize(const float* S0, const float* S1, const float* Gx, const float* Ainv, const float* gamma,
                           int G, int H, float* coef, float* dgamma, float* dbeta, mpmae_stream_t s) {
  const int gpb = (H <= 4096 && G >= 64) ? 8 : 1;       // groups per workgroup (dense decoder: one group per sample)
  LAUNCH(grn_bwd_finalize_kernel, dim3(cdiv(G, gpb)), dim3(256), 0, S_(s), S0, S1, Gx, Ainv, gamma, H, coef, dgamma, dbeta, G, gpb);
  RET();
}

int mpmae_grn_stats_from_wgrad(int dt, const float* T, const float* dbt, const void* W2s, int ldw, const float* scale, const float* beta,
                               float* dW2, float* db2, float* S0, float* S1, int C, int H, mpmae_stream_t s) {
  if (!T || !dbt || !W2s || !scale || !beta || !dW2 || !db2 || !S0 || !S1 || C < 1 || H < 1 || ldw < H) return (int)hipErrorInvalidValue;
  if (dt == 0) LAUNCH(grn_stats_from_wgrad_kernel<float>, dim3(cdiv(H, 16)), dim3(256), 0, S_(s), T, dbt, (const float*)W2s, ldw, scale, beta, dW2, db2, S0, S1, C, H);
  else LAUNCH(grn_stats_from_wgrad_kernel<bf16_t>, dim3(cdiv(H, 16)), dim3(256), 0, S_(s), T, dbt, (const bf16_t*)W2s, ldw, scale, beta, dW2, db2, S0, S1, C, H);
  RET();
}

// ------------------------------------------------------------------------------------------
static size_t dw_lds_bytes(int CC, bool wgrad) {
  size_t b = (size_t)DW_HP * CC * sizeof(float) + (DW_HP + 4) * sizeof(int);
  if (wgrad) b += (size_t)50 * CC * sizeof(float);
  return b;
}

// v5 (one sample's whole map in LDS): returns false when the map does not fit / the attribute cannot be raised
template <typename T, int S>
static bool launch_dw_v5(const MpmaeDwArgs& a, hipStream_t st) {
  constexpr int CW = 64 / S;
  const size_t lds = dw5_map_bytes<T, S>(a.g.grid) + 49 * CW * sizeof(float);
  if (lds > 160 * 1024 - 512) return false;
  static size_t cur = 64 * 1024;
  if (lds > cur) {
    if (hipFuncSetAttribute((const void*)dwconv7_v5_kernel<T, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    cur = lds;
  }
  dim3 g(a.g.N, a.C / CW);
  int nt8;
  nt8 = 512 /* DW_NT8: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
  // S = 8: the 62x62 map takes 61 KB, so two workgroups per CU; 8 waves each keep 4 waves per SIMD busy
  LAUNCH((dwconv7_v5_kernel<T, S>), g, dim3(S == 8 ? nt8 : 256), lds, st, a);
  return true;
}

template <typename T, int S>
static bool launch_dwwg_v5(const MpmaeDwWgArgs& a, int nblocks, hipStream_t st, const DwWgGroupP* grp = nullptr) {
  constexpr int CW = 64 / S;
  size_t lds = dw5_map_bytes<T, S>(a.g.grid);
  int nt8;
  nt8 = 512 /* DW_NT8: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
  const int nthreads = S == 8 ? nt8 : 256;
  const size_t red = (size_t)(nthreads / 64) * 50 * CW * sizeof(float);
  if (red > lds) lds = red;
  if (lds > 160 * 1024 - 512) return false;
  static size_t cur = 64 * 1024;
  if (lds > cur) {
    if (hipFuncSetAttribute((const void*)dwconv7_wgrad_v5_kernel<T, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)dwconv7_wgrad_v5_kernel<T, S, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    cur = lds;
  }
  DwWgGroupP gr;
  if (grp) gr = *grp; else gr.count = 0;
  dim3 g(nblocks, a.C / CW, gr.count > 0 ? gr.count : 1);
  if (a.g.grid == 7) LAUNCH((dwconv7_wgrad_v5_kernel<T, S, 7>), g, dim3(nthreads), lds, st, a, gr);
  else LAUNCH((dwconv7_wgrad_v5_kernel<T, S>), g, dim3(nthreads), lds, st, a, gr);
  return true;
}

static int dw6_threads(int S) {     // waves per workgroup (each wave walks two patches at a time)
  int t8 = -1, t4 = -1, t2 = -1;
  t8 = 320 /* DW6_T8: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
  t4 = 320 /* DW6_T4: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
  t2 = 320 /* DW6_T2: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
  return S == 8 ? t8 : S == 4 ? t4 : t2;
}

template <int S>
static bool launch_dw_v6(const MpmaeDwArgs& a, hipStream_t st) {
  constexpr int CW = 64 / S;
  const size_t lds = dw5_map_bytes<bf16_t, S>(a.g.grid) + 49 * CW * sizeof(float);
  if (lds > 64 * 1024) return false;
  dim3 g(a.g.N, a.C / CW);
  int gc;
  gc = 1 /* DW6_GC: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
  // compile-time map pitch: measured faster only at S = 2 for the forward / data-gradient kernel (17.7 vs 18.8 us; slower at
  // S = 4, 8), decisive for the weight-gradient kernel (220 -> 128 VGPRs)
  if (gc && S == 2 && a.g.grid == 7) LAUNCH((dwconv7_v6_kernel<S, 7>), g, dim3(dw6_threads(S)), lds, st, a);
  else LAUNCH((dwconv7_v6_kernel<S>), g, dim3(dw6_threads(S)), lds, st, a);
  return true;
}

static int dw_variant() {      // MPMAE_DW=4 forces the per-patch kernels (A/B measurements)
  int v;
  v = g_opt[MPMAE_OPT_DW];
  return v;
}

static bool dw_v4_ok(int C, int S) {
  // per-visible-patch tiles pay for S >= 2; at S == 1 (stage 3, dense decoder) every output would drag
  // its own 49-point halo, so the positional tiles of dwconv.cuh are used there
  if (S != 8 && S != 4 && S != 2) return false;
  return C % (64 / S) == 0;
}

template <int S, int CCH>
static int launch_dw_mfma(const DwP& a, hipStream_t st) {
  using D = DwMfma<S, CCH>;
  const size_t lds = D::lds(a.g.keep);
  if (lds > 160 * 1024) return -1;
  static size_t cur = 0;
  if (lds > cur) {
    if (hipFuncSetAttribute((const void*)dwconv7_mfma_kernel<S, CCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return (int)hipGetLastError();      // (launch_status() only reports LAUNCH errors)
    cur = lds;
  }
  LAUNCH((dwconv7_mfma_kernel<S, CCH>), dim3(a.g.N, a.C / CCH), dim3(D::NT), lds, st, a);
  return launch_status();
}

// matrix-core depthwise (dwmfma.cuh): sparse stages with S = 8 / 4 whose sample fits the LDS planes; -1 = not taken
// (S = 2 was tried as a dense-map product - zero-padded 14 x 14 planar map per sample and channel, 7 MFMAs per channel with the 1-D
// Toeplitz fragment of a tap row: correct, and 28 vs 14.7 us at atto stage 2, 51 vs 31 us at tiny: a sample is 28 MFMAs per wave
// behind the same fixed costs - loads, two barriers, 224 two-byte tap reads per lane - as a stage-0 sample; profiles/r04/dw_mfma_probes.txt)
static int try_dw_mfma(const DwP& a, hipStream_t st) {
  if (!a.g.inv || !a.g.vis || (a.g.S != 8 && a.g.S != 4) || a.g.keep < 1 || a.g.keep > 62 || a.g.grid > 8) return -1;
  if ((((uintptr_t)a.x | (uintptr_t)a.out | (uintptr_t)a.add) & 15) || (a.C & 7)) return -1;
  if (a.C % 40 == 0) return a.g.S == 8 ? launch_dw_mfma<8, 40>(a, st) : launch_dw_mfma<4, 40>(a, st);
  if (a.C % 32 == 0) return a.g.S == 8 ? launch_dw_mfma<8, 32>(a, st) : launch_dw_mfma<4, 32>(a, st);
  return -1;
}

int mpmae_dwconv7_fwd(int dt, const MpmaeDwArgs* a, mpmae_stream_t s) {
  if (!a || a->CC < 1 || a->CC > 256 || a->TP * a->g.S > 8) return (int)hipErrorInvalidValue;
  const MpmaeDwArgs& A = *a;      // LAUNCH captures the referenced struct BY VALUE: a launch recorded into a program must not read the caller's struct at replay time
  if (dt == 1 && dw_variant() >= 8) {
    const int r = try_dw_mfma(*a, S_(s));
    if (r >= 0) return r;
  }
  if (dt == 1 && a->g.S == 1 && a->g.grid == 7 && (a->C & 15) == 0 && dw_variant() >= 6 &&
      (((uintptr_t)a->x) & 15) == 0 && (((uintptr_t)a->out | (uintptr_t)a->add) & 3) == 0) {
    dim3 g(a->g.N, cdiv(a->C, 64));
    LAUNCH((dwconv7_v6s1_kernel<7>), g, dim3(256), 0, S_(s), A);
    RET();
  }
  if (dw_v4_ok(a->C, a->g.S) && dw_variant() >= 5) {
    bool ok = false;
    if (dt == 1 && dw_variant() >= 6 && (a->C & 1) == 0 && (((uintptr_t)a->x | (uintptr_t)a->out | (uintptr_t)a->add) & 3) == 0) {
      switch (a->g.S) { case 8: ok = launch_dw_v6<8>(*a, S_(s)); break; case 4: ok = launch_dw_v6<4>(*a, S_(s)); break;
                        default: ok = launch_dw_v6<2>(*a, S_(s)); }
      if (ok) RET();
    }
#define DW5(TT) do { switch (a->g.S) { case 8: ok = launch_dw_v5<TT, 8>(*a, S_(s)); break; case 4: ok = launch_dw_v5<TT, 4>(*a, S_(s)); break; \
                                      default: ok = launch_dw_v5<TT, 2>(*a, S_(s)); } } while (0)
    if (dt == 0) DW5(float); else DW5(bf16_t);
#undef DW5
    if (ok) RET();
  }
  // (the wave-granular 8 x 8-tile kernels of dwconv3.cuh - fp32 at S = 1, bf16 on grids other than 7 x 7 - were removed in round 6: those cases
  //  run the generic tile kernels below)
  const size_t lds = dw_lds_bytes(a->CC, false);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  dim3 g(a->g.N * a->tiles_side * a->tiles_side, cdiv(a->C, a->CC));
  if (dt == 0) {
    { static size_t cur = 0; if (lds > cur) { if (hipFuncSetAttribute((const void*)dwconv7_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } }
    LAUNCH(dwconv7_fwd_kernel<float>, g, dim3(256), lds, S_(s), A);
  } else {
    { static size_t cur = 0; if (lds > cur) { if (hipFuncSetAttribute((const void*)dwconv7_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } }
    LAUNCH(dwconv7_fwd_kernel<bf16_t>, g, dim3(256), lds, S_(s), A);
  }
  RET();
}

template <int CCH>
static int launch_dwwg_mfma(const DwWgP& a, hipStream_t st, const DwWgGroupP& gr, int count) {
  using D = DwMfmaWg<CCH>;
  const size_t lds = D::lds(a.g.keep);
  if (lds > 160 * 1024) return -1;
  static size_t cur = 0;
  if (lds > cur) {
    if (hipFuncSetAttribute((const void*)dwconv7_wgrad_mfma_kernel<CCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return (int)hipGetLastError();
    cur = lds;
  }
  LAUNCH((dwconv7_wgrad_mfma_kernel<CCH>), dim3(a.g.N, a.C / CCH, count), dim3(D::NT), lds, st, a, gr);
  return launch_status();
}

template <int CCH>
static int launch_dwwg_mfma4(const DwWgP& a, hipStream_t st, const DwWgGroupP& gr, int count) {
  using D = DwMfmaWg4<CCH>;
  const size_t lds = D::lds(a.g.keep);
  if (lds > 160 * 1024) return -1;
  static size_t cur = 0;
  if (lds > cur) {
    if (hipFuncSetAttribute((const void*)dwconv7_wgrad_mfma4_kernel<CCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return (int)hipGetLastError();
    cur = lds;
  }
  LAUNCH((dwconv7_wgrad_mfma4_kernel<CCH>), dim3(a.g.N, a.C / CCH, count), dim3(D::NT), lds, st, a, gr);
  return launch_status();
}

// matrix-core depthwise weight gradient (dwmfma_wg.cuh): S = 8 / 4, one workgroup (= one slab) per sample; -1 = not taken
static int try_dwwg_mfma(const DwWgP& a, hipStream_t st, const DwWgGroupP& gr, int count, size_t ws_floats) {
  if (g_opt[MPMAE_OPT_DWW] < 7 || !a.g.inv || !a.g.vis || (a.g.S != 8 && a.g.S != 4) || a.g.keep < 1 || a.g.keep > 62 || a.g.grid > 8) return -1;
  if ((((uintptr_t)a.x | (uintptr_t)a.dd) & 15) || (a.C & 7)) return -1;
  if ((size_t)a.g.N * 50 * a.C * count > ws_floats) return -1;
  if (a.g.S == 4) {
    if (a.C % 40 == 0) return launch_dwwg_mfma4<40>(a, st, gr, count);
    if (a.C % 32 == 0) return launch_dwwg_mfma4<32>(a, st, gr, count);
    return -1;
  }
  if (a.C % 40 == 0) return launch_dwwg_mfma<40>(a, st, gr, count);
  if (a.C % 32 == 0) return launch_dwwg_mfma<32>(a, st, gr, count);
  return -1;
}

int mpmae_dwconv7_wgrad(int dt, const MpmaeDwWgArgs* a, int nblocks, mpmae_stream_t s) {
  if (!a || a->CC < 1 || a->CC > 256 || a->TP * a->g.S > 8) return (int)hipErrorInvalidValue;
  const MpmaeDwWgArgs& A = *a;      // (captured by value, as above: the group entry below passes a stack copy)
  if (dt == 1 && a->ws) {
    DwWgGroupP gr;
    gr.count = 0;
    const int r = try_dwwg_mfma(*a, S_(s), gr, 1, a->ws_floats);
    if (r > 0) return r;
    if (r == 0) {
      launch_reduce(2, a->ws, a->g.N, 50 * a->C, a->dw, a->db, a->C, a->s_kh, a->s_kw, a->s_c, S_(s));
      RET();
    }
  }
  if (dt == 1 && a->g.S == 1 && a->g.grid == 7 && (a->C & 15) == 0 && dw_variant() >= 6 &&
      (((uintptr_t)a->x) & 15) == 0 && (((uintptr_t)a->dd) & 3) == 0) {
    const size_t per = (size_t)50 * a->C;
    if (!a->ws || a->ws_floats < per) return (int)hipErrorInvalidValue;
    int nbs1;
    nbs1 = 0 /* DWW_S1_NB: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
    const int want = nbs1 > 0 ? nbs1 : (cdiv(a->C, 64) <= 5 ? 128 : 64);      // ~512-640 workgroups in total (measured)
    int nb = a->g.N < want ? a->g.N : want;
    if ((size_t)nb * per > a->ws_floats) nb = (int)(a->ws_floats / per);
    dim3 g(nb, cdiv(a->C, 64));
    DwWgGroupP gr;
    gr.count = 0;
    LAUNCH((dwconv7_wgrad_v6s1_kernel<7>), g, dim3(256), 0, S_(s), A, gr);
    launch_reduce(2, a->ws, nb, 50 * a->C, a->dw, a->db, a->C, a->s_kh, a->s_kw, a->s_c, S_(s));
    RET();
  }
  if (dw_v4_ok(a->C, a->g.S) && dw_variant() >= 5) {
    const size_t per = (size_t)50 * a->C;
    if (!a->ws || a->ws_floats < per) return (int)hipErrorInvalidValue;
    int nbmax;
    nbmax = 128 /* DWW_NB: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
    int nb = a->g.N < nbmax ? a->g.N : nbmax;
    if ((size_t)nb * per > a->ws_floats) nb = (int)(a->ws_floats / per);
    bool ok = false;
#define DWW5(TT) do { switch (a->g.S) { case 8: ok = launch_dwwg_v5<TT, 8>(*a, nb, S_(s)); break; case 4: ok = launch_dwwg_v5<TT, 4>(*a, nb, S_(s)); break; \
                                       default: ok = launch_dwwg_v5<TT, 2>(*a, nb, S_(s)); } } while (0)
    if (dt == 0) DWW5(float); else DWW5(bf16_t);
#undef DWW5
    if (ok) {
      launch_reduce(2, a->ws, nb, 50 * a->C, a->dw, a->db, a->C, a->s_kh, a->s_kw, a->s_c, S_(s));
      RET();
    }
  }
  const size_t lds = dw_lds_bytes(a->CC, true);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (nblocks > a->ntiles_total) nblocks = a->ntiles_total;
  dim3 g(nblocks, cdiv(a->C, a->CC));
  if (dt == 0) {
    { static size_t cur = 0; if (lds > cur) { if (hipFuncSetAttribute((const void*)dwconv7_wgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } }
    LAUNCH(dwconv7_wgrad_kernel<float>, g, dim3(256), lds, S_(s), A);
  } else {
    { static size_t cur = 0; if (lds > cur) { if (hipFuncSetAttribute((const void*)dwconv7_wgrad_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } }
    LAUNCH(dwconv7_wgrad_kernel<bf16_t>, g, dim3(256), lds, S_(s), A);
  }
  RET();
}

// All depthwise weight gradients of a stage in one launch (grid.z = problem) + one fold: identical geometry / width / tap strides,
// bf16, the per-sample LDS-map kernels (v5: S >= 2, v6s1: S = 1 on the 7 x 7 grid). Anything else: one mpmae_dwconv7_wgrad each.
int mpmae_dwconv7_wgrad_group(int dt, const MpmaeDwWgArgs* probs, int count, float* ws, size_t ws_floats, mpmae_stream_t s) {
  if (!probs || count < 1 || !ws) return (int)hipErrorInvalidValue;
  const MpmaeDwWgArgs& a0 = probs[0];
  bool ok = dt == 1 && count <= DWG_MAX && dw_variant() >= 6 && a0.CC >= 1 && a0.TP * a0.g.S <= 8;
  for (int i = 0; ok && i < count; ++i) {
    const MpmaeDwWgArgs& a = probs[i];
    ok = a.C == a0.C && a.g.N == a0.g.N && a.g.keep == a0.g.keep && a.g.grid == a0.g.grid && a.g.S == a0.g.S && a.g.vis == a0.g.vis &&
         a.g.inv == a0.g.inv && a.s_kh == a0.s_kh && a.s_kw == a0.s_kw && a.s_c == a0.s_c && a.act == a0.act && a.x && a.dd && a.dw &&
         (((uintptr_t)a.x) & 15) == 0 && (((uintptr_t)a.dd) & 3) == 0;
  }
  const bool s1 = ok && a0.g.S == 1 && a0.g.grid == 7 && (a0.C & 15) == 0;
  const bool v5 = ok && !s1 && a0.g.S >= 2 && dw_v4_ok(a0.C, a0.g.S);
  const size_t per = (size_t)50 * a0.C;
  if (ok && count > 1 && (a0.g.S == 8 || a0.g.S == 4)) {           // matrix-core kernel: grid.z = problem, one slab per sample and problem
    DwWgGroupP gr;
    ReduceGroupP rg;
    gr.count = rg.count = count;
    for (int i = 0; i < count; ++i) {
      gr.x[i] = probs[i].x; gr.dd[i] = probs[i].dd; gr.ws[i] = ws + (size_t)i * a0.g.N * per;
      rg.part[i] = gr.ws[i]; rg.out[i] = probs[i].dw; rg.out2[i] = probs[i].db;
    }
    const int r = try_dwwg_mfma(a0, S_(s), gr, count, ws_floats);
    if (r > 0) return r;
    if (r == 0) {
      const int W = 50 * a0.C;
      LAUNCH(reduce_partials_group2_kernel, dim3(cdiv(W, 64), 16, count), dim3(256), 0, S_(s), rg, a0.g.N, W, a0.C, a0.s_kh, a0.s_kw, a0.s_c);
      RET();
    }
  }
  int nb = 0;
  if (s1) {
    const int want = 0 /* DWW_S1_NB: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ > 0 ? 0 /* DWW_S1_NB: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ : (cdiv(a0.C, 64) <= 5 ? 128 : 64);
    nb = a0.g.N < want ? a0.g.N : want;
  } else if (v5) {
    nb = a0.g.N < 128 /* DWW_NB: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ ? a0.g.N : 128 /* DWW_NB: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
  }
  if (nb > 0 && (size_t)nb * per * count > ws_floats) nb = (int)(ws_floats / (per * count));
  if (nb < 1 || count == 1) {                    // one by one, each on the scratch given here
    for (int i = 0; i < count; ++i) {
      MpmaeDwWgArgs a = probs[i];
      a.ws = ws; a.ws_floats = ws_floats;
      const int err = mpmae_dwconv7_wgrad(dt, &a, 2048, s);
      if (err) return err;
    }
    return 0;
  }
  DwWgGroupP gr;
  ReduceGroupP rg;
  gr.count = rg.count = count;
  for (int i = 0; i < count; ++i) {
    gr.x[i] = probs[i].x; gr.dd[i] = probs[i].dd; gr.ws[i] = ws + (size_t)i * nb * per;
    rg.part[i] = gr.ws[i]; rg.out[i] = probs[i].dw; rg.out2[i] = probs[i].db;
  }
  if (s1) {
    dim3 g(nb, cdiv(a0.C, 64), count);
    LAUNCH((dwconv7_wgrad_v6s1_kernel<7>), g, dim3(256), 0, S_(s), a0, gr);
  } else {
    bool done;
    switch (a0.g.S) { case 8: done = launch_dwwg_v5<bf16_t, 8>(a0, nb, S_(s), &gr); break; case 4: done = launch_dwwg_v5<bf16_t, 4>(a0, nb, S_(s), &gr); break;
                      default: done = launch_dwwg_v5<bf16_t, 2>(a0, nb, S_(s), &gr); }
    if (!done) return (int)hipErrorInvalidValue;
  }
  const int W = 50 * a0.C;
  int R = nb / 16;
  if (R < 1) R = 1;
  if (R > 32) R = 32;
  LAUNCH(reduce_partials_group2_kernel, dim3(cdiv(W, 64), R, count), dim3(256), 0, S_(s), rg, nb, W, a0.C, a0.s_kh, a0.s_kw, a0.s_c);
  RET();
}

int mpmae_dwstride_fwd(int dt, const void* in, void* out, const float* w, const float* b, int Mout, int C, int S, int k,
                       const uint8_t* act_in, const uint8_t* act_out, mpmae_stream_t s) {
  if (k < 1 || k > 2) return (int)hipErrorInvalidValue;
  if (k == 2 && dt == 1 && (C & 7) == 0) {
    LAUNCH(dwstride2_fwd_kernel<bf16_t>, dim3(grid1d((long long)Mout * (C / 8), 256, 8192)), dim3(256), 0, S_(s), (const bf16_t*)in, (bf16_t*)out, w, b,
           Mout, C, S, act_in, act_out);
    RET();
  }
  const int g = grid1d((long long)Mout * C);
  if (dt == 0) LAUNCH(dwstride_fwd_kernel<float>, dim3(g), dim3(256), 0, S_(s), (const float*)in, (float*)out, w, b, Mout, C, S, k, act_in, act_out);
  else LAUNCH(dwstride_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, S_(s), (const bf16_t*)in, (bf16_t*)out, w, b, Mout, C, S, k, act_in, act_out);
  RET();
}

int mpmae_dwstride_bwd(int dt, const void* dout, const void* in, void* din, const float* w, float* dw, float* db,
                       int Mout, int C, int S, int k, const uint8_t* act_in, float* ws, size_t ws_floats, mpmae_stream_t s) {
  if (k < 1 || k > 2) return (int)hipErrorInvalidValue;
  const int cpb = C < 256 ? C : 256;
  const int rows_par = 256 / cpb > 0 ? 256 / cpb : 1;
  const size_t per = (size_t)(k * k + 1) * C;
  if (k == 1 && (C & 7) == 0 && C / 8 <= 256 && ws && ws_floats >= per) {
    int g1 = cdiv(Mout, (256 / (C / 8)) * 8);              // ~8 rows per thread
    if (g1 > 2048) g1 = 2048;
    while ((size_t)g1 * per > ws_floats && g1 > 1) g1 /= 2;
    if (dt == 0) LAUNCH(dwstride1_bwd_kernel<float>, dim3(g1), dim3(256), 0, S_(s), (const float*)dout, (const float*)in, (float*)din, w, ws, Mout, C, act_in);
    else LAUNCH(dwstride1_bwd_kernel<bf16_t>, dim3(g1), dim3(256), 0, S_(s), (const bf16_t*)dout, (const bf16_t*)in, (bf16_t*)din, w, ws, Mout, C, act_in);
    launch_reduce(3, ws, g1, (int)per, dw, db, C, 0, 0, 0, S_(s));
    RET();
  }
  if (k == 2 && dt == 1 && (C & 7) == 0 && C / 8 <= 256 && ws && ws_floats >= per) {      // vectorised 2x2/2 stem (patch 16)
    int g2 = cdiv(Mout, (256 / (C / 8)) * 4);
    if (g2 > 2048) g2 = 2048;
    while ((size_t)g2 * per > ws_floats && g2 > 1) g2 /= 2;
    LAUNCH(dwstride2_bwd_kernel<bf16_t>, dim3(g2), dim3(256), 0, S_(s), (const bf16_t*)dout, (const bf16_t*)in, (bf16_t*)din, w, ws, Mout, C, S, act_in);
    launch_reduce(3, ws, g2, (int)per, dw, db, 4 * C, 0, 0, 0, S_(s));
    RET();
  }
  int g = 512;
  if (!ws || ws_floats < per * rows_par) return (int)hipErrorInvalidValue;
  while ((size_t)g * rows_par * per > ws_floats && g > 1) g /= 2;
  if (dt == 0) LAUNCH(dwstride_bwd_kernel<float>, dim3(g), dim3(256), 0, S_(s), (const float*)dout, (const float*)in, (float*)din, w, ws, Mout, C, S, k, act_in);
  else LAUNCH(dwstride_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, S_(s), (const bf16_t*)dout, (const bf16_t*)in, (bf16_t*)din, w, ws, Mout, C, S, k, act_in);
  // slab row = [k*k taps][C] then [C] bias: e = t*C + c -> t < k*k ? dw[t*C + c] : db[c]; dw is contiguous (k*k, C)
  launch_reduce(3, ws, g * rows_par, (int)per, dw, db, k * k * C, 0, 0, 0, S_(s));
  RET();
}

int mpmae_fill_mask_token(int dt, void* xdec, const float* token, const int* inv, int rows, int D, const void* vis_rows, int keep,
                          int L, mpmae_stream_t s) {
  if (vis_rows) {
    if ((D & 7) || keep < 1 || L < keep || rows % L) return (int)hipErrorInvalidValue;
    const int g = grid1d((long long)rows * (D / 8));
    if (dt == 0) LAUNCH(assemble_tokens_kernel<float>, dim3(g), dim3(256), 0, S_(s), (float*)xdec, token, inv, (const float*)vis_rows, rows, D, keep, L);
    else LAUNCH(assemble_tokens_kernel<bf16_t>, dim3(g), dim3(256), 0, S_(s), (bf16_t*)xdec, token, inv, (const bf16_t*)vis_rows, rows, D, keep, L);
    RET();
  }
  const int g = grid1d((long long)rows * D);
  if (dt == 0) LAUNCH(fill_mask_token_kernel<float>, dim3(g), dim3(256), 0, S_(s), (float*)xdec, token, inv, rows, D);
  else LAUNCH(fill_mask_token_kernel<bf16_t>, dim3(g), dim3(256), 0, S_(s), (bf16_t*)xdec, token, inv, rows, D);
  RET();
}

int mpmae_mask_token_bwd(int dt, const void* dxdec, const int* inv, float* dtoken, int rows, int D, void* vis_rows_out, int keep,
                         int L, mpmae_stream_t s) {
  if ((D & 7) || D / 8 > 256) return (int)hipErrorInvalidValue;
  if (vis_rows_out && (keep < 1 || L < keep || rows % L)) return (int)hipErrorInvalidValue;
  const int rl_n = 256 / (D / 8);
  int blocks = cdiv(rows, rl_n * 8);             // ~8 rows per thread
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  if (dt == 0) LAUNCH(mask_token_bwd_kernel<float>, dim3(blocks), dim3(256), 0, S_(s), (const float*)dxdec, inv, dtoken, rows, D, (float*)vis_rows_out, keep, L);
  else LAUNCH(mask_token_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, S_(s), (const bf16_t*)dxdec, inv, dtoken, rows, D, (bf16_t*)vis_rows_out, keep, L);
  RET();
}

int mpmae_pool_rows(int dt, const void* x, void* pooled, int N, int L, int C, mpmae_stream_t s) {
  const int g = grid1d((long long)N * C);
  if (dt == 0) LAUNCH(pool_rows_kernel<float>, dim3(g), dim3(256), 0, S_(s), (const float*)x, (float*)pooled, N, L, C);
  else LAUNCH(pool_rows_kernel<bf16_t>, dim3(g), dim3(256), 0, S_(s), (const bf16_t*)x, (bf16_t*)pooled, N, L, C);
  RET();
}

// ------------------------------------------------------------------------------------------
int mpmae_loss_pix_cont(int dt, int bwd, const MpmaePixContArgs* a, int npatches, mpmae_stream_t s) {
  if (!a || a->L < 1) return (int)hipErrorInvalidValue;
  const MpmaePixContArgs& A = *a;      // (captured by value by LAUNCH)
  dim3 g(bwd ? npatches : npatches / a->L), b(bwd ? 256 : 512) ;   // forward: one 16-wave block per sample
  if (dt == 0) { if (bwd) LAUNCH((loss_pix_cont_kernel<float, true>), g, b, 0, S_(s), A);
                 else LAUNCH((loss_pix_cont_kernel<float, false>), g, b, 0, S_(s), A); }
  else { if (bwd) LAUNCH((loss_pix_cont_kernel<bf16_t, true>), g, b, 0, S_(s), A);
         else LAUNCH((loss_pix_cont_kernel<bf16_t, false>), g, b, 0, S_(s), A); }
  RET();
}

int mpmae_loss_pix_cat(int dt, int bwd, const MpmaePixCatArgs* a, int npatches, mpmae_stream_t s) {
  if (!a || a->K > 16 || a->L < 1) return (int)hipErrorInvalidValue;
  const MpmaePixCatArgs& A = *a;      // (captured by value by LAUNCH)
  dim3 g(bwd ? npatches : npatches / a->L), b(bwd ? 256 : 1024);   // forward: one 16-wave block per sample
  if (dt == 0) { if (bwd) LAUNCH((loss_pix_cat_kernel<float, true>), g, b, 0, S_(s), A);
                 else LAUNCH((loss_pix_cat_kernel<float, false>), g, b, 0, S_(s), A); }
  else { if (bwd) LAUNCH((loss_pix_cat_kernel<bf16_t, true>), g, b, 0, S_(s), A);
         else LAUNCH((loss_pix_cat_kernel<bf16_t, false>), g, b, 0, S_(s), A); }
  RET();
}

int mpmae_loss_img(int dt, int bwd, const MpmaeImgArgs* a, mpmae_stream_t s) {
  if (!a || a->N < 1) return (int)hipErrorInvalidValue;
  const MpmaeImgArgs& A = *a;      // (captured by value by LAUNCH)
  dim3 g(a->N), b(256);
  if (dt == 0) { if (bwd) LAUNCH((loss_img_kernel<float, true>), g, b, 0, S_(s), A);
                 else LAUNCH((loss_img_kernel<float, false>), g, b, 0, S_(s), A); }
  else { if (bwd) LAUNCH((loss_img_kernel<bf16_t, true>), g, b, 0, S_(s), A);
         else LAUNCH((loss_img_kernel<bf16_t, false>), g, b, 0, S_(s), A); }
  RET();
}

int mpmae_loss_multi(int dt, int bwd, int kind, const void* dev_args, int count, int gridx, mpmae_stream_t s) {
  if (!dev_args || count < 1 || gridx < 1 || kind < 0 || kind > 2) return (int)hipErrorInvalidValue;
  dim3 g(gridx, count);
#define LM(KERN, ARGT, TH) do { \
    if (dt == 0) { if (bwd) LAUNCH((KERN<float, true>), g, dim3(256), 0, S_(s), (const ARGT*)dev_args); \
                   else LAUNCH((KERN<float, false>), g, dim3(TH), 0, S_(s), (const ARGT*)dev_args); } \
    else { if (bwd) LAUNCH((KERN<bf16_t, true>), g, dim3(256), 0, S_(s), (const ARGT*)dev_args); \
           else LAUNCH((KERN<bf16_t, false>), g, dim3(TH), 0, S_(s), (const ARGT*)dev_args); } } while (0)
  if (kind == 0) LM(loss_pix_cont_multi_kernel, MpmaePixContArgs, 512);
  else if (kind == 1) LM(loss_pix_cat_multi_kernel, MpmaePixCatArgs, 1024);
  else LM(loss_img_multi_kernel, MpmaeImgArgs, 256);
#undef LM
  RET();
}

static int loss_pix_cont_rows_impl(int dt, int bwd, const void* dev_args, int count, int N, int maxC, int p, int H, mpmae_stream_t s) {
  const int gx = N, split = 0;      // (k workgroups per sample, each walking a share of the patch rows: measured slower in round 5 and removed; the kernel keeps the parameter)
  if (bwd < 0 || bwd > 2) return (int)hipErrorInvalidValue;
  if (!dev_args || count < 1 || N < 1 || maxC < 1 || p < 1 || (H & 3) || ((p * p) & 3)) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)maxC * (p * H + 4) * 4;
  const int nvec = maxC * p * (H / 4), npv = maxC * p * p / 4;
  const int mv = cdiv(nvec, 512), mp = cdiv(npv, 64);
  if (lds > 150 * 1024 || mv > 12 || mp > 12) return (int)hipErrorInvalidValue;
  const auto* tab = (const MpmaePixContArgs*)dev_args;
#define LPR(TT, MV, MP, BW) do { \
    static size_t cur = 48 * 1024; \
    if (lds > cur) { if (hipFuncSetAttribute((const void*)loss_pix_cont_rows_kernel<TT, MV, MP, BW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } \
    LAUNCH((loss_pix_cont_rows_kernel<TT, MV, MP, BW>), dim3(gx, count), dim3(512), lds, S_(s), tab, split); } while (0)
#define LPR2(TT, MV, MP) do { if (bwd == 1) LPR(TT, MV, MP, 1); else if (bwd == 2) LPR(TT, MV, MP, 2); else LPR(TT, MV, MP, 0); } while (0)
  if (mv <= 3 && mp <= 3) { if (dt == 0) LPR2(float, 3, 3); else LPR2(bf16_t, 3, 3); }
  else { if (dt == 0) LPR2(float, 12, 12); else LPR2(bf16_t, 12, 12); }
#undef LPR2
#undef LPR
  RET();
}

int mpmae_loss_pix_cont_rows(int dt, const void* dev_args, int count, int N, int maxC, int p, int H, mpmae_stream_t s) {
  return loss_pix_cont_rows_impl(dt, 0, dev_args, count, N, maxC, p, H, s);
}

int mpmae_loss_pix_cont_rows_bwd(int dt, const void* dev_args, int count, int N, int maxC, int p, int H, mpmae_stream_t s) {
  return loss_pix_cont_rows_impl(dt, 1, dev_args, count, N, maxC, p, H, s);
}

int mpmae_loss_pix_cont_rows_fused(int dt, const void* dev_args, int count, int N, int maxC, int p, int H, mpmae_stream_t s) {
  return loss_pix_cont_rows_impl(dt, 2, dev_args, count, N, maxC, p, H, s);
}

int mpmae_loss_pix_cat_waves(int dt, int bwd, const void* dev_args, int count, int N, int max_pk, mpmae_stream_t s) {
  if (!dev_args || count < 1 || N < 1 || max_pk < 4 || (max_pk & 3) || bwd < 0 || bwd > 2) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)16 * max_pk * (dt == 0 ? 4 : 2);
  if (lds > 150 * 1024) return (int)hipErrorInvalidValue;
  const auto* tab = (const MpmaePixCatArgs*)dev_args;
#define LCW(TT, BW) do { \
    static size_t cur = 48 * 1024; \
    if (lds > cur) { if (hipFuncSetAttribute((const void*)loss_pix_cat_waves_kernel<TT, BW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } \
    LAUNCH((loss_pix_cat_waves_kernel<TT, BW>), dim3(N, count), dim3(1024), lds, S_(s), tab, max_pk); } while (0)
  if (dt == 0) { if (bwd == 1) LCW(float, 1); else if (bwd == 2) LCW(float, 2); else LCW(float, 0); }
  else { if (bwd == 1) LCW(bf16_t, 1); else if (bwd == 2) LCW(bf16_t, 2); else LCW(bf16_t, 0); }
#undef LCW
  RET();
}

int mpmae_loss_finalize_guarded(const float* acc, int N, const float* log_vars, int T, float loss_scale, float* losses,
                                float* weighted, float* total, float* coef, float* dlog_vars, const unsigned* err_words, int n_err,
                                int err_stride, mpmae_stream_t s) {
  if (T > 16 || T < 1 || n_err < 0 || (n_err > 0 && (!err_words || err_stride < 3))) return (int)hipErrorInvalidValue;
  LAUNCH(loss_finalize_kernel, dim3(1), dim3(64 * T), 0, S_(s), acc, N, log_vars, T, loss_scale, losses, weighted,
                     total, coef, dlog_vars, err_words, err_words ? n_err : 0, err_stride);
  RET();
}

int mpmae_loss_finalize(const float* acc, int N, const float* log_vars, int T, float loss_scale, float* losses,
                        float* weighted, float* total, float* coef, float* dlog_vars, mpmae_stream_t s) {
  return mpmae_loss_finalize_guarded(acc, N, log_vars, T, loss_scale, losses, weighted, total, coef, dlog_vars, nullptr, 0, 0, s);
}

int mpmae_adamw(float* p, const float* g, float* m, float* v, const float* hp, float beta1, float beta2, float eps,
                float wd, size_t n, const uint8_t* decay, float* gnorm2, mpmae_stream_t s) {
  const int nb = grid1d((long long)n, 256, 4096);
  LAUNCH(adamw_kernel, dim3(nb), dim3(256), 0, S_(s), p, g, m, v, hp, beta1, beta2, eps, wd, n, decay, gnorm2, 0, nb);
  RET();
}

int mpmae_sumsq(const float* x, size_t n, float* out, mpmae_stream_t s) {
  LAUNCH(sumsq_kernel, dim3(grid1d((long long)n, 256, 1024)), dim3(256), 0, S_(s), x, n, out);
  RET();
}


int mpmae_crop(const void* src, void* dst, int elem_bytes, int N, int C, int H, int S, const int* ty, const int* tx, mpmae_stream_t s) {
  if (!src || !dst || !ty || !tx || N < 1 || C < 1 || S < 1 || H < S || (elem_bytes != 4 && elem_bytes != 8)) return (int)hipErrorInvalidValue;
  const int g = grid1d((long long)N * C * S * S, 256, 16384);
  if (elem_bytes == 4) LAUNCH(crop_kernel<uint32_t>, dim3(g), dim3(256), 0, S_(s), (const uint32_t*)src, (uint32_t*)dst, N, C, H, S, ty, tx);
  else LAUNCH(crop_kernel<unsigned long long>, dim3(g), dim3(256), 0, S_(s), (const unsigned long long*)src, (unsigned long long*)dst, N, C, H, S, ty, tx);
  RET();
}

int mpmae_crop_norm(const void* src, int src_type, float* dst, int N, int C, int H, int S, const int* ty, const int* tx,
                    const float* mean, const float* stdv, float nodata, mpmae_stream_t s) {
  if (!src || !dst || !mean || !stdv || N < 1 || C < 1 || S < 1 || H < S || ((ty == nullptr) != (tx == nullptr))) return (int)hipErrorInvalidValue;
  const int g = grid1d((long long)N * C * S * S, 256, 16384);
  if (src_type == 0) LAUNCH(crop_norm_kernel<float>, dim3(g), dim3(256), 0, S_(s), (const float*)src, dst, N, C, H, S, ty, tx, mean, stdv, nodata);
  else if (src_type == 1) LAUNCH(crop_norm_kernel<uint16_t>, dim3(g), dim3(256), 0, S_(s), (const uint16_t*)src, dst, N, C, H, S, ty, tx, mean, stdv, nodata);
  else if (src_type == 2) LAUNCH(crop_norm_kernel<uint8_t>, dim3(g), dim3(256), 0, S_(s), (const uint8_t*)src, dst, N, C, H, S, ty, tx, mean, stdv, nodata);
  else return (int)hipErrorInvalidValue;
  RET();
}

int mpmae_crop_lut(const uint8_t* src, long long* dst, int N, int H, int S, const int* ty, const int* tx, const int* lut256, mpmae_stream_t s) {
  if (!src || !dst || !lut256 || N < 1 || S < 1 || H < S || ((ty == nullptr) != (tx == nullptr))) return (int)hipErrorInvalidValue;
  LAUNCH(crop_lut_kernel, dim3(grid1d((long long)N * S * S, 256, 16384)), dim3(256), 0, S_(s), src, dst, N, H, S, ty, tx, lut256);
  RET();
}

int mpmae_im2col3(int dt, const float* img, const int* vis, const int* inv, void* out, int ldo, int N, int keep, int grid,
                  int S, int Cseg, int H, mpmae_stream_t s) {
  const int epv = dt == 0 ? 4 : 8;
  if (!img || !vis || !inv || !out || ldo < 9 * Cseg || (ldo % epv) || S < 1 || S > 16) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)(S + 2) * (S + 2) * (Cseg | 1) * sizeof(float);      // odd channel pitch in LDS
  if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
  if (dt == 0) LAUNCH(im2col3_kernel<float>, dim3(N * keep), dim3(256), lds, S_(s), img, vis, inv, (float*)out, ldo, keep, grid, S, Cseg, H);
  else LAUNCH(im2col3_kernel<bf16_t>, dim3(N * keep), dim3(256), lds, S_(s), img, vis, inv, (bf16_t*)out, ldo, keep, grid, S, Cseg, H);
  RET();
}

int mpmae_gather_kxk(int dt, const float* img, const int* vis, const int* inv, void* out, int ldo, int N, int keep, int grid, int p, int k, int Cseg, int H,
                     mpmae_stream_t s) {
  if (!img || !out || N < 1 || keep < 1 || k < 1 || p != 8 * k || H != grid * p || ldo < k * k * Cseg) return (int)hipErrorInvalidValue;
  const int rows = N * keep * 64;
  const int g = grid1d((long long)rows * ldo, 256, 8192);
  if (dt == 0) LAUNCH(gather_kxk_kernel<float>, dim3(g), dim3(256), 0, S_(s), img, vis, inv, (float*)out, ldo, keep, grid, p, k, Cseg, H, rows);
  else LAUNCH(gather_kxk_kernel<bf16_t>, dim3(g), dim3(256), 0, S_(s), img, vis, inv, (bf16_t*)out, ldo, keep, grid, p, k, Cseg, H, rows);
  RET();
}

int mpmae_head_scale(int dt, const void* B, void* Bout, int ldb, int D, int W, const uint8_t* col_mod, const float* coef, float* rowscale, mpmae_stream_t s) {
  if (!B || !Bout || !col_mod || !coef || !rowscale || D < 1 || W < 1 || ldb < W) return (int)hipErrorInvalidValue;
  const int g = grid1d((long long)D * W, 256, 2048);
  if (dt == 0) LAUNCH(head_scale_kernel<float>, dim3(g), dim3(256), 0, S_(s), (const float*)B, (float*)Bout, ldb, D, W, col_mod, coef, rowscale);
  else LAUNCH(head_scale_kernel<bf16_t>, dim3(g), dim3(256), 0, S_(s), (const bf16_t*)B, (bf16_t*)Bout, ldb, D, W, col_mod, coef, rowscale);
  RET();
}

int mpmae_strided_add(float* dst, const float* src, int rows, int cols, int src_ld, int dst_sr, int dst_sc, mpmae_stream_t s) {
  if (!dst || !src || rows < 1 || cols < 1) return (int)hipErrorInvalidValue;
  LAUNCH(strided_add_kernel, dim3(grid1d((long long)rows * cols, 256, 256)), dim3(256), 0, S_(s), dst, src, rows, cols, src_ld, dst_sr, dst_sc);
  RET();
}

int mpmae_stem_front(const MpmaeStemFrontArgs* a, mpmae_stream_t s) {
  if (!a || !a->img || !a->vis || !a->inv || (!a->W && !a->W_master) || !a->bias || !a->out || !a->xhat1 || !a->xhat2 || !a->rstd1 || !a->rstd2)
    return (int)hipErrorInvalidValue;
  if (a->Cin < 1 || a->Cin > 12 || a->C0 < 4 || a->C0 > 48 || (a->C0 & 3) || (a->W && ((a->ldw & 7) || a->ldw < 9 * a->Cin)) || a->N < 1 ||
      a->keep < 1 || a->H != a->grid * 8)
    return (int)hipErrorInvalidValue;
  StemFrontP p;
  p.img = a->img; p.vis = a->vis; p.inv = a->inv; p.W = reinterpret_cast<const bf16_t*>(a->W); p.ldw = a->ldw; p.Wm = a->W_master; p.bias = a->bias;
  p.xhat1 = a->xhat1; p.rstd1 = a->rstd1; p.xhat2 = a->xhat2; p.rstd2 = a->rstd2; p.out = a->out;
  p.g1 = a->g1; p.b1 = a->b1; p.w = a->w; p.wb = a->wb; p.g2 = a->g2; p.b2 = a->b2;
  if (a->col && ((a->ldc & 7) || a->ldc < 9 * a->Cin || a->ldc > 128)) return (int)hipErrorInvalidValue;
  p.col = reinterpret_cast<bf16_t*>(a->col); p.ldc = a->ldc;
  p.keep = a->keep; p.grid = a->grid; p.H = a->H; p.Cin = a->Cin; p.C0 = a->C0; p.track = a->track_activity;
  p.npatch = a->N * a->keep;
  p.act_out = a->act_out;
  static int per_cu[2] = {0, 0};                                   // resident workgroups per CU of the two instantiations (asked once)
  const int which = a->Cin == 12 ? 1 : 0;
  if (!per_cu[which]) {
    int nb = 0;
    const hipError_t e = which ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stem_front_kernel<12>, 256, 0)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stem_front_kernel<0>, 256, 0);
    if (e != hipSuccess) (void)hipGetLastError();
    per_cu[which] = (e == hipSuccess && nb > 0) ? nb : 2;
  }
  const int blocks = std::min(p.npatch, per_cu[which] * ps_num_cus());      // persistent: one resident wave of workgroups walks all patches
  if (a->Cin == 12) LAUNCH(stem_front_kernel<12>, dim3(blocks), dim3(256), 0, S_(s), p);
  else LAUNCH(stem_front_kernel<0>, dim3(blocks), dim3(256), 0, S_(s), p);
  RET();
}

// ONE launch per FOLDG_MAX mode-1 fold records (blockIdx.z = record) instead of one launch per record
static void launch_fold_records(const MpmaeFoldDesc* descs, int count, hipStream_t st) {
  for (int i0 = 0; i0 < count; i0 += FOLDG_MAX) {
    const int n = count - i0 < FOLDG_MAX ? count - i0 : FOLDG_MAX;
    FoldGroupP g;
    g.count = n; g.pad = 0;
    int wmax = 1, pmax = 1;
    for (int i = 0; i < n; ++i) {
      const MpmaeFoldDesc& d = descs[i0 + i];
      g.part[i] = d.part; g.out[i] = d.out; g.P[i] = d.P; g.W[i] = d.W; g.a[i] = d.a; g.b[i] = d.b; g.c[i] = d.c;
      if (d.W > wmax) wmax = d.W;
      if (d.P > pmax) pmax = d.P;
    }
    for (int i = n; i < FOLDG_MAX; ++i) { g.part[i] = nullptr; g.out[i] = nullptr; g.P[i] = g.W[i] = 0; g.a[i] = 1; g.b[i] = g.c[i] = 0; }
    int R = pmax / 16;                 // as launch_reduce: >= 4 rows per thread
    if (R < 1) R = 1;
    if (R > 32) R = 32;
    if (g_opt[MPMAE_OPT_DET] > 0) R = 1;      // (as launch_reduce: plain `+=` in a fixed order)
    LAUNCH(reduce_partials_group1_kernel, dim3(cdiv(wmax, 64), R, n), dim3(256), 0, st, g);
  }
}

int mpmae_stem_tail(int dt, int bwd, const MpmaeStemTailArgs* a, mpmae_stream_t s) {
  if (!a || (a->C & 7) || a->C > 512 || a->M < 1) return (int)hipErrorInvalidValue;
  const int nvec = a->C / 8;
  const int G = nvec <= 8 ? 8 : nvec <= 16 ? 16 : nvec <= 32 ? 32 : 64;
  const int rpw = 64 / G;
  StemTailP p;
  p.x = a->x; p.xhat1 = a->xhat1; p.rstd1 = a->rstd1; p.xhat2 = a->xhat2; p.rstd2 = a->rstd2; p.out = a->out;
  p.g1 = a->g1; p.b1 = a->b1; p.w = a->w; p.wb = a->wb; p.g2 = a->g2; p.b2 = a->b2;
  p.act_in = a->act_in; p.act_out = a->act_out; p.ws = a->ws; p.M = a->M; p.C = a->C;
  int stcap;
  stcap = 512 /* STB_BLOCKS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
  int blocks = grid1d((long long)cdiv(a->M, rpw) * 64, 256, bwd ? stcap : 4096);
  if (bwd) {
    if (!a->ws || !a->dg1 || !a->db1 || !a->dw || !a->dwb || !a->dg2 || !a->db2) return (int)hipErrorInvalidValue;
    while ((size_t)blocks * 4 * 6 * a->C > a->ws_floats && blocks > 1) blocks /= 2;
    if ((size_t)blocks * 4 * 6 * a->C > a->ws_floats) return (int)hipErrorInvalidValue;
  }
#define ST(TT, GG) do { if (bwd) LAUNCH((stem_tail_bwd_kernel<TT, GG>), dim3(blocks), dim3(256), 0, S_(s), p); \
                        else LAUNCH((stem_tail_fwd_kernel<TT, GG>), dim3(blocks), dim3(256), 0, S_(s), p); } while (0)
#define ST_T(TT) do { if (G == 8) ST(TT, 8); else if (G == 16) ST(TT, 16); else if (G == 32) ST(TT, 32); else ST(TT, 64); } while (0)
  if (dt == 0) ST_T(float); else ST_T(bf16_t);
#undef ST_T
#undef ST
  if (bwd) {
    const int nw = blocks * 4;
    float* outs[3][2] = {{a->dg2, a->db2}, {a->dw, a->dwb}, {a->dg1, a->db1}};
    MpmaeFoldDesc fd[3];
    for (int k = 0; k < 3; ++k) {      // slabs [k][wave][2][C]: e = n*C + c -> n == 0 ? first[c] : second[c]
      const long long delta = outs[k][1] - outs[k][0];
      if (delta > 2147483647LL || delta < -2147483647LL) return (int)hipErrorInvalidValue;
      fd[k].part = a->ws + (size_t)k * nw * 2 * a->C; fd[k].out = outs[k][0];
      fd[k].P = nw; fd[k].W = 2 * a->C; fd[k].a = a->C; fd[k].b = (int)delta; fd[k].c = 1;
    }
    // the three folds of ONE producer kernel in one launch (round 5: they are the exposed tail of the step on the main lane; grouping the
    // folds of DIFFERENT producers was slower, MPMAE_OPT_FOLD_GROUP); FOLD_GROUP < 0 keeps the three launches
    if (g_opt[MPMAE_OPT_FOLD_GROUP] >= 0) launch_fold_records(fd, 3, S_(s));
    else for (int k = 0; k < 3; ++k) launch_reduce(1, fd[k].part, fd[k].P, fd[k].W, fd[k].out, nullptr, fd[k].a, fd[k].b, fd[k].c, 0, S_(s));
  }
  RET();
}

// ------------------------------------------------------------------------------------------
// launch programs (see LAUNCH above)
// ------------------------------------------------------------------------------------------
MpmaeProgram* mpmae_program_create(void) { return new MpmaeProgram(); }

void mpmae_program_destroy(MpmaeProgram* p) {
  if (!p) return;
  if (g_rec == p) g_rec = nullptr;
  for (auto s : p->side) (void)hipStreamDestroy(s);
  for (auto e : p->events) if (e) (void)hipEventDestroy(e);
  for (auto e : p->join) if (e) (void)hipEventDestroy(e);
  if (p->fork) (void)hipEventDestroy(p->fork);
  delete p;
}

int mpmae_program_begin_op(MpmaeProgram* p, int lane, const int* waits, int nwaits, int signal) {
  if (!p || lane < 0 || lane > 7 || signal < 0 || nwaits < 0) return (int)hipErrorInvalidValue;
  p->ops.emplace_back();
  ProgOp& op = p->ops.back();
  op.lane = lane;
  op.signal = signal;
  for (int i = 0; i < nwaits; ++i) op.waits.push_back(waits[i]);
  if (lane + 1 > p->nlanes) p->nlanes = lane + 1;
  g_rec = p;
  return 0;
}

int mpmae_program_end(MpmaeProgram* p) {
  if (!p) return (int)hipErrorInvalidValue;
  g_rec = nullptr;
  int maxid = 0;
  for (auto& op : p->ops) {
    if (op.signal > maxid) maxid = op.signal;
    for (int w : op.waits) if (w > maxid) maxid = w;
  }
  while ((int)p->events.size() <= maxid) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
    p->events.push_back(e);
    p->epoch.push_back(0u);
  }
  while ((int)p->side.size() < p->nlanes - 1) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return (int)hipGetLastError();
    p->side.push_back(s);
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
    p->join.push_back(e);
  }
  if (!p->fork && hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
  p->sig_op.assign(maxid + 1, -1);
  p->sig_lane.assign(maxid + 1, -1);
  p->waited.assign(maxid + 1, 0);
  for (size_t i = 0; i < p->ops.size(); ++i)
    if (p->ops[i].signal > 0) { p->sig_op[p->ops[i].signal] = (int)i; p->sig_lane[p->ops[i].signal] = p->ops[i].lane; }
  for (auto& op : p->ops)
    for (int w : op.waits) if (w > 0 && p->sig_lane[w] != op.lane) p->waited[w] = 1;
  return 0;
}

int mpmae_program_num_ops(const MpmaeProgram* p) { return p ? (int)p->ops.size() : -1; }

// Export an op's `signal` event to a stream outside the program (the gradient exchange): export_signal() keeps the event recorded even
// though no op of the program waits for it (call after mpmae_program_end), stream_wait() makes `stream` wait for the event as recorded
// by the MOST RECENT run() call (a no-op when that call did not reach the op).
int mpmae_program_export_signal(MpmaeProgram* p, int signal) {
  if (!p || signal <= 0 || signal >= (int)p->waited.size() || p->sig_op[signal] < 0) return (int)hipErrorInvalidValue;
  p->waited[signal] = 1;
  return 0;
}

int mpmae_program_stream_wait(MpmaeProgram* p, int signal, mpmae_stream_t stream) {
  if (!p || signal <= 0 || signal >= (int)p->events.size()) return (int)hipErrorInvalidValue;
  if (p->epoch[signal] != p->run) return 0;
  return hipStreamWaitEvent(S_(stream), p->events[signal], 0) == hipSuccess ? 0 : (int)hipGetLastError();
}

// ---- a side lane must not share the main stream's HARDWARE queue ------------------------------------------------------------------
// ROCm maps HIP streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues round-robin in creation order; streams that land on the
// same queue are serialised. Whether the weight-gradient lane overlaps with the main lane therefore depended on how many streams
// torch / RCCL had created before the program: measured 4.38 ms per step when the lanes sat on different queues and 5.38 ms when a
// one-rank RCCL communicator had shifted the side lane onto the main stream's queue (kernel trace: both lanes on queue 4). At the
// first replay against a given main stream every side lane is therefore PROBED - two 30 us spin kernels, one per stream, started
// together: concurrent streams finish in ~30 us, a shared queue in ~60 - and a lane that does not overlap is replaced by a freshly
// created stream (the next queue in the rotation), up to eight times. One host synchronisation, once per (program, main stream).
__global__ void lane_probe_spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// elapsed time (us) of one spin on `main` [+ one on `side`, started together]; < 0 on any runtime error
static float lane_probe_us(hipStream_t main, hipStream_t side, hipEvent_t e0, hipEvent_t e1, hipEvent_t es) {
  const unsigned long long ticks = 3000;                                // 30 us
  if (hipStreamSynchronize(main) != hipSuccess || (side && hipStreamSynchronize(side) != hipSuccess)) return -1.f;
  if (hipEventRecord(e0, main) != hipSuccess || (side && hipStreamWaitEvent(side, e0, 0) != hipSuccess)) return -1.f;
  hipLaunchKernelGGL(lane_probe_spin_kernel, dim3(1), dim3(64), 0, main, ticks);
  if (side) {
    hipLaunchKernelGGL(lane_probe_spin_kernel, dim3(1), dim3(64), 0, side, ticks);
    if (hipEventRecord(es, side) != hipSuccess || hipStreamWaitEvent(main, es, 0) != hipSuccess) return -1.f;
  }
  if (hipEventRecord(e1, main) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1.f;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.f;
  return ms * 1e3f;
}

static bool lane_overlaps(hipStream_t main, hipStream_t side, hipEvent_t e0, hipEvent_t e1, hipEvent_t es, float alone_us) {
  const float both = lane_probe_us(main, side, e0, e1, es);
  return both < 0.f || both < alone_us + 33.f;      // measured: +19 us (the cross-stream event) when concurrent, +48 us on a shared hardware queue
}

// (once per distinct main stream, and for at most 8 of them: a caller that rotates main streams is not probed - and synchronised - forever)
static bool lanes_need_check(const MpmaeProgram* p, hipStream_t main) {
  if (p->lanes_checked_for.size() >= 8) return false;
  for (auto m : p->lanes_checked_for) if (m == main) return false;
  return true;
}

static void lanes_overlap_check(MpmaeProgram* p, hipStream_t main) {
  p->lanes_checked_for.push_back(main);
  hipEvent_t e0 = nullptr, e1 = nullptr, es = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreateWithFlags(&es, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  std::vector<hipStream_t> rejected;
  (void)lane_probe_us(main, nullptr, e0, e1, es);
  float alone = lane_probe_us(main, nullptr, e0, e1, es);
  for (int i = 0; i < 2; ++i) { const float t = lane_probe_us(main, nullptr, e0, e1, es); if (t > 0.f && t < alone) alone = t; }
  if (alone > 0.f) {
    for (auto& lane : p->side) {
      (void)lane_probe_us(main, lane, e0, e1, es);                     // warm-up (the first launch on a new stream creates its queue)
      for (int attempt = 0; attempt < 8 && !lane_overlaps(main, lane, e0, e1, es, alone); ++attempt) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        rejected.push_back(lane);                                       // kept alive until the end: the rotation must move on
        lane = s;
        (void)lane_probe_us(main, lane, e0, e1, es);
      }
    }
  }
  for (auto s : rejected) (void)hipStreamDestroy(s);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(es);
  (void)hipGetLastError();
}

// `other` (a stream outside the program: the gradient-exchange stream, the input-stage stream) against the main stream and every side lane
int mpmae_program_stream_overlaps(MpmaeProgram* p, mpmae_stream_t main_, mpmae_stream_t other_) {
  if (g_rec) return -(int)hipErrorInvalidValue;
  hipStream_t main = S_(main_), other = S_(other_);
  if (other == main) return 0;
  if (p && !p->side.empty() && lanes_need_check(p, main)) lanes_overlap_check(p, main);
  hipEvent_t e0 = nullptr, e1 = nullptr, es = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreateWithFlags(&es, hipEventDisableTiming) != hipSuccess)
    return -(int)hipGetLastError();
  int ok = 1;
  std::vector<hipStream_t> against{main};
  if (p) for (auto s : p->side) against.push_back(s);
  for (auto a : against) {
    (void)lane_probe_us(a, nullptr, e0, e1, es);
    float alone = lane_probe_us(a, nullptr, e0, e1, es);
    const float t = lane_probe_us(a, nullptr, e0, e1, es);
    if (t > 0.f && t < alone) alone = t;
    (void)lane_probe_us(a, other, e0, e1, es);
    if (alone > 0.f && !lane_overlaps(a, other, e0, e1, es, alone)) ok = 0;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(es);
  (void)hipGetLastError();
  return ok;
}

int mpmae_program_run(MpmaeProgram* p, int first, int count, mpmae_stream_t main_) {
  if (!p || g_rec || first < 0 || count < 0 || first + count > (int)p->ops.size()) return (int)hipErrorInvalidValue;
  hipStream_t main = S_(main_);
  bool lanes = false;
  for (int i = first; i < first + count; ++i) lanes |= p->ops[i].lane != 0;
  ++p->run;
  if (lanes && lanes_need_check(p, main)) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) lanes_overlap_check(p, main);
  }
  if (lanes) {                                   // fork the side lanes from the main stream
    if (hipEventRecord(p->fork, main) != hipSuccess) return (int)hipGetLastError();
    for (auto s : p->side) if (hipStreamWaitEvent(s, p->fork, 0) != hipSuccess) return (int)hipGetLastError();
  }
  // A lane is an in-order stream: once lane A has waited for the event of op k of lane B, every earlier op of lane B is
  // implied, and so is everything of lane A itself. A wait is a barrier packet in the queue whether or not the event is
  // long complete, so the implied ones are dropped (atto step: 33 -> 20 on the main lane), as are records nobody waits for.
  int seen[8][8];
  for (auto& r : seen) for (int& v : r) v = -1;
  for (int i = first; i < first + count; ++i) {
    ProgOp& op = p->ops[i];
    hipStream_t st = op.lane == 0 ? main : p->side[op.lane - 1];
    for (int w : op.waits) {                     // only events recorded in THIS run (earlier ones were joined)
      if (w <= 0 || p->epoch[w] != p->run) continue;
      const int sl = p->sig_lane[w], so = p->sig_op[w];
      if (sl == op.lane || seen[op.lane][sl] >= so) continue;
      if (hipStreamWaitEvent(st, p->events[w], 0) != hipSuccess) return (int)hipGetLastError();
      seen[op.lane][sl] = so;
    }
    const bool sig = op.signal > 0 && p->waited[op.signal];
    const size_t nl = op.launches.size();
    for (size_t j = 0; j < nl; ++j) {
      if (sig && j + 1 == nl && g_opt[MPMAE_OPT_EVX] > 0) g_stop_ev = p->events[op.signal];      // the last launch carries the signal itself
      op.launches[j](st);
    }
    if (sig) {
      if (g_stop_ev || nl == 0 || g_opt[MPMAE_OPT_EVX] <= 0) {      // not consumed (the last launch is no kernel) / no launch / option off
        g_stop_ev = nullptr;
        if (hipEventRecord(p->events[op.signal], st) != hipSuccess) return (int)hipGetLastError();
      }
      p->epoch[op.signal] = p->run;
    }
  }
  if (lanes) {                                   // join
    for (size_t l = 0; l < p->side.size(); ++l) {
      // ... unless the main lane has ALREADY waited for this lane's last op of the run (the step: the optimizer's first op waits for the last
      // weight gradient of every lane): the record + wait pair would be two more barrier packets at the step boundary for an implied order
      int last = -1;
      for (int i = first + count - 1; i >= first && last < 0; --i) if (p->ops[i].lane == (int)l + 1) last = i;
      if (last < 0 || seen[0][l + 1] >= last) continue;
      if (hipEventRecord(p->join[l], p->side[l]) != hipSuccess) return (int)hipGetLastError();
      if (hipStreamWaitEvent(main, p->join[l], 0) != hipSuccess) return (int)hipGetLastError();
    }
  }
  return launch_status();
}

int mpmae_hp_fetch(const float* ring_pinned, int slots, int* counter, float* hp, const float* total, const MpmaeMeters* meters,
                   mpmae_stream_t s) {
  if (!ring_pinned || slots < 1 || !counter || !hp) return (int)hipErrorInvalidValue;
  MeterP mt{nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, 0};
  if (meters && meters->ring) {
    if (!meters->losses || !meters->sums || !meters->gnorm2 || meters->T < 1 || meters->window < 1) return (int)hipErrorInvalidValue;
    mt = MeterP{meters->losses, meters->weighted, meters->T, meters->ring, meters->window, meters->sums, meters->gnorm2,
                meters->err_words, meters->err_words ? meters->n_err : 0, meters->err_stride};
  }
  LAUNCH(hp_fetch_kernel, dim3(1), dim3(1024), 0, S_(s), ring_pinned, slots, counter, hp, total, mt);
  RET();
}

int mpmae_fold_group(const MpmaeFoldDesc* descs, int count, mpmae_stream_t s) {
  if (!descs || count < 0) return (int)hipErrorInvalidValue;
  for (int i = 0; i < count; ++i) {
    const MpmaeFoldDesc& d = descs[i];
    if (!d.part || !d.out || d.P < 1 || d.W < 1 || d.a < 1) return (int)hipErrorInvalidValue;
  }
  if (!g_opt[MPMAE_OPT_FOLD_GROUP]) {
    for (int i = 0; i < count; ++i) {
      const MpmaeFoldDesc& d = descs[i];
      launch_reduce(1, d.part, d.P, d.W, d.out, nullptr, d.a, d.b, d.c, 0, S_(s));
    }
    RET();
  }
  launch_fold_records(descs, count, S_(s));
  RET();
}

int mpmae_memset_async(void* ptr, int value, size_t bytes, mpmae_stream_t s) {
  submit(S_(s), [=](hipStream_t st) { (void)hipMemsetAsync(ptr, value, bytes, st); });
  RET();
}

int mpmae_memcpy_h2d_async(void* dst, const void* src_pinned, size_t bytes, mpmae_stream_t s) {
  submit(S_(s), [=](hipStream_t st) { (void)hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, st); });
  RET();
}
