// extern "C" surface of libmpmae_hip.so (see include/mpmae_hip.h). gfx950 only.
#include "common.cuh"
#include "gemm.cuh"
#include "rows.cuh"
#include "dwconv.cuh"
#include "misc.cuh"
#include "loss.cuh"
#include "gemm_fast.cuh"
#include "dwconv2.cuh"
#include "dwconv3.cuh"
#include "rows2.cuh"
#include "rs.cuh"
#include "dwconv6.cuh"
#include "gemm_tn2.cuh"
#include "rsc.cuh"
#include "stemtail.cuh"
#include "gemm_nt3.cuh"
#include "dwband.cuh"
#include "dwmfma.cuh"
#include "dwmfma_wg.cuh"
#include "ps.cuh"
#include "gemm_tn3.cuh"
#include "gemm_tng.cuh"
#include "gemm_nt4.cuh"
#include "grn_group.cuh"

static bool gemm_fast_ok(int dt, int pro, int epi, const GemmP& a);
static int launch_gemm_fast(int epi, GemmP a, hipStream_t st);
static bool wgrad_fast_ok(int dt, int ppro, int qpro, const WgradP& a);
static int launch_wgrad_fast(WgradP a, hipStream_t st, bool qgrn = false);

#define S_(s) reinterpret_cast<hipStream_t>(s)

// ------------------------------------------------------------------------------------------
// Launch programs. Every kernel launch of this library goes through LAUNCH(): normally it is
// issued at once; while a program is being recorded on this thread (mpmae_program_begin_op) the
// fully-resolved launch (kernel, grid, block, LDS, by-value arguments) is appended to the program
// instead. mpmae_program_run() replays a recorded step from C with one HIP stream per lane and
// event ordering between lanes — no Python, no ctypes marshalling, no graph instantiation.
// ------------------------------------------------------------------------------------------
#include <functional>
#include <vector>
struct ProgOp {
  int lane = 0, signal = 0;
  std::vector<int> waits;
  std::vector<std::function<void(hipStream_t)>> launches;
};
struct MpmaeProgram {
  std::vector<ProgOp> ops;
  std::vector<hipStream_t> side;          // lanes 1..n
  std::vector<hipEvent_t> events;         // by signal id
  std::vector<unsigned> epoch;            // run in which events[id] was last recorded
  std::vector<hipEvent_t> join;
  hipEvent_t fork = nullptr;
  unsigned run = 0;
  int nlanes = 1;
  std::vector<int> sig_op, sig_lane;      // by signal id: index / lane of the op that records it (program_end)
  std::vector<char> waited;               // by signal id: some op of another lane waits for it
  std::vector<hipStream_t> lanes_checked_for;   // main streams the side lanes were probed against (see lanes_overlap_check), at most 8
};
static thread_local MpmaeProgram* g_rec = nullptr;

template <typename F>
static inline void submit(hipStream_t st, F&& f) {
  if (g_rec) g_rec->ops.back().launches.emplace_back(std::forward<F>(f));
  else f(st);
}
// Launch status: hipGetLastError() is a per-thread STICKY value that any earlier runtime call of the process may have set (PyTorch
// probes pointers / pinned memory and leaves hipErrorInvalidValue behind: seen as flaky "launch failed" returns of whichever entry
// point ran next). Every launch therefore clears the stale value first and folds ITS OWN status into a thread-local accumulator that
// the entry point returns and resets (launch_status()).
static thread_local int g_launch_err = 0;
static inline int launch_status() { const int e = g_launch_err; g_launch_err = 0; return e; }
#define LAUNCH(kern, g, b, lds, st, ...) \
  submit(st, [=](hipStream_t st__) { (void)hipGetLastError(); hipLaunchKernelGGL(kern, g, b, lds, st__, __VA_ARGS__); \
                                     const hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess && !g_launch_err) g_launch_err = (int)e__; })
#define RET() return launch_status()

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int grid1d(long long total, int per_block = 256, int cap = 16384) {
  long long g = (total + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// second stage of the two-stage reductions (see reduce_partials_kernel in misc.cuh)
static void launch_reduce(int mode, const float* part, int P, int W, float* out, float* out2, int a, int b, int c, int d,
                          hipStream_t st) {
  int R = P / 16;                 // >= 4 rows per thread (4 row lanes per block)
  if (R < 1) R = 1;
  if (R > 32) R = 32;
  dim3 g(cdiv(W, 64), R);
  if (mode == 0) LAUNCH(reduce_partials_kernel<0>, g, dim3(256), 0, st, part, P, W, out, out2, a, b, c, d);
  else if (mode == 1) LAUNCH(reduce_partials_kernel<1>, g, dim3(256), 0, st, part, P, W, out, out2, a, b, c, d);
  else if (mode == 3) LAUNCH(reduce_partials_kernel<3>, g, dim3(256), 0, st, part, P, W, out, out2, a, b, c, d);
  else LAUNCH(reduce_partials_kernel<2>, g, dim3(256), 0, st, part, P, W, out, out2, a, b, c, d);
}

// ------------------------------------------------------------------------------------------
// Library options (mpmae_set_option): explicit, process-wide A/B switches of kernel selection. They replace environment
// variables read inside the library; defaults are the measured-best choices.
// ------------------------------------------------------------------------------------------
static int g_opt[MPMAE_OPT_COUNT_] = {
    /* MPMAE_OPT_LNB_BLOCKS */ 512,
    /* MPMAE_OPT_DW_NT8 */ 512,
    /* MPMAE_OPT_DW6_T8 */ 320,
    /* MPMAE_OPT_DW6_T4 */ 320,
    /* MPMAE_OPT_DW6_T2 */ 320,
    /* MPMAE_OPT_DW6_GC */ 1,
    /* MPMAE_OPT_DW */ 8,
    /* MPMAE_OPT_DWW_S1_NB */ 0,
    /* MPMAE_OPT_DWW_NB */ 128,
    /* MPMAE_OPT_DWW */ 7,
    /* MPMAE_OPT_NT_GLDS64 */ 1,
    /* MPMAE_OPT_NT_BK32 */ 1,
    /* MPMAE_OPT_NT_GLDS */ 1,
    /* MPMAE_OPT_TN */ 2,
    /* MPMAE_OPT_TN_BLOCKS */ 512,
    /* MPMAE_OPT_TN_MINROWS */ 256,
    /* MPMAE_OPT_TN_BLOCKS_BIG */ 512,
    /* MPMAE_OPT_CS_SPLIT */ 1,
    /* MPMAE_OPT_RSC_BLOCKS */ 1536,
    /* MPMAE_OPT_RSC_PF */ 1,
    /* MPMAE_OPT_RSC_NC32 */ 1,
    /* MPMAE_OPT_RSC_SMALL */ 1,
    /* MPMAE_OPT_RSC_N40 */ 2,
    /* MPMAE_OPT_RSC_N80 */ 1,
    /* MPMAE_OPT_STB_BLOCKS */ 512,
    /* MPMAE_OPT_TN3_BLOCKS */ 128,
    /* MPMAE_OPT_TNG_BLOCKS */ 512,
    /* MPMAE_OPT_NT4 */ 1,
    /* MPMAE_OPT_FOLD_GROUP */ 0,
    /* MPMAE_OPT_RSC_W5 */ 1,
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
static int launch_gemm(int pro, int epi, const GemmP& a, hipStream_t st) {
  dim3 g(cdiv(a.M, GBM), cdiv(a.N, GBN)), b(256);
#define GEMM_CASE(P, E)                                                            \
  if (pro == P && epi == E) {                                                      \
    LAUNCH((gemm_kernel<T, P, E>), g, b, 0, st, a);                    \
    return launch_status();                                                 \
  }
  GEMM_CASE(PRO_NONE, EPI_STORE)
  GEMM_CASE(PRO_NONE, EPI_RESID)
  GEMM_CASE(PRO_NONE, EPI_GELU_SUMSQ)
  GEMM_CASE(PRO_NONE, EPI_SCATTER_ROWS)
  GEMM_CASE(PRO_NONE, EPI_DZ_STATS)
  GEMM_CASE(PRO_NONE, EPI_DOWN_DGRAD)
  GEMM_CASE(PRO_LN_AFFINE, EPI_GELU_SUMSQ)
  GEMM_CASE(PRO_LN_AFFINE, EPI_STORE)
  GEMM_CASE(PRO_GRN, EPI_RESID)
  GEMM_CASE(PRO_GRN_BWD, EPI_STORE)
  GEMM_CASE(PRO_DOWN_GATHER, EPI_STORE)
  GEMM_CASE(PRO_ROW_GATHER, EPI_STORE)
  GEMM_CASE(PRO_IM2COL3, EPI_STORE)
#undef GEMM_CASE
  return (int)hipErrorInvalidValue;
}

int mpmae_gemm(int dt, int pro, int epi, const MpmaeGemmArgs* args, mpmae_stream_t s) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return (int)hipErrorInvalidValue;
  if ((epi == EPI_GELU_SUMSQ || epi == EPI_DZ_STATS) && args->rpg < args->M && args->rpg < 43)
    return (int)hipErrorInvalidValue;   // a 128-row tile may span at most GMAXG statistics groups
  if (gemm_fast_ok(dt, pro, epi, *args)) return launch_gemm_fast(epi, *args, S_(s));
  const bool stats = (epi == EPI_GELU_SUMSQ || epi == EPI_DZ_STATS);
  const bool single = stats && args->rpg >= args->M;
  const int mblocks = cdiv(args->M, GBM);
  if (single) {
    const size_t need = (size_t)mblocks * args->N * (epi == EPI_DZ_STATS ? 2 : 1);
    if (!args->ws || args->ws_floats < need) return (int)hipErrorInvalidValue;
  }
  int err = dt == 0 ? launch_gemm<float>(pro, epi, *args, S_(s)) : launch_gemm<bf16_t>(pro, epi, *args, S_(s));
  if (err == 0 && single) {
    launch_reduce(0, args->ws, mblocks, args->N, args->s0, nullptr, 0, 0, 0, 0, S_(s));
    if (epi == EPI_DZ_STATS)
      launch_reduce(0, args->ws + (size_t)mblocks * args->N, mblocks, args->N, args->s1, nullptr, 0, 0, 0, 0, S_(s));
    err = launch_status();
  }
  return err;
}

template <typename T>
static int launch_wgrad(int ppro, int qpro, const WgradP& a, int splits, hipStream_t st) {
  dim3 g(cdiv(a.Nn, WBN), cdiv(a.Kk, WBK), splits), b(256);
#define WG_CASE(P, Q)                                                              \
  if (ppro == P && qpro == Q) {                                                    \
    LAUNCH((wgrad_kernel<T, P, Q>), g, b, 0, st, a);                   \
    return launch_status();                                                 \
  }
  WG_CASE(PRO_NONE, PRO_NONE)
  WG_CASE(PRO_NONE, PRO_GRN)
  WG_CASE(PRO_NONE, PRO_LN_AFFINE)
  WG_CASE(PRO_GRN_BWD, PRO_LN_AFFINE)
  WG_CASE(PRO_NONE, PRO_DOWN_GATHER)
  WG_CASE(PRO_ROW_GATHER, PRO_NONE)
  WG_CASE(PRO_NONE, PRO_IM2COL3)
#undef WG_CASE
  return (int)hipErrorInvalidValue;
}

int mpmae_wgrad(int dt, int ppro, int qpro, const MpmaeWgradArgs* args, int splits, mpmae_stream_t s) {
  if (!args || splits < 1) return (int)hipErrorInvalidValue;
  if (wgrad_fast_ok(dt, ppro, qpro, *args)) return launch_wgrad_fast(*args, S_(s), qpro == PRO_GRN);
  WgradP a = *args;
  const size_t per = (size_t)a.Nn * a.Kk + a.Nn;
  if (!a.ws || a.ws_floats < per) return (int)hipErrorInvalidValue;
  const int maxs = (int)(a.ws_floats / per);
  if (splits > maxs) splits = maxs;
  int rps = cdiv(a.M, splits);
  rps = cdiv(rps, WBM) * WBM;
  a.rows_per_split = rps;
  splits = cdiv(a.M, rps);
  int err = dt == 0 ? launch_wgrad<float>(ppro, qpro, a, splits, S_(s)) : launch_wgrad<bf16_t>(ppro, qpro, a, splits, S_(s));
  if (err) return err;
  launch_reduce(1, a.ws, splits, a.Nn * a.Kk, a.dW, nullptr, a.Kk, a.sn, a.sk, 0, S_(s));
  if (a.db) launch_reduce(0, a.ws + (size_t)splits * a.Nn * a.Kk, splits, a.Nn, a.db, nullptr, 0, 0, 0, 0, S_(s));
  return launch_status();
}

// Grouped weight gradients (gemm_tng.cuh): count problems of one shape, one launch + one fold.
template <int RX, int RY, int NST>
static int launch_tng(const TngP& g, int blocks, hipStream_t st) {
  using Cf = TngCfg<RX, RY, NST>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_tng_kernel<RX, RY, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS) != hipSuccess)
      return (int)hipGetLastError();
    attr = true;
  }
  LAUNCH((gemm_tng_kernel<RX, RY, NST>), dim3(blocks), dim3(256), Cf::LDS, st, g);
  return 0;
}

template <int RX, int RY, int NST>
static int launch_tng48(const TngP& g, int blocks, hipStream_t st) {
  using Cf = Tng48Cfg<RX, RY, NST>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_tng48_kernel<RX, RY, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS) != hipSuccess)
      return (int)hipGetLastError();
    attr = true;
  }
  LAUNCH((gemm_tng48_kernel<RX, RY, NST>), dim3(blocks), dim3(256), Cf::LDS, st, g);
  return 0;
}

static bool tng_ok(int dt, const WgradP* pr, int count, int* WX_, int* WY_) {
  if (dt != 1 || count < 1 || count > TNG_MAXP || g_opt[MPMAE_OPT_TNG_BLOCKS] <= 0) return false;
  const int WX = pr[0].Nn < pr[0].Kk ? pr[0].Nn : pr[0].Kk, WY = pr[0].Nn < pr[0].Kk ? pr[0].Kk : pr[0].Nn;
  // 80-column regions (atto / nano widths) or 48-column regions (tiny / large widths, femto from stage 1)
  if (!((WX == 80 && WY % 320 == 0) || (WX % 160 == 0 && WY % 160 == 0) || (WX == 96 && WY % 384 == 0) || (WX % 192 == 0 && WY % 192 == 0)))
    return false;
  if (((size_t)WX * WY + WY) % 4 || ((size_t)WX * WY + WX) % 4) return false;
  for (int i = 0; i < count; ++i) {
    const WgradP& a = pr[i];
    const int wx = a.Nn < a.Kk ? a.Nn : a.Kk, wy = a.Nn < a.Kk ? a.Kk : a.Nn;
    if (wx != WX || wy != WY || a.M != pr[0].M || a.M < 1) return false;
    if (a.P2 || a.pp0 || a.pp1 || a.qp0 || a.qp1 || !a.P || !a.Q || !a.dW) return false;
    if (a.sn < 1 || a.sk < 1) return false;
    if ((a.ldp | a.ldq) & 7) return false;
    if (((uintptr_t)a.P | (uintptr_t)a.Q) & 15) return false;
    if (a.ldp < a.Nn || a.ldq < a.Kk) return false;
  }
  *WX_ = WX; *WY_ = WY;
  return true;
}

int mpmae_wgrad_group(int dt, const MpmaeWgradArgs* probs, int count, float* ws, size_t ws_floats, mpmae_stream_t s) {
  if (!probs || count < 1 || !ws) return (int)hipErrorInvalidValue;
  int WX = 0, WY = 0;
  if (!tng_ok(dt, probs, count, &WX, &WY)) {            // one call per problem (each with the scratch given here)
    for (int i = 0; i < count; ++i) {
      MpmaeWgradArgs a = probs[i];
      a.ws = ws; a.ws_floats = ws_floats;
      const int tiles = cdiv(a.Nn, 64) * cdiv(a.Kk, 64);
      int splits = cdiv(768, tiles);
      if (splits > cdiv(a.M, 256)) splits = cdiv(a.M, 256);
      const int err = mpmae_wgrad(dt, PRO_NONE, PRO_NONE, &a, splits < 1 ? 1 : splits, s);
      if (err) return err;
    }
    return 0;
  }
  const int M = probs[0].M;
  const bool r48 = WX % 80 != 0;                        // 48-column regions, a wave owns 2 x 2 of them (gemm_tng48_kernel)
  const bool narrow = WX == 80 || WX == 96;             // (1, 4) waves: 80 x 320 / 96 x 384 tile; else (2, 2): 160 x 160 / 192 x 192
  const int tx = r48 ? (narrow ? 96 : 192) : (narrow ? 80 : 160), ty = r48 ? (narrow ? 384 : 192) : (narrow ? 320 : 160);
  const int xt = WX / tx, yt = WY / ty;
  const size_t per_max = (size_t)WX * WY + WY;          // slab stride: the larger of the two bias lengths
  int splits = g_opt[MPMAE_OPT_TNG_BLOCKS] / (count * xt * yt);
  const int maxs = M / (8 * TNG_SL);                    // >= 8 k-steps per split
  if (splits > maxs) splits = maxs;
  if (splits >= 8) splits -= splits % 8;                // one row range per XCD
  if ((size_t)splits * count * per_max > ws_floats) splits = (int)(ws_floats / ((size_t)count * per_max));
  if (splits < 1) splits = 1;
  if ((size_t)count * per_max > ws_floats) return (int)hipErrorInvalidValue;
  const int rps = cdiv(cdiv(M, splits), TNG_SL) * TNG_SL;
  splits = cdiv(M, rps);
  TngP g;
  TngFoldP f;
  g.nprob = count; g.M = M; g.WX = WX; g.WY = WY; g.rps = rps; g.splits = splits; g.xt = xt; g.yt = yt;
  f.nprob = count; f.splits = splits;
  size_t maxper = 0;
  for (int i = 0; i < count; ++i) {
    const WgradP& a = probs[i];
    const bool swap = a.Kk < a.Nn;                      // X = Q (pwconv1: P = dh is the wide operand)
    TngProb& p = g.p[i];
    p.X = reinterpret_cast<const bf16_t*>(swap ? a.Q : a.P);
    p.Y = reinterpret_cast<const bf16_t*>(swap ? a.P : a.Q);
    p.ldx = swap ? a.ldq : a.ldp; p.ldy = swap ? a.ldp : a.ldq;
    p.slab = ws + (size_t)i * splits * per_max;
    p.swap = swap ? 1 : 0; p.want_db = a.db ? 1 : 0;
    const size_t per = (size_t)a.Nn * a.Kk + a.Nn;
    f.p[i].slab = p.slab; f.p[i].dW = a.dW; f.p[i].db = a.db; f.p[i].nk = a.Nn * a.Kk; f.p[i].per = (int)per; f.p[i].Kk = a.Kk; f.p[i].sn = a.sn; f.p[i].sk = a.sk;
    if (per > maxper) maxper = per;
  }
  const int blocks = count * xt * yt * splits;
  int err;
  if (r48) err = narrow ? launch_tng48<1, 4, 3>(g, blocks, S_(s)) : launch_tng48<2, 2, 3>(g, blocks, S_(s));
  else if (narrow) err = launch_tng<1, 4, 3>(g, blocks, S_(s));
  else err = launch_tng<2, 2, 3>(g, blocks, S_(s));
  if (err) return err;
  int fb = cdiv((long long)(maxper / 4), 256);
  if (fb > 256) fb = 256;
  LAUNCH(wgrad_group_fold_kernel, dim3(fb, count), dim3(256), 0, S_(s), f);
  RET();
}

// ------------------------------------------------------------------------------------------
template <typename T>
static void launch_ln_fwd(const void* x, void* xhat, float* rstd, void* y, const float* gamma, const float* beta,
                          int act, float eps, int M, int C, const uint8_t* rowmask, hipStream_t st) {
  const int blocks = grid1d((long long)M * 64, 256, 8192);
  LAUNCH(ln_fwd_kernel<T>, dim3(blocks), dim3(256), 0, st, (const T*)x, (T*)xhat, rstd, (T*)y, gamma, beta,
                     act, eps, M, C, rowmask);
}

static int ln_fwd_impl(int dt, const void* x, void* xhat, float* rstd, void* y, const float* gamma, const float* beta, int act,
                       float eps, int M, int C, const uint8_t* rowmask, int down_S, mpmae_stream_t s) {
  if (C > 64 * LN_MAXPER) return (int)hipErrorInvalidValue;
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
  if (C > 64 * LN_MAXPER) return (int)hipErrorInvalidValue;
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
    lncap = g_opt[MPMAE_OPT_LNB_BLOCKS];   // measured: 512 -> 50 us, 1024 -> 36 us, 2048 -> 40 us (slab reduce grows)
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

// ------------------------------------------------------------------------------------------
static size_t dw_lds_bytes(int CC, bool wgrad) {
  size_t b = (size_t)DW_HP * CC * sizeof(float) + (DW_HP + 4) * sizeof(int);
  if (wgrad) b += (size_t)50 * CC * sizeof(float);
  return b;
}

template <typename T>
static void launch_dw_v2(const MpmaeDwArgs& a, hipStream_t st) {
  const bool c40 = (a.C % 40 == 0);
  const int cc = c40 ? 40 : 32;
  dim3 g(a.g.N * a.tiles_side * a.tiles_side, cdiv(a.C, cc));
  if (c40) LAUNCH((dwconv7_v2_kernel<T, 40>), g, dim3(320), 0, st, a);
  else LAUNCH((dwconv7_v2_kernel<T, 32>), g, dim3(256), 0, st, a);
}

template <typename T>
static void launch_dwwg_v2(const MpmaeDwWgArgs& a, int nblocks, hipStream_t st) {
  const bool c40 = (a.C % 40 == 0);
  const int cc = c40 ? 40 : 32;
  dim3 g(nblocks, cdiv(a.C, cc));
  if (c40) LAUNCH((dwconv7_wgrad_v2_kernel<T, 40>), g, dim3(320), 0, st, a);
  else LAUNCH((dwconv7_wgrad_v2_kernel<T, 32>), g, dim3(256), 0, st, a);
}

template <typename T, int S>
static void launch_dw_v4(const MpmaeDwArgs& a, hipStream_t st) {
  using D = Dw4<S>;
  const int chunks = a.C / D::CW;
  int nw = 1;
  for (int d = 8; d >= 1; --d) if (chunks % d == 0) { nw = d; break; }
  const size_t lds = ((D::HP + 3) & ~3) * sizeof(int) +
                     (size_t)nw * (D::HP * D::CW * sizeof(T) + (S > 1 ? 49 * D::CW * sizeof(float) : 0));
  dim3 g(a.g.N * a.g.keep, chunks / nw);
  LAUNCH((dwconv7_v4_kernel<T, S>), g, dim3(64 * nw), lds, st, a);
}

template <typename T, int S>
static void launch_dwwg_v4(const MpmaeDwWgArgs& a, int nblocks, hipStream_t st) {
  using D = Dw4<S>;
  dim3 g(nblocks, a.C / D::CW);
  LAUNCH((dwconv7_wgrad_v4_kernel<T, S>), g, dim3(64), 0, st, a);
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
  nt8 = g_opt[MPMAE_OPT_DW_NT8];
  // S = 8: the 62x62 map takes 61 KB, so two workgroups per CU; 8 waves each keep 4 waves per SIMD busy
  LAUNCH((dwconv7_v5_kernel<T, S>), g, dim3(S == 8 ? nt8 : 256), lds, st, a);
  return true;
}

template <typename T, int S>
static bool launch_dwwg_v5(const MpmaeDwWgArgs& a, int nblocks, hipStream_t st, const DwWgGroupP* grp = nullptr) {
  constexpr int CW = 64 / S;
  size_t lds = dw5_map_bytes<T, S>(a.g.grid);
  int nt8;
  nt8 = g_opt[MPMAE_OPT_DW_NT8];
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
  t8 = g_opt[MPMAE_OPT_DW6_T8];
  t4 = g_opt[MPMAE_OPT_DW6_T4];
  t2 = g_opt[MPMAE_OPT_DW6_T2];
  return S == 8 ? t8 : S == 4 ? t4 : t2;
}

template <int S>
static bool launch_dw_v6(const MpmaeDwArgs& a, hipStream_t st) {
  constexpr int CW = 64 / S;
  const size_t lds = dw5_map_bytes<bf16_t, S>(a.g.grid) + 49 * CW * sizeof(float);
  if (lds > 64 * 1024) return false;
  dim3 g(a.g.N, a.C / CW);
  int gc;
  gc = g_opt[MPMAE_OPT_DW6_GC];
  // compile-time map pitch: measured faster only at S = 2 for the forward / data-gradient kernel (17.7 vs 18.8 us; slower at
  // S = 4, 8), decisive for the weight-gradient kernel (220 -> 128 VGPRs)
  if (gc && S == 2 && a.g.grid == 7) LAUNCH((dwconv7_v6_kernel<S, 7>), g, dim3(dw6_threads(S)), lds, st, a);
  else LAUNCH((dwconv7_v6_kernel<S>), g, dim3(dw6_threads(S)), lds, st, a);
  return true;
}

template <int S>
static bool launch_dwwg_v6(const MpmaeDwWgArgs& a, int nblocks, hipStream_t st) {
  constexpr int CW = 64 / S;
  const int nthreads = dw6_threads(S);
  size_t lds = dw5_map_bytes<bf16_t, S>(a.g.grid);
  const size_t red = (size_t)(nthreads / 64) * 50 * CW * sizeof(float);
  if (red > lds) lds = red;
  if (lds > 64 * 1024) return false;
  dim3 g(nblocks, a.C / CW);
  LAUNCH((dwconv7_wgrad_v6_kernel<S>), g, dim3(nthreads), lds, st, a);
  return true;
}

static int dw_variant() {      // MPMAE_DW=4 forces the per-patch kernels (A/B measurements)
  int v;
  v = g_opt[MPMAE_OPT_DW];
  return v;
}

static bool dw_v4_ok(int C, int S) {
  // per-visible-patch tiles pay for S >= 2; at S == 1 (stage 3, dense decoder) every output would drag
  // its own 49-point halo, so the positional 8x8 tiles of dwconv3.cuh are used there
  if (S != 8 && S != 4 && S != 2) return false;
  return C % (64 / S) == 0;
}

template <int S, int C, int BR>
static int launch_dw_band(const DwP& a, hipStream_t st) {
  using D = DwBand<S, C, BR>;
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute((const void*)dwconv7_band_kernel<S, C, BR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)D::LDS) != hipSuccess)
      return (int)hipGetLastError();      // (launch_status() only reports LAUNCH errors: it would return success without a launch)
    once = true;
  }
  LAUNCH((dwconv7_band_kernel<S, C, BR>), dim3(a.g.N, cdiv(a.g.grid, BR)), dim3(512), D::LDS, st, a);
  return launch_status();
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
  if (dt == 1 && dw_variant() >= 8) {
    const int r = try_dw_mfma(*a, S_(s));
    if (r >= 0) return r;
  }
  if (dt == 1 && dw_variant() >= 7 && a->g.grid == 7 && a->g.inv && (((uintptr_t)a->x | (uintptr_t)a->out | (uintptr_t)a->add) & 15) == 0) {
    // band kernel (dwband.cuh): (sample, patch row) per workgroup, all channels, whole-line loads. Measured at bs 256: stage 0
    // (S = 8, C = 40) 47 / 60 us forward / data gradient against 60 / 67 us for the per-sample kernels; at stage 1 (S = 4, C = 80)
    // it LOSES (36 / 40 vs 25 / 27 us) and with two patch rows per workgroup (one workgroup per CU) it loses everywhere
    if (a->g.S == 8 && a->C == 40) return launch_dw_band<8, 40, 1>(*a, S_(s));
  }
  if (dt == 1 && a->g.S == 1 && a->g.grid == 7 && (a->C & 15) == 0 && dw_variant() >= 6 &&
      (((uintptr_t)a->x) & 15) == 0 && (((uintptr_t)a->out | (uintptr_t)a->add) & 3) == 0) {
    dim3 g(a->g.N, cdiv(a->C, 64));
    LAUNCH((dwconv7_v6s1_kernel<7>), g, dim3(256), 0, S_(s), *a);
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
  if (dw_v4_ok(a->C, a->g.S)) {
#define DW4(TT) do { switch (a->g.S) { case 8: launch_dw_v4<TT, 8>(*a, S_(s)); break; case 4: launch_dw_v4<TT, 4>(*a, S_(s)); break; \
                                      case 2: launch_dw_v4<TT, 2>(*a, S_(s)); break; default: launch_dw_v4<TT, 1>(*a, S_(s)); } } while (0)
    if (dt == 0) DW4(float); else DW4(bf16_t);
#undef DW4
    RET();
  }
  if ((a->C & 7) == 0 && a->g.grid * a->g.grid <= W64_MAXL) {
    const int chunks = a->C / 8;
    const int nw = chunks <= 8 ? chunks : (chunks % 5 == 0 ? 5 : 8);      // waves per block sharing the tables
    const size_t esz = dt == 0 ? 4 : 2;
    const size_t lds = (DW_HP + W64_MAXL) * sizeof(int) + (size_t)nw * (DW_HP * 8 * esz + 49 * 8 * sizeof(float));
    dim3 g(a->g.N * a->tiles_side * a->tiles_side, cdiv(chunks, nw));
    if (dt == 0) LAUNCH(dwconv7_w64_kernel<float>, g, dim3(64 * nw), lds, S_(s), *a);
    else LAUNCH(dwconv7_w64_kernel<bf16_t>, g, dim3(64 * nw), lds, S_(s), *a);
    RET();
  }
  if ((a->C & 7) == 0) {
    if (dt == 0) launch_dw_v2<float>(*a, S_(s)); else launch_dw_v2<bf16_t>(*a, S_(s));
    RET();
  }
  const size_t lds = dw_lds_bytes(a->CC, false);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  dim3 g(a->g.N * a->tiles_side * a->tiles_side, cdiv(a->C, a->CC));
  if (dt == 0) {
    { static size_t cur = 0; if (lds > cur) { if (hipFuncSetAttribute((const void*)dwconv7_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } }
    LAUNCH(dwconv7_fwd_kernel<float>, g, dim3(256), lds, S_(s), *a);
  } else {
    { static size_t cur = 0; if (lds > cur) { if (hipFuncSetAttribute((const void*)dwconv7_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } }
    LAUNCH(dwconv7_fwd_kernel<bf16_t>, g, dim3(256), lds, S_(s), *a);
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
    nbs1 = g_opt[MPMAE_OPT_DWW_S1_NB];
    const int want = nbs1 > 0 ? nbs1 : (cdiv(a->C, 64) <= 5 ? 128 : 64);      // ~512-640 workgroups in total (measured)
    int nb = a->g.N < want ? a->g.N : want;
    if ((size_t)nb * per > a->ws_floats) nb = (int)(a->ws_floats / per);
    dim3 g(nb, cdiv(a->C, 64));
    DwWgGroupP gr;
    gr.count = 0;
    LAUNCH((dwconv7_wgrad_v6s1_kernel<7>), g, dim3(256), 0, S_(s), *a, gr);
    launch_reduce(2, a->ws, nb, 50 * a->C, a->dw, a->db, a->C, a->s_kh, a->s_kw, a->s_c, S_(s));
    RET();
  }
  if (dw_v4_ok(a->C, a->g.S) && dw_variant() >= 5) {
    const size_t per = (size_t)50 * a->C;
    if (!a->ws || a->ws_floats < per) return (int)hipErrorInvalidValue;
    int nbmax;
    nbmax = g_opt[MPMAE_OPT_DWW_NB];
    int nb = a->g.N < nbmax ? a->g.N : nbmax;
    if ((size_t)nb * per > a->ws_floats) nb = (int)(a->ws_floats / per);
    bool ok = false;
    // the packed weight-gradient kernel needs 98 accumulator VGPRs per lane and measured slower than v5
    // (62 vs 45 us at stage 1); it stays available for experiments (MPMAE_DWW=6)
    int wg6;
    wg6 = g_opt[MPMAE_OPT_DWW] == 6 ? 1 : 0;
    if (wg6 && dt == 1 && (a->C & 1) == 0 && (((uintptr_t)a->x | (uintptr_t)a->dd) & 3) == 0) {
      switch (a->g.S) { case 8: ok = launch_dwwg_v6<8>(*a, nb, S_(s)); break; case 4: ok = launch_dwwg_v6<4>(*a, nb, S_(s)); break;
                        default: ok = launch_dwwg_v6<2>(*a, nb, S_(s)); }
    }
    if (!ok) {
#define DWW5(TT) do { switch (a->g.S) { case 8: ok = launch_dwwg_v5<TT, 8>(*a, nb, S_(s)); break; case 4: ok = launch_dwwg_v5<TT, 4>(*a, nb, S_(s)); break; \
                                       default: ok = launch_dwwg_v5<TT, 2>(*a, nb, S_(s)); } } while (0)
    if (dt == 0) DWW5(float); else DWW5(bf16_t);
#undef DWW5
    }
    if (ok) {
      launch_reduce(2, a->ws, nb, 50 * a->C, a->dw, a->db, a->C, a->s_kh, a->s_kw, a->s_c, S_(s));
      RET();
    }
  }
  if (dw_v4_ok(a->C, a->g.S)) {
    const size_t per = (size_t)50 * a->C;
    if (!a->ws || a->ws_floats < per) return (int)hipErrorInvalidValue;
    const int npatches = a->g.N * a->g.keep;
    if (nblocks > npatches) nblocks = npatches;
    if ((size_t)nblocks * per > a->ws_floats) nblocks = (int)(a->ws_floats / per);
#define DWW4(TT) do { switch (a->g.S) { case 8: launch_dwwg_v4<TT, 8>(*a, nblocks, S_(s)); break; case 4: launch_dwwg_v4<TT, 4>(*a, nblocks, S_(s)); break; \
                                       case 2: launch_dwwg_v4<TT, 2>(*a, nblocks, S_(s)); break; default: launch_dwwg_v4<TT, 1>(*a, nblocks, S_(s)); } } while (0)
    if (dt == 0) DWW4(float); else DWW4(bf16_t);
#undef DWW4
    launch_reduce(2, a->ws, nblocks, 50 * a->C, a->dw, a->db, a->C, a->s_kh, a->s_kw, a->s_c, S_(s));
    RET();
  }
  if ((a->C & 7) == 0 && a->g.grid * a->g.grid <= W64_MAXL) {
    const size_t per = (size_t)50 * a->C;
    if (!a->ws || a->ws_floats < per) return (int)hipErrorInvalidValue;
    if (nblocks > a->ntiles_total) nblocks = a->ntiles_total;
    if ((size_t)nblocks * per > a->ws_floats) nblocks = (int)(a->ws_floats / per);
    dim3 g(nblocks, a->C / 8);
    if (dt == 0) LAUNCH(dwconv7_wgrad_w64_kernel<float>, g, dim3(64), 0, S_(s), *a);
    else LAUNCH(dwconv7_wgrad_w64_kernel<bf16_t>, g, dim3(64), 0, S_(s), *a);
    launch_reduce(2, a->ws, nblocks, 50 * a->C, a->dw, a->db, a->C, a->s_kh, a->s_kw, a->s_c, S_(s));
    RET();
  }
  if ((a->C & 7) == 0) {
    const size_t per = (size_t)50 * a->C;
    if (!a->ws || a->ws_floats < per) return (int)hipErrorInvalidValue;
    if (nblocks > a->ntiles_total) nblocks = a->ntiles_total;
    if ((size_t)nblocks * per > a->ws_floats) nblocks = (int)(a->ws_floats / per);
    if (dt == 0) launch_dwwg_v2<float>(*a, nblocks, S_(s)); else launch_dwwg_v2<bf16_t>(*a, nblocks, S_(s));
    launch_reduce(2, a->ws, nblocks, 50 * a->C, a->dw, a->db, a->C, a->s_kh, a->s_kw, a->s_c, S_(s));
    RET();
  }
  const size_t lds = dw_lds_bytes(a->CC, true);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (nblocks > a->ntiles_total) nblocks = a->ntiles_total;
  dim3 g(nblocks, cdiv(a->C, a->CC));
  if (dt == 0) {
    { static size_t cur = 0; if (lds > cur) { if (hipFuncSetAttribute((const void*)dwconv7_wgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } }
    LAUNCH(dwconv7_wgrad_kernel<float>, g, dim3(256), lds, S_(s), *a);
  } else {
    { static size_t cur = 0; if (lds > cur) { if (hipFuncSetAttribute((const void*)dwconv7_wgrad_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } }
    LAUNCH(dwconv7_wgrad_kernel<bf16_t>, g, dim3(256), lds, S_(s), *a);
  }
  RET();
}

// All depthwise weight gradients of a stage in one launch (grid.z = problem) + one fold: identical geometry / width / tap strides,
// bf16, the per-sample LDS-map kernels (v5: S >= 2, v6s1: S = 1 on the 7 x 7 grid). Anything else: one mpmae_dwconv7_wgrad each.
int mpmae_dwconv7_wgrad_group(int dt, const MpmaeDwWgArgs* probs, int count, float* ws, size_t ws_floats, mpmae_stream_t s) {
  if (!probs || count < 1 || !ws) return (int)hipErrorInvalidValue;
  const MpmaeDwWgArgs& a0 = probs[0];
  bool ok = dt == 1 && count <= DWG_MAX && dw_variant() >= 6 && g_opt[MPMAE_OPT_DWW] != 6 && a0.CC >= 1 && a0.TP * a0.g.S <= 8;
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
    const int want = g_opt[MPMAE_OPT_DWW_S1_NB] > 0 ? g_opt[MPMAE_OPT_DWW_S1_NB] : (cdiv(a0.C, 64) <= 5 ? 128 : 64);
    nb = a0.g.N < want ? a0.g.N : want;
  } else if (v5) {
    nb = a0.g.N < g_opt[MPMAE_OPT_DWW_NB] ? a0.g.N : g_opt[MPMAE_OPT_DWW_NB];
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
  dim3 g(bwd ? npatches : npatches / a->L), b(bwd ? 256 : 512) ;   // forward: one 16-wave block per sample
  if (dt == 0) { if (bwd) LAUNCH((loss_pix_cont_kernel<float, true>), g, b, 0, S_(s), *a);
                 else LAUNCH((loss_pix_cont_kernel<float, false>), g, b, 0, S_(s), *a); }
  else { if (bwd) LAUNCH((loss_pix_cont_kernel<bf16_t, true>), g, b, 0, S_(s), *a);
         else LAUNCH((loss_pix_cont_kernel<bf16_t, false>), g, b, 0, S_(s), *a); }
  RET();
}

int mpmae_loss_pix_cat(int dt, int bwd, const MpmaePixCatArgs* a, int npatches, mpmae_stream_t s) {
  if (a->K > 16) return (int)hipErrorInvalidValue;
  dim3 g(bwd ? npatches : npatches / a->L), b(bwd ? 256 : 1024);   // forward: one 16-wave block per sample
  if (dt == 0) { if (bwd) LAUNCH((loss_pix_cat_kernel<float, true>), g, b, 0, S_(s), *a);
                 else LAUNCH((loss_pix_cat_kernel<float, false>), g, b, 0, S_(s), *a); }
  else { if (bwd) LAUNCH((loss_pix_cat_kernel<bf16_t, true>), g, b, 0, S_(s), *a);
         else LAUNCH((loss_pix_cat_kernel<bf16_t, false>), g, b, 0, S_(s), *a); }
  RET();
}

int mpmae_loss_img(int dt, int bwd, const MpmaeImgArgs* a, mpmae_stream_t s) {
  dim3 g(a->N), b(256);
  if (dt == 0) { if (bwd) LAUNCH((loss_img_kernel<float, true>), g, b, 0, S_(s), *a);
                 else LAUNCH((loss_img_kernel<float, false>), g, b, 0, S_(s), *a); }
  else { if (bwd) LAUNCH((loss_img_kernel<bf16_t, true>), g, b, 0, S_(s), *a);
         else LAUNCH((loss_img_kernel<bf16_t, false>), g, b, 0, S_(s), *a); }
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
  if (!dev_args || count < 1 || N < 1 || maxC < 1 || p < 1 || (H & 3) || ((p * p) & 3)) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)maxC * (p * H + 4) * 4;
  const int nvec = maxC * p * (H / 4), npv = maxC * p * p / 4;
  const int mv = cdiv(nvec, 512), mp = cdiv(npv, 64);
  if (lds > 150 * 1024 || mv > 12 || mp > 12) return (int)hipErrorInvalidValue;
  const auto* tab = (const MpmaePixContArgs*)dev_args;
#define LPR(TT, MV, MP, BW) do { \
    static size_t cur = 48 * 1024; \
    if (lds > cur) { if (hipFuncSetAttribute((const void*)loss_pix_cont_rows_kernel<TT, MV, MP, BW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } \
    LAUNCH((loss_pix_cont_rows_kernel<TT, MV, MP, BW>), dim3(N, count), dim3(512), lds, S_(s), tab); } while (0)
#define LPR2(TT, MV, MP) do { if (bwd) LPR(TT, MV, MP, true); else LPR(TT, MV, MP, false); } while (0)
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

int mpmae_loss_pix_cat_waves(int dt, int bwd, const void* dev_args, int count, int N, int max_pk, mpmae_stream_t s) {
  if (!dev_args || count < 1 || N < 1 || max_pk < 4 || (max_pk & 3)) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)16 * max_pk * (dt == 0 ? 4 : 2);
  if (lds > 150 * 1024) return (int)hipErrorInvalidValue;
  const auto* tab = (const MpmaePixCatArgs*)dev_args;
#define LCW(TT, BW) do { \
    static size_t cur = 48 * 1024; \
    if (lds > cur) { if (hipFuncSetAttribute((const void*)loss_pix_cat_waves_kernel<TT, BW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } \
    LAUNCH((loss_pix_cat_waves_kernel<TT, BW>), dim3(N, count), dim3(1024), lds, S_(s), tab, max_pk); } while (0)
  if (dt == 0) { if (bwd) LCW(float, true); else LCW(float, false); }
  else { if (bwd) LCW(bf16_t, true); else LCW(bf16_t, false); }
#undef LCW
  RET();
}

int mpmae_loss_finalize(const float* acc, int N, const float* log_vars, int T, float loss_scale, float* losses,
                        float* weighted, float* total, float* coef, float* dlog_vars, mpmae_stream_t s) {
  if (T > 16 || T < 1) return (int)hipErrorInvalidValue;
  LAUNCH(loss_finalize_kernel, dim3(1), dim3(64 * T), 0, S_(s), acc, N, log_vars, T, loss_scale, losses, weighted,
                     total, coef, dlog_vars);
  RET();
}

int mpmae_adamw(float* p, const float* g, float* m, float* v, const float* hp, float beta1, float beta2, float eps,
                float wd, size_t n, const uint8_t* decay, float* gnorm2, mpmae_stream_t s) {
  LAUNCH(adamw_kernel, dim3(grid1d((long long)n, 256, 4096)), dim3(256), 0, S_(s), p, g, m, v, hp, beta1, beta2,
                     eps, wd, n, decay, gnorm2);
  RET();
}

int mpmae_sumsq(const float* x, size_t n, float* out, mpmae_stream_t s) {
  LAUNCH(sumsq_kernel, dim3(grid1d((long long)n, 256, 1024)), dim3(256), 0, S_(s), x, n, out);
  RET();
}


// ------------------------------------------------------------------------------------------
// fast bf16 paths (compute-shaped layers) and the element-wise GRN kernels
// ------------------------------------------------------------------------------------------
static bool gemm_fast_ok(int dt, int pro, int epi, const GemmP& a) {
  if (dt != 1 || pro != PRO_NONE) return false;
  const bool stats = (epi == EPI_GELU_SUMSQ || epi == EPI_DZ_STATS);
  if (epi != EPI_STORE && epi != EPI_RESID && !stats) return false;
  const bool grouped = stats && a.rpg > 0 && a.rpg < a.M;
  if (grouped && (a.rpg < 43 || a.M % a.rpg)) return false;    // a 128-row tile may overlap at most 4 groups
  if (stats && (!a.ws || a.ws_floats < (size_t)((a.M + 127) / 128) * a.N * 2 * (grouped ? 4 : 1))) return false;
  if (epi == EPI_DZ_STATS && (a.ldr & 7)) return false;
  if ((a.K | a.N | a.lda | a.ldb | a.ldc) & 7) return false;
  if (epi == EPI_RESID && (a.ldr & 7)) return false;
  return true;
}

static bool glds_bn64() { int v; v = g_opt[MPMAE_OPT_NT_GLDS64]; return v != 0; }

template <int BN>
static int launch_gemm_fast_bn(int epi, const GemmP& a, hipStream_t st) {
  size_t lds = (size_t)(2 * FBM * FLD + 2 * BN * FLD) * sizeof(bf16_t);
  if (epi == EPI_GELU_SUMSQ || epi == EPI_DZ_STATS) {         // epilogue image: fp32 stage | h tile | colacc[4][2][BN] | activity
    const size_t epi_lds = (size_t)128 * 68 * 4 + (size_t)128 * (BN + 8) * 2 + (size_t)8 * BN * 4 + 128;
    if (epi_lds > lds) lds = epi_lds;
  }
  dim3 g(cdiv(a.M, FBM), cdiv(a.N, BN));
  int bk32;
  bk32 = g_opt[MPMAE_OPT_NT_BK32];
  int glds;
  glds = g_opt[MPMAE_OPT_NT_GLDS];
  if (glds && (epi == EPI_STORE || epi == EPI_RESID) && (BN == 128 || glds_bn64()) && a.M >= 4096 && a.K % 64 == 0) {
    // direct global -> LDS slabs, swizzled unpadded rows
    if (glds == 2 || a.K <= 512) {
      const size_t l = (size_t)(2 * FBM * 32 + 2 * BN * 32) * sizeof(bf16_t);
      const size_t need = l > (size_t)128 * 68 * 4 ? l : (size_t)128 * 68 * 4;
      LAUNCH((gemm_nt_bf16_kernel<BN, EPI_STORE, 32, true>), g, dim3(256), need, st, a);
    } else {
      const size_t l = (size_t)(2 * FBM * 64 + 2 * BN * 64) * sizeof(bf16_t);
      static bool once = false;
      if (!once && l > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel<BN, EPI_STORE, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l) != hipSuccess) return (int)hipGetLastError();
        once = true;
      }
      LAUNCH((gemm_nt_bf16_kernel<BN, EPI_STORE, 64, true>), g, dim3(256), l, st, a);
    }
    return launch_status();
  }
  if (bk32 && (epi == EPI_STORE || epi == EPI_RESID) && BN == 128 && a.M >= 4096 && a.K <= 512) {
    // short K, wide N (decoder pw1 / pw2.dgrad, pixel heads): half-depth K slabs, 41 KB of LDS instead of 74 KB ->
    // 3-4 workgroups per CU (measured 84 -> 67, 72 -> 58, 85 -> 77 us; for K = 2048 the 64-deep slabs stay faster)
    const size_t lds32 = (size_t)(2 * FBM * (32 + FPAD) + 2 * BN * (32 + FPAD)) * sizeof(bf16_t);
    LAUNCH((gemm_nt_bf16_kernel<BN, EPI_STORE, 32>), g, dim3(256), lds32, st, a);
    return launch_status();
  }
#define FAST_CASE(E)                                                                                   \
  if (epi == E || (E == EPI_STORE && epi == EPI_RESID)) {                                              \
    static bool once = false;                                                                          \
    if (!once && lds > 64 * 1024) {                                                                    \
      if (hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel<BN, E>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds) != hipSuccess) return (int)hipGetLastError();                  \
      once = true;                                                                                     \
    }                                                                                                  \
    LAUNCH((gemm_nt_bf16_kernel<BN, E>), g, dim3(256), lds, st, a);                        \
    return launch_status();                                                                     \
  }
  FAST_CASE(EPI_STORE)
  FAST_CASE(EPI_GELU_SUMSQ)
  FAST_CASE(EPI_DZ_STATS)
#undef FAST_CASE
  return (int)hipErrorInvalidValue;
}

template <int EPI>
static int launch_nt3_k(const GemmP& a, const Nt3Scales& sc, hipStream_t st) {
  const size_t lds = (size_t)N3_ST * N3_STAGE_B;
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute((const void*)gemm_nt3_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return (int)hipGetLastError();
    once = true;
  }
  dim3 g(cdiv(a.M, N3_BM), cdiv(a.N, N3_BN));
  LAUNCH((gemm_nt3_kernel<EPI>), g, dim3(N3_T), lds, st, a, sc);
  return launch_status();
}

// 256-row tiles, 8 waves, 32 x 32 x 16 MFMA, DMA double buffer (gemm_nt4.cuh): the decoder / head shapes
template <int BN>
static int launch_nt4(const GemmP& a, hipStream_t st) {
  using Cf = Nt4Cfg<BN>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_nt4_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS) != hipSuccess)
      return (int)hipGetLastError();
    attr = true;
  }
  const int tm = cdiv(a.M, NT4_BM), tn = cdiv(a.N, BN);
  LAUNCH((gemm_nt4_kernel<BN>), dim3(cdiv(tm, 8) * 8 * tn), dim3(512), Cf::LDS, st, a, tm, tn);
  return launch_status();
}

static int launch_gemm_fast(int epi, GemmP a, hipStream_t st) {
  if (epi == EPI_STORE) a.R = nullptr;
  // measured (tools/gemm_probe.py, profiles/r04/gemm_probe.txt): the 256 x 256 tile wins where its tile count fills the 256 CUs in
  // at most two even rounds (decoder pw1 / pw2.dgrad, N = 2048: 49.5 vs 55.3 us) and loses on 539 tiles (pixel heads, N = 2816:
  // 73 vs 65 us); the 256 x 128 variant ties with the 128 x 128 kernels (N = 512). NT4 = 2 takes it for every eligible shape.
  const int nt4 = g_opt[MPMAE_OPT_NT4];
  if (nt4 && (epi == EPI_STORE || epi == EPI_RESID) && a.K % NT4_BK == 0 && a.M >= 8192 && a.N >= 256 &&
      !(((uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.C | (uintptr_t)a.R) & 15) &&
      (nt4 >= 2 || (a.N % 256 == 0 && a.N >= 1024 && a.N <= 2048))) {
    return a.N >= 1024 ? launch_nt4<256>(a, st) : launch_nt4<128>(a, st);
  }
  const int w128 = cdiv(a.N, 128) * 128 - a.N, w64 = cdiv(a.N, 64) * 64 - a.N;
  int err = (w64 < w128) ? launch_gemm_fast_bn<64>(epi, a, st) : launch_gemm_fast_bn<128>(epi, a, st);
  if (err) return err;
  if (epi == EPI_GELU_SUMSQ || epi == EPI_DZ_STATS) {
    const int mblocks = cdiv(a.M, FBM);
    if (a.rpg > 0 && a.rpg < a.M) {                            // per-group sums (dense decoder): fold the tiles' group slots
      const int G = a.M / a.rpg;
      const int rg = grid1d((long long)G * a.N, 256, 2048);
      LAUNCH(reduce_tile_groups_kernel, dim3(rg), dim3(256), 0, st, (const float*)a.ws, mblocks, a.N, a.rpg, G, a.s0);
      if (epi == EPI_DZ_STATS)
        LAUNCH(reduce_tile_groups_kernel, dim3(rg), dim3(256), 0, st, (const float*)(a.ws + (size_t)mblocks * 4 * a.N), mblocks, a.N,
               a.rpg, G, a.s1);
      return launch_status();
    }
    launch_reduce(0, a.ws, mblocks, a.N, a.s0, nullptr, 0, 0, 0, 0, st);
    if (epi == EPI_DZ_STATS) launch_reduce(0, a.ws + (size_t)mblocks * a.N, mblocks, a.N, a.s1, nullptr, 0, 0, 0, 0, st);
    err = launch_status();
  }
  return err;
}

static bool tn2_ok(const WgradP& a);
static bool wgrad_fast_ok(int dt, int ppro, int qpro, const WgradP& a) {
  if (dt != 1 || ppro != PRO_NONE || ((a.ldp | a.ldq | a.Nn | a.Kk) & 1)) return false;
  if (qpro == PRO_NONE) return true;
  // GRN prologue on the wide operand (pwconv2's weight gradient from h instead of a stored z): transpose-read kernel only, one GRN group,
  // narrow side = P, not a decoder / head shape (the DMA-ring kernel moves its operands global -> LDS untouched)
  return qpro == PRO_GRN && a.qp0 && a.qp1 && a.rpg >= a.M && a.Nn <= a.Kk && a.Nn < 256 && tn2_ok(a) &&
         !(((uintptr_t)a.qp0 | (uintptr_t)a.qp1) & 3);
}

static int tn_variant() {      // MPMAE_TN=1 forces the register-transposing kernel (A/B measurements)
  int v;
  v = g_opt[MPMAE_OPT_TN];
  return v;
}

// transpose-read kernel: 16-byte row vectors, narrow side <= wide side
template <int NT, int KT>
static void launch_tn2(const WgradP& a, bool swap, int splits, hipStream_t st, bool qgrn) {
  const int WX = swap ? a.Kk : a.Nn, WY = swap ? a.Nn : a.Kk;
  dim3 g(cdiv(WX, 16 * NT), cdiv(WY, 64 * KT), splits);
  if (swap) LAUNCH((gemm_tn2_kernel<NT, KT, true>), g, dim3(256), 0, st, a, splits);
  else if (qgrn) LAUNCH((gemm_tn2_kernel<NT, KT, false, true>), g, dim3(256), 0, st, a, splits);
  else LAUNCH((gemm_tn2_kernel<NT, KT, false>), g, dim3(256), 0, st, a, splits);
}

static bool tn2_ok(const WgradP& a) {
  if (tn_variant() < 2) return false;
  if ((a.ldp | a.ldq | a.Nn | a.Kk) & 7) return false;
  if (((uintptr_t)a.P | (uintptr_t)a.Q) & 15) return false;
  const int WX = a.Nn < a.Kk ? a.Nn : a.Kk;
  return WX >= 32;
}

// decoder / head shapes: DMA ring + 128 x 256 tiles (gemm_tn3.cuh); one workgroup per CU (144 KB of LDS)
static int launch_wgrad_tn3(WgradP a, bool swap, hipStream_t st) {
  const size_t per = (size_t)a.Nn * a.Kk + a.Nn;
  const int WX = swap ? a.Kk : a.Nn, WY = swap ? a.Nn : a.Kk;
  const int tiles = (WX / TN3_BX) * (WY / TN3_BY);
  int splits = g_opt[MPMAE_OPT_TN3_BLOCKS] / tiles;
  if (tiles * 8 <= g_opt[MPMAE_OPT_TN3_BLOCKS] && splits < 8) splits = 8;      // one row range per XCD
  if (splits > a.M / (4 * TN3_SL)) splits = a.M / (4 * TN3_SL);
  if (splits > (int)(a.ws_floats / per)) splits = (int)(a.ws_floats / per);
  if (splits < 1) splits = 1;
  const int rps = cdiv(cdiv(a.M, splits), TN3_SL) * TN3_SL;
  a.rows_per_split = rps;
  splits = cdiv(a.M, rps);
  static bool attr[2] = {false, false};
  if (!attr[swap]) {
    const void* f = swap ? (const void*)gemm_tn3_kernel<true> : (const void*)gemm_tn3_kernel<false>;
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, TN3_LDS) != hipSuccess) return (int)hipGetLastError();
    attr[swap] = true;
  }
  dim3 g(splits, WX / TN3_BX, WY / TN3_BY);        // split fastest: one row range per XCD when splits == 8 (gemm_tn3.cuh)
  if (swap) LAUNCH((gemm_tn3_kernel<true>), g, dim3(256), TN3_LDS, st, a, splits);
  else LAUNCH((gemm_tn3_kernel<false>), g, dim3(256), TN3_LDS, st, a, splits);
  const int nk = a.Nn * a.Kk;
  if (a.sn == a.Kk && a.sk == 1) {
    launch_reduce(3, a.ws, splits, nk + a.Nn, a.dW, a.db, nk, 0, 0, 0, st);
  } else {
    launch_reduce(1, a.ws, splits, nk + a.Nn, a.dW, nullptr, a.Kk, a.sn, a.sk, nk, st);
    if (a.db) launch_reduce(3, a.ws, splits, nk + a.Nn, nullptr, a.db, nk, 1, 0, 0, st);
  }
  return launch_status();
}

static int launch_wgrad_tn2(WgradP a, hipStream_t st, bool qgrn) {
  const size_t per = (size_t)a.Nn * a.Kk + a.Nn;
  const bool swap = a.Kk < a.Nn;
  const int WX = swap ? a.Kk : a.Nn, WY = swap ? a.Nn : a.Kk;
  if (g_opt[MPMAE_OPT_TN3_BLOCKS] > 0 && WX >= 256 && WX % TN3_BX == 0 && WY % TN3_BY == 0 && a.M % TN3_SL == 0 && a.M >= 16 * TN3_SL)
    return launch_wgrad_tn3(a, swap, st);
  int nt, kt;
  if (WX <= 48) { nt = 3; kt = 3; }
  else if (WX % 80 == 0) { nt = 5; kt = 5; }
  else { nt = 4; kt = 4; }
  const int tiles = cdiv(WX, 16 * nt) * cdiv(WY, 64 * kt);
  int target = -1, minrows = -1;
  target = g_opt[MPMAE_OPT_TN_BLOCKS];
  minrows = g_opt[MPMAE_OPT_TN_MINROWS];
  int bigt;
  bigt = g_opt[MPMAE_OPT_TN_BLOCKS_BIG];   // measured in-step: 512 -> 5.59, 256 -> 5.54, 128 -> 5.86 ms
  // large dW (stage 2+, decoder, heads): every split writes and the second stage re-reads a full fp32 copy of dW
  const int tgt = (bigt > 0 && (size_t)a.Nn * a.Kk >= 65536) ? bigt : target;
  int splits = cdiv(tgt, tiles);
  const int maxs = cdiv(a.M, minrows);
  if (splits > maxs) splits = maxs;
  if (splits > (int)(a.ws_floats / per)) splits = (int)(a.ws_floats / per);
  if (splits < 1) splits = 1;
  int rps = cdiv(cdiv(a.M, splits), 32) * 32;
  a.rows_per_split = rps;
  splits = cdiv(a.M, rps);
  if (nt == 3) launch_tn2<3, 3>(a, swap, splits, st, qgrn);
  else if (nt == 5) launch_tn2<5, 5>(a, swap, splits, st, qgrn);
  else launch_tn2<4, 4>(a, swap, splits, st, qgrn);
  const int nk = a.Nn * a.Kk;
  if (a.sn == a.Kk && a.sk == 1) {               // contiguous dW: weights and bias fold in one launch
    launch_reduce(3, a.ws, splits, nk + a.Nn, a.dW, a.db, nk, 0, 0, 0, st);
  } else {
    launch_reduce(1, a.ws, splits, nk + a.Nn, a.dW, nullptr, a.Kk, a.sn, a.sk, nk, st);
    if (a.db) launch_reduce(3, a.ws, splits, nk + a.Nn, nullptr, a.db, nk, 1, 0, 0, st);
  }
  return launch_status();
}

static int launch_wgrad_fast(WgradP a, hipStream_t st, bool qgrn) {
  const size_t per = (size_t)a.Nn * a.Kk + a.Nn;
  if (!a.ws || a.ws_floats < per) return (int)hipErrorInvalidValue;
  if (tn2_ok(a)) return launch_wgrad_tn2(a, st, qgrn);
  const int tiles = cdiv(a.Nn, 128) * cdiv(a.Kk, 128);
  int splits = cdiv(512, tiles);                 // ~2 workgroups per CU
  if (splits > 128) splits = 128;               // bound the second-stage reduction
  const int maxs = cdiv(a.M, 256);               // at least 8 reduction slabs per workgroup
  if (splits > maxs) splits = maxs;
  if (splits > (int)(a.ws_floats / per)) splits = (int)(a.ws_floats / per);
  if (splits < 1) splits = 1;
  int rps = cdiv(cdiv(a.M, splits), TBM) * TBM;
  a.rows_per_split = rps;
  splits = cdiv(a.M, rps);
  dim3 g(cdiv(a.Nn, 128), cdiv(a.Kk, 128), splits);
  LAUNCH(gemm_tn_bf16_kernel, g, dim3(256), 0, st, a);
  launch_reduce(1, a.ws, splits, a.Nn * a.Kk, a.dW, nullptr, a.Kk, a.sn, a.sk, 0, st);
  if (a.db) launch_reduce(0, a.ws + (size_t)splits * a.Nn * a.Kk, splits, a.Nn, a.db, nullptr, 0, 0, 0, 0, st);
  return launch_status();
}

// Dense-decoder GRN, one kernel per direction (grn_group.cuh): bf16, H == 2048, rpg <= 52 rows per group, M == G * rpg.
int mpmae_grn_group_ok(int dt, int M, int H, int rpg) {
  return dt == 1 && H == GrnGroupCfg<13>::H && rpg >= 1 && rpg <= GrnGroupCfg<13>::MAXROWS && M > 0 && M % rpg == 0;
}

template <int MAXR>
static int launch_grn_group(bool bwd, const void* a0, void* a1, const float* p0, const float* p1, const float* p2, const float* p3,
                            float eps, int G, int rpg, float* o0, float* o1, float* o2, hipStream_t st) {
  using Cf = GrnGroupCfg<MAXR>;
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute((const void*)grn_group_fwd_kernel<MAXR>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS_F) != hipSuccess ||
        hipFuncSetAttribute((const void*)grn_group_bwd_kernel<MAXR>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS_B) != hipSuccess)
      return (int)hipGetLastError();
    once = true;
  }
  if (!bwd) LAUNCH(grn_group_fwd_kernel<MAXR>, dim3(G), dim3(1024), Cf::LDS_F, st, (const bf16_t*)a0, (bf16_t*)a1, p0, p1, eps, rpg, o0, o1, o2);
  else LAUNCH(grn_group_bwd_kernel<MAXR>, dim3(G), dim3(1024), Cf::LDS_B, st, (bf16_t*)a1, (const bf16_t*)a0, p0, p1, p2, p3, rpg, o0);
  return 0;
}

int mpmae_grn_group_fwd(int dt, const void* h, void* z, const float* gamma, const float* beta, float eps, int M, int H,
                        int rpg, float* Gx, float* Ainv, float* scale, mpmae_stream_t s) {
  if (!mpmae_grn_group_ok(dt, M, H, rpg) || !h || !z || !gamma || !beta || !Gx || !Ainv || !scale) return (int)hipErrorInvalidValue;
  const int e = rpg <= 28 ? launch_grn_group<7>(false, h, z, gamma, beta, nullptr, nullptr, eps, M / rpg, rpg, Gx, Ainv, scale, S_(s))
                          : launch_grn_group<13>(false, h, z, gamma, beta, nullptr, nullptr, eps, M / rpg, rpg, Gx, Ainv, scale, S_(s));
  if (e) return e;
  RET();
}

// dh over dz; slab[G][2H] receives the per-group gamma / beta gradient rows (fold: mpmae_fold_group{slab, G, 2H, dgamma, H, dbeta - dgamma, 1})
int mpmae_grn_group_bwd(int dt, void* dz, const void* h, const float* scale, const float* Gx, const float* Ainv,
                        const float* gamma, int M, int H, int rpg, float* slab, mpmae_stream_t s) {
  if (!mpmae_grn_group_ok(dt, M, H, rpg) || !dz || !h || !scale || !Gx || !Ainv || !gamma || !slab) return (int)hipErrorInvalidValue;
  const int e = rpg <= 28 ? launch_grn_group<7>(true, h, dz, scale, Gx, Ainv, gamma, 0.f, M / rpg, rpg, slab, nullptr, nullptr, S_(s))
                          : launch_grn_group<13>(true, h, dz, scale, Gx, Ainv, gamma, 0.f, M / rpg, rpg, slab, nullptr, nullptr, S_(s));
  if (e) return e;
  RET();
}

int mpmae_grn_apply(int dt, const void* h, void* z, const float* scale, const float* beta, int M, int H, int rpg,
                    const uint8_t* act, mpmae_stream_t s) {
  if (H & 7) return (int)hipErrorInvalidValue;
  const int g = grid1d((long long)M * H / 8, 256, 8192);
  if (dt == 0) LAUNCH(grn_apply_kernel<float>, dim3(g), dim3(256), 0, S_(s), (const float*)h, (float*)z, scale, beta, M, H, rpg, act);
  else LAUNCH(grn_apply_kernel<bf16_t>, dim3(g), dim3(256), 0, S_(s), (const bf16_t*)h, (bf16_t*)z, scale, beta, M, H, rpg, act);
  RET();
}

int mpmae_grn_bwd_apply(int dt, void* dz, const void* h, const float* scale, const float* coef, int M, int H, int rpg,
                        mpmae_stream_t s) {
  if (H & 7) return (int)hipErrorInvalidValue;
  const int g = grid1d((long long)M * H / 8, 256, 8192);
  if (dt == 0) LAUNCH(grn_bwd_apply_kernel<float>, dim3(g), dim3(256), 0, S_(s), (float*)dz, (const float*)h, scale, coef, M, H, rpg);
  else LAUNCH(grn_bwd_apply_kernel<bf16_t>, dim3(g), dim3(256), 0, S_(s), (bf16_t*)dz, (const bf16_t*)h, scale, coef, M, H, rpg);
  RET();
}

int mpmae_colstats(int dt, const void* h, const void* dz, int mode, float* s0, float* s1, int M, int H, int rpg,
                   float* ws, size_t ws_floats, mpmae_stream_t s) {
  if ((H & 7) == 0 && cdiv(H / 8, 64) <= 6 && (size_t)4 * (mode + 1) * H * sizeof(float) <= 64 * 1024) {
    const bool single1 = rpg >= M;
    int rpw = single1 ? 8 : rpg;
    if (!single1 && M % rpg != 0) return (int)hipErrorInvalidValue;
    const size_t per = (size_t)H * (mode == 1 ? 2 : 1);
    if (single1) {
      if (!ws || ws_floats < per) return (int)hipErrorInvalidValue;
      while ((size_t)cdiv(M, rpw) * per > ws_floats || cdiv(M, rpw) > 4096) rpw *= 2;
    }
    if (single1) { rpw = 64; while ((size_t)cdiv(M, rpw) * per > ws_floats) rpw *= 2; }
    const int nblk = cdiv(M, rpw);
    float* o0 = single1 ? ws : s0;
    float* o1 = single1 ? ws + (size_t)nblk * H : s1;
    int vpl = cdiv(H / 8, 64), ysplit = 1;
    int cs_split;
    cs_split = g_opt[MPMAE_OPT_CS_SPLIT];
    if (cs_split && nblk * 2 <= 1024 && vpl > 1) { ysplit = vpl; vpl = 1; }      // few row slabs: split the columns over gridDim.y
    const size_t lds = (size_t)4 * (mode + 1) * H * sizeof(float);
#define CS3(TT, VV) LAUNCH((colstats_v3_kernel<TT, VV>), dim3(nblk, ysplit), dim3(256), lds, S_(s), (const TT*)h, (const TT*)dz, mode, o0, o1, M, H, rpw)
#define CS3_T(TT) do { if (vpl == 1) CS3(TT, 1); else if (vpl == 2) CS3(TT, 2); else if (vpl <= 4) CS3(TT, 4); else CS3(TT, 6); } while (0)
    if (dt == 0) CS3_T(float); else CS3_T(bf16_t);
#undef CS3_T
#undef CS3
    if (single1) {
      launch_reduce(0, ws, nblk, H, s0, nullptr, 0, 0, 0, 0, S_(s));
      if (mode == 1) launch_reduce(0, ws + (size_t)nblk * H, nblk, H, s1, nullptr, 0, 0, 0, 0, S_(s));
    }
    RET();
  }
  const bool single = rpg >= M;
  int rpb = single ? 256 : rpg;
  if (single) {
    const size_t per = (size_t)H * (mode == 1 ? 2 : 1);
    if (!ws || ws_floats < per) return (int)hipErrorInvalidValue;
    while ((size_t)cdiv(M, rpb) * per > ws_floats) rpb *= 2;
  } else if (M % rpg != 0) return (int)hipErrorInvalidValue;
  const int rblocks = cdiv(M, rpb);
  dim3 g(cdiv(H, 64), rblocks);
  if (dt == 0) LAUNCH(colstats_kernel<float>, g, dim3(256), 0, S_(s), (const float*)h, (const float*)dz, mode, s0, s1, M, H, rpg, rpb, ws);
  else LAUNCH(colstats_kernel<bf16_t>, g, dim3(256), 0, S_(s), (const bf16_t*)h, (const bf16_t*)dz, mode, s0, s1, M, H, rpg, rpb, ws);
  if (single) {
    launch_reduce(0, ws, rblocks, H, s0, nullptr, 0, 0, 0, 0, S_(s));
    if (mode == 1) launch_reduce(0, ws + (size_t)rblocks * H, rblocks, H, s1, nullptr, 0, 0, 0, 0, S_(s));
  }
  RET();
}

// ------------------------------------------------------------------------------------------
// row-streaming fused pointwise kernels
// ------------------------------------------------------------------------------------------
template <int KC, int HN>
static int launch_rs(int which, const MpmaeRsArgs& a, hipStream_t st) {
  RsP p;
  p.A = (const bf16_t*)a.A; p.A2 = (const bf16_t*)a.A2; p.W = (const bf16_t*)a.W; p.ldw = a.ldw;
  p.bias = a.bias; p.v0 = a.v0; p.v1 = a.v1; p.out = (bf16_t*)a.out; p.xhat = (bf16_t*)a.xhat; p.xn = (bf16_t*)a.xn;
  p.rstd = a.rstd; p.R = (const bf16_t*)a.R; p.lng = a.lng; p.ws = a.ws; p.act = a.act; p.M = a.M;
  p.fin_sum = nullptr; p.D = nullptr; p.W2 = nullptr; p.ldw2 = 0; p.hb = nullptr;
  if (a.fin_sum || a.dz_dout) return (int)hipErrorInvalidValue;      // folded GRN finalisation / dz recomputation: chunked kernels only
  if (which < 2 && !a.out) return (int)hipErrorInvalidValue;
  const int ngroups = a.M / 16;
  if (which == 0 || which == 1) {
    constexpr int KS = (KC + 31) / 32, LDW = KS * 32 + 8, SLD = HN + 8;
    int nw = 8;
    size_t lds = (size_t)HN * LDW * 2 + (size_t)nw * 16 * SLD * 2 + 2 * HN * 4;
    if (lds > 160 * 1024) { nw = 4; lds = (size_t)HN * LDW * 2 + (size_t)nw * 16 * SLD * 2 + 2 * HN * 4; }
    const int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
    int blocks = 256 * (per_cu > 2 ? 2 : per_cu);
    if (blocks > cdiv(ngroups, nw)) blocks = cdiv(ngroups, nw);
    const size_t need = (size_t)blocks * HN * (which == 1 ? 2 : 1);
    if (!a.ws || a.ws_floats < need) return (int)hipErrorInvalidValue;
    if (which == 0) {
      static bool once = false;
      if (!once) { if (hipFuncSetAttribute((const void*)rs_wide_kernel<KC, HN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); once = true; }
      LAUNCH((rs_wide_kernel<KC, HN, 0>), dim3(blocks), dim3(64 * nw), lds, st, p);
      launch_reduce(0, a.ws, blocks, HN, a.s0, nullptr, 0, 0, 0, 0, st);
    } else {
      static bool once = false;
      if (!once) { if (hipFuncSetAttribute((const void*)rs_wide_kernel<KC, HN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); once = true; }
      LAUNCH((rs_wide_kernel<KC, HN, 1>), dim3(blocks), dim3(64 * nw), lds, st, p);
      launch_reduce(0, a.ws, blocks, HN, a.s0, nullptr, 0, 0, 0, 0, st);
      launch_reduce(0, a.ws + (size_t)blocks * HN, blocks, HN, a.s1, nullptr, 0, 0, 0, 0, st);
    }
  } else {
    constexpr int NT = (KC + 15) / 16, NP = NT * 16, LDW = HN + 8, SLD = NP + 8;
    const int nw = 8;
    const size_t lds = (size_t)NP * LDW * 2 + 2 * HN * 4 + 2 * NP * 4 + (size_t)nw * 16 * SLD * 2;
    const int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
    int blocks = 256 * (per_cu > 2 ? 2 : per_cu);
    if (blocks > cdiv(ngroups, nw)) blocks = cdiv(ngroups, nw);
    if (which == 2) {
      static bool once = false;
      if (!once) { if (hipFuncSetAttribute((const void*)rs_narrow_kernel<KC, HN, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); once = true; }
      LAUNCH((rs_narrow_kernel<KC, HN, 0, false>), dim3(blocks), dim3(64 * nw), lds, st, p);
    } else {
      if (!a.ws || a.ws_floats < (size_t)blocks * 2 * KC) return (int)hipErrorInvalidValue;
      static bool once = false;
      if (!once) { if (hipFuncSetAttribute((const void*)rs_narrow_kernel<KC, HN, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); once = true; }
      LAUNCH((rs_narrow_kernel<KC, HN, 1, false>), dim3(blocks), dim3(64 * nw), lds, st, p);
      const long long delta = a.s1 - a.s0;        // s0 = dgamma, s1 = dbeta (same flat gradient buffer)
      if (delta > 2147483647LL || delta < -2147483647LL) return (int)hipErrorInvalidValue;
      launch_reduce(1, a.ws, blocks, 2 * KC, a.s0, nullptr, KC, (int)delta, 1, 0, st);
    }
  }
  return launch_status();
}

// chunked variants (rsc.cuh): weights streamed through LDS, any M

static int ps_num_cus();
template <int KC, int RT, int NC, int KCH, int RTN = RT, int PFN = 0>
static int launch_rsc(int which, const MpmaeRsArgs& a, hipStream_t st) {
  RsP p;
  p.A = (const bf16_t*)a.A; p.A2 = (const bf16_t*)a.A2; p.W = (const bf16_t*)a.W; p.ldw = a.ldw;
  p.bias = a.bias; p.v0 = a.v0; p.v1 = a.v1; p.out = (bf16_t*)a.out; p.xhat = (bf16_t*)a.xhat; p.xn = (bf16_t*)a.xn;
  p.rstd = a.rstd; p.R = (const bf16_t*)a.R; p.lng = a.lng; p.ws = a.ws; p.act = a.act; p.M = a.M;
  p.fin_sum = a.fin_sum; p.fin_sum0 = a.fin_sum0; p.fin_gamma = a.fin_gamma; p.fin_gx = a.fin_gx; p.fin_ainv = a.fin_ainv;
  p.fin_out = a.fin_out; p.fin_dgamma = a.fin_dgamma; p.fin_dbeta = a.fin_dbeta; p.fin_eps = a.fin_eps;
  p.D = (const bf16_t*)a.dz_dout; p.W2 = (const bf16_t*)a.dz_w2t; p.ldw2 = a.dz_ldw2; p.hb = a.dz_bias;
  const int HN = a.H;
  if (HN != 4 * KC) return (int)hipErrorInvalidValue;      // the kernels assume H = 4C (compile-time row pitch)
  if (((uintptr_t)a.bias | (uintptr_t)a.v0 | (uintptr_t)a.v1 | (uintptr_t)a.lng | (uintptr_t)a.W) & 15) return (int)hipErrorInvalidValue;
  if ((a.ldw & 7) || HN % NC || HN % KCH) return (int)hipErrorInvalidValue;
  const int rowblocks = cdiv(a.M, 64 * RT);
  if (which == 0 || which == 1) {
    if (a.fin_sum) return (int)hipErrorInvalidValue;
    // split the N range so that ~3 workgroups per CU exist; a split must be a whole number of chunks
    const int target = g_opt[MPMAE_OPT_RSC_BLOCKS];
    int nsplit = 1;
    while (rowblocks * nsplit * 2 <= target && (HN / NC) % (nsplit * 2) == 0) nsplit *= 2;
    const int cps = HN / nsplit;
    constexpr int KP = ((KC + 31) / 32) * 32;
    const size_t lds = (size_t)2 * NC * (KP + RSC_PAD) * 2 + (size_t)2 * cps * 4;
    const size_t need = (size_t)rowblocks * HN * (which == 1 ? 2 : 1);
    if (!a.ws || a.ws_floats < need || lds > 160 * 1024 - 512) return (int)hipErrorInvalidValue;
    dim3 g(rowblocks, nsplit);
    if (which == 0) {
      static size_t cur = 64 * 1024;
      if (lds > cur) { if (hipFuncSetAttribute((const void*)rsc_wide_kernel<KC, 0, RT, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; }
      LAUNCH((rsc_wide_kernel<KC, 0, RT, NC>), g, dim3(256), lds, st, p, HN, cps);
      launch_reduce(0, a.ws, rowblocks, HN, a.s0, nullptr, 0, 0, 0, 0, st);
    } else {
      static size_t cur = 64 * 1024;
      if (lds > cur) { if (hipFuncSetAttribute((const void*)rsc_wide_kernel<KC, 1, RT, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; }
      LAUNCH((rsc_wide_kernel<KC, 1, RT, NC>), g, dim3(256), lds, st, p, HN, cps);
      if (a.s1 == a.s0 + HN) {        // adjacent outputs (the engine's layout): one launch over [2*HN]
        launch_reduce(0, a.ws, rowblocks, 2 * HN, a.s0, nullptr, 0, 0, 0, 0, st);
      } else {                        // e = which*HN + j -> which == 0 ? s0[j] : s1[j]
        const long long delta = a.s1 - a.s0;
        if (delta > 2147483647LL || delta < -2147483647LL) return (int)hipErrorInvalidValue;
        launch_reduce(1, a.ws, rowblocks, 2 * HN, a.s0, nullptr, HN, (int)delta, 1, 0, st);
      }
    }
  } else if (which == 4 || which == 5) {
    const int rpg = a.rpg > 0 ? a.rpg : a.M;
    // 80-row tiles (5 waves) where 64-row tiles are more than one round of workgroups and 80-row tiles are not (C = 160 at bs 256:
    // 304 -> 244 workgroups on 256 CUs): the per-workgroup time is the weight stream, not the rows
    const bool w5 = KC == 160 && g_opt[MPMAE_OPT_RSC_W5] && cdiv(a.M, 64 * RTN) > ps_num_cus() && cdiv(a.M, 80 * RTN) <= ps_num_cus();
    const int rowblocks = cdiv(a.M, (w5 ? 80 : 64) * RTN);
    constexpr int NP = ((KC + 15) / 16) * 16;
    const int pf_on = g_opt[MPMAE_OPT_RSC_PF];
    const bool pf = PFN && pf_on && rpg >= a.M;       // LDS-staged GRN vectors (+ early issue): single GRN group only
    const bool dzr = a.dz_dout != nullptr;            // which 5: dz recomputed from dout; which 4: h recomputed from xn (never read)
    if (dzr && which == 4 && (!a.dz_bias || ((uintptr_t)a.dz_bias & 15))) return (int)hipErrorInvalidValue;
    if (dzr && (!(PFN & 4) || !pf || !a.dz_w2t || (a.dz_ldw2 & 7) || (((uintptr_t)a.dz_dout | (uintptr_t)a.dz_w2t) & 15))) return (int)hipErrorInvalidValue;
    constexpr int KP2 = ((KC + 31) / 32) * 32;
    const size_t lds = (size_t)2 * NP * (KCH + RSC_PAD) * 2 + (size_t)2 * KC * 4 + (pf ? (size_t)2 * HN * 4 + 32 : 0) +
                       (dzr ? (size_t)2 * KCH * (KP2 + RSC_PAD) * 2 : 0);
    if (lds > 160 * 1024 - 512) return (int)hipErrorInvalidValue;
    if (a.fin_sum) {                                   // folded GRN finalisation
      if (!pf || !a.fin_gamma || !a.fin_gx || !a.fin_ainv) return (int)hipErrorInvalidValue;
      if (which == 4 && (!a.fin_out || !a.v1)) return (int)hipErrorInvalidValue;
      if (which == 5 && (!a.fin_sum0 || !a.fin_dgamma || !a.fin_dbeta || !a.v0)) return (int)hipErrorInvalidValue;
      if (((uintptr_t)a.fin_sum | (uintptr_t)a.fin_gamma) & 15) return (int)hipErrorInvalidValue;
    }
    if (which == 5 && (!a.ws || a.ws_floats < (size_t)rowblocks * 2 * KC)) return (int)hipErrorInvalidValue;
#define RSC_NARROW_W(MODE_, PF_, NWV_) do { \
      static size_t cur = 64 * 1024; \
      if (lds > cur) { if (hipFuncSetAttribute((const void*)rsc_narrow_kernel<KC, MODE_, RTN, KCH, PF_, NWV_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; } \
      LAUNCH((rsc_narrow_kernel<KC, MODE_, RTN, KCH, PF_, NWV_>), dim3(rowblocks), dim3(64 * NWV_), lds, st, p, HN, rpg); } while (0)
#define RSC_NARROW(MODE_, PF_) do { \
      if constexpr (KC == 160) { if (w5) RSC_NARROW_W(MODE_, PF_, 5); else RSC_NARROW_W(MODE_, PF_, 4); } \
      else RSC_NARROW_W(MODE_, PF_, 4); } while (0)
    if (which == 4) { if (dzr) RSC_NARROW(0, (PFN & 4) ? (PFN & 6) : 0); else if (pf) RSC_NARROW(0, (PFN & 3)); else RSC_NARROW(0, 0); }
    else {
      if (dzr) RSC_NARROW(1, (PFN & 4) ? (PFN & 6) : 0);
      else if (pf) RSC_NARROW(1, (PFN & 3)); else RSC_NARROW(1, 0);
      // (direct float atomics into dgamma / dbeta instead of slab rows + this launch: measured 5.06 vs 4.92 ms per step - the
      // gridDim.x colliding updates per address land together at the kernel's tail)
      const long long delta = a.s1 - a.s0;        // s0 = dgamma, s1 = dbeta (same flat gradient buffer)
      if (delta > 2147483647LL || delta < -2147483647LL) return (int)hipErrorInvalidValue;
      if (a.defer_fold) *a.defer_fold = MpmaeFoldDesc{a.ws, rowblocks, 2 * KC, a.s0, KC, (int)delta, 1};
      else launch_reduce(1, a.ws, rowblocks, 2 * KC, a.s0, nullptr, KC, (int)delta, 1, 0, st);
    }
#undef RSC_NARROW
#undef RSC_NARROW_W
  } else {
    return (int)hipErrorInvalidValue;
  }
  return launch_status();
}

int mpmae_rs(int which, const MpmaeRsArgs* a, mpmae_stream_t s) {
  if (!a || which < 0 || which > 5) return (int)hipErrorInvalidValue;
  if (a->C == 160 && a->H == 640) return launch_rsc<160, 1, 32, 64, 1, 3>(which, *a, S_(s));      // 32-column chunks: the N range splits 4 ways (27.7 -> 24.1 us vs 64)
  if (a->C == 320 && a->H == 1280) return launch_rsc<320, 1, 32, 32>(which, *a, S_(s));
  // ConvNeXtV2-tiny widths (BASELINE config 4: 96 / 192 / 384; 768 stays on the tiled GEMMs)
  if (a->C == 96 && a->H == 384 && g_opt[MPMAE_OPT_RSC_SMALL]) return launch_rsc<96, 2, 64, 64, 1, 3>(which, *a, S_(s));
  if (a->C == 192 && a->H == 768) return launch_rsc<192, 1, 32, 64, 1, 3>(which, *a, S_(s));
  if (a->C == 384 && a->H == 1536) return launch_rsc<384, 1, 32, 32, 1, 3>(which, *a, S_(s));
  const int small = g_opt[MPMAE_OPT_RSC_SMALL];      // 0: keep the LDS-resident-weights kernels for which 0-3
  if (which > 3 || (small && which < 2)) {
    const int v40 = g_opt[MPMAE_OPT_RSC_N40], v80 = g_opt[MPMAE_OPT_RSC_N80];
    // (variants without the staged-vector prologue - no folded GRN finalisation, no operand recomputation - were removed in round 3:
    // the engine's program needs both, and nothing tested them)
    if (a->C == 40 && a->H == 160) {
      if (v40 == 1) return launch_rsc<40, 4, 160, 32, 2, 6>(which, *a, S_(s));
      return launch_rsc<40, 4, 160, 32, 1, 6>(which, *a, S_(s));
    }
    if (a->C == 80 && a->H == 320) {
      if (v80 == 0) return launch_rsc<80, 2, 64, 64, 2, 6>(which, *a, S_(s));
      return launch_rsc<80, 2, 64, 64, 1, 6>(which, *a, S_(s));
    }
  }
  if (which > 3 || (a->M & 15)) return (int)hipErrorInvalidValue;
  if (a->C == 40 && a->H == 160) return launch_rs<40, 160>(which, *a, S_(s));
  if (a->C == 80 && a->H == 320) return launch_rs<80, 320>(which, *a, S_(s));
  if (a->C == 96 && a->H == 384) return launch_rs<96, 384>(which, *a, S_(s));
  return (int)hipErrorInvalidValue;
}


// ------------------------------------------------------------------------------------------
// persistent per-sample stage kernels (ps.cuh)
// ------------------------------------------------------------------------------------------
static int ps_num_cus() {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
    (void)hipGetLastError();
    cus = v;
  }
  return cus;
}

static int ps_check(const MpmaePsArgs& a, int S, int keep_max) {
  if (!a.x_in || !a.g.vis || !a.g.inv || !a.sync || a.nblk < 1 || a.nblk > MPMAE_PS_MAXBLK || a.ng < 1 || a.ng > 16) return (int)hipErrorInvalidValue;
  if (a.g.S != S || a.g.keep < 1 || a.g.keep > keep_max || a.g.N < 1 || a.g.N > ps_num_cus()) return (int)hipErrorInvalidValue;
  if ((size_t)a.g.keep * S * S * a.C * 4 >= 65535u) return (int)hipErrorInvalidValue;       // 16-bit LDS offsets of the neighbour table
  return 0;
}

template <int C, int S>
static int launch_ps_fwd(const MpmaePsArgs& a, hipStream_t st) {
  using K = ps::Cfg<C, S>;
  if (const int e = ps_check(a, S, S == 2 ? K::RP / 4 : 32)) return e;
  for (int b = 0; b < a.nblk; ++b) {
    const MpmaePsBlock& B = a.blk[b];
    if (!B.dw_w || !B.dw_b || !B.ln_g || !B.ln_b || !B.W1 || !B.b1 || !B.grn_g || !B.grn_b || !B.W2 || !B.b2 || !B.dhat || !B.rstd ||
        !B.xn || !B.h || !B.z || !B.out || !B.G2 || !B.Gx || !B.Ainv || !B.scale || (B.ldw1 & 7) || (B.ldw2 & 7) || B.ldw1 < C || B.ldw2 < 4 * C)
      return (int)hipErrorInvalidValue;
    if (((uintptr_t)B.W1 | (uintptr_t)B.W2 | (uintptr_t)B.dhat | (uintptr_t)B.xn | (uintptr_t)B.h | (uintptr_t)B.z | (uintptr_t)B.out |
         (uintptr_t)B.dw_w | (uintptr_t)B.dw_b | (uintptr_t)B.ln_g | (uintptr_t)B.ln_b | (uintptr_t)B.b1 | (uintptr_t)B.b2) & 15)
      return (int)hipErrorInvalidValue;
  }
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)ps::ps_fwd_kernel<C, S>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS) != hipSuccess) return (int)hipGetLastError();
    attr = true;
  }
  LAUNCH((ps::ps_fwd_kernel<C, S>), dim3(a.g.N), dim3(ps::NTHR), K::LDS, st, a);
  return launch_status();
}

template <int C, int S>
static int launch_ps_bwd(const MpmaePsBwdArgs& a, hipStream_t st) {
  using K = ps::Cfg<C, S>;
  if (!a.dout_in || !a.g.vis || !a.g.inv || !a.sync || a.nblk < 1 || a.nblk > MPMAE_PS_MAXBLK || a.ng < 1 || a.ng > 16) return (int)hipErrorInvalidValue;
  if (a.g.S != S || a.g.keep < 1 || a.g.keep > (S == 2 ? K::RP / 4 : 32) || a.g.N < 1 || a.g.N > ps_num_cus()) return (int)hipErrorInvalidValue;
  if ((size_t)a.g.keep * S * S * C * 4 >= 65535u || ((uintptr_t)a.dout_in & 15)) return (int)hipErrorInvalidValue;
  for (int b = 0; b < a.nblk; ++b) {
    const MpmaePsBwdBlock& B = a.blk[b];
    if (!B.dw_w || !B.ln_g || !B.grn_g || !B.W2T || !B.W1T || !B.h || !B.dhat || !B.rstd || !B.Gx || !B.Ainv || !B.scale || !B.S0 || !B.S1 ||
        !B.d_grn_g || !B.d_grn_b || !B.d_ln_g || !B.d_ln_b || !B.dh || !B.dd || !B.dx || !a.ln_slab || (B.ldw2t & 7) || (B.ldw1t & 7) || B.ldw2t < C || B.ldw1t < 4 * C)
      return (int)hipErrorInvalidValue;
    if (((uintptr_t)B.W2T | (uintptr_t)B.W1T | (uintptr_t)B.h | (uintptr_t)B.dhat | (uintptr_t)B.dh | (uintptr_t)B.dd | (uintptr_t)B.dx |
         (uintptr_t)B.dw_w | (uintptr_t)B.ln_g) & 15)
      return (int)hipErrorInvalidValue;
  }
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)ps::ps_bwd_kernel<C, S>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS) != hipSuccess) return (int)hipGetLastError();
    attr = true;
  }
  LAUNCH((ps::ps_bwd_kernel<C, S>), dim3(a.g.N), dim3(ps::NTHR), K::LDS, st, a);
  const int nwg = a.g.N;
  LAUNCH(ps::ps_ln_reduce_kernel, dim3((2 * C + 63) / 64, a.nblk), dim3(256), 0, st, a, nwg);
  return launch_status();
}

int mpmae_ps_bwd(const MpmaePsBwdArgs* a, mpmae_stream_t s) {
  if (!a) return (int)hipErrorInvalidValue;
  if (a->C == 160 && a->g.S == 2) return launch_ps_bwd<160, 2>(*a, S_(s));
  if (a->C == 320 && a->g.S == 1) return launch_ps_bwd<320, 1>(*a, S_(s));
  return (int)hipErrorInvalidValue;
}

int mpmae_ps_fwd(const MpmaePsArgs* a, mpmae_stream_t s) {
  if (!a) return (int)hipErrorInvalidValue;
  if (a->C == 160 && a->g.S == 2) return launch_ps_fwd<160, 2>(*a, S_(s));
  if (a->C == 320 && a->g.S == 1) return launch_ps_fwd<320, 1>(*a, S_(s));
  return (int)hipErrorInvalidValue;
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

int mpmae_quant_mx(const void* x, int ld, int rows, int K, void* q, uint32_t* scales, int lds, mpmae_stream_t s) {
  if (!x || !q || !scales || rows < 1 || K < 128 || (K % 128) || (ld & 7) || lds < rows || (((uintptr_t)x | (uintptr_t)q) & 15))
    return (int)hipErrorInvalidValue;
  const long long blocks = (long long)rows * (K / 32);
  LAUNCH(quant_mx_kernel, dim3(grid1d(blocks, 256, 8192)), dim3(256), 0, S_(s), (const bf16_t*)x, ld, rows, K, (unsigned char*)q, scales, lds);
  RET();
}

int mpmae_gemm_mx(int epi, const MpmaeGemmArgs* a, const uint32_t* sa, int lsa, const uint32_t* sb, int lsb, mpmae_stream_t s) {
  if (!a || !sa || !sb || a->M < 1 || a->N < 1 || a->K < 128 || (a->K % 128) || (a->lda & 15) || (a->ldb & 15) || (a->ldc & 7) || (a->N & 7) ||
      lsa < a->M || lsb < a->N || (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15))
    return (int)hipErrorInvalidValue;
  if (epi == EPI_RESID && (!a->R || (a->ldr & 7))) return (int)hipErrorInvalidValue;
  const Nt3Scales sc{sa, sb, lsa, lsb};
  if (epi == EPI_STORE) { GemmP g = *a; g.R = nullptr; return launch_nt3_k<EPI_STORE>(g, sc, S_(s)); }
  if (epi == EPI_RESID) return launch_nt3_k<EPI_RESID>(*a, sc, S_(s));
  return (int)hipErrorInvalidValue;
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
  stcap = g_opt[MPMAE_OPT_STB_BLOCKS];
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
    for (int k = 0; k < 3; ++k) {      // slabs [k][wave][2][C]: e = n*C + c -> n == 0 ? first[c] : second[c]
      const long long delta = outs[k][1] - outs[k][0];
      if (delta > 2147483647LL || delta < -2147483647LL) return (int)hipErrorInvalidValue;
      launch_reduce(1, a->ws + (size_t)k * nw * 2 * a->C, nw, 2 * a->C, outs[k][0], nullptr, a->C, (int)delta, 1, 0, S_(s));
    }
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
    for (auto& l : op.launches) l(st);
    if (op.signal > 0 && p->waited[op.signal]) {
      if (hipEventRecord(p->events[op.signal], st) != hipSuccess) return (int)hipGetLastError();
      p->epoch[op.signal] = p->run;
    }
  }
  if (lanes) {                                   // join
    for (size_t l = 0; l < p->side.size(); ++l) {
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
  // ONE launch per FOLDG_MAX records (blockIdx.z = record) instead of one per record
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
    LAUNCH(reduce_partials_group1_kernel, dim3(cdiv(wmax, 64), R, n), dim3(256), 0, S_(s), g);
  }
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
