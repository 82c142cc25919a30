// extern "C" surface of libmpmae_hip.so, GEMM unit: dense / fused GEMMs, weight gradients (single, grouped, DMA-ring), the MX-fp8
// path and the GRN element-wise / column-statistics entry points (include/mpmae_hip.h). gfx950 only.
#include "capi_common.h"
#include "gemm.cuh"
#include "gemm_fast.cuh"
#include "rows2.cuh"
#include "gemm_tn2.cuh"
#include "gemm_nt3.cuh"
#include "gemm_tn3.cuh"
#include "gemm_tng.cuh"
#include "grn_group.cuh"

static bool gemm_fast_ok(int dt, int pro, int epi, const GemmP& a);
static int launch_gemm_fast(int epi, GemmP a, hipStream_t st);
static bool wgrad_fast_ok(int dt, int ppro, int qpro, const WgradP& a);
static int launch_wgrad_fast(WgradP a, hipStream_t st, bool qgrn = false);

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
  // rowscale (one-pass losses): folded by the second stage of the bf16 kernels only, contiguous dW
  if (args->rowscale && (args->sn != args->Kk || args->sk != 1 || !wgrad_fast_ok(dt, ppro, qpro, *args))) return (int)hipErrorInvalidValue;
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
  // (the grouped kernel needs one slab set per problem in the scratch: large / huge stage 3 - 6 problems of 1536 x 6144 - do not fit and go
  // one by one like the shapes the kernel does not take)
  if (!tng_ok(dt, probs, count, &WX, &WY) || (size_t)count * ((size_t)WX * WY + WY) > ws_floats) {            // one call per problem (each with the scratch given here)
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
  if (glds && (epi == EPI_STORE || epi == EPI_RESID) && (BN == 128 || glds_bn64()) && a.M >= 4096 && (a.K % 64 == 0 || (a.K % 32 == 0 && a.K <= 512))) {      // (K = 160: downsample 0 - the 32-deep kernels)
    // direct global -> LDS slabs, swizzled unpadded rows
    int ring = g_opt[MPMAE_OPT_NT_RING];
    if (ring == 1) {
      // auto (stand-alone numbers of the 22 step shapes: profiles/r06/gemm_ring_probe.txt): a grid of at most one tile per CU has nobody to hide a
      // round trip behind - three 64-deep stages (stage-3 pwconv2 / pwconv1.dgrad 14.5 -> 11.9 us, tiny's K = 3072 products 37.4 -> 33.1; four stages
      // are 0.6 / 1.2 us faster alone and 0.007 ms slower in the step); short K on a large grid - three 32-deep stages (decoder pwconv1 47.6 -> 44.9,
      // pixel heads 57.9 -> 52.5); long K on a large grid stays on the two-buffer kernel, whose 64 KB let two workgroups share a CU (decoder pwconv2
      // 33.1 vs 37.4 / 44.7 us for the rings). In the step: 3.375 vs 3.425 ms (atto), 14.70 vs 14.78 (tiny) - profiles/r06/ab_nt_ring.txt
      const int tiles = (int)(g.x * g.y);
      ring = (tiles <= ps_num_cus() && a.K >= 256) ? 364 : (a.K <= 512 ? 332 : 0);
    }
    if ((ring == 364 || ring == 464) && (a.K & 63)) ring = 332;      // (K = 160: 32-deep slabs only)
    if (ring > 1) {      // NST * 100 + BK: the ring form (gemm_nt_ring_kernel), more than one slab in flight per workgroup
#define NT_RING(BK_, NST_) do { \
        const size_t l = (size_t)NST_ * (FBM + BN) * BK_ * sizeof(bf16_t); \
        static bool once = false; \
        if (!once && l > 64 * 1024) { \
          if (hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<BN, BK_, NST_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l) != hipSuccess) return (int)hipGetLastError(); \
          once = true; \
        } \
        LAUNCH((gemm_nt_ring_kernel<BN, BK_, NST_>), g, dim3(256), l, st, a); \
        return launch_status(); } while (0)
      if (ring == 332) NT_RING(32, 3);
      if (ring == 432) NT_RING(32, 4);
      if (ring == 632) NT_RING(32, 6);
      if (ring == 364) NT_RING(64, 3);
      if (ring == 464) NT_RING(64, 4);
#undef NT_RING
    }
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

static int launch_gemm_fast(int epi, GemmP a, hipStream_t st) {
  if (epi == EPI_STORE) a.R = nullptr;
  // (a 256 x 256-tile 8-wave kernel, a deep-K 128 x 256 ring kernel and a stream-K schedule for these products were built in rounds 4-5, measured
  //  no faster in the step once the 128 x 128 kernel mapped its tiles XCD-aware, and removed in round 6: profiles/r05/ab_nt4_after_xcd.txt, ab_nt4_nt5.txt,
  //  stream_k_probe.txt)
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
  if (a.rowscale) {                              // (checked by mpmae_wgrad: contiguous dW)
    launch_reduce_rowscale(a.ws, splits, nk + a.Nn, a.dW, a.db, nk, a.Kk, a.rowscale, st);
  } else if (a.sn == a.Kk && a.sk == 1) {
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
  target = 512 /* TN_BLOCKS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
  minrows = 256 /* TN_MINROWS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
  int bigt;
  bigt = 512 /* TN_BLOCKS_BIG: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;   // measured in-step: 512 -> 5.59, 256 -> 5.54, 128 -> 5.86 ms
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
  if (a.rowscale) {
    launch_reduce_rowscale(a.ws, splits, nk + a.Nn, a.dW, a.db, nk, a.Kk, a.rowscale, st);
  } else if (a.sn == a.Kk && a.sk == 1) {               // contiguous dW: weights and bias fold in one launch
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
  if (a.rowscale) {                              // (contiguous dW, checked by mpmae_wgrad)
    launch_reduce_rowscale(a.ws, splits, a.Nn * a.Kk, a.dW, nullptr, a.Nn * a.Kk, a.Kk, a.rowscale, st);
    if (a.db) launch_reduce_rowscale(a.ws + (size_t)splits * a.Nn * a.Kk, splits, a.Nn, nullptr, a.db, 0, a.Kk, a.rowscale, st);
    return launch_status();
  }
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

// 2 * H floats of dynamic LDS beside the kernels' static reduction scratch, inside the default 64 KiB (no hipFuncSetAttribute): H <= 8160
#define MPMAE_GRN_FIN_MAXH 8160
int mpmae_grn_apply_fin(int dt, const void* h, void* z, const float* G2, const float* gamma, const float* beta, float eps, int M, int H,
                        const uint8_t* act, float* Gx, float* Ainv, float* scale, mpmae_stream_t s) {
  if ((H & 7) || H > MPMAE_GRN_FIN_MAXH || !G2 || !gamma || !beta || !Gx || !Ainv || !scale) return (int)hipErrorInvalidValue;
  const int g = grid1d((long long)M * H / 8, 256, 2048);
  const size_t lds = (size_t)2 * H * 4;
  if (dt == 0) LAUNCH(grn_apply_fin_kernel<float>, dim3(g), dim3(256), lds, S_(s), (const float*)h, (float*)z, G2, gamma, beta, eps, M, H, act, Gx, Ainv, scale);
  else LAUNCH(grn_apply_fin_kernel<bf16_t>, dim3(g), dim3(256), lds, S_(s), (const bf16_t*)h, (bf16_t*)z, G2, gamma, beta, eps, M, H, act, Gx, Ainv, scale);
  RET();
}

int mpmae_grn_bwd_apply_fin(int dt, void* dz, const void* h, const float* scale, const float* S0, const float* S1, const float* Gx,
                            const float* Ainv, const float* gamma, int M, int H, float* coef, float* dgamma, float* dbeta, mpmae_stream_t s) {
  if ((H & 7) || H > MPMAE_GRN_FIN_MAXH || !scale || !S0 || !S1 || !Gx || !Ainv || !gamma || !coef || !dgamma || !dbeta) return (int)hipErrorInvalidValue;
  const int g = grid1d((long long)M * H / 8, 256, 2048);
  const size_t lds = (size_t)2 * H * 4;
  if (dt == 0) LAUNCH(grn_bwd_apply_fin_kernel<float>, dim3(g), dim3(256), lds, S_(s), (float*)dz, (const float*)h, scale, S0, S1, Gx, Ainv, gamma, M, H, coef, dgamma, dbeta);
  else LAUNCH(grn_bwd_apply_fin_kernel<bf16_t>, dim3(g), dim3(256), lds, S_(s), (bf16_t*)dz, (const bf16_t*)h, scale, S0, S1, Gx, Ainv, gamma, M, H, coef, dgamma, dbeta);
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

