// extern "C" surface of libmpmae_hip.so, row-streaming unit: the fused pointwise kernels of a sparse block (mpmae_rs, rsc.cuh) and the
// persistent per-sample stage kernel (mpmae_ps_fwd, ps.cuh). gfx950 only.
#include "capi_common.h"
#include "rs.cuh"
#include "rsc.cuh"
#include "rsc1.cuh"
#include "rsp.cuh"
#include "rsn3.cuh"
#include "rst.cuh"
#include "ps.cuh"

// ------------------------------------------------------------------------------------------
// row-streaming fused pointwise kernels
// ------------------------------------------------------------------------------------------
// chunked variants (rsc.cuh): weights streamed through LDS, any M

template <int KC, int RT, int NC, int KCH, int RTN = RT, int PFN = 0>
static int launch_rsc(int which, const MpmaeRsArgs& a, hipStream_t st) {
  RsP p;
  p.A = (const bf16_t*)a.A; p.A2 = (const bf16_t*)a.A2; p.W = (const bf16_t*)a.W; p.ldw = a.ldw;
  p.bias = a.bias; p.v0 = a.v0; p.v1 = a.v1; p.out = (bf16_t*)a.out; p.xhat = (bf16_t*)a.xhat; p.xn = (bf16_t*)a.xn;
  p.rstd = a.rstd; p.R = (const bf16_t*)a.R; p.lng = a.lng; p.ws = a.ws; p.act = a.act; p.M = a.M;
  p.fin_sum = a.fin_sum; p.fin_sum0 = a.fin_sum0; p.fin_gamma = a.fin_gamma; p.fin_gx = a.fin_gx; p.fin_ainv = a.fin_ainv;
  p.fin_out = a.fin_out; p.fin_dgamma = a.fin_dgamma; p.fin_dbeta = a.fin_dbeta; p.fin_eps = a.fin_eps;
  p.D = (const bf16_t*)a.dz_dout; p.W2 = (const bf16_t*)a.dz_w2t; p.ldw2 = a.dz_ldw2; p.hb = a.dz_bias;
  p.s0a = nullptr; p.s1a = nullptr; p.perwave = 0; p.wg_ws = a.wg_ws;
  p.dn_xhat = (bf16_t*)a.dn_xhat; p.dn_rstd = a.dn_rstd; p.dn_y = (bf16_t*)a.dn_y; p.dn_g = a.dn_gamma; p.dn_b = a.dn_beta; p.dn_S = a.dn_S;
  if (a.dn_y && (which != 4 || !a.dn_xhat || !a.dn_rstd || !a.dn_gamma || !a.dn_beta || a.dn_S < 2 || (a.dn_S & 1) || KC > 96 ||
                 (((uintptr_t)a.dn_gamma | (uintptr_t)a.dn_beta) & 15))) return (int)hipErrorInvalidValue;
  if (which == 4 && !a.out && !a.dn_y) return (int)hipErrorInvalidValue;
  const int HN = a.H;
  if (HN != 4 * KC) return (int)hipErrorInvalidValue;      // the kernels assume H = 4C (compile-time row pitch)
  if (((uintptr_t)a.bias | (uintptr_t)a.v0 | (uintptr_t)a.v1 | (uintptr_t)a.lng | (uintptr_t)a.W) & 15) return (int)hipErrorInvalidValue;
  if ((a.ldw & 7) || HN % NC || HN % KCH) return (int)hipErrorInvalidValue;
  const int rowblocks = cdiv(a.M, 64 * RT);
  if (which == 6) {
    // T = dout^T gelu(h) and db2 = sum_rows dout (rst.cuh): A = dout [M][C], R = h [M][H]; ws = one slab row [C * H + C] per persistent workgroup.
    // With W = staged W2 [C][ldw]: s0 = S0 [H], s1 = S1 [H] (the GRN backward statistics, ADDED to) - see below. Without W: s0 = T [C * H], s1 = db2 [C] raw
    // (added to; mpmae_grn_stats_from_wgrad is the separate second step)
    if constexpr (KC == 40 || KC == 80) {
      if (!a.A || !a.R || !a.s0 || !a.s1 || !a.ws || HN % 160) return (int)hipErrorInvalidValue;
      if (((uintptr_t)a.A | (uintptr_t)a.R) & 15) return (int)hipErrorInvalidValue;
      const int nw = g_opt[MPMAE_OPT_RST_NW] == 4 ? 4 : 16, ny = HN / 160;
      const int ntiles = cdiv(a.M, 16 * nw);
      // (tools/probes/rst_probe.py. NW = 4: C = 40 flat between 2 and 3 per CU, 33.9-34.5 us; C = 80 best at 2 per CU over both column slices: 27.1 vs 33.4 us at 3)
      const int wgs = 0 /* RST_WGS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ > 0 ? 0 /* RST_WGS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ : (nw == 16 ? 1 : (KC == 40 ? 3 : 2)) * ps_num_cus();
      int gx = wgs / ny > 0 ? wgs / ny : 1;
      if (gx > ntiles) gx = ntiles;
      const size_t W = (size_t)KC * HN + KC;
      if (a.ws_floats < (size_t)gx * W) return (int)hipErrorInvalidValue;
      constexpr int BXv = ((KC + 15) / 16) * 16;
      const size_t ldst = (size_t)16 * nw * (tn2_ld(BXv) + tn2_ld(160)) * 2;
      // fused (W = staged W2): every workgroup also writes its share of S0 / S1 as a small slab row [2][H] behind the big ones; only THAT fold runs here
      // (the next main-lane kernel waits for it); the big slabs [C * H | C] stay in ws for mpmae_rs_wgrad_fold(C, H, ...) -> dW2, db2 on the weight-gradient
      // lane, *wg_rows = slab rows. Without W: raw T / db2 folded into s0 / s1 here.
      const bool fused = a.W != nullptr;
      if (fused && (!a.wg_rows || a.ldw < HN || a.ws_floats < (size_t)gx * (W + 2 * HN))) return (int)hipErrorInvalidValue;
      if (fused) { p.s0a = a.ws + (size_t)gx * W; *a.wg_rows = gx; }
      if (nw == 16) {
        static bool attr = false;
        if (!attr) { if (hipFuncSetAttribute((const void*)rst_kernel<KC, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldst) != hipSuccess) return (int)hipGetLastError(); attr = true; }
        LAUNCH((rst_kernel<KC, 16>), dim3(gx, ny), dim3(1024), ldst, st, p, ntiles);
      } else {
        LAUNCH((rst_kernel<KC, 4>), dim3(gx, ny), dim3(256), ldst, st, p, ntiles);
      }
      if (!fused) launch_reduce(3, a.ws, gx, (int)W, a.s0, a.s1, KC * HN, 0, 0, 0, st);
      else if (a.s1 == a.s0 + HN) launch_reduce(0, p.s0a, gx, 2 * HN, a.s0, nullptr, 0, 0, 0, 0, st);
      else {
        const long long delta = a.s1 - a.s0;
        if (delta > 2147483647LL || delta < -2147483647LL) return (int)hipErrorInvalidValue;
        launch_reduce(1, p.s0a, gx, 2 * HN, a.s0, nullptr, HN, (int)delta, 1, 0, st);
      }
      return launch_status();
    } else {
      return (int)hipErrorInvalidValue;
    }
  }
  if (which == 0 || which == 1) {
    if (a.fin_sum) return (int)hipErrorInvalidValue;
    // split the N range so that ~3 workgroups per CU exist; a split must be a whole number of chunks
    const int target = 1536 /* RSC_BLOCKS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */;
    int nsplit = 1;
    while (rowblocks * nsplit * 2 <= target && (HN / NC) % (nsplit * 2) == 0) nsplit *= 2;
    const int cps = HN / nsplit;
    constexpr int KP = ((KC + 31) / 32) * 32;
    // weight chunks + statistic rows: one per wave (fixed summation order, no LDS atomics) unless that costs a resident workgroup per CU
    // (tiny stage 2, C = 384: 53 -> 62 KB = 3 -> 2 workgroups per CU, +1.5 % on the step); MPMAE_OPT_DET forces the per-wave rows
    const size_t lds_w = (size_t)2 * NC * (KP + RSC_PAD) * 2, lds1 = lds_w + (size_t)2 * cps * 4, lds4 = lds_w + (size_t)4 * 2 * cps * 4;
    const size_t cap = 160 * 1024;
    const int det = g_opt[MPMAE_OPT_DET];      // (-1: the shared row + LDS atomics everywhere, for A/B)
    p.perwave = (det > 0 || (det == 0 && (cap / lds4 == cap / lds1 || cap / lds1 > 4))) ? 1 : 0;
    const size_t lds = p.perwave ? lds4 : lds1;
    if constexpr (KC == 40 || KC == 80) {
      // persistent burst-load form (rsp.cuh): resident weights, every operand of the next row tile requested under this tile's arithmetic
      const int rp = g_opt[MPMAE_OPT_RSP];
      if (rp > 0) {
        const int rtv = (rp >= 2 && which == 0) ? 2 : 1;      // (which 1 with two row tiles per wave: > 256 VGPRs)
        const int ntiles = cdiv(a.M, 64 * rtv), ny = HN / 160;
        constexpr int KPv = ((KC + 31) / 32) * 32;
        const size_t ldsp = (size_t)160 * (KPv + RSC_PAD) * 2 + (size_t)(2 * KPv + 160) * 4 + (size_t)4 * 2 * 160 * 4;
        // (tools/probes/rs1_probe.py small: which 0 best at 3 workgroups per CU - 43.5 / 29.0 us at C = 40 / 80 against 50.6 / 33.6 chunked -, which 1 at 2: 42.1 / 29.3 against 56.0 / 36.8)
        const int wgs = 0 /* RSP_WGS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ > 0 ? 0 /* RSP_WGS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ : (which == 0 ? 3 : 2) * ps_num_cus();
        int gx = wgs / ny > 0 ? wgs / ny : 1;
        if (gx > ntiles) gx = ntiles;
        const size_t needp = (size_t)gx * HN * (which == 1 ? 2 : 1);
        if (!a.ws || a.ws_floats < needp || HN % 160) return (int)hipErrorInvalidValue;
        dim3 gp(gx, ny);
#define RSP_WIDE(MODE_, RT_) LAUNCH((rsp_wide_kernel<KC, MODE_, RT_>), gp, dim3(256), ldsp, st, p, ntiles)
        if (which == 0) { if (rtv == 2) RSP_WIDE(0, 2); else RSP_WIDE(0, 1); }
        else { if (rtv == 2) RSP_WIDE(1, 2); else RSP_WIDE(1, 1); }
#undef RSP_WIDE
        if (which == 0) launch_reduce(0, a.ws, gx, HN, a.s0, nullptr, 0, 0, 0, 0, st);
        else if (a.s1 == a.s0 + HN) launch_reduce(0, a.ws, gx, 2 * HN, a.s0, nullptr, 0, 0, 0, 0, st);
        else {
          const long long delta = a.s1 - a.s0;
          if (delta > 2147483647LL || delta < -2147483647LL) return (int)hipErrorInvalidValue;
          launch_reduce(1, a.ws, gx, 2 * HN, a.s0, nullptr, HN, (int)delta, 1, 0, st);
        }
        return launch_status();
      }
    }
    if constexpr (KC == 160 || KC == 320 || KC == 192 || KC == 384) {
      // one-shot form (rsc1.cuh): a workgroup keeps its weight slice (DMA, once) and walks row tiles; every operand requested in one burst
      const int rt1 = g_opt[MPMAE_OPT_RSC1];
      if (rt1 > 0) {
        const int rtv = (rt1 >= 2 && !(KC >= 320 && which == 0 && !a.ln_done)) ? 2 : 1;      // (LayerNorm mode at C = 320 with two row tiles per wave spills)
        const int ntiles = cdiv(a.M, 64 * rtv);
        // measured (tools/probes/rs1_probe.py, incl. the fold): C = 160: which 0 24.0 (slice 128) / 27.4 (64) vs 29.9 us chunked, which 1 25.7 (128) / 19.1 (64) vs
        // 23.8; C = 320: which 0 25.3 vs 30.8, which 1 18.7 vs 22.5; without the fold launch (atomics) 22.5 / 16.4 at C = 320
        // (tiny widths, BASELINE config 4: C = 192 in 96-column slices, C = 384 in 64-column slices)
        const int cpsv = (KC == 192) ? 96 : (KC == 384) ? 64 : (0 /* RSC1_CPS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ > 0 ? 0 /* RSC1_CPS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ : ((KC == 160 && which == 0) ? 128 : 64));
        if (cpsv != 64 && !(KC == 160 && cpsv == 128) && !(KC == 192 && cpsv == 96)) return (int)hipErrorInvalidValue;
        const int ny = HN / cpsv;
        const int wgs = 0 /* RSC1_WGS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ > 0 ? 0 /* RSC1_WGS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ : 3 * ps_num_cus();
        const int gxmax = wgs / ny > 0 ? wgs / ny : 1;
        const int tpw = cdiv(ntiles, gxmax), gx = cdiv(ntiles, tpw);
        const size_t lds1s = (size_t)cpsv * KC * 2 + (size_t)4 * 2 * cpsv * 4 + (size_t)2 * KC * 4;
        const size_t need1 = (size_t)gx * HN * (which == 1 ? 2 : 1);
        // few workgroup rows (38-76 here, 76-304 row blocks in the chunked kernels): the column statistics go straight into s0 / s1 with
        // no-return float atomics - no slab, no fold launch (not in the reproducible mode)
        const int atmax = g_opt[MPMAE_OPT_DET] > 0 ? 0 : (g_opt[MPMAE_OPT_RSC_ATOMIC] > g_opt[MPMAE_OPT_RSC1_ATOMIC] ? g_opt[MPMAE_OPT_RSC_ATOMIC] : g_opt[MPMAE_OPT_RSC1_ATOMIC]);
        const bool at1 = gx <= atmax && a.s0 && (which == 0 || a.s1);
        if (at1) { p.s0a = a.s0; p.s1a = a.s1; }
        if ((!at1 && (!a.ws || a.ws_floats < need1)) || lds1s > 160 * 1024 - 512) return (int)hipErrorInvalidValue;
        p.perwave = 1;
        dim3 g1(gx, ny);
#define RSC_WIDE1(MODE_, RT_, CPS_) do { \
          static size_t cur = 64 * 1024; \
          if (lds1s > cur) { if (hipFuncSetAttribute((const void*)rsc_wide1_kernel<KC, MODE_, RT_, CPS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1s) != hipSuccess) return (int)hipGetLastError(); cur = lds1s; } \
          LAUNCH((rsc_wide1_kernel<KC, MODE_, RT_, CPS_>), g1, dim3(256), lds1s, st, p, ntiles); } while (0)
#define RSC_WIDE1_M(MODE_) do { \
          if constexpr (KC == 192) { if (rtv == 2) RSC_WIDE1(MODE_, 2, 96); else RSC_WIDE1(MODE_, 1, 96); } \
          else if constexpr (KC == 160) { if (cpsv == 128) { if (rtv == 2) RSC_WIDE1(MODE_, 2, 128); else RSC_WIDE1(MODE_, 1, 128); } \
                                     else { if (rtv == 2) RSC_WIDE1(MODE_, 2, 64); else RSC_WIDE1(MODE_, 1, 64); } } \
          else { if (rtv == 2) RSC_WIDE1(MODE_, 2, 64); else RSC_WIDE1(MODE_, 1, 64); } } while (0)
        if (which == 0) { if (a.ln_done) RSC_WIDE1_M(2); else RSC_WIDE1_M(0); }
        else RSC_WIDE1_M(1);
#undef RSC_WIDE1_M
#undef RSC_WIDE1
        if (!at1) {
          if (which == 0) launch_reduce(0, a.ws, gx, HN, a.s0, nullptr, 0, 0, 0, 0, st);
          else if (a.s1 == a.s0 + HN) launch_reduce(0, a.ws, gx, 2 * HN, a.s0, nullptr, 0, 0, 0, 0, st);
          else {
            const long long delta = a.s1 - a.s0;
            if (delta > 2147483647LL || delta < -2147483647LL) return (int)hipErrorInvalidValue;
            launch_reduce(1, a.ws, gx, 2 * HN, a.s0, nullptr, HN, (int)delta, 1, 0, st);
          }
        }
        return launch_status();
      }
    }
    const size_t need = (size_t)rowblocks * HN * (which == 1 ? 2 : 1);
    // MPMAE_OPT_RSC_ATOMIC = largest row-block count whose statistics are accumulated with atomics instead of slab + fold launch
    const bool at = rowblocks <= g_opt[MPMAE_OPT_RSC_ATOMIC] && a.s0 && (which == 0 || a.s1);
    if (at) { p.s0a = a.s0; p.s1a = a.s1; }
    if ((!at && (!a.ws || a.ws_floats < need)) || lds > 160 * 1024 - 512) return (int)hipErrorInvalidValue;
    dim3 g(rowblocks, nsplit);
    if (which == 0) {
      static size_t cur = 64 * 1024;
      if (lds > cur) { if (hipFuncSetAttribute((const void*)rsc_wide_kernel<KC, 0, RT, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; }
      LAUNCH((rsc_wide_kernel<KC, 0, RT, NC>), g, dim3(256), lds, st, p, HN, cps);
      if (!at) launch_reduce(0, a.ws, rowblocks, HN, a.s0, nullptr, 0, 0, 0, 0, st);
    } else {
      static size_t cur = 64 * 1024;
      if (lds > cur) { if (hipFuncSetAttribute((const void*)rsc_wide_kernel<KC, 1, RT, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError(); cur = lds; }
      LAUNCH((rsc_wide_kernel<KC, 1, RT, NC>), g, dim3(256), lds, st, p, HN, cps);
      if (at) {
      } else if (a.s1 == a.s0 + HN) {        // adjacent outputs (the engine's layout): one launch over [2*HN]
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
    if constexpr (KC == 160) {
      // ring-pipelined backward (rsn3.cuh): weight slabs by DMA into a three-slot ring, rows two chunks ahead, one bare barrier per chunk
      const int r3 = g_opt[MPMAE_OPT_RSN3];
      if (r3 > 0 && which == 5 && pf && !dzr) {
        // (two K-groups per row tile - half as long a dependent chain per wave, 126 KB of ring: measured SLOWER, 42.5 vs 32.0 us, and the 640-thread
        //  instantiation spills; the kernel keeps the template parameter, the launcher does not offer it)
        const int nkg = 1, nrt = r3 == 4 ? 4 : 5;
        const int rb3 = cdiv(a.M, 16 * nrt);
        const size_t lds3 = (size_t)3 * nkg * KC * 64 * 2 + (size_t)(2 * KC + 2 * HN + 8 + KC) * 4;
        if (!a.ws || a.ws_floats < (size_t)rb3 * 2 * KC) return (int)hipErrorInvalidValue;
#define RSN3(NRT_, NKG_) do { \
          static bool attr = false; \
          if (!attr) { if (hipFuncSetAttribute((const void*)rsn3_bwd_kernel<KC, NRT_, NKG_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3) != hipSuccess) return (int)hipGetLastError(); attr = true; } \
          LAUNCH((rsn3_bwd_kernel<KC, NRT_, NKG_>), dim3(rb3), dim3(64 * NRT_ * NKG_), lds3, st, p); } while (0)
        if (nrt == 5) RSN3(5, 1); else RSN3(4, 1);
#undef RSN3
        const long long delta = a.s1 - a.s0;
        if (delta > 2147483647LL || delta < -2147483647LL) return (int)hipErrorInvalidValue;
        if (a.defer_fold) *a.defer_fold = MpmaeFoldDesc{a.ws, rb3, 2 * KC, a.s0, KC, (int)delta, 1};
        else launch_reduce(1, a.ws, rb3, 2 * KC, a.s0, nullptr, KC, (int)delta, 1, 0, st);
        return launch_status();
      }
    }
    if constexpr (KC == 40 || KC == 80) {
      // persistent burst-load form (rsp.cuh): which 4 without h recomputation, which 5 with dz recomputation, single GRN group
      const int rp = g_opt[MPMAE_OPT_RSP];
      // measured (tools/probes/rsp_narrow_probe.py, profiles/r05/rsp_narrow_probe.txt): only which 5 at C = 40 gains (88.6 -> 70 us at 2 workgroups per CU);
      // which 4 ties at C = 40 (36 us = 4.2 TB/s: HBM + GELU, not latency) and both lose at C = 80 (28 vs 25, 52 vs 43 us: 55-127 KB of resident weights
      // leave one or two workgroups per CU). MPMAE_OPT_RSP_NARROW: bit 0 = which 4 at C = 40, bit 1 = which 5 at C = 40, bit 2 / 3 = the same at C = 80
      const int nbit = (KC == 80 ? 4 : 1) << (which == 5 ? 1 : 0);
      if (rp > 0 && (g_opt[MPMAE_OPT_RSP_NARROW] & nbit) && pf && !a.dn_y && ((which == 4 && !dzr) || (which == 5 && dzr))) {      // (the fused downsample LayerNorm lives in rsc_narrow)
        const int nwv = g_opt[MPMAE_OPT_RSP_NWV] > 0 ? g_opt[MPMAE_OPT_RSP_NWV] : ((KC == 80 && which == 5) ? 8 : 4);
        if (nwv != 4 && nwv != 8) return (int)hipErrorInvalidValue;
        const int ntiles = cdiv(a.M, 16 * nwv);
        constexpr int NPv = ((KC + 15) / 16) * 16, KP2v = ((KC + 31) / 32) * 32;
        const bool wg = a.wg_ws != nullptr;      // pwconv1's weight gradient inside the kernel (C = 40, 4 waves: 64-row staging tiles of dh and x-hat)
        if (wg && (KC != 40 || which != 5 || nwv != 4 || !a.wg_rows)) return (int)hipErrorInvalidValue;
        const size_t ldsn = ((size_t)NPv * (HN + RSC_PAD) + (which == 5 ? (size_t)HN * (KP2v + RSC_PAD) : 0)) * 2 + (size_t)(2 * HN + NPv + 8) * 4 +
                            (which == 5 ? (size_t)nwv * 2 * NPv * 4 : 0) + (wg ? (size_t)16 * nwv * (tn2_ld(HN) + tn2_ld(NPv)) * 2 : 0);
        const int wgs = 0 /* RSP_NWGS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ > 0 ? 0 /* RSP_NWGS: launch shape frozen in round 6 (swept flat, profiles/r05/option_sweep.txt) */ : (int)((160 * 1024) / ldsn > 2 ? 2 : (160 * 1024) / ldsn) * ps_num_cus();
        const int gx = ntiles < wgs ? ntiles : wgs;
        if (ldsn > 160 * 1024 - 512 || (which == 5 && (!a.ws || a.ws_floats < (size_t)gx * 2 * KC))) return (int)hipErrorInvalidValue;
#define RSP_NARROW(MODE_, NWV_) do { \
          static size_t cur = 64 * 1024; \
          if (ldsn > cur) { if (hipFuncSetAttribute((const void*)rsp_narrow_kernel<KC, MODE_, NWV_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn) != hipSuccess) return (int)hipGetLastError(); cur = ldsn; } \
          LAUNCH((rsp_narrow_kernel<KC, MODE_, NWV_>), dim3(gx), dim3(64 * NWV_), ldsn, st, p, ntiles); } while (0)
        if (wg) { if (a.wg_ws_floats < (size_t)gx * ((size_t)HN * KC + HN)) return (int)hipErrorInvalidValue; *a.wg_rows = gx; }
        if (which == 4) { if (nwv == 8) RSP_NARROW(0, 8); else RSP_NARROW(0, 4); }
        else {
          if constexpr (KC == 40) {
            if (wg) {
              static size_t curw = 64 * 1024;
              if (ldsn > curw) { if (hipFuncSetAttribute((const void*)rsp_narrow_kernel<40, 1, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn) != hipSuccess) return (int)hipGetLastError(); curw = ldsn; }
              LAUNCH((rsp_narrow_kernel<40, 1, 4, true>), dim3(gx), dim3(256), ldsn, st, p, ntiles);
            } else if (nwv == 8) RSP_NARROW(1, 8); else RSP_NARROW(1, 4);
          } else {
            if (nwv == 8) RSP_NARROW(1, 8); else RSP_NARROW(1, 4);
          }
          const long long delta = a.s1 - a.s0;        // s0 = dgamma, s1 = dbeta (same flat gradient buffer)
          if (delta > 2147483647LL || delta < -2147483647LL) return (int)hipErrorInvalidValue;
          if (a.defer_fold) *a.defer_fold = MpmaeFoldDesc{a.ws, gx, 2 * KC, a.s0, KC, (int)delta, 1};
          else launch_reduce(1, a.ws, gx, 2 * KC, a.s0, nullptr, KC, (int)delta, 1, 0, st);
        }
#undef RSP_NARROW
        return launch_status();
      }
    }
    if (a.wg_ws) return (int)hipErrorInvalidValue;      // (the fused weight gradient only exists in the persistent C = 40 backward kernel above)
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

int mpmae_rs_wgrad_fold(int A, int B, const float* slabs, int rows, const float* v0, const float* v1, float* dW, float* db, mpmae_stream_t s) {
  if (A < 1 || B < 32 || !slabs || rows < 1 || !v0 || !v1 || !dW || !db) return (int)hipErrorInvalidValue;
  int R = rows / 16 < 1 ? 1 : (rows / 16 > 32 ? 32 : rows / 16);
  if (g_opt[MPMAE_OPT_DET] > 0) R = 1;
  LAUNCH(wg_fold_kernel, dim3(cdiv((long long)A * B, 64), R), dim3(256), 0, S_(s), slabs, rows, A, B, v0, v1, dW, db);
  RET();
}

int mpmae_rs(int which, const MpmaeRsArgs* a, mpmae_stream_t s) {
  if (!a || which < 0 || which > 6) return (int)hipErrorInvalidValue;
  if (a->C == 160 && a->H == 640) return launch_rsc<160, 1, 32, 64, 1, 3>(which, *a, S_(s));      // 32-column chunks: the N range splits 4 ways (27.7 -> 24.1 us vs 64)
  if (a->C == 320 && a->H == 1280) return launch_rsc<320, 1, 32, 32>(which, *a, S_(s));
  // ConvNeXtV2-tiny widths (BASELINE config 4: 96 / 192 / 384; 768 stays on the tiled GEMMs)
  if (a->C == 96 && a->H == 384) return launch_rsc<96, 2, 64, 64, 1, 3>(which, *a, S_(s));
  if (a->C == 192 && a->H == 768) return launch_rsc<192, 1, 32, 64, 1, 3>(which, *a, S_(s));
  if (a->C == 384 && a->H == 1536) return launch_rsc<384, 1, 32, 32, 1, 3>(which, *a, S_(s));
  if (which == 2 || which == 3) return (int)hipErrorInvalidValue;      // (the resident-weights kernels on materialised z / dh: removed in round 4)
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
  return (int)hipErrorInvalidValue;
}


// ------------------------------------------------------------------------------------------
// persistent per-sample stage kernels (ps.cuh)
// ------------------------------------------------------------------------------------------
int ps_num_cus() {
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
  if (!a.x_in || !a.g.vis || !a.g.inv || !a.sync || a.sync_words < 4 || a.nblk < 1 || a.nblk > MPMAE_PS_MAXBLK || a.ng < 1 || a.ng > 16) return (int)hipErrorInvalidValue;
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

int mpmae_ps_fwd(const MpmaePsArgs* a, mpmae_stream_t s) {
  if (!a) return (int)hipErrorInvalidValue;
  if (a->C == 160 && a->g.S == 2) return launch_ps_fwd<160, 2>(*a, S_(s));
  if (a->C == 320 && a->g.S == 1) return launch_ps_fwd<320, 1>(*a, S_(s));
  return (int)hipErrorInvalidValue;
}

