// Submanifold depthwise 7x7 (MinkowskiDepthwiseConvolution, convnextv2_sparse.py:37-39) on the MATRIX cores, S = 8 / 4 (round 4).
//
// A depthwise filter differs per channel, so an MFMA cannot share its filter operand across channels - but it can share it across
// PATCHES: for one channel, the S x S outputs of a visible patch are a fixed linear map of the (S + 6)^2 window around it, the same
// map for every patch of every sample. So per channel
//      D[i = output point of the patch][n = patch] = sum_k  A_c[i][k = window point] * B_c[k][n]
// with A_c the (2-D Toeplitz) matrix of channel c's 49 taps and B_c the windows of 16 patches in channel-PLANAR form. 49 of the
// 128 products of a row are real taps: 38 % of 2.5 PFLOP/s against the VALU kernels' 22-25 TFLOP/s (dwconv6.cuh:
// 4 vector instructions per multiply-add).
//
//   S = 8: an MFMA row block is two output rows (16 points), its k range the 8 window rows x 16 columns they see = 4 k-steps of
//          (2 window rows x 16 columns); a patch is 4 row blocks yo, and k-step kk of block yo reads window rows 2 (yo + kk) - 3 ..:
//          the SAME B fragment serves every (yo, kk) with yo + kk = t, so a channel is 7 fragments (t = 0..6) for 16 MFMAs.
//   S = 4: the row block is the whole 4 x 4 patch, k = 10 window rows x 3 pieces of 4 columns = 30 pieces in 4 k-steps of 8.
//
// Planar form: the activations are channels-last rows [row][C] in HBM. A workgroup = (sample, chunk of CCH channels: all 40 at
// atto) loads the sample's visible patches once (16-byte vectors: 8 channels of one point, whole contiguous patches), transposes
// 4 points x 8 channels in registers (v_perm_b32) and writes 8-byte PIECES (4 consecutive columns of one channel) to LDS as
// [channel][patch row y][slot][columns]: the 16 patches of a fragment are 16 consecutive granules, so a ds_read_b64 of 32 lanes
// covers 64 distinct banks. A window row is gathered per lane from the 3 x 3 neighbour patches' lines (masked neighbours -> a zero
// granule at slot = keep); the k order inside a step is chosen so that every lane reads whole pieces: lane group g = (rr, h) reads
// row rr of the step, pieces {left.1, centre.1} (h = 0) or {centre.0, right.0} (h = 1) at S = 8. The Toeplitz fragments of a
// wave's 4 channels are built once from a bf16 tap table and stay in registers (64 VGPRs); the taps are ROUNDED TO bf16 (the
// activations already are; tolerance in the tests). A wave owns a channel quad.
//
// Passes. The patches are worked in groups of 16 (the MFMA's n), at S = 8 two row-block PAIRS per group (fragments t = 0..4 and
// 2..6: 32 accumulator registers instead of 64); a last group of <= 4 patches (19 = 16 + 3 visible patches at mask ratio 0.6) runs
// as ONE pass with n = (patch, row block) - 4 MFMAs per channel instead of 16 for 3 useful columns of 16.
//
// Output. A lane's accumulators are 4 consecutive columns x 4 channels: 8 bytes per point. Stored like that (and the residual
// operand of the data gradient loaded like that) the kernel spent 20 of its 38 us in the stores and 8 in those loads (probe:
// phases switched off one by one, profiles/r04/dw_mfma_probes.txt) - an 8-byte piece per 80-byte row is a request per lane.
// So every pass goes through a channels-last STAGING tile in LDS (patch stride + 16 bytes: bank spread): the residual rows are
// loaded as whole 16-byte vectors by all threads, the epilogue adds / masks / rounds in place, and the tile leaves as contiguous
// runs (32 points x C channels = 2.5 KB per patch and pass at atto stage 0).
// grid = (N, C / CCH), block = 64 * CCH / 4 (one workgroup per CU at CCH = 40, S = 8: 147 KB of LDS).
#pragma once
#include "dwconv6.cuh"

__device__ __forceinline__ uint32_t dwm_dw(const uint4& v, int d) { return d == 0 ? v.x : d == 1 ? v.y : d == 2 ? v.z : v.w; }   // d folds under unrolling

// LDS-only barrier: __syncthreads() carries an s_waitcnt vmcnt(0) - in the pass loop that is a wait for the previous pass's global
// stores (and for the fill loads in front of the Toeplitz build) at every one of 2-3 barriers per pass
__device__ __forceinline__ void dwm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// cycle stamps of the phases (tools/probes/dwm_stamps.hip compiles this file with -DDWM_STAMPS; the library never does)
#ifdef DWM_STAMPS
__device__ unsigned long long dwm_stamp_buf[1 << 20];
#define DWM_STAMP(k) do { if (lane == 0) dwm_stamp_buf[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wave) * 48 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DWM_STAMP(k) do { } while (0)
#endif

template <int S, int CCH> struct DwMfma {
  static constexpr int NQ = CCH / 4, NT = 64 * NQ, NV = CCH / 8;
  static constexpr int NPOS = S == 8 ? 32 : 16;               // points of a patch in a full-group pass
  static constexpr int PSTR = NPOS * CCH * 2 + 16;            // staging bytes per patch (full-group pass)
  static constexpr int PSTR_T = S * S * CCH * 2 + 16;         // staging bytes per patch of the tail pass (whole patches, <= 4)
  static constexpr int STG = 16 * PSTR;
  static_assert(4 * PSTR_T <= STG, "tail pass fits the staging tile");
  static size_t lds(int keep) { return (size_t)CCH * S * (keep + 1) * S * 2 + 16 * NV + STG + 16 + (size_t)CCH * 64 * 2 + (size_t)(keep + 1) * 9 * 4 + 64 * 4 + 64 * 4 + (size_t)(keep + 1) * S * S; }
};

// The MFMAs of one pass, channel by channel: the 4-5 fragments of a channel (10 reads, 20 registers), then its 4-8 MFMAs. Measured
// with the cycle stamps (tools/probes/dwm_stamps.hip): fragment-major order (8 reads of one fragment for all 4 channels per step) and
// a second fragment set in flight take the same 2 200-3 500 cycles per pass - with 10 waves on 4 SIMDs the phase is bound by the
// matrix pipe of the SIMDs that carry three waves (3 x 32 MFMAs x 16 cycles) plus the offset arithmetic - and the variants that
// need more registers spill at the 168-register budget of a 10-wave workgroup (a spilled residual vector = a wait for its load
// in front of the MFMAs: data gradient 24 -> 34 us).
// Tail pass: n = (patch, row block) - step tt multiplies Toeplitz step tt into the single accumulator of the lane's own row block.
template <int S, int CCH>
__global__ __launch_bounds__(64 * (CCH / 4)) void dwconv7_mfma_kernel(const DwP p) {
  static_assert(S == 8 || S == 4, "patch side");
  static_assert(CCH % 8 == 0, "channel chunk");
  using D = DwMfma<S, CCH>;
  constexpr int NT = D::NT, NV = D::NV, PPR = S / 4, GB = S * 2;
  constexpr int NPAIR = S == 8 ? 2 : 1;               // passes per full group (S = 8: two pairs of row blocks)
  constexpr int NYO = S == 8 ? 2 : 1;                 // row blocks per full-group pass
  constexpr int NB = S == 8 ? 5 : 4;                  // B fragments per channel and full-group pass
  extern __shared__ __attribute__((aligned(16))) unsigned char dwm_smem[];
  const int keep = p.g.keep, SL = keep + 1, G = p.g.grid, L = G * G;
  const int ROWB = SL * GB, PLB = S * ROWB;           // bytes of a (channel, y) line / of a channel plane
  unsigned char* data = dwm_smem;
  // (the planes of channel octet o start 16 o bytes late: the fill's 16-lane store groups span the octets of one point, and planes that
  // are a multiple of 128 bytes apart put all of them on one bank - a 5-way conflict on 24 stores per lane)
  unsigned char* stg = dwm_smem + (size_t)CCH * PLB + 16 * NV;
  bf16_t* wt = reinterpret_cast<bf16_t*>(stg + D::STG + 16);
  int* nbt = reinterpret_cast<int*>(stg + D::STG + 16 + CCH * 64 * 2);
  int* invl = nbt + (keep + 1) * 9;                   // [64] slot of each patch of this sample (L <= 64)
  int* visl = invl + 64;                              // [64] patch of each slot (keep < 64: wave 0 writes and reads both tables)
  unsigned char* actl = reinterpret_cast<unsigned char*>(visl + 64);      // [keep * S * S] activity bytes of the sample's rows
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = blockIdx.x, c0 = blockIdx.y * CCH, C = p.C;
  const bf16_t* x = reinterpret_cast<const bf16_t*>(p.x);
  DWM_STAMP(0);

  // ---- prologue. Every global load is issued up front and none depends on another (geometry: the sample's inv / vis tables go to
  // LDS, the neighbour table is built from there). The small operands are written first (each wave its own taps), and the Toeplitz
  // fragments are built while the sample's rows are still in flight; then the register transposes and the piece stores. Fill task = (slot, y, piece, octet), octet fastest: a wave reads whole contiguous lines.
  const int li = lane & 15, lg = lane >> 4;
  const int q = wave;
  bf16x8_t A[4][4];
  {
    const int tasks = keep * S * PPR * NV;
    constexpr int U = S == 8 ? 3 : 1;                  // 19 patches: 1520 / 380 tasks on 640 threads
    constexpr int WU = 4;                              // a wave stages the taps of ITS quad: 4 channels x 64 entries on 64 lanes
    const int inv_v = p.g.inv[n * L + (tid < L ? tid : 0)];
    const int vis_v = p.g.vis[n * keep + (tid < keep ? tid : 0)];
    float wv[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int i = u * 64 + lane, c = 4 * q + (i & 3), k = i >> 2;
      int kh = k / 7, kw = k - kh * 7;
      if (p.flip) { kh = 6 - kh; kw = 6 - kw; }
      wv[u] = p.w[(k < 49 ? kh * p.s_kh + kw * p.s_kw : 0) + (c0 + c) * p.s_c];
    }
    const int nact = keep * S * S / 4;
    const uint32_t act_v = *(p.act ? reinterpret_cast<const uint32_t*>(p.act + (size_t)n * keep * (S * S)) + (tid < nact ? tid : 0)
                                   : reinterpret_cast<const uint32_t*>(p.x));
    // (the LAST batch of rows is requested behind the Toeplitz build: with all three in flight across it the kernel spilled)
    constexpr int UE = U;
    uint4 v[U][4];
    int dst[U];
    auto fill_ld = [&](int u) {
      const int tk = u * NT + tid;
      const int tc = tk < tasks ? tk : 0;
      const int o = tc % NV, r1 = tc / NV, xp = r1 % PPR, r2 = r1 / PPR, y = r2 % S, slot = r2 / S;
      const bf16_t* src = x + ((size_t)(n * keep + slot) * (S * S) + y * S + 4 * xp) * C + c0 + 8 * o;
      return src;
    };
#pragma unroll
    for (int u = 0; u < UE; ++u) {
      const bf16_t* src = fill_ld(u);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[u][r] = *reinterpret_cast<const uint4*>(src + (size_t)r * C);
    }
    DWM_STAMP(1);
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int i = u * 64 + lane, c = 4 * q + (i & 3), k = i >> 2;
      wt[c * 64 + k] = k < 49 ? f2bf(wv[u]) : (bf16_t)0;
    }
    // geometry, activity bytes, zero granules. invl / visl are written AND read (neighbour table, below) by wave 0 only (L, keep < 64)
    if (tid < L) invl[tid] = inv_v;
    if (tid < keep) visl[tid] = vis_v;
    if (tid < nact) reinterpret_cast<uint32_t*>(actl)[tid] = p.act ? act_v : 0x01010101u;
    for (int i = NT + tid; i < nact; i += NT)
      reinterpret_cast<uint32_t*>(actl)[i] = p.act ? reinterpret_cast<const uint32_t*>(p.act + (size_t)n * keep * (S * S))[i] : 0x01010101u;
    for (int i = tid; i < CCH * S * PPR; i += NT) {                                      // zero granules (slot = keep)
      const int xp = i % PPR, r1 = i / PPR, y = r1 % S, c = r1 / S;
      *reinterpret_cast<uint2*>(data + c * PLB + 16 * (c >> 3) + y * ROWB + keep * GB + xp * 8) = make_uint2(0u, 0u);
    }
    // the tap table slice and invl / visl are wave-private: in-order LDS, a wait instead of a barrier - the Toeplitz build (4 400
    // cycles, issue-bound) runs while the sample's rows are in flight and no wave waits for another before the fill barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DWM_STAMP(2);

    // ---- Toeplitz fragments of this wave's channel quad: A[cc][kk], lane (i = lane & 15, g = lane >> 4) holds A[i][8 g .. 8 g + 7]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      int kv = kk;
      asm volatile("" : "+v"(kv));                  // the tap indices of step kk are worked out HERE (hoisted to the top they cost 32 registers across the build)
      int idx[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = e >> 2, e4 = e & 3;
        int ky, kx;
        bool ok = true;
        if constexpr (S == 8) {
          const int yl = li >> 3, xo = li & 7, h = lg & 1, rr = lg >> 1;
          ky = 2 * kv + rr - yl;
          const int xin = h == 0 ? (j == 0 ? -4 + e4 : 4 + e4) : (j == 0 ? e4 : 8 + e4);
          kx = xin - xo + 3;
        } else {
          const int yl = li >> 2, xo = li & 3;
          const int pi = kv * 8 + lg * 2 + j, row = pi / 3, col = pi - row * 3;
          ok = pi < 30;
          ky = row - yl;
          kx = 4 * (col - 1) + e4 - xo + 3;
        }
        idx[e] = (ok && ky >= 0 && ky < 7 && kx >= 0 && kx < 7) ? ky * 7 + kx : 63;
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const bf16_t* wc = wt + (4 * q + cc) * 64;
        uint32_t d[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) d[e >> 1] = (uint32_t)wc[idx[e]] | ((uint32_t)wc[idx[e + 1]] << 16);
        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));      // packed HERE: hipcc sank the packing to the first use and kept 128 halves live
        A[cc][kk] = __builtin_bit_cast(bf16x8_t, make_uint4(d[0], d[1], d[2], d[3]));
        __builtin_amdgcn_sched_barrier(0);        // 8 two-byte reads in flight (hipcc gives each its own register; 32 in flight take the same 4 400 cycles: the build is issue-bound)
      }
    }
#pragma unroll
    for (int u = UE; u < U; ++u) {
      const bf16_t* src = fill_ld(u);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[u][r] = *reinterpret_cast<const uint4*>(src + (size_t)r * C);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tk = u * NT + tid;
      const int o = tk % NV, r1 = tk / NV, xp = r1 % PPR, r2 = r1 / PPR, y = r2 % S, slot = r2 / S;
      dst[u] = tk < tasks ? (8 * o) * PLB + 16 * o + y * ROWB + slot * GB + xp * 8 : -1;
    }
    DWM_STAMP(3);
    // ---- neighbour table from the LDS copies of vis / inv (masked / outside -> the zero granule; row `keep`: the idle lanes' patch)
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
    // ---- register transposes and piece stores of the first batch, then (more than 3 tasks per thread: rare) the rest
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (dst[u] >= 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
          const uint32_t lo = __builtin_amdgcn_perm(dwm_dw(v[u][1], j >> 1), dwm_dw(v[u][0], j >> 1), sel);
          const uint32_t hi = __builtin_amdgcn_perm(dwm_dw(v[u][3], j >> 1), dwm_dw(v[u][2], j >> 1), sel);
          *reinterpret_cast<uint2*>(data + dst[u] + j * PLB) = make_uint2(lo, hi);
        }
      }
    }
    for (int tk = U * NT + tid; tk < tasks; tk += NT) {
      const int o = tk % NV, r1 = tk / NV, xp = r1 % PPR, r2 = r1 / PPR, y = r2 % S, slot = r2 / S;
      const bf16_t* src = x + ((size_t)(n * keep + slot) * (S * S) + y * S + 4 * xp) * C + c0 + 8 * o;
      uint4 w[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const uint4*>(src + (size_t)r * C);
      const int d0 = (8 * o) * PLB + 16 * o + y * ROWB + slot * GB + xp * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
        const uint32_t lo = __builtin_amdgcn_perm(dwm_dw(w[1], j >> 1), dwm_dw(w[0], j >> 1), sel);
        const uint32_t hi = __builtin_amdgcn_perm(dwm_dw(w[3], j >> 1), dwm_dw(w[2], j >> 1), sel);
        *reinterpret_cast<uint2*>(data + d0 + j * PLB) = make_uint2(lo, hi);
      }
    }
    DWM_STAMP(4);
  }
  dwm_lds_barrier();
  DWM_STAMP(5);

  float b4[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) b4[cc] = p.bias[c0 + 4 * q + cc];
  }
  bf16_t* out = reinterpret_cast<bf16_t*>(p.out);
  const bf16_t* add = reinterpret_cast<const bf16_t*>(p.add);
  const unsigned add_m = opaque_mask(add != nullptr);
  const bool has_add = add != nullptr;

  // pass list: full groups of 16 patches (NPAIR passes each), then one tail pass for a last group of <= 4 patches (S = 8)
  const int rem = keep & 15;
  const int nfull = (S == 8 && rem > 0 && rem <= 4) ? keep >> 4 : (keep + 15) >> 4;
  const int tailn = (S == 8 && rem > 0 && rem <= 4) ? rem : 0;
  const int npass = nfull * NPAIR + (tailn > 0 ? 1 : 0);
  constexpr int AU = (16 * D::NPOS * NV + NT - 1) / NT;
  static_assert(AU <= 4, "residual vectors per thread");
  // residual rows of pass `ps` as whole 16-byte vectors (named registers, not an array: hipcc kept an av[AU] array in scratch - with a
  // wait for the loads in front of the MFMAs). They are requested one pass AHEAD, in front of the previous pass's copy-out stores:
  // vmcnt retires in order, so a load issued behind those stores could only be awaited together with them.
  uint4 av0 = make_uint4(0u, 0u, 0u, 0u), av1 = av0, av2 = av0, av3 = av0;
  auto av_load = [&](int ps) {
    const bool tl = ps >= nfull * NPAIR;
    const int ng = tl ? nfull : ps / NPAIR, yp = tl ? 0 : ps - ng * NPAIR;
    const int npatch = tl ? tailn : min(16, keep - 16 * ng), npp = tl ? S * S : D::NPOS, posbase = tl ? 0 : D::NPOS * yp;
    const int nvec = npatch * npp * NV;
    auto src = [&](int u) -> const uint4* {
      const int vi = u * NT + tid, vc = vi < nvec ? vi : 0;
      const int o = vc % NV, r1 = vc / NV, pos = r1 % npp, pi = r1 / npp;
      return reinterpret_cast<const uint4*>(add + ((size_t)(n * keep + 16 * ng + pi) * (S * S) + posbase + pos) * C + c0 + 8 * o);
    };
    av0 = *src(0);
    if constexpr (AU > 1) av1 = *src(1);
    if constexpr (AU > 2) av2 = *src(2);
    if constexpr (AU > 3) av3 = *src(3);
  };
  if (has_add) av_load(0);
#pragma unroll 1
  for (int ps = 0; ps < npass; ++ps) {
    const bool is_tail = ps >= nfull * NPAIR;
    const int ng = is_tail ? nfull : ps / NPAIR, yp = is_tail ? 0 : ps - ng * NPAIR;
    const int npatch = is_tail ? tailn : min(16, keep - 16 * ng);
    const int npp = is_tail ? S * S : D::NPOS;                 // points per patch in this pass
    const int pstr = is_tail ? D::PSTR_T : D::PSTR;
    const int posbase = is_tail ? 0 : D::NPOS * yp;
    const int pidx = is_tail ? li >> 2 : li;                   // this lane's patch within the pass
    const int yo_l = li & 3;                                   // ... and its row block (tail pass)
    const bool valid = pidx < npatch;
    const int s = 16 * ng + pidx;
    const int nvec = npatch * npp * NV;

    uint32_t actw[NYO];
#pragma unroll
    for (int yo = 0; yo < NYO; ++yo) {
      const int yoa = is_tail ? yo_l : NYO * yp + yo;
      const int pos0 = S == 8 ? (2 * yoa + (lg >> 1)) * 8 + 4 * (lg & 1) : lg * 4;
      actw[yo] = *reinterpret_cast<const uint32_t*>(actl + (valid ? s : 0) * (S * S) + pos0);
    }
    int nb[3][3];
#pragma unroll
    for (int k = 0; k < 9; ++k) nb[k / 3][k % 3] = nbt[(valid ? s : keep) * 9 + k];
    // ---- byte offsets (within a channel plane) of the two pieces of every B fragment of this pass
    int off[NB][2];
#pragma unroll
    for (int tt = 0; tt < NB; ++tt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (S == 8) {
          const int t = is_tail ? yo_l + tt : 2 * yp + tt;        // tail: k-step tt of the lane's own row block (tt < 4)
          const int h = lg & 1, rr = lg >> 1;
          const int yin = 2 * t - 3 + rr;
          const int dy = yin < 0 ? 0 : (yin > 7 ? 2 : 1);
          const int dx = h == 0 ? (j == 0 ? 0 : 1) : (j == 0 ? 1 : 2);
          const int r0 = dx == 0 ? nb[0][0] : (dx == 1 ? nb[0][1] : nb[0][2]);
          const int r1 = dx == 0 ? nb[1][0] : (dx == 1 ? nb[1][1] : nb[1][2]);
          const int r2 = dx == 0 ? nb[2][0] : (dx == 1 ? nb[2][1] : nb[2][2]);
          const int sl = dy == 0 ? r0 : (dy == 1 ? r1 : r2);
          off[tt][j] = (yin & 7) * ROWB + sl * GB + (h ? 0 : 8);
        } else {
          const int pi = tt * 8 + lg * 2 + j, row = pi / 3, col = pi - row * 3;
          const int yin = row - 3;
          const int dy = yin < 0 ? 0 : (yin > 3 ? 2 : 1);
          const int r0 = col == 0 ? nb[0][0] : (col == 1 ? nb[0][1] : nb[0][2]);
          const int r1 = col == 0 ? nb[1][0] : (col == 1 ? nb[1][1] : nb[1][2]);
          const int r2 = col == 0 ? nb[2][0] : (col == 1 ? nb[2][1] : nb[2][2]);
          const int sl = pi < 30 ? (dy == 0 ? r0 : (dy == 1 ? r1 : r2)) : keep;
          off[tt][j] = (yin & 3) * ROWB + sl * GB;
        }
      }
    f32x4_t acc[4][NYO];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
      for (int yo = 0; yo < NYO; ++yo) acc[cc][yo] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      __builtin_amdgcn_sched_barrier(0);        // keep the next channel's fragment reads out of this channel's registers
      const unsigned char* plane = data + (4 * q + cc) * PLB + 16 * ((4 * q + cc) >> 3);
      uint2 raw[NB][2];
#pragma unroll
      for (int tt = 0; tt < NB; ++tt) {
        raw[tt][0] = *reinterpret_cast<const uint2*>(plane + off[tt][0]);
        raw[tt][1] = *reinterpret_cast<const uint2*>(plane + off[tt][1]);
      }
      if (S == 8 && is_tail) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const bf16x8_t B = __builtin_bit_cast(bf16x8_t, make_uint4(raw[tt][0].x, raw[tt][0].y, raw[tt][1].x, raw[tt][1].y));
          acc[cc][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[cc][tt], B, acc[cc][0], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int tt = 0; tt < NB; ++tt) {
          const bf16x8_t B = __builtin_bit_cast(bf16x8_t, make_uint4(raw[tt][0].x, raw[tt][0].y, raw[tt][1].x, raw[tt][1].y));
          if constexpr (S == 8) {
#pragma unroll
            for (int yo = 0; yo < 2; ++yo) {
              const int kk = tt - yo;            // t - (2 yp + yo) with t = 2 yp + tt
              if (kk >= 0 && kk < 4) acc[cc][yo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[cc][kk], B, acc[cc][yo], 0, 0, 0);
            }
          } else {
            acc[cc][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[cc][tt], B, acc[cc][0], 0, 0, 0);
          }
        }
      }
    }
    DWM_STAMP(6 + 6 * ps);
    dwm_lds_barrier();                           // the previous pass's copy-out has read the staging tile
    DWM_STAMP(7 + 6 * ps);
    if (has_add) {
      // unconditional stores (idle threads hit a spare 16 bytes behind the tile)
      auto av_dst = [&](int u) -> uint4* {
        const int vi = u * NT + tid;
        const int o = vi % NV, r1 = vi / NV, pos = r1 % npp, pi = r1 / npp;
        return reinterpret_cast<uint4*>(stg + (vi < nvec ? pi * pstr + (pos * NV + o) * 16 : D::STG));
      };
      *av_dst(0) = av0;
      if constexpr (AU > 1) *av_dst(1) = av1;
      if constexpr (AU > 2) *av_dst(2) = av2;
      if constexpr (AU > 3) *av_dst(3) = av3;
      dwm_lds_barrier();
    }
    DWM_STAMP(8 + 6 * ps);
    // ---- epilogue in the staging tile: lane (patch, g) holds 4 consecutive columns x 4 channels of every row block
    if (valid) {
      const int nyo = is_tail ? 1 : NYO;
#pragma unroll
      for (int yo = 0; yo < NYO; ++yo) {
        if (yo < nyo) {
          const int yoa = is_tail ? yo_l : NYO * yp + yo;
          const int pos0 = (S == 8 ? (2 * yoa + (lg >> 1)) * 8 + 4 * (lg & 1) : lg * 4) - posbase;
          unsigned char* sp = stg + pidx * pstr + (pos0 * CCH + 4 * q) * 2;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t live = ((actw[yo] >> (8 * r)) & 0xffu) ? 0xffffffffu : 0u;
            const uint2 ar = *reinterpret_cast<const uint2*>(sp + r * CCH * 2);        // stale bits without a residual: masked, not multiplied
            const f32x2_t a01 = bf2x2_to_f2(ar.x & add_m), a23 = bf2x2_to_f2(ar.y & add_m);
            const float o0 = acc[0][yo][r] + b4[0] + a01.x, o1 = acc[1][yo][r] + b4[1] + a01.y;
            const float o2 = acc[2][yo][r] + b4[2] + a23.x, o3 = acc[3][yo][r] + b4[3] + a23.y;
            *reinterpret_cast<uint2*>(sp + r * CCH * 2) = make_uint2(f2bf2(o0, o1) & live, f2bf2(o2, o3) & live);
          }
        }
      }
    }
    DWM_STAMP(9 + 6 * ps);
    dwm_lds_barrier();
    DWM_STAMP(10 + 6 * ps);
    if (has_add && ps + 1 < npass) av_load(ps + 1);       // in front of this pass's stores (see above)
    // ---- copy-out: contiguous runs of npp * CCH * 2 bytes per patch. (Rolled on purpose: all LDS reads first and four stores in a
    // row took 2 900 instead of 1 400 cycles - the store queue back-pressures the wave either way.)
    {
      for (int vi = tid; vi < nvec; vi += NT) {
        const int o = vi % NV, r1 = vi / NV, pos = r1 % npp, pi = r1 / npp;
        const uint4 w = *reinterpret_cast<const uint4*>(stg + pi * pstr + (pos * NV + o) * 16);
        *reinterpret_cast<uint4*>(out + ((size_t)(n * keep + 16 * ng + pi) * (S * S) + posbase + pos) * C + c0 + 8 * o) = w;
      }
    }
    DWM_STAMP(11 + 6 * ps);
  }
}
