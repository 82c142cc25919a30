// Persistent per-sample stage kernels ("ps"): ONE launch runs the FORWARD of every ConvNeXtV2 block of a sparse stage
// (a backward twin was built in round 3, measured no faster stand-alone and slower in the step - a one-workgroup-per-CU kernel cannot
// share the GPU with the weight-gradient lane - and removed in round 4).
//
// Replaces, per block, the launch chain  dw7x7 -> [LN + pw1 + GELU^2 sums] -> reduce -> [GRN + pw2 + residual]  of the
// row-streaming kernels (rsc.cuh, dwconv6.cuh) at the compute-shaped stages, where those launches are latency chains
// (profiles/r02: 17 + 33 + 33 us per block for ~4 us of roofline work). Reference semantics: models/convnextv2_sparse.py:47-56,
// models/sparse_norm_layers.py:24-33 (batch-global GRN) and :61-77.
//
// Decomposition: one 512-thread workgroup per SAMPLE (grid = N <= number of CUs, one workgroup per CU by LDS footprint). A sample's
// rows of the stage (keep * S^2 <= 80) live in LDS for the whole stage:
//   XA  [RP][C+8]   bf16   depthwise output d -> LayerNorm output xn (A operand of pw1)
//   HA  [RP][4C+16] bf16   h -> z = GRN(gelu(h)) (A operand of pw2); aliased by XF [R][C] fp32, the block input x, while no
//                          hidden tensor is live (written by the pw2 epilogue, read by the next block's depthwise gather)
// Weights never touch LDS: a wave owns a slice of the output columns, so every weight element is needed by exactly one wave and
// streams global (L2) -> registers directly in MFMA fragment layout through a ring of k-steps. The MFMA is issued transposed
// (D[n][m] = W A^T) as in rsc.cuh: a lane ends with 4 consecutive output columns of one row, bias / GELU / residual are lane-local.
// The batch-global GRN statistics cross workgroups through device-scope float atomics into the block's G2 vector and ONE grid
// barrier per block: a monotonic arrival counter polled with relaxed agent-scope loads. Both sides of every exchange are atomics /
// sc1 loads (MI355X_MICROARCH.md "valid forms": {agent atomics both sides}), so no release / acquire fence - whose L2 write-back
// would have to flush the ~270 KB of saved activations each workgroup streams out per block - is needed.
// The depthwise 7x7 is a patch-granular gather: per visible patch a table of the (2 NB + 1)^2 neighbouring patch slots; absent
// (masked) neighbours are skipped wave-uniformly, present ones are S^2 contiguous rows at immediate offsets.
#pragma once
#include <cstddef>
#include "rsc.cuh"

typedef MpmaePsArgs PsP;

namespace ps {

constexpr int NTHR = 512;
constexpr unsigned ABSENT = 0xFFFFu;

// -DPS_STAMPS: workgroup 0 writes shader-cycle stamps of block 1's phases to sync[8 + i] (tools/ps_check.py --stamps)
#ifdef PS_STAMPS
#define PS_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && b == 1) a.sync[8 + (i)] = (unsigned)clock64(); } while (0)
#else
#define PS_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ float atomic_ld(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// all workgroups of the launch; `target` = arrivals expected so far (monotonic counter, reset by the last workgroup to leave)
__device__ __forceinline__ void grid_barrier(unsigned* sync, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 21)) {          // a workgroup never became resident (another persistent kernel on the GPU): fail, do not hang
        __hip_atomic_store(&sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}

// XCD-hierarchical form (round 6; MI355X_MICROARCH.md "barrier-xcd": 4.1-4.8 us at 256 workgroups against 7.4 for the flat counter, whose 255
// arrivals serialise on one word). Workgroup b arrives on the counter of group x = b % 8 (the dispatcher deals workgroups round-robin over the 8
// XCDs; only the speed depends on that); the LAST arriver of a group arrives on the top counter and polls it (8 pollers), then publishes the
// generation in its group's flag word, which the group's other workgroups poll (<= 32 pollers per 128-byte line instead of 256 on one).
// All counters are monotonic over the launch (generation gen = 1, 2, ...: a workgroup cannot arrive for gen + 1 before gen is complete) and
// every word sits on its own 128-byte line: sync[32] top, sync[64 + 32 x] group counters, sync[320 + 32 x] group flags (MPMAE_PS_SYNC_WORDS = 640).
// Atomics / atomic loads on both sides of every exchange, as above; same spin bound and error word.
constexpr int HB_TOP = 32, HB_CNT = 64, HB_FLAG = 320, HB_WORDS = 640;
// Split in two so that a workgroup can ARRIVE as soon as its statistics are out and WAIT only after it has issued what does not depend on them
// (pw2's first weight slabs, the residual rows and bias of its own tiles): `role` (thread 0 only) = 1 for the last arriver of its group.
__device__ __forceinline__ unsigned grid_arrive_xcd(unsigned* sync, unsigned gen, unsigned nwg) {
  __syncthreads();
  unsigned role = 0;
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7u;
    const unsigned gsz = (nwg >> 3) + ((nwg & 7u) > x ? 1u : 0u);
    const unsigned old = __hip_atomic_fetch_add(&sync[HB_CNT + 32 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == gen * gsz) {                    // last of its group in this generation
      __hip_atomic_fetch_add(&sync[HB_TOP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      role = 1;
    }
  }
  return role;
}
__device__ __forceinline__ void grid_wait_xcd(unsigned* sync, unsigned gen, unsigned nwg, unsigned role) {
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7u, ngroups = nwg < 8u ? nwg : 8u;
    bool failed = false;
    auto wait_for = [&](unsigned* w, unsigned target) {
      unsigned spins = 0;
      while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 21)) { __hip_atomic_store(&sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); failed = true; break; }
      }
    };
    if (role) {
      wait_for(&sync[HB_TOP], gen * ngroups);
      if (!failed) __hip_atomic_store(&sync[HB_FLAG + 32 * x], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      wait_for(&sync[HB_FLAG + 32 * x], gen);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void grid_exit(unsigned* sync, unsigned nwg, bool hier) {
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nwg - 1) {                          // the last workgroup to leave: every other one is past its last barrier
      if (hier) {
        __hip_atomic_store(&sync[HB_TOP], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int x = 0; x < 8; ++x) {
          __hip_atomic_store(&sync[HB_CNT + 32 * x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&sync[HB_FLAG + 32 * x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int C, int S> struct Cfg {
  static constexpr int H = 4 * C, SS = S * S;
  static constexpr int NRG_ = NTHR / (H / 8);
  static constexpr int MT = (S == 2) ? 5 : 2, RP = 16 * MT;            // row tiles: keep * S^2 <= RP
  static constexpr int NB = (S == 2) ? 2 : 3, NBW = 2 * NB + 1, NBN = NBW * NBW;   // neighbour patches of the 7x7 window
  static constexpr int MASKW = (NBN * 2 + 3) / 4;                       // dword of the presence mask inside a table row
  static constexpr int TABB = ((MASKW + (NBN > 32 ? 2 : 1)) * 4 + 15) / 16 * 16;   // bytes per table row: uint16 XF byte offsets + mask
  static constexpr int ZOFF = 16 * ((S == 2) ? 5 : 2) * C * 4;          // zero patch behind the XF rows (absent neighbours read it)
  static constexpr int LDX = C + 8, LDH = H + 16;
  static constexpr int XA_B = RP * LDX * 2, HA_B = RP * LDH * 2;
  static constexpr int OFF_XA = 0, OFF_HA = XA_B, OFF_VEC = OFF_HA + HA_B, OFF_CS = OFF_VEC + 2 * H * 4,
                       OFF_TAB = OFF_CS + 2 * H * 4, OFF_LIVE = OFF_TAB + 32 * TABB, OFF_RED = OFF_LIVE + RP, OFF_ROW = OFF_RED + 64,
                       LDS = OFF_ROW + RP * 8;                           // OFF_ROW: LayerNorm-backward row sums [RP][2]
  static constexpr int NCHK = (RP + NRG_ - 1) / NRG_;                  // rows per thread of the element-wise passes
  static_assert(ZOFF + SS * C * 4 <= HA_B && ZOFF + SS * C * 4 < 65536, "XF alias + zero patch");
  static_assert(LDS <= 160 * 1024, "LDS");
  static constexpr int CP = C / 2;                                     // depthwise work items (patch, channel pair): a wave owns whole
  static constexpr int NFG = CP / 64, CPF = NFG * 64, CPR = CP - CPF;  // patches for the first CPF pairs (wave-uniform neighbour skip) and
  static constexpr int PPR = CPR ? 64 / CPR : 1;                       // passes of PPR patches x CPR pairs for the remaining CPR pairs
  static_assert(8 % NFG == 0 && (CPR == 0 || 64 % CPR == 0), "depthwise item mapping");
  static constexpr int NCC = H / 8, NRG = NTHR / NCC;                  // z pass: 8-column chunks x row groups
};

// D[n][m] += sum_k W[n][k] A[m][k] for NTL 16-column tiles (rows of W) x MT 16-row tiles of A (LDS), k = KS steps of 32.
// wl: this lane's weight pointer (&W[(n_first + lr) * ldw + lg * 8]); tile j is wl + j * tstride. W streams global -> registers through a
// ring of D k-steps; gemm_prefetch() issues the first D steps (call it a phase EARLY: weights do not depend on anything the kernel
// computes, so their L2 latency hides behind the preceding LayerNorm / barrier), gemm_run() consumes and refills the ring.
// `rot` (workgroup-uniform, 0 .. KS-1) rotates the order in which the k-steps are visited: every workgroup of the launch streams the
// SAME weight matrix at the same moment, and in lock step they would all hit the same L2 channel at once.
template <int KS> __device__ __forceinline__ int kstep(int ks, int rot) { const int t = ks + rot; return (t >= KS ? t - KS : t) * 32; }
template <int NTL, int KS, int D>
__device__ __forceinline__ void gemm_prefetch(const bf16_t* __restrict__ wl, const int tstride, uint4 (&wq)[D][NTL], const int rot = 0) {
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < KS) {
      const int ko = kstep<KS>(s, rot);
#pragma unroll
      for (int j = 0; j < NTL; ++j) wq[s][j] = *reinterpret_cast<const uint4*>(wl + (size_t)j * tstride + ko);
    }
}
template <int NTL, int MT, int KS, int D, int LDA>
__device__ __forceinline__ void gemm_run(const bf16_t* __restrict__ wl, const int tstride, const bf16_t* al, uint4 (&wq)[D][NTL],
                                         f32x4_t (&acc)[NTL][MT], const int rot = 0) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8_t af[MT];
    const int ka = kstep<KS>(ks, rot);
#pragma unroll
    for (int m = 0; m < MT; ++m) af[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(al + m * 16 * LDA + ka));
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wq[ks % D][j]);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[m], acc[j][m], 0, 0, 0);
    }
    if (ks + D < KS) {
      const int ko = kstep<KS>(ks + D, rot);
#pragma unroll
      for (int j = 0; j < NTL; ++j) wq[ks % D][j] = *reinterpret_cast<const uint4*>(wl + (size_t)j * tstride + ko);
    }
  }
}

template <int N> __device__ __forceinline__ unsigned tab_dword(const uint4 (&t)[N], int i) {      // i: compile-time after unrolling
  const uint4& q = t[i >> 2];
  return (i & 3) == 0 ? q.x : (i & 3) == 1 ? q.y : (i & 3) == 2 ? q.z : q.w;
}
template <int N> __device__ __forceinline__ unsigned tab_entry(const uint4 (&t)[N], int i) {
  const unsigned d = tab_dword(t, i >> 1);
  return (i & 1) ? (d >> 16) : (d & 0xFFFFu);
}

// neighbour (nb, row q of that patch) feeds output o of the patch with tap (ky, kx): compile-time geometry of the 7x7 window
template <int S> __device__ __forceinline__ constexpr bool dw_used(int nb, int q) {
  constexpr int NB = (S == 2) ? 2 : 3, NBW = 2 * NB + 1;
  const int wy = S * (nb / NBW - NB) + q / S, wx = S * (nb % NBW - NB) + q % S;
  for (int o = 0; o < S * S; ++o) {
    const int ky = wy - o / S + 3, kx = wx - o % S + 3;
    if (ky >= 0 && ky < 7 && kx >= 0 && kx < 7) return true;
  }
  return false;
}

// One depthwise 7x7 of patch slot k over the sample's LDS-resident fp32 rows XF (channel pair cp), accumulated into accout[o].
// The table row of the patch holds, per neighbour patch, the XF byte offset of its S^2 rows (absent neighbours point at a zero patch)
// and a presence mask. The LDS reads are UNCONDITIONAL and run one GROUP of neighbours (one row of the 5x5 patch window at S = 2,
// two rows of the 7x7 window at S = 1) ahead of the multiply-adds; only the multiply-adds of an absent neighbour are skipped. With
// the reads inside that branch every present neighbour cost a full LDS round trip (19 k of the first version's 100 k cycles per
// block), one NEIGHBOUR ahead still 20 k: most neighbours are masked, so there is nothing to hide a round trip behind.
// FLIP = 1: data gradient (taps mirrored).
template <int C, int S, int FLIP>
__device__ __forceinline__ void dw_gather(const unsigned char* smem, f32x2_t (&accout)[4], int k /*patch slot*/, int cp,
                                          const f32x2_t (&w)[49]) {
  using K = Cfg<C, S>;
  constexpr int SS = K::SS, NB = K::NB, NBW = K::NBW, NBN = K::NBN;
  constexpr int GR = (S == 2) ? 1 : 2, GN = GR * NBW, NG = (NBW + GR - 1) / GR;      // window rows / neighbours per group, groups
  uint4 tabv[K::TABB / 16];
#pragma unroll
  for (int i = 0; i < K::TABB / 16; ++i) tabv[i] = *reinterpret_cast<const uint4*>(smem + K::OFF_TAB + k * K::TABB + i * 16);
  const unsigned mlo = tab_dword(tabv, K::MASKW), mhi = (NBN > 32) ? tab_dword(tabv, K::MASKW + 1) : 0u;
  const unsigned char* xf = smem + K::OFF_HA + cp * 8;
  f32x2_t v[2][GN * SS];
#pragma unroll
  for (int i = 0; i < GN; ++i)
#pragma unroll
    for (int q = 0; q < SS; ++q)
      if (dw_used<S>(i, q)) v[0][i * SS + q] = *reinterpret_cast<const f32x2_t*>(xf + tab_entry(tabv, i) + q * C * 4);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
#pragma unroll
      for (int i = 0; i < GN; ++i) {
        const int nb = (g + 1) * GN + i;
        if (nb < NBN) {
          const unsigned off = tab_entry(tabv, nb);
#pragma unroll
          for (int q = 0; q < SS; ++q)
            if (dw_used<S>(nb, q)) v[(g + 1) & 1][i * SS + q] = *reinterpret_cast<const f32x2_t*>(xf + off + q * C * 4);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < GN; ++i) {
      const int nb = g * GN + i;
      if (nb < NBN) {
        const bool present = ((nb < 32 ? mlo >> nb : mhi >> (nb - 32)) & 1u) != 0;
        if (present) {
          const int dpy = nb / NBW - NB, dpx = nb % NBW - NB;
#pragma unroll
          for (int q = 0; q < SS; ++q) {
            const int wy = S * dpy + q / S, wx = S * dpx + q % S;          // input position relative to the patch origin
#pragma unroll
            for (int o = 0; o < SS; ++o) {
              const int ky = wy - o / S + 3, kx = wx - o % S + 3;
              if (ky >= 0 && ky < 7 && kx >= 0 && kx < 7) {
                const int t = FLIP ? (6 - ky) * 7 + (6 - kx) : ky * 7 + kx;
                accout[o] = __builtin_elementwise_fma(w[t], v[g & 1][i * SS + q], accout[o]);      // v_pk_fma_f32 (scalar fmacs made hipcc spill half of w)
              }
            }
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the reads ONE group ahead: hoisted all at once they need 200 registers
  }
}

// neighbour table of the sample's visible patches: per patch slot k a row of TABB bytes = NBN uint16 XF byte offsets (absent -> ZOFF,
// the zero patch) and the presence mask at dword MASKW
template <int C, int S>
__device__ __forceinline__ void build_tab(const MpmaeGeom& g_, unsigned char* smem, int n) {
  using K = Cfg<C, S>;
  unsigned short* tab = reinterpret_cast<unsigned short*>(smem + K::OFF_TAB);
  unsigned* tabw = reinterpret_cast<unsigned*>(smem + K::OFF_TAB);
  const int keep = g_.keep, grid = g_.grid, L = grid * grid;
  for (int i = threadIdx.x; i < keep * (K::TABB / 4); i += NTHR) {
    const int k = i / (K::TABB / 4), wd = i - k * (K::TABB / 4);
    if (wd >= K::MASKW) tabw[i] = 0u;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < keep * K::NBN; i += NTHR) {
    const int k = i / K::NBN, nb = i - k * K::NBN;
    const int patch = g_.vis[n * keep + k];
    const int qy = patch / grid + nb / K::NBW - K::NB, qx = patch % grid + nb % K::NBW - K::NB;
    int slot = -1;
    if (qy >= 0 && qx >= 0 && qy < grid && qx < grid) slot = g_.inv[n * L + qy * grid + qx];
    tab[k * (K::TABB / 2) + nb] = slot >= 0 ? (unsigned short)(slot * K::SS * C * 4) : (unsigned short)K::ZOFF;
    if (slot >= 0) atomicOr(&tabw[k * (K::TABB / 4) + K::MASKW + (nb >> 5)], 1u << (nb & 31));
  }
}

// =====================================================================================
// forward: blocks 0 .. nblk-1 of one stage
template <int C, int S>
__global__ __launch_bounds__(512) void ps_fwd_kernel(const PsP a) {
  using K = Cfg<C, S>;
  using T = bf16_t;
  constexpr int H = K::H, SS = K::SS, MT = K::MT, RP = K::RP, LDX = K::LDX, LDH = K::LDH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* XA = reinterpret_cast<bf16_t*>(smem + K::OFF_XA);
  bf16_t* HA = reinterpret_cast<bf16_t*>(smem + K::OFF_HA);
  float* XF = reinterpret_cast<float*>(smem + K::OFF_HA);
  float* vec = reinterpret_cast<float*>(smem + K::OFF_VEC);
  float* csum = reinterpret_cast<float*>(smem + K::OFF_CS);
  unsigned char* live = smem + K::OFF_LIVE;
  float* red = reinterpret_cast<float*>(smem + K::OFF_RED);

  const int tid0 = threadIdx.x;
  const int n = blockIdx.x, nwg = gridDim.x;
  const bool hier = a.sync_words >= HB_WORDS;      // (the XCD-hierarchical grid barrier needs its 640 sync words; 4 words: the flat counter)
  const int keep = a.g.keep, R = keep * SS;
  const size_t rowbase = (size_t)n * R;

  // ---- stage prologue: neighbour table, activity, input rows
  {
  const int tid = tid0;
  for (int i = tid; i < K::XA_B / 16; i += NTHR) reinterpret_cast<uint4*>(XA)[i] = make_uint4(0u, 0u, 0u, 0u);
  build_tab<C, S>(a.g, smem, n);
  for (int i = tid; i < RP; i += NTHR) live[i] = (i < R) ? (a.act ? a.act[rowbase + i] : (unsigned char)1) : (unsigned char)0;
  {
    const T* xin = reinterpret_cast<const T*>(a.x_in) + rowbase * C;
    for (int i = tid; i < R * (C / 8); i += NTHR) {
      float v[8];
      ld8<T>(xin + (size_t)i * 8, v);
      *reinterpret_cast<float4*>(XF + i * 8) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(XF + i * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    for (int i = tid; i < SS * C / 4; i += NTHR) *reinterpret_cast<float4*>(smem + K::OFF_HA + K::ZOFF + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  }
  __syncthreads();

  // the per-block records are read from the kernel-argument segment itself (constant address space, scalar loads with a dynamic
  // offset): indexing the by-value copy `a.blk[b]` with a runtime b makes hipcc spill the whole argument struct to scratch
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(4))) const char* kchar_p;
  typedef __attribute__((address_space(4))) const MpmaePsBlock* kblk_p;
  const kblk_p blks = (kblk_p)((kchar_p)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MpmaePsArgs, blk));
#else
  const MpmaePsBlock* blks = a.blk;
#endif
  const void* prev_out = a.x_in;
  for (int b = 0; b < a.nblk; ++b) {
    const MpmaePsBlock B = blks[b];
    const T* xres = reinterpret_cast<const T*>(prev_out);
    prev_out = B.out;
    // thread coordinates behind an opaque copy: otherwise every address of the (fully unrolled) block body is loop-invariant, gets
    // hoisted out of the block loop and lives - spilled - across the whole kernel (72 scratch stores in the prologue, 200 reloads)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lg = lane >> 4;

    PS_STAMP(0);
    // ---- P1: depthwise 7x7 (XF -> XA). Items = (patch, 64 channel pairs), dealt round-robin over the 8 waves: the first keep * NFG
    // items cover pairs [0, CPF) of ONE patch each (masked neighbours are skipped by the whole wave), the rest cover the remaining CPR
    // pairs of PPR patches at a time.
    {
      const int nU = keep * K::NFG, nT = nU + (K::CPR ? (keep + K::PPR - 1) / K::PPR : 0);
      auto run_items = [&](int i0, int i1, int istep, bool rem, const f32x2_t (&w)[49], const f32x2_t bias) {
        const int cp = rem ? K::CPF + lane % (K::CPR ? K::CPR : 1) : (wave % K::NFG) * 64 + lane;
#pragma unroll 1
        for (int i = i0; i < i1; i += istep) {
          const int kk = rem ? (i - nU) * K::PPR + lane / (K::CPR ? K::CPR : 1) : i / K::NFG;
          const bool on = kk < keep;
          const int k = on ? kk : keep - 1;
          f32x2_t acc[4];
#pragma unroll
          for (int o = 0; o < 4; ++o) acc[o] = bias;
          dw_gather<C, S, 0>(smem, acc, k, cp, w);
          if (on) {
#pragma unroll
            for (int o = 0; o < SS; ++o) {
              const int row = k * SS + o;
              const bool lv = live[row] != 0;
              *reinterpret_cast<unsigned*>(XA + row * LDX + 2 * cp) = lv ? f2bf2(acc[o].x, acc[o].y) : 0u;
            }
          }
        }
      };
      auto run_set = [&](int i0, int i1, int istep, bool rem) {      // (requesting the first weight set one block ahead moved its wait into pw2: no gain)
        const int cpS = rem ? K::CPF + lane % (K::CPR ? K::CPR : 1) : (wave % K::NFG) * 64 + lane;
        f32x2_t w[49];
#pragma unroll
        for (int t = 0; t < 49; ++t) w[t] = *reinterpret_cast<const f32x2_t*>(B.dw_w + ((t % 7) * 7 + t / 7) * C + 2 * cpS);   // t = ky*7 + kx -> (kw*7 + kh)*C
        run_items(i0, i1, istep, rem, w, *reinterpret_cast<const f32x2_t*>(B.dw_b + 2 * cpS));
      };
      if constexpr (K::NFG == 1 && K::CPR > 0) {
        // C = 160: waves 0-5 take the uniform items, waves 6-7 the remainder items - ONE tap set (49 register pairs, a round trip behind the previous
        // block's stores) per wave. Dealt round-robin over all 8 waves, five waves needed both sets one after the other: at 19 patches the phase was
        // set + 2 items + set + 1 item against set + 4 items here (sub-stamps of round 6: a set 6.2 k cycles behind the store drain, an item 2.6 k);
        // 3.296 vs 3.304 ms in the step (profiles/r06/ab_ps_dw_split.txt). Requesting the next block's set in front of this block's `out` stores
        // (one vmcnt for loads and stores on gfx950) spilled 91 registers and lost 0.09 ms.
        constexpr int WU = 6;
        if (wave < WU) { if (wave < nU) run_set(wave, nU, WU, false); }
        else if (nU + wave - WU < nT) run_set(nU + wave - WU, nT, 8 - WU, true);
      } else {
        if (wave < nU) run_set(wave, nU, 8, false);
        const int r0 = nU + ((wave - nU % 8) + 8) % 8;        // this wave's first remainder item (items continue round-robin behind the uniform ones)
        if (r0 < nT) run_set(r0, nT, 8, true);
      }
    }
    // LayerNorm vectors and pw1's first weight slabs: requested before the barrier in front of the LayerNorm phase
    constexpr int NCH = C / 8, NI = (NCH + 15) / 16;
    float ga[NI][8], be[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = min(lr + 16 * i, NCH - 1);
      ld8<float>(B.ln_g + q * 8, ga[i]);
      ld8<float>(B.ln_b + q * 8, be[i]);
    }
    constexpr int NT1 = C / 32, KS1 = C / 32;            // 16-column tiles per wave, k-steps
    constexpr int NTL1 = (NT1 % 5 == 0) ? 5 : NT1;       // tiles processed together
    constexpr int D1 = (KS1 < 4) ? KS1 : 4;
    const int n0 = wave * (H / 8);
    const T* W1 = reinterpret_cast<const T*>(B.W1);
    uint4 wq1[D1][NTL1];
    gemm_prefetch<NTL1, KS1, D1>(W1 + (size_t)(n0 + lr) * B.ldw1 + lg * 8, 16 * B.ldw1, wq1, n % KS1);
    __syncthreads();

    PS_STAMP(1);
    // ---- P3: LayerNorm in place on XA (16 lanes per row), x-hat / rstd / xn saved for the backward
    {
      T* dhat = reinterpret_cast<T*>(B.dhat) + rowbase * C;
      T* xng = reinterpret_cast<T*>(B.xn) + rowbase * C;
      constexpr int NPASS = (RP + 31) / 32;
#pragma unroll
      for (int ps_ = 0; ps_ < NPASS; ++ps_) {
        const int row = ps_ * 32 + wave * 4 + lg;
        const bool inb = row < R;
        const int rowc = min(row, RP - 1);
        float v[NI][8];
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int q = lr + 16 * i;
          const uint4 raw = and4(*reinterpret_cast<const uint4*>(XA + rowc * LDX + min(q, NCH - 1) * 8), q < NCH);
          unpack8(raw, v[i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) s1 += v[i][e];
        }
        s1 = sum16(s1);
        const float mean = s1 / C;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = (lr + 16 * i < NCH) ? v[i][e] - mean : 0.f;
            s2 += d * d;
          }
        s2 = sum16(s2);
        const float rstd = rsqrtf(s2 / C + 1e-6f);
        const bool lv = inb && live[rowc] != 0;
        if (inb && lr == 0) B.rstd[rowbase + row] = lv ? rstd : 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int q = lr + 16 * i;
          float xh[8], xn[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            xh[e] = lv ? bf2f(f2bf((v[i][e] - mean) * rstd)) : 0.f;
            xn[e] = lv ? xh[e] * ga[i][e] + be[i][e] : 0.f;
          }
          if (inb && q < NCH) {
            st8<T>(dhat + (size_t)row * C + q * 8, xh);
            const bf16x8_t pk = pack_bf16x8(xn);
            *reinterpret_cast<uint4*>(xng + (size_t)row * C + q * 8) = __builtin_bit_cast(uint4, pk);
            *reinterpret_cast<uint4*>(XA + row * LDX + q * 8) = __builtin_bit_cast(uint4, pk);
          }
        }
      }
    }
    __syncthreads();

    PS_STAMP(2);
    // ---- P4: pw1 + bias -> h (global, bf16); g = gelu(h) -> HA (bf16) with its squared column sums
    {
      T* hg = reinterpret_cast<T*>(B.h) + rowbase * H;
#pragma unroll 1
      for (int jg = 0; jg < NT1 / NTL1; ++jg) {
        const int nb0 = n0 + jg * NTL1 * 16;
        const T* wl = W1 + (size_t)(nb0 + lr) * B.ldw1 + lg * 8;
        if (jg > 0) gemm_prefetch<NTL1, KS1, D1>(wl, 16 * B.ldw1, wq1, n % KS1);
        float4 b4[NTL1];
#pragma unroll
        for (int j = 0; j < NTL1; ++j) b4[j] = *reinterpret_cast<const float4*>(B.b1 + nb0 + j * 16 + lg * 4);
        f32x4_t acc[NTL1][MT];
#pragma unroll
        for (int j = 0; j < NTL1; ++j)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[j][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        gemm_run<NTL1, MT, KS1, D1, LDX>(wl, 16 * B.ldw1, XA + lr * LDX + lg * 8, wq1, acc, n % KS1);
        bool lv[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) lv[m] = live[m * 16 + lr] != 0;
#pragma unroll
        for (int j = 0; j < NTL1; ++j) {
          const int nc = nb0 + j * 16 + lg * 4;
          const float bb[4] = {b4[j].x, b4[j].y, b4[j].z, b4[j].w};
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int row = m * 16 + lr;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = lv[m] ? acc[j][m][r] + bb[r] : 0.f;
            *reinterpret_cast<uint2*>(HA + row * LDH + nc) = pack_bf16x4(o);      // h, staged: stored and turned into g = gelu(h) by the pass below
          }
        }
      }
    }
    __syncthreads();
    // ---- P4b: h leaves for global memory in whole 16-byte row pieces (a thread = 8 columns of every NRG-th row: consecutive lanes write
    // consecutive bytes). Stored straight from the accumulators - 8 bytes per lane, 32 contiguous bytes per row and instruction - the h
    // stores were 10 k of pw1's 27 k cycles per block (stage 2: 317 -> 267 us with the stores removed, profiles/r05/ps_stamps.txt).
    // g = gelu(h) replaces h in HA; its squared column sums fold over the row groups through the (dead) xn rows of XA.
    {
      T* hg = reinterpret_cast<T*>(B.h) + rowbase * H;
      const int cc = tid % K::NCC, rg = tid / K::NCC;
      float* part = reinterpret_cast<float*>(smem + K::OFF_XA);              // [NRG][H]
      static_assert(K::NRG * H * 4 <= K::XA_B, "column-sum partials fit the xn rows");
      if (rg < K::NRG) {
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int m = rg; m < R; m += K::NRG) {
          const uint4 hv = *reinterpret_cast<const uint4*>(HA + m * LDH + cc * 8);
          *reinterpret_cast<uint4*>(hg + (size_t)m * H + cc * 8) = hv;
          float h8[8], g8[8];
          unpack8(hv, h8);                                                  // the backward (and the row-streaming path) see the stored bf16 value
          gelu_n<T, 8>(h8, g8);
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[e] += g8[e] * g8[e];
          *reinterpret_cast<uint4*>(HA + m * LDH + cc * 8) = __builtin_bit_cast(uint4, pack_bf16x8(g8));
        }
        *reinterpret_cast<float4*>(part + rg * H + cc * 8) = make_float4(cs[0], cs[1], cs[2], cs[3]);
        *reinterpret_cast<float4*>(part + rg * H + cc * 8 + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
      }
      __syncthreads();
      for (int j = tid; j < H; j += NTHR) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < K::NRG; ++q) a += part[q * H + j];
        csum[j] = a;
      }
    }
    __syncthreads();

    PS_STAMP(3);
    // ---- P5 .. P8 per wave tile count (waves w, w + 8, ... own the 16-column tiles of pw2)
    constexpr int NT2 = C / 16, NTW = (NT2 + 7) / 8, KS2 = H / 32;
    auto tail = [&](auto ntl_) {
      constexpr int NTL = decltype(ntl_)::value;
      constexpr int D2 = (S == 2) ? 8 : 6;
      const T* W2 = reinterpret_cast<const T*>(B.W2);
      const T* wl2 = W2 + (size_t)(wave * 16 + lr) * B.ldw2 + lg * 8;
      uint4 wq2[D2][NTL];
      // ---- P5: batch-global GRN statistics: atomics into G2, one grid barrier, finalisation by every workgroup
      constexpr int NJ = (H + NTHR - 1) / NTHR;
      float gg[NJ], gb[NJ];
#pragma unroll
      for (int u = 0; u < NJ; ++u) {
        const int j = tid + NTHR * u, jc = min(j, H - 1);
        gg[u] = B.grn_g[jc]; gb[u] = B.grn_b[jc];
        if (j < H) (void)unsafeAtomicAdd(B.G2 + (size_t)(n % a.ng) * H + j, csum[j]);      // ng accumulator copies: fewer colliding adds per line
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the adds are performed (acknowledged) before this workgroup arrives
      unsigned role = 0;
      if (hier) role = grid_arrive_xcd(a.sync, (unsigned)(b + 1), nwg);      // arrive early ...
      gemm_prefetch<NTL, KS2, D2>(wl2, 8 * 16 * B.ldw2, wq2, n % KS2);      // pw2's first weight slabs travel while the barrier is waited for
      // ... and so do the residual rows and the bias of this wave's output tiles (consumed behind pw2)
      const T* xr = xres + rowbase * C;
      uint2 xraw[NTL][MT];
      float4 b4[NTL];
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        b4[j] = *reinterpret_cast<const float4*>(B.b2 + (wave + 8 * j) * 16 + lg * 4);
#pragma unroll
        for (int m = 0; m < MT; ++m)
          xraw[j][m] = *reinterpret_cast<const uint2*>(xr + (size_t)min(m * 16 + lr, R - 1) * C + (wave + 8 * j) * 16 + lg * 4);
      }
      if (hier) grid_wait_xcd(a.sync, (unsigned)(b + 1), nwg, role); else grid_barrier(a.sync, (unsigned)(b + 1) * nwg);      // ... wait late
      PS_STAMP(4);
      float gx[NJ], s = 0.f;
#pragma unroll
      for (int u = 0; u < NJ; ++u) {
        const int j = tid + NTHR * u;
        float t = 0.f;
        for (int q = 0; q < a.ng; ++q) t += atomic_ld(B.G2 + (size_t)q * H + min(j, H - 1));
        gx[u] = sqrtf(t);
        s += (j < H) ? gx[u] : 0.f;
      }
      s = wave_sum(s);
      if (lane == 0) red[wave] = s;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) tot += red[w8];
      const float ainv = 1.f / (tot / H + a.eps);
      if (n == 0 && tid == 0) B.Ainv[0] = ainv;
#pragma unroll
      for (int u = 0; u < NJ; ++u) {
        const int j = tid + NTHR * u;
        if (j < H) {
          const float sc = 1.f + gg[u] * (gx[u] * ainv);
          vec[j] = sc;
          vec[H + j] = gb[u];
          if (n == 0) { B.Gx[j] = gx[u]; B.scale[j] = sc; }
        }
      }
      __syncthreads();

      PS_STAMP(5);
      // ---- P7: z = g * scale + beta in place on HA, saved for the backward (operand of pw2's weight gradient)
      {
        const int cc = tid % K::NCC, rg = tid / K::NCC;
        if (rg < K::NRG) {
          float sc[8], bt[8];
          ld8<float>(vec + cc * 8, sc);
          ld8<float>(vec + H + cc * 8, bt);
          T* zg = reinterpret_cast<T*>(B.z) + rowbase * H + cc * 8;
          for (int m = rg; m < R; m += K::NRG) {
            float g[8], z[8];
            unpack8(*reinterpret_cast<const uint4*>(HA + m * LDH + cc * 8), g);
            const bool lvm = live[m] != 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = lvm ? g[e] * sc[e] + bt[e] : 0.f;
            const uint4 zp = __builtin_bit_cast(uint4, pack_bf16x8(z));
            *reinterpret_cast<uint4*>(zg + (size_t)m * H) = zp;
            *reinterpret_cast<uint4*>(HA + m * LDH + cc * 8) = zp;
          }
        }
      }
      __syncthreads();

      PS_STAMP(6);
      // ---- P8: pw2 + bias + residual -> out (global, bf16) and XF (fp32 copy of the bf16 value: next block's depthwise input)
      f32x4_t acc[NTL][MT];
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      gemm_run<NTL, MT, KS2, D2, LDH>(wl2, 8 * 16 * B.ldw2, HA + lr * LDH + lg * 8, wq2, acc, n % KS2);
      __syncthreads();                               // every wave is done reading z: HA's bytes become XF
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        const int nc = (wave + 8 * j) * 16 + lg * 4;
        const float bb[4] = {b4[j].x, b4[j].y, b4[j].z, b4[j].w};
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int row = m * 16 + lr;
          const bool lvm = live[row] != 0;
          float x[4], o[4];
          unpack4(xraw[j][m], x);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = lvm ? acc[j][m][r] + bb[r] + x[r] : 0.f;
          if (row < R) {
            const uint2 pk = pack_bf16x4(o);
            float of[4];
            unpack4(pk, of);                              // (the bf16 value; it leaves for global memory in 16-byte pieces behind the block's last barrier)
            *reinterpret_cast<float4*>(XF + row * C + nc) = make_float4(of[0], of[1], of[2], of[3]);
          }
        }
      }
      for (int i = tid; i < SS * C / 4; i += NTHR) *reinterpret_cast<float4*>(smem + K::OFF_HA + K::ZOFF + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if (wave + 8 * (NTW - 1) < NT2) tail(std::integral_constant<int, NTW>{});
    else tail(std::integral_constant<int, (NTW > 1 ? NTW - 1 : 1)>{});
    __syncthreads();
    {      // out: XF (exact bf16 values in fp32) -> global, whole 16-byte row pieces (like h in P4b)
      T* outg = reinterpret_cast<T*>(B.out) + rowbase * C;
      for (int i = tid; i < R * (C / 8); i += NTHR) {
        const float4 a0 = *reinterpret_cast<const float4*>(XF + i * 8), a1 = *reinterpret_cast<const float4*>(XF + i * 8 + 4);
        const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        *reinterpret_cast<uint4*>(outg + (size_t)i * 8) = __builtin_bit_cast(uint4, pack_bf16x8(v));
      }
    }
    PS_STAMP(7);
  }
  grid_exit(a.sync, nwg, hier);
}

}  // namespace ps
