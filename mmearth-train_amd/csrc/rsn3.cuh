// Ring-pipelined form of the fused pointwise BACKWARD kernel of the compute-shaped stages (C = 160: rsc_narrow<160, 1, ...>, rsc.cuh):
//   dh = (dz * scale + coef * gelu(h)) * gelu'(h) (stored over dz), dd = LayerNorm-backward(dh W1), dgamma / dbeta partials,
//   optional folded GRN backward finalisation in the prologue (p.fin_sum).
//
// Why. rsc_narrow walks K = H = 640 in 10 chunks of 64; each chunk's weight slab goes global -> staging VGPRs -> ds_write -> barrier and
// is requested ONE iteration ahead, like the chunk's dz / h rows. An iteration is short (20 MFMAs per wave), a round trip under load is
// 1-3 us (profiles/r05/rs1_stamps*.txt), one workgroup per CU (244 workgroups): the kernel is a chain of 10 exposed round trips, 37 us for
// 100 MB of L2 -> CU traffic (11 us at the fill rate of this chip). Here
//   * the weight slabs go global -> LDS by DMA (global_load_lds_dwordx4, no staging registers, no ds_write pass) into a ring of THREE slots,
//     requested TWO iterations ahead; the lane-linear image is chunk-swizzled (chunk ^ ((row & 3) | ((row >> 3) & 1) << 2), 128-byte rows:
//     conflict-free under the ds_read_b128 lane groups, the swizzle of gemm_fast.cuh);
//   * the dz / h rows of a chunk travel in registers, requested two iterations ahead too (three register sets);
//   * ONE bare s_barrier per iteration, no vmcnt(0) drain: a wave waits for ITS OWN rows of chunk kc (issued behind its share of slab kc, so
//     the slab has landed too - vector memory operations complete in order), then the barrier publishes every wave's share.
// Everything else - fragment layouts, prologue arithmetic, LayerNorm-backward epilogue, the finalisation - is rsc_narrow's.
#pragma once
#include "rsc.cuh"
typedef const __attribute__((address_space(1))) void* rsn3_gptr_t;
typedef __attribute__((address_space(3))) void* rsn3_lptr_t;

// NRT row tiles of 16 rows per workgroup x NKG K-groups: wave w works on row tile w % NRT and on the chunks kc = NKG t + (w / NRT), t = 0 ..
// NKC / NKG - 1. NKG = 2 halves the length of a wave's dependent chain (a wave is an in-order stream: per chunk 16 ds_reads with their waits,
// ~100 VALU of GELU / GELU', 20 MFMAs - ~1200 cycles, and 244 workgroups of 5 waves are 1.2 waves per SIMD: nothing else to issue); the two
// partial accumulators of a row tile are added through LDS before the LayerNorm-backward epilogue, which group 0 runs. A ring stage holds
// the NKG slabs of one step.
template <int KC, int NRT, int NKG>
__global__ __launch_bounds__(64 * NRT * NKG) void rsn3_bwd_kernel(const RsP p) {
  using T = bf16_t;
  constexpr int NWV = NRT * NKG, NTH = 64 * NWV, HN = 4 * KC, KCH = 64, NKC = HN / KCH, KSC = KCH / 32, NT = KC / 16, NP = KC, D = 3, CPR = KCH / 8;
  constexpr int NST = NKC / NKG, NINST = NKG * NP * CPR / 64, SLAB = NP * KCH;
  static_assert(KC % 16 == 0 && NINST % NWV == 0 && HN % KCH == 0 && NKC % NKG == 0 && (NP * CPR / 64) % (NINST / NWV) == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char rsc_smem[];
  bf16_t* Wr = reinterpret_cast<bf16_t*>(rsc_smem);                                             // [D][NKG][NP][KCH], lane-linear, chunk-swizzled
  float* red = reinterpret_cast<float*>(rsc_smem + (size_t)D * NKG * NP * KCH * sizeof(bf16_t));      // [2][KC]
  float* vec = red + 2 * KC;                                                                     // [2][HN]: scale | coef
  float* fsh = vec + 2 * HN;                                                                     // [8]
  float* lgv = fsh + 8;                                                                          // [KC] LayerNorm gamma (epilogue)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int kg = wave / NRT;                        // K-group of this wave
  const int row = blockIdx.x * (16 * NRT) + (wave - kg * NRT) * 16 + lr;
  const bool inb = row < p.M;
  const int rowc = min(row, p.M - 1);
  for (int i = tid; i < 2 * KC; i += NTH) red[i] = 0.f;

  // The chunk rows are loaded by inline asm as well, and awaited by EXPLICIT counted waits: left to hipcc, the first use of a row register
  // is preceded by s_waitcnt vmcnt(0) (it does not count across the exec-masked store branches), which also drains the slab and rows
  // requested for chunk kc + 1 - the ring would be one iteration deep again.
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 araw[D][KSC], hraw[D][KSC];
  static_assert(KSC == 2, "two k-steps per chunk");
  const bf16_t* pa = p.A + (size_t)rowc * HN + kg * KCH + lg * 8;
  const bf16_t* ph = p.A2 + (size_t)rowc * HN + kg * KCH + lg * 8;
  auto swz = [](int r) { return (r & 3) | (((r >> 3) & 1) << 2); };
#define RSN3_GLD(dst, ptr, imm) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(imm) : "memory")
  auto dma = [&](int slot, int t) {                 // this wave's share of the NKG slabs of step t: NINST / NWV wave instructions of 64 x 16 bytes
#pragma unroll
    for (int ii = 0; ii < NINST / NWV; ++ii) {
      const int i = ii * NWV + wave, sb = i / (NP * CPR / 64), i2 = i - sb * (NP * CPR / 64);      // slab of the stage, instruction inside it
      const int sl = i2 * 64 + lane, r = sl / CPR, chs = sl - r * CPR;
      const bf16_t* src = p.W + (size_t)r * p.ldw + (NKG * t + sb) * KCH + ((chs ^ swz(r)) << 3);
      bf16_t* dst = Wr + ((size_t)slot * NKG + sb) * SLAB + (size_t)i2 * 64 * 8;
      __builtin_amdgcn_global_load_lds((rsn3_gptr_t)src, (rsn3_lptr_t)dst, 16, 0, 0);
    }
    asm volatile("" ::: "memory");
  };
  constexpr int NVM = NINST / NWV + 2 * KSC;         // vector-memory operations of one stage (slab share + rows): the counted waits below
  // stage kc: slab by DMA first, then this wave's rows (in-order completion: rows ready => the wave's slab share has landed)
#define RSN3_ISSUE(SL, KCI) do { dma(SL, KCI); \
    RSN3_GLD(araw[SL][0], pa, (KCI) * NKG * 128); RSN3_GLD(araw[SL][1], pa, (KCI) * NKG * 128 + 64); \
    RSN3_GLD(hraw[SL][0], ph, (KCI) * NKG * 128); RSN3_GLD(hraw[SL][1], ph, (KCI) * NKG * 128 + 64); } while (0)
  const uint8_t abl = *(p.act ? p.act + rowc : reinterpret_cast<const uint8_t*>(p.W));          // pointer select, not a branch
  RSN3_ISSUE(0, 0);
  RSN3_ISSUE(1, 1);

  // ---- staged GRN vectors / folded backward finalisation (rsc_narrow, PF bit 1)
  for (int i = tid; i < KC; i += NTH) lgv[i] = p.lng[i];
  if (!p.fin_sum) {
    for (int i = tid; i < HN / 4; i += NTH) {
      reinterpret_cast<float4*>(vec)[i] = reinterpret_cast<const float4*>(p.v0)[i];
      reinterpret_cast<float4*>(vec + HN)[i] = reinterpret_cast<const float4*>(p.v1)[i];
    }
  } else {                                        // grn_bwd_finalize_kernel
    constexpr int NJ = (HN + NTH - 1) / NTH;
    float fs[NJ], fg[NJ], fx[NJ], f0[NJ], fv[NJ];
    const float* s0p = p.fin_sum0 ? p.fin_sum0 : p.fin_sum;
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
      const int jc = min(tid + NTH * u, HN - 1);
      fs[u] = p.fin_sum[jc]; fg[u] = p.fin_gamma[jc]; fx[u] = p.fin_gx[jc]; f0[u] = s0p[jc]; fv[u] = p.v0[jc];
    }
    const float ainv = p.fin_ainv[0];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NJ; ++u) s += (tid + NTH * u < HN) ? fg[u] * fs[u] * fx[u] : 0.f;
    s = wave_sum(s);
    if (lane == 0) fsh[wave] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) tot += fsh[w];
    const float T2 = tot * ainv * ainv / HN;
    const bool pub = blockIdx.x == 0;
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
      const int j = tid + NTH * u;
      if (j < HN) {
        const float gx = fx[u], s1 = fs[u];
        const float dGx = fg[u] * s1 * ainv - T2;
        const float cf = (gx > 0.f) ? dGx / gx : 0.f;
        vec[j] = fv[u];
        vec[HN + j] = cf;
        if (pub) {
          if (p.fin_out) p.fin_out[j] = cf;
          atomicAdd(p.fin_dgamma + j, gx * ainv * s1);
          atomicAdd(p.fin_dbeta + j, f0[u]);
        }
      }
    }
  }
  // epilogue operands, requested up front as well (one round trip less at the end)
  uint2 xraw[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) xraw[j] = *reinterpret_cast<const uint2*>(p.xhat + (size_t)rowc * KC + j * 16 + lg * 4);
  const float rsl = p.rstd[rowc];
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(araw[0][0]), "+v"(araw[0][1]), "+v"(hraw[0][0]), "+v"(hraw[0][1]), "+v"(araw[1][0]), "+v"(araw[1][1]), "+v"(hraw[1][0]), "+v"(hraw[1][1]));
  __syncthreads();                                  // vec / red visible; the first two stages have landed for every wave
  const bool live = inb && (p.act ? abl != 0 : true);

  f32x4_t acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int fsw = swz(lr);                          // weight rows of tile j: j * 16 + lr (bits 0, 1, 3 of lr)

  // LDS reads inside the ring loop are inline asm with counted lgkmcnt waits: hipcc puts an s_waitcnt vmcnt(0) in front of every LDS access it can
  // see once a DMA is in flight (it cannot prove that the read does not alias the DMA's destination), which would drain the ring every iteration
#define RSN3_RD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  const unsigned lds0 = (unsigned)(uintptr_t)(rsn3_lptr_t)rsc_smem;
  constexpr unsigned OFF_VEC = (unsigned)D * NKG * NP * KCH * 2 + 2 * KC * 4;
  const unsigned wofs0 = (unsigned)(kg * SLAB * 2 + lr * KCH * 2 + (((0 * 4 + lg) ^ fsw) << 4)), wofs1 = (unsigned)(kg * SLAB * 2 + lr * KCH * 2 + (((1 * 4 + lg) ^ fsw) << 4));
  auto fr = [](const u32x4& v) { return __builtin_bit_cast(bf16x8_t, v); };
  auto f4 = [](const u32x4& v, float (&o)[8], int half) {
    o[half * 4 + 0] = __uint_as_float(v.x); o[half * 4 + 1] = __uint_as_float(v.y); o[half * 4 + 2] = __uint_as_float(v.z); o[half * 4 + 3] = __uint_as_float(v.w);
  };

  auto step = [&](auto slot_, auto kc_) {
    constexpr int SL = decltype(slot_)::value;
    const int kc = NKG * decltype(kc_)::value + kg;            // this wave's chunk of the step
    const unsigned va = lds0 + OFF_VEC + (unsigned)(kc * KCH + lg * 8) * 4;
    const unsigned wa0 = lds0 + (unsigned)SL * NKG * SLAB * 2 + wofs0, wa1 = lds0 + (unsigned)SL * NKG * SLAB * 2 + wofs1;
    u32x4 vv[KSC][4], wf[2][4];
    static_assert(KSC == 2, "two k-steps per chunk");
    RSN3_RD(vv[0][0], va, 0);   RSN3_RD(vv[0][1], va, 16);  RSN3_RD(vv[0][2], va, HN * 4);       RSN3_RD(vv[0][3], va, HN * 4 + 16);
    RSN3_RD(vv[1][0], va, 128); RSN3_RD(vv[1][1], va, 144); RSN3_RD(vv[1][2], va, HN * 4 + 128); RSN3_RD(vv[1][3], va, HN * 4 + 144);
    RSN3_RD(wf[0][0], wa0, 0); RSN3_RD(wf[0][1], wa1, 0); RSN3_RD(wf[0][2], wa0, 2048); RSN3_RD(wf[0][3], wa1, 2048);      // tiles 0, 1
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vv[0][0]), "+v"(vv[0][1]), "+v"(vv[0][2]), "+v"(vv[0][3]), "+v"(vv[1][0]), "+v"(vv[1][1]), "+v"(vv[1][2]), "+v"(vv[1][3]));
    bf16x8_t af[KSC];
#pragma unroll
    for (int s = 0; s < KSC; ++s) {
      const int k = kc * KCH + s * 32 + lg * 8;
      float sc[8], tc[8];
      f4(vv[s][0], sc, 0); f4(vv[s][1], sc, 1); f4(vv[s][2], tc, 0); f4(vv[s][3], tc, 1);
      float a[8], h[8], gl[8], dg[8], z[8];
      unpack8(__builtin_bit_cast(uint4, araw[SL][s]), a);          // (rows beyond M re-read the last row: nothing of theirs is stored or summed - a mask here becomes a
      unpack8(__builtin_bit_cast(uint4, hraw[SL][s]), h);          //  branch that hipcc merges with the store's and hoists, with an s_waitcnt vmcnt(0), two iterations up)
      gelu_both_n<T, 8>(h, gl, dg);
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (a[e] * sc[e] + tc[e] * gl[e]) * dg[e];                 // dh
      af[s] = pack_bf16x8(z);
      if (inb) *reinterpret_cast<uint4*>(const_cast<bf16_t*>(p.A) + (size_t)row * HN + k) = __builtin_bit_cast(uint4, af[s]);
    }
    // tile pairs (j, j + 1): the fragments of pair g + 1 are requested before pair g is awaited
#define RSN3_PAIR(G, CUR, NXT) do { \
      if ((G) + 1 < NT / 2) { RSN3_RD(wf[NXT][0], wa0, ((G) + 1) * 4096); RSN3_RD(wf[NXT][1], wa1, ((G) + 1) * 4096); \
                              RSN3_RD(wf[NXT][2], wa0, ((G) + 1) * 4096 + 2048); RSN3_RD(wf[NXT][3], wa1, ((G) + 1) * 4096 + 2048); \
                              asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wf[CUR][0]), "+v"(wf[CUR][1]), "+v"(wf[CUR][2]), "+v"(wf[CUR][3])); } \
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[CUR][0]), "+v"(wf[CUR][1]), "+v"(wf[CUR][2]), "+v"(wf[CUR][3])); \
      acc[2 * (G)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr(wf[CUR][0]), af[0], acc[2 * (G)], 0, 0, 0); \
      acc[2 * (G)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr(wf[CUR][1]), af[1], acc[2 * (G)], 0, 0, 0); \
      acc[2 * (G) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr(wf[CUR][2]), af[0], acc[2 * (G) + 1], 0, 0, 0); \
      acc[2 * (G) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr(wf[CUR][3]), af[1], acc[2 * (G) + 1], 0, 0, 0); } while (0)
    static_assert(NT == 10, "five tile pairs");
    RSN3_PAIR(0, 0, 1); RSN3_PAIR(1, 1, 0); RSN3_PAIR(2, 0, 1); RSN3_PAIR(3, 1, 0); RSN3_PAIR(4, 0, 1);
#undef RSN3_PAIR
  };
  // step kc: (kc >= 1) this wave's rows of chunk kc are here - and with them its share of slab kc; the barrier publishes every wave's share
  // and says that every wave is done with slot (kc + 2) % 3 = the slot of chunk kc - 1; then stage kc + 2 is requested into that slot.
  // Counted wait: the operations issued after stage kc are stage kc + 1 (NVM of them) and the exec-masked dh stores (possibly none).
#define RSN3_STEP(KCI) do { \
    constexpr int SL_ = (KCI) % 3, NX_ = ((KCI) + 2) % 3; \
    if ((KCI) >= 2) { \
      if ((KCI) + 1 < NST) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(araw[SL_][0]), "+v"(araw[SL_][1]), "+v"(hraw[SL_][0]), "+v"(hraw[SL_][1]) : "n"(NVM)); \
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(araw[SL_][0]), "+v"(araw[SL_][1]), "+v"(hraw[SL_][0]), "+v"(hraw[SL_][1])); \
    } \
    if ((KCI) >= 1) asm volatile("s_barrier" ::: "memory"); \
    if ((KCI) + 2 < NST) RSN3_ISSUE(NX_, ((KCI) + 2 < NST ? (KCI) + 2 : 0)); \
    step(std::integral_constant<int, SL_>{}, std::integral_constant<int, (KCI)>{}); } while (0)
  static_assert(NST == 10 || NST == 5, "ten chunks, one or two K-groups");
  RSN3_STEP(0); RSN3_STEP(1); RSN3_STEP(2); RSN3_STEP(3); RSN3_STEP(4);
  if constexpr (NST > 5) { RSN3_STEP(5); RSN3_STEP(6); RSN3_STEP(7); RSN3_STEP(8); RSN3_STEP(9); }
#undef RSN3_STEP
#undef RSN3_ISSUE
#undef RSN3_GLD
#undef RSN3_RD
  if constexpr (NKG > 1) {      // the partial sums of K-groups 1.. join group 0's through LDS (the ring is dead: every slab has been consumed)
    float* xch = reinterpret_cast<float*>(rsc_smem);                     // [NKG - 1][NRT][NT][64 lanes][4]
    __syncthreads();
    if (kg > 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
        *reinterpret_cast<float4*>(xch + ((((size_t)(kg - 1) * NRT + (wave - kg * NRT)) * NT + j) * 64 + lane) * 4) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int g = 1; g < NKG; ++g)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(xch + ((((size_t)(g - 1) * NRT + wave) * NT + j) * 64 + lane) * 4);
          acc[j][0] += v.x; acc[j][1] += v.y; acc[j][2] += v.z; acc[j][3] += v.w;
        }
    }
  }
  // ---- epilogue (K-group 0): LayerNorm backward; lane holds row m = lr, columns n = j*16 + lg*4 + r
  if (kg == 0) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n4 = j * 16 + lg * 4;
      const float4 g = *reinterpret_cast<const float4*>(lgv + n4);
      xraw[j] = and2(xraw[j], inb);
      float xh[4];
      unpack4(xraw[j], xh);
      const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dxn = live ? bf2f(f2bf(acc[j][r])) : 0.f;      // bf16 like the unfused path
        const float ga = sum16(dxn * xh[r]), gb = sum16(dxn);
        if (lr == 0) { atomicAdd(&red[n4 + r], ga); atomicAdd(&red[KC + n4 + r], gb); }
        const float gq = dxn * gg[r];
        acc[j][r] = gq;
        s1 += gq;
        s2 += gq * xh[r];
      }
    }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    s1 /= KC; s2 /= KC;
    const float rs = inb ? rsl : 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n4 = j * 16 + lg * 4;
      float xh[4], o[4];
      unpack4(xraw[j], xh);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = live ? rs * (acc[j][r] - s1 - xh[r] * s2) : 0.f;
      if (inb) *reinterpret_cast<uint2*>(p.out + (size_t)row * KC + n4) = pack_bf16x4(o);
    }
  }
  __syncthreads();
  for (int i = tid; i < KC; i += NTH) {
    p.ws[((size_t)blockIdx.x * 2 + 0) * KC + i] = red[i];
    p.ws[((size_t)blockIdx.x * 2 + 1) * KC + i] = red[KC + i];
  }
}
