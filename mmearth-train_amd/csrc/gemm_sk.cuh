// Stream-K dense NT GEMM: C[M][N] = A[M][K] W[N][K]^T (+ bias, + R, row mask), bf16 in / fp32 accumulate / bf16 out.
// (decoder pwconv2 and pwconv1 data gradient N = 512, K = 2048; heads data gradient N = 512, K = 2816; the stage-3 pwconv2 /
// pwconv1 data gradient N = 320, K = 1280: models/convnextv2.py:46-52, models/convnextv2_sparse.py:47-56, models/fcmae.py:126-151.)
//
// gemm_nt5.cuh is the tile (128 x 256, 64 x 128 wave tiles, 3-stage DMA ring, transposed issue); what it lacked against the vendor's
// kernels for these shapes was the SCHEDULE (profiles/r04/blas_yardstick.txt): 196 whole tiles on 256 CUs leave
// 60 CUs idle for the whole launch, and at stage 3 (M = 4864) 38-76 tiles leave most of the GPU idle. Here the unit of work is one
// 64-deep K ITERATION of a tile:
//   * the grid is 8 x Q workgroups (Q per XCD, all resident: one per CU); workgroup b runs on XCD b % 8 (observed placement - used
//     for SPEED only, see below). XCD x owns the contiguous tile range [T x / 8, T (x + 1) / 8) in row-block-major order (the column
//     tiles of a row block are neighbours: they share the A slab in that XCD's L2) and its workgroup q the iteration range
//     [I q / Q, I (q + 1) / Q) of the XCD's I = tiles x K / 64 iterations - every workgroup does the same number of iterations +- 1;
//   * a workgroup walks its range tile by tile: (tail of its first tile) (whole tiles)* (head of its last tile). A whole tile goes
//     straight to the epilogue. The workgroup that owns a tile's HEAD (k = 0) is its FINISHER: it computes its segment LAST in its
//     own timeline, by which time the other segments - the next workgroup's tail, computed FIRST in that workgroup's timeline -
//     have long been published; it adds their fp32 partials and runs the epilogue. Nobody who publishes ever waits, so the
//     schedule cannot deadlock whatever the residency;
//   * a partial is the 128 x 256 fp32 accumulator image in REGISTER order (lane-linear 16-byte pieces: coalesced both ways), one slot
//     per workgroup (a workgroup publishes at most one segment). Publish = write-through `sc1` stores -> every wave drains (vmcnt(0))
//     -> barrier -> one relaxed agent-scope flag store; consume = one lane polls the flag (relaxed, agent) -> barrier -> `sc1` loads
//     (MI355X_MICROARCH.md "valid forms", cdna_hip_programming.md section 6 G16 R1: correct for ANY placement of the two workgroups;
//     the XCD mapping above only makes the hand-off same-die). The finisher resets the flag it consumed: the flags are zero between
//     launches without a memset.
// Needs K % 64 == 0, N % 8 == 0, 16-byte aligned rows. Rows / columns beyond M / N are clamped on load and not stored.
#pragma once
#include "gemm_nt5.cuh"

struct SkP {
  float* part;          // [gridDim.x][128 * 256] fp32 partial slots
  unsigned* flags;      // [gridDim.x], zero at launch (self-resetting)
  int mtiles, ntiles;
};
constexpr int SK_SLOT_FLOATS = NT5_BM * NT5_BN;

__global__ __launch_bounds__(256) void gemm_sk_kernel(const GemmP p, const SkP sk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nt5_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  // Q = G * ntiles workgroups per XCD: workgroup q = (group q / ntiles, column tile q % ntiles). The stream-K iteration space of an XCD
  // runs over its ROW BLOCKS [rb_lo, rb_hi) x K / 64; the ntiles workgroups of a group take the SAME range [I grp / G, I (grp + 1) / G), each for
  // its own column tile: they walk the same A slabs at the same time and share them in the XCD's L2, as the column tiles of a row block
  // do in gemm_nt5's whole-tile order. (A first version that dealt out (tile, k) ranges workgroup by workgroup ran the two column tiles
  // of a row block at different k phases: every A slab came from HBM / MALL twice and the kernel was SLOWER than whole tiles, 51 vs 46 us.)
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, Q = gridDim.x >> 3;
  const int nk = p.K / NT5_BK;
  const int ntc = sk.ntiles, G = Q / ntc, grp = q / ntc, ct = q - grp * ntc;
  const int rb_lo = (int)((long long)sk.mtiles * xcd / 8), rb_hi = (int)((long long)sk.mtiles * (xcd + 1) / 8);
  const long long I = (long long)(rb_hi - rb_lo) * nk;
  const int i0 = (int)(I * grp / G), i1 = (int)(I * (grp + 1) / G);
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* W = reinterpret_cast<const bf16_t*>(p.B);
  auto swz = [](int row) { return (row & 3) | (((row >> 3) & 1) << 2); };

  const int browl = (lr >> 2) * 8 + (lr & 3);                   // W rows of a pair: + t * 4
  const int swa = swz(lr), swb = swz(browl);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)nt5_smem;
  unsigned aa[2], wa[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    aa[ks] = lds0 + (wm * 64 + lr) * 128 + (((ks * 4 + lg) ^ swa) << 4);
    wa[ks] = lds0 + NT5_AB + (wn * 128 + browl) * 128 + (((ks * 4 + lg) ^ swb) << 4);
  }
#define SK_RD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  auto fr = [](const nt5_u32x4_t& v) { return __builtin_bit_cast(bf16x8_t, v); };
  bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C);
  const bf16_t* Rg = reinterpret_cast<const bf16_t*>(p.R);
  const unsigned act_m = opaque_mask(p.act != nullptr) & 0xffu, r_m = opaque_mask(Rg != nullptr), b_m = opaque_mask(p.bias != nullptr);

  int i = i0;
  bool pending = false;
  while (i < i1) {
    const int tl = i / nk, kb = i - tl * nk, ke = min(nk, kb + (i1 - i));
    const int mt = rb_lo + tl, nt = ct;
    const int m0 = mt * NT5_BM, n0 = nt * NT5_BN;

    auto dma = [&](int stage, int k0) {
      unsigned char* as = nt5_smem + stage * NT5_STAGE_B;
      unsigned char* ws = as + NT5_AB;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int sl = u * 256 + tid, row = sl >> 3, ch = (sl & 7) ^ swz(row);
        const bf16_t* src = A + (size_t)min(m0 + row, p.M - 1) * p.lda + k0 + ch * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(as + (u * 256 + wave * 64) * 16), 16, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sl = u * 256 + tid, row = sl >> 3, ch = (sl & 7) ^ swz(row);
        const bf16_t* src = W + (size_t)min(n0 + row, p.N - 1) * p.ldb + k0 + ch * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ws + (u * 256 + wave * 64) * 16), 16, 0, 0);
      }
    };

    f32x4_t acc[4][8];
#pragma unroll
    for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[a_][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // `pending`: the previous segment published a partial whose write-through stores are still draining; this segment's first two
    // stages are requested BEHIND them, one vmcnt(0) covers both, then the flag goes out - the drain hides in the pipeline fill
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // every wave has left the previous segment's stages
    asm volatile("" ::: "memory");
    const int ns = ke - kb;
    dma(0, kb * NT5_BK);
    if (ns > 1) dma(1, (kb + 1) * NT5_BK);
    if (pending) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid == 0) __hip_atomic_store(sk.flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pending = false;
    }
    unsigned so = 0;
    for (int s = 0; s < ns; ++s) {
      if (s + 1 < ns) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + 2 < ns) dma((s + 2) % NT5_ST, (kb + s + 2) * NT5_BK);
      nt5_u32x4_t af[2][4], wf[2][8];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned a_ = aa[ks] + so, w_ = wa[ks] + so;
        SK_RD(af[ks][0], a_, 0); SK_RD(af[ks][1], a_, 2048); SK_RD(af[ks][2], a_, 4096); SK_RD(af[ks][3], a_, 6144);
        SK_RD(wf[ks][0], w_, 0);     SK_RD(wf[ks][1], w_, 512);   SK_RD(wf[ks][2], w_, 4096);  SK_RD(wf[ks][3], w_, 4608);
        SK_RD(wf[ks][4], w_, 8192);  SK_RD(wf[ks][5], w_, 8704);  SK_RD(wf[ks][6], w_, 12288); SK_RD(wf[ks][7], w_, 12800);
      }
      asm volatile("s_waitcnt lgkmcnt(12)"
                   : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]), "+v"(wf[0][3]),
                     "+v"(wf[0][4]), "+v"(wf[0][5]), "+v"(wf[0][6]), "+v"(wf[0][7]));
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int a_ = 0; a_ < 4; ++a_) acc[a_][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr(wf[0][j]), fr(af[0][a_]), acc[a_][j], 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]), "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[1][2]), "+v"(wf[1][3]),
                     "+v"(wf[1][4]), "+v"(wf[1][5]), "+v"(wf[1][6]), "+v"(wf[1][7]));
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int a_ = 0; a_ < 4; ++a_) acc[a_][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr(wf[1][j]), fr(af[1][a_]), acc[a_][j], 0, 0, 0);
      so = (so == (NT5_ST - 1) * NT5_STAGE_B) ? 0u : so + NT5_STAGE_B;
    }
    i += ns;

    if (kb != 0) {
      // ---- publish this segment's partial (register order: piece (a_, j) of lane `tid` at ((a_ * 8 + j) * 256 + tid) * 16 bytes)
      float* slot = sk.part + (size_t)blockIdx.x * SK_SLOT_FLOATS;
#pragma unroll
      for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float* dst = slot + ((size_t)(a_ * 8 + j) * 256 + tid) * 4;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(acc[a_][j]) : "memory");
        }
      if (i < i1) { pending = true; continue; }                  // (a publishing segment is a workgroup's first: another one follows unless its range ends here)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid == 0) __hip_atomic_store(sk.flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    if (ke != nk) {
      // ---- finisher: the other segments of this tile belong to the same column's workgroups of groups grp + 1, grp + 2, ... of this XCD, in k order
      const long long tile_end = (long long)(tl + 1) * nk;
      for (int g2 = grp + 1; g2 < G; ++g2) {
        const long long s2 = I * g2 / G, e2 = I * (g2 + 1) / G;
        if (s2 >= tile_end) break;
        if (e2 <= s2) continue;                                   // (an empty range publishes nothing)
        const int b2 = (g2 * ntc + ct) * 8 + xcd;
        if (tid == 0) {
          while (__hip_atomic_load(sk.flags + b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
          __hip_atomic_store(sk.flags + b2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // consumed: zero for the next launch
        }
        __syncthreads();
        const float* slot = sk.part + (size_t)b2 * SK_SLOT_FLOATS;
        // two halves of 16 pieces, all 16 loads of a half in flight (64 VGPRs): two latencies per partial instead of four
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          f32x4_t pv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const float* src = slot + ((size_t)(hh * 16 + u) * 256 + tid) * 4;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(pv[u]) : "v"(src) : "memory");
          }
          asm volatile("s_waitcnt vmcnt(0)"
                       : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]),
                         "+v"(pv[8]), "+v"(pv[9]), "+v"(pv[10]), "+v"(pv[11]), "+v"(pv[12]), "+v"(pv[13]), "+v"(pv[14]), "+v"(pv[15]));
#pragma unroll
          for (int u = 0; u < 16; ++u) acc[(hh * 16 + u) >> 3][(hh * 16 + u) & 7] += pv[u];
        }
      }
    }

    // ---- epilogue (gemm_nt5.cuh): lane = row m0 + wm*64 + a_*16 + lr, columns n0 + wn*128 + jp*32 + lg*8 + (t*4 + r)
    uint8_t lv[4];
#pragma unroll
    for (int a_ = 0; a_ < 4; ++a_) {
      const int rowc = min(m0 + wm * 64 + a_ * 16 + lr, p.M - 1);
      lv[a_] = *(p.act ? p.act + rowc : reinterpret_cast<const uint8_t*>(p.B));
    }
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const int col = n0 + wn * 128 + jp * 32 + lg * 8;
      const int colc = min(col, p.N - 8);
      const float* bp = p.bias ? p.bias + colc : reinterpret_cast<const float*>(p.B);
      const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
      uint4 rraw[4];
#pragma unroll
      for (int a_ = 0; a_ < 4; ++a_) {
        const int rowc = min(m0 + wm * 64 + a_ * 16 + lr, p.M - 1);
        rraw[a_] = *reinterpret_cast<const uint4*>(Rg ? Rg + (size_t)rowc * p.ldr + colc : reinterpret_cast<const bf16_t*>(p.B));
      }
      if (col >= p.N) continue;                                   // N % 8 == 0 guaranteed by the dispatcher
      const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int a_ = 0; a_ < 4; ++a_) {
        const int row = m0 + wm * 64 + a_ * 16 + lr;
        if (row >= p.M) continue;
        const bool live = ((lv[a_] & act_m) | (~act_m & 1u)) != 0;
        const uint4 rm = make_uint4(rraw[a_].x & r_m, rraw[a_].y & r_m, rraw[a_].z & r_m, rraw[a_].w & r_m);
        float v[8], rr[8];
        rr[0] = __uint_as_float(rm.x << 16); rr[1] = __uint_as_float(rm.x & 0xffff0000u);
        rr[2] = __uint_as_float(rm.y << 16); rr[3] = __uint_as_float(rm.y & 0xffff0000u);
        rr[4] = __uint_as_float(rm.z << 16); rr[5] = __uint_as_float(rm.z & 0xffff0000u);
        rr[6] = __uint_as_float(rm.w << 16); rr[7] = __uint_as_float(rm.w & 0xffff0000u);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = live ? acc[a_][2 * jp + (e >> 2)][e & 3] + __uint_as_float(__float_as_uint(bv[e]) & b_m) + rr[e] : 0.f;
        st8<bf16_t>(Cg + (size_t)row * p.ldc + col, v);
      }
    }
  }
#undef SK_RD
}
