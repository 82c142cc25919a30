// MX-fp8 NT GEMM for the pointwise layers with K % 128 == 0 (decoder Block pwconv1 / pwconv2, forward and data gradient):
//   C[M,N] (bf16) = deq(A)[M,K] * deq(B)[N,K]^T + bias (+ R, row mask)          EPI_STORE / EPI_RESID
// Operands are OCP e4m3 bytes with E8M0 block scales (one per 32 consecutive k, OCP MX v1.0), multiplied by
// v_mfma_scale_f32_16x16x128_f8f6f4 - the only fp8 MFMA on gfx950 that runs above the bf16 rate (MI355X_MICROARCH.md).
// 256 x 128 output tiles, 8 waves (4 along M x 2 along N, 64 x 64 per wave = 4 x 4 MFMA tiles), K slabs of 128 bytes per
// row that go global -> LDS directly (global_load_lds_dwordx4, no staging VGPRs) into a THREE-slab ring: at the top of
// iteration kt the slab kt + 2 is requested, a counted `s_waitcnt vmcnt(7)` retires exactly slab kt (7 DMA instructions per
// thread and slab: 4 A + 2 B + 1 scale line; two slabs stay in flight) and ONE raw s_barrier per slab both publishes slab
// kt and frees the ring slot slab kt + 2 lands in (cdna_hip_programming.md: __syncthreads() would drain the ring).
// The LDS image of a DMA is lane-linear, so rows are unpadded 128-byte lines; bank conflicts of the fragment reads are
// removed by XOR-ing the 16-byte chunk index with a function of the row on the SOURCE address and again on the read
// (gemm_fast.cuh). MFMAs are issued transposed over interleaved column-tile pairs (D[n][m] = B . A^T): a lane ends with 8
// consecutive output columns of one row, so the epilogue is lane-local and leaves as 16-byte stores from the accumulators.
// Scales are stored slab-major, sc[K/128][rows] dwords (the 4 block scales of a row and slab in one dword): a slab's scales
// are one contiguous line and ride in the same ring.
// The same skeleton with bf16 operands (16x16x32 MFMA, 64-element slabs) was measured SLOWER than the 128-row-tile kernels
// of gemm_fast.cuh on every decoder / head / tiny-stage shape (416 vs 482, 610 vs 702, 474 vs 561 TFLOP/s ...,
// profiles/r02/gemm_probe.txt) and is not kept: at 26-GFLOP problems the ring buys nothing over two resident 128 x 128 blocks.
// Reference ops: models/convnextv2.py:46-52 (decoder Block pwconv1/2).
#pragma once
#include "gemm_fast.cuh"

constexpr int N3_BM = 256, N3_BN = 128, N3_RB = 128, N3_ST = 3, N3_T = 512;
constexpr int N3_A_B = N3_BM * N3_RB, N3_B_B = N3_BN * N3_RB;             // bytes per slab
constexpr int N3_SC_B = (N3_BM + N3_BN + 128) * 4;                         // scale dwords of a slab (FP8) + 2 waves' dummy lines
constexpr int N3_STAGE_B = N3_A_B + N3_B_B + N3_SC_B;

typedef int v8i_t __attribute__((ext_vector_type(8)));

struct Nt3Scales { const uint32_t* sa; const uint32_t* sb; int lsa, lsb; };      // [K/128][lsa] / [K/128][lsb] dwords

template <int EPI>
__global__ __launch_bounds__(N3_T) void gemm_nt3_kernel(const GemmP p, const Nt3Scales sc) {
  constexpr int ESZ = 1;
  constexpr bool FP8 = true;
  extern __shared__ __attribute__((aligned(16))) unsigned char n3_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * N3_BM, n0 = blockIdx.y * N3_BN;
  const unsigned char* A = reinterpret_cast<const unsigned char*>(p.A);
  const unsigned char* B = reinterpret_cast<const unsigned char*>(p.B);
  const size_t lda = (size_t)p.lda * ESZ, ldb = (size_t)p.ldb * ESZ;
  const int nk = (p.K * ESZ) / N3_RB;                       // K * ESZ % 128 == 0 (dispatcher)

  auto swz = [](int row) { return (row & 3) | (((row >> 3) & 1) << 2); };
  // per-thread DMA sources (row clamped: products of rows / columns beyond M / N are never stored)
  const unsigned char* srcA[4];
  const unsigned char* srcB[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int sl = i * N3_T + tid, row = sl >> 3, ch = (sl & 7) ^ swz(row);
    srcA[i] = A + (size_t)min(m0 + row, p.M - 1) * lda + ch * 16;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int sl = i * N3_T + tid, row = sl >> 3, ch = (sl & 7) ^ swz(row);
    srcB[i] = B + (size_t)min(n0 + row, p.N - 1) * ldb + ch * 16;
  }
  auto dma = [&](int slot, int kt) {
    unsigned char* base = n3_smem + slot * N3_STAGE_B;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + (size_t)kt * N3_RB), (lptr_t)(base + (i * N3_T + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcB[i] + (size_t)kt * N3_RB), (lptr_t)(base + N3_A_B + (i * N3_T + wave * 64) * 16), 16, 0, 0);
    if (FP8) {      // 384 scale dwords of the slab: waves 0-3 the A rows, waves 4-5 the B rows (4 bytes per lane)
      if (wave < 4)
        __builtin_amdgcn_global_load_lds((gptr_t)(sc.sa + (size_t)kt * sc.lsa + min(m0 + wave * 64 + lane, p.M - 1)),
                                         (lptr_t)(base + N3_A_B + N3_B_B + wave * 256), 4, 0, 0);
      else      // waves 6-7 repeat the B scales into a dummy line: every wave issues the SAME number of DMAs per slab (vmcnt counts)
        __builtin_amdgcn_global_load_lds((gptr_t)(sc.sb + (size_t)kt * sc.lsb + min(n0 + ((wave - 4) & 1) * 64 + lane, p.N - 1)),
                                         (lptr_t)(base + N3_A_B + N3_B_B + N3_BM * 4 + (wave - 4) * 256), 4, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int swa = swz(lr);                                   // A rows are lr (mod 16)
  const int browl = (lr >> 2) * 8 + (lr & 3);                // B rows of a pair: + t * 4
  const int swb = swz(browl);
  const int arow = (wm * 64 + lr) * N3_RB, brow = (wn * 64 + browl) * N3_RB;

  dma(0, 0);
  if (nk > 1) dma(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    // slab kt has landed for this wave's own requests once at most one younger slab (6 or 7 instructions) is outstanding
    if (kt + 1 < nk) {
      if (FP8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                              // everyone's slab kt is in LDS; everyone is done with slab kt - 1
    if (kt + 2 < nk) dma((kt + 2) % N3_ST, kt + 2);
    const unsigned char* as = n3_smem + (kt % N3_ST) * N3_STAGE_B + arow;
    const unsigned char* bs = n3_smem + (kt % N3_ST) * N3_STAGE_B + N3_A_B + brow;
    {
      const uint32_t* scs = reinterpret_cast<const uint32_t*>(n3_smem + (kt % N3_ST) * N3_STAGE_B + N3_A_B + N3_B_B);
      // operand layout of the 8-bit 16x16x128 form (probed on hardware, tools/mx_debug.py): lane group g holds k = 16g..16g+15 in
      // its first 16 bytes and k = 64+16g..64+16g+15 in its second, and supplies the scale of k-block g (k = 32g..32g+31)
      const int c0a = (lg ^ swa) * 16, c1a = ((lg + 4) ^ swa) * 16;
      const int c0b = (lg ^ swb) * 16, c1b = ((lg + 4) ^ swb) * 16;
      v8i_t af[4], bfr[4];
      int sa[4], sb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4 lo = *reinterpret_cast<const uint4*>(as + i * 16 * N3_RB + c0a), hi = *reinterpret_cast<const uint4*>(as + i * 16 * N3_RB + c1a);
        af[i] = (v8i_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        sa[i] = (int)(scs[wm * 64 + i * 16 + lr] >> (8 * lg));            // this lane's block scale in byte 0
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = ((j >> 1) * 32 + (j & 1) * 4);
        const uint4 lo = *reinterpret_cast<const uint4*>(bs + r * N3_RB + c0b), hi = *reinterpret_cast<const uint4*>(bs + r * N3_RB + c1b);
        bfr[j] = (v8i_t){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        sb[j] = (int)(scs[N3_BM + wn * 64 + r + browl] >> (8 * lg));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, sb[j], 0, sa[i]);
    }
  }

  // ---- epilogue: lane = row m0 + wm*64 + i*16 + lr, columns n0 + wn*64 + jp*32 + lg*8 + (t*4 + r)
  bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C);
  const bf16_t* Rg = reinterpret_cast<const bf16_t*>(p.R);
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    const int cl = wn * 64 + jp * 32 + lg * 8, col = n0 + cl;
    const bool cin = col < p.N;                                // N % 8 == 0 (dispatcher)
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias && cin) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col), b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + wm * 64 + i * 16 + lr;
      const bool rin = row < p.M && cin;
      const bool live = rin && (!p.act || p.act[row]);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = live ? acc[i][2 * jp + (e >> 2)][e & 3] + bv[e] : 0.f;
      if (EPI == EPI_RESID && rin && Rg) {
        float rr[8];
        ld8<bf16_t>(Rg + (size_t)row * p.ldr + col, rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = live ? v[e] + rr[e] : 0.f;
      }
      if (rin) st8<bf16_t>(Cg + (size_t)row * p.ldc + col, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// MX quantisation of a bf16 matrix [rows][K] (row stride ld): q[rows][K] e4m3 bytes, block scales (E8M0, one per 32
// consecutive k: shared exponent = floor(log2(amax)) - 8, OCP MX v1.0) in the slab-major layout the GEMM rides,
// sc[K/128][lds] dwords. One lane = one 32-element block (4 x 16-byte loads), 4 lanes = one 128-element slab of a row.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_mx_kernel(const bf16_t* __restrict__ x, int ld, int rows, int K,
                                                       unsigned char* __restrict__ q, uint32_t* __restrict__ sc, int lds) {
  const int nblk = K / 32;
  const long long total = (long long)rows * nblk;
  for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < total; b += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(b / nblk), kb = (int)(b - (long long)row * nblk);
    const bf16_t* src = x + (size_t)row * ld + kb * 32;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t[8];
      ld8<bf16_t>(src + i * 8, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i * 8 + e] = t[e]; amax = fmaxf(amax, fabsf(t[e])); }
    }
    // shared exponent: 2^se with se = floor(log2(amax)) - 8 (e4m3 emax), clamped to the E8M0 range; amax = 0 -> smallest scale
    int se = -127;
    if (amax > 0.f && amax < 3.0e38f) se = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
    se = max(-127, min(127, se));
    const float inv = __uint_as_float((uint32_t)(127 - se) << 23);          // 2^-se (se in [-127, 127] -> exponent field 0..254)
    uint32_t packed[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int w = 0;
      // amax / 2^se lies in [256, 512): the MX conversion saturates at the e4m3 maximum (448) instead of producing NaN
      auto sat = [&](float f) { return __builtin_amdgcn_fmed3f(f * inv, -448.f, 448.f); };
      w = __builtin_amdgcn_cvt_pk_fp8_f32(sat(v[4 * i]), sat(v[4 * i + 1]), w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(sat(v[4 * i + 2]), sat(v[4 * i + 3]), w, true);
      packed[i] = (uint32_t)w;
    }
    uint4* dst = reinterpret_cast<uint4*>(q + (size_t)row * K + kb * 32);
    dst[0] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    dst[1] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
    // the 4 block scales of (row, slab) form one dword: byte (kb & 3); assembled across the 4 adjacent lanes
    uint32_t sb = (uint32_t)(se + 127) << (8 * (kb & 3));
    sb |= __shfl_xor(sb, 1, 64);
    sb |= __shfl_xor(sb, 2, 64);
    if ((kb & 3) == 0) sc[(size_t)(kb >> 2) * lds + row] = sb;
  }
}
