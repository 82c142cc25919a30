// rst: the statistics pass of a fused block's backward at C = 40 / 80, recast as the pwconv2 WEIGHT GRADIENT (round 6).
//
// Where dz is never materialised (stages 0-1: the fused backward kernel recomputes dz = dout W2 chunk by chunk) the main lane ran a pass
// over dout [M][C] and h [M][4C] whose only products were the two GRN-backward H-vectors S0 = sum_m dz, S1 = sum_m dz * gelu(h)
// (rsp_wide MODE 1, out = NULL), and the weight-gradient lane read the SAME two tensors again for dW2 = dout^T GRN(gelu(h)) (gemm_tn2 with the
// GRN operand prologue: 72 us + fold per stage-0 block). Both are functions of ONE product, T = dout^T gelu(h) [C][4C] (+ db2 = sum_m dout):
//     S0[j] = sum_c W2[c][j] db2[c]        S1[j] = sum_c W2[c][j] T[c][j]        dW2[c][j] = scale[j] T[c][j] + beta[j] db2[c]
// (rows.cuh: grn_stats_from_wgrad_kernel, which also adds the parameter gradients). Round 5 produced T with the generic transpose-read
// weight-gradient kernel on the main lane and lost (3.76 vs 3.66 ms: 72 + 12 us in place of the 48 us statistics pass). This kernel produces
// T at the statistics pass's price - the same 125 MB (stage 0) read once, persistent workgroups, the next tile's rows in flight under the
// products - so the second read of dout and h, the gemm_tn2 launch and its fold leave the step.
//
// Contraction over ROWS: an MFMA operand wants 8 consecutive m per lane while rows are channels-last. As in gemm_tn2.cuh the row tile is
// copied to LDS row-major (16-byte vectors, coalesced: a 64-row tile of h is one contiguous 20 KB run; GELU applied on the way) and gfx950's
// transposing LDS read (ds_read_b64_tr_b16) hands every lane its 4 consecutive m of one column; LDS rows are an odd multiple of 16 elements
// (tn2_ld) so the 8 rows a half-wave touches fall into 8 distinct bank groups. One MFMA per (16 c, 16 j, 32 m); a wave owns every third-ish
// j tile of the workgroup's 160-column slice for ALL c tiles and keeps those accumulators for the whole kernel (9 / 15 tiles at C = 40 / 80);
// the idle slot of wave 3 carries the bias gradient (dout^T ones). One slab row [C * 4C | C] per workgroup (-> dW2, db2 on the
// weight-gradient lane: wg_fold_kernel below) and one small row [2][4C] of the workgroup's share of S0 / S1 (-> the main lane's next kernel).
// Workgroup shape (NW waves, 16 NW rows per tile): every workgroup ends with a slab row of 25-100 KB that a second stage must read again, so FEW, FAT
// workgroups: NW = 16 (1024 threads, 256-row tiles, one workgroup per CU, 100 KB of rows in flight per CU) writes 256 slab rows where NW = 4 at three
// workgroups per CU wrote 768 (the fold of those cost 22 / 50 us at C = 40 / 80 in the step - more than the product itself at C = 80).
// grid = (GX persistent workgroups, 4C / 160); block = 64 NW; LDS = 16 NW (LDX + 176) 2 bytes
#pragma once
#include "rsc.cuh"
#include "gemm_tn2.cuh"

template <int KC, int NW>
__global__ __launch_bounds__(64 * NW) void rst_kernel(const RsP p, int ntiles) {
  constexpr int NTH = 64 * NW, HN = 4 * KC, CPS = 160, NX = (KC + 15) / 16, BX = NX * 16, LDX = tn2_ld(BX), LDY = tn2_ld(CPS), RTL = 16 * NW;
  constexpr int XVR = KC / 8, YVR = CPS / 8;                         // 16-byte vectors per tile row
  constexpr int XV = (RTL * XVR + NTH - 1) / NTH, YV = (RTL * YVR + NTH - 1) / NTH;
  constexpr int NJ = CPS / 16, JU = (NJ + NW - 1) / NW;              // 10 column tiles: wave w owns w, w + NW, ... (< 10)
  static_assert(KC % 8 == 0 && HN % CPS == 0 && RTL % 32 == 0 && (RTL * YVR) % NTH == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char rsc_smem[];
  bf16_t* Xs = reinterpret_cast<bf16_t*>(rsc_smem);                  // [64][LDX]  dout rows (columns KC..BX stay zero)
  bf16_t* Ys = Xs + RTL * LDX;                                       // [64][LDY]  gelu(h) rows of this workgroup's column slice
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int n_begin = blockIdx.y * CPS;

  uint4 xr[XV], yr[YV];
  auto request = [&](int tile) {          // clamped addresses; rows beyond M are zeroed when consumed
    const int rb = tile * RTL;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int v = min(tid + NTH * i, RTL * XVR - 1), r = v / XVR, c = (v - r * XVR) * 8;
      xr[i] = *reinterpret_cast<const uint4*>(p.A + (size_t)min(rb + r, p.M - 1) * KC + c);
    }
#pragma unroll
    for (int i = 0; i < YV; ++i) {
      const int v = tid + NTH * i, r = v / YVR, c = (v - r * YVR) * 8;
      yr[i] = *reinterpret_cast<const uint4*>(p.R + (size_t)min(rb + r, p.M - 1) * HN + n_begin + c);
    }
  };
  request(blockIdx.x);
  if (BX != KC) {                          // zero padding columns of the narrow operand, written once
    for (int i = tid; i < RTL * (BX - KC) / 8; i += NTH) {
      const int r = i / ((BX - KC) / 8), c = KC + (i - r * ((BX - KC) / 8)) * 8;
      *reinterpret_cast<uint4*>(Xs + r * LDX + c) = make_uint4(0u, 0u, 0u, 0u);
    }
  }

  // (the bias gradient dout^T ones accumulates in the LAST tile slot of the last wave, which has no column tile: 3 + 4 * 2 = 11 / 15 >= 10)
  static_assert((NW - 1) + NW * (JU - 1) >= NJ, "the last wave's last slot is idle");
  f32x4_t acc[NX][JU];
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int u = 0; u < JU; ++u) acc[i][u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);
  const bool db_wave = wave == NW - 1, do_db = blockIdx.y == 0 && db_wave;      // (a wave with an idle tile slot; every column slice needs db2 for S0, slice 0 stores it)
  const int xoff = (lg * 4 + (lr >> 2)) * LDX + 4 * (lr & 3);
  const int yoff = (lg * 4 + (lr >> 2)) * LDY + 4 * (lr & 3);

#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int rb = tile * RTL;
    // ---- this tile's rows: registers -> LDS (h through GELU, rounded to bf16 like every operand of the bf16 mode)
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int v = tid + NTH * i, r = v / XVR, c = (v - r * XVR) * 8;
      if (v < RTL * XVR) *reinterpret_cast<uint4*>(Xs + r * LDX + c) = and4(xr[i], rb + r < p.M);
    }
#pragma unroll
    for (int i = 0; i < YV; ++i) {
      const int v = tid + NTH * i, r = v / YVR, c = (v - r * YVR) * 8;
      float h[8], g[8];
      unpack8(yr[i], h);
      gelu_n<bf16_t, 8>(h, g);
      *reinterpret_cast<uint4*>(Ys + r * LDY + c) = __builtin_bit_cast(uint4, pack_bf16x8(g));      // (rows beyond M: their dout rows are zero)
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);       // the next tile's rows travel under this tile's products
#pragma unroll
    for (int ks = 0; ks < RTL / 32; ++ks) {
      const bf16_t* xs = Xs + ks * 32 * LDX + xoff;
      const bf16_t* ys = Ys + ks * 32 * LDY + yoff;
      bf16x8_t xf[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) xf[i] = tn2_frag(xs + i * 16, 16 * LDX);
#pragma unroll
      for (int u = 0; u < JU; ++u) {
        const int jt = wave + NW * u;
        if (jt < NJ) {                      // (wave-uniform)
          const bf16x8_t yf = tn2_frag(ys + jt * 16, 16 * LDY);
#pragma unroll
          for (int i = 0; i < NX; ++i) acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf, acc[i][u], 0, 0, 0);
        } else if (u == JU - 1 && db_wave) {
#pragma unroll
          for (int i = 0; i < NX; ++i) acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], ones, acc[i][u], 0, 0, 0);
        }
      }
    }
    __syncthreads();                        // every wave has read the tile before the next one overwrites it
  }

  // ---- one slab row per workgroup: [KC * HN | KC]; D layout: row (c) = lg * 4 + r, column (j) = lr
  float* slab = p.ws + (size_t)blockIdx.x * ((size_t)KC * HN + KC);
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int u = 0; u < JU; ++u) {
      const int jt = wave + NW * u;
      if (jt < NJ) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = i * 16 + lg * 4 + r;
          if (c < KC) slab[(size_t)c * HN + n_begin + jt * 16 + lr] = acc[i][u][r];
        }
      }
    }
  if (do_db && lr == 0) {
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = i * 16 + lg * 4 + r;
        if (c < KC) slab[(size_t)KC * HN + c] = acc[i][JU - 1][r];
      }
  }
  // ---- this workgroup's share of the GRN backward statistics, straight from the accumulators (everything downstream is linear in T):
  //   S1'[j] = sum_c W2[c][j] T'[c][j],  S0'[j] = sum_c W2[c][j] db2'[c]  ->  small slab row [2][HN] behind the big slabs (p.s0a): the fold the NEXT
  // main-lane kernel waits for reads gridDim.x * 2 HN floats instead of gridDim.x * (KC HN + KC); the big slabs fold into dW2 / db2 on the weight-gradient lane
  if (p.s0a) {
    float* dbs = reinterpret_cast<float*>(rsc_smem);               // (the row tiles are dead: last barrier of the loop)
    if (wave == NW - 1 && lr == 0) {
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) dbs[i * 16 + lg * 4 + r] = acc[i][JU - 1][r];
    }
    __syncthreads();
    float* srow = p.s0a + (size_t)blockIdx.x * 2 * HN;
#pragma unroll
    for (int u = 0; u < JU; ++u) {
      const int jt = wave + NW * u;
      if (jt < NJ) {
        const int j = n_begin + jt * 16 + lr;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = i * 16 + lg * 4 + r;
            const float w = c < KC ? bf2f(p.W[(size_t)min(c, KC - 1) * p.ldw + j]) : 0.f;
            s0 += w * dbs[c];
            s1 += w * acc[i][u][r];
          }
        s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        if (lg == 0) { srow[j] = s0; srow[HN + j] = s1; }
      }
    }
  }
}

// Second stage of the weight gradients that are accumulated INSIDE main-lane kernels (rst_kernel: T = dout^T gelu(h) | db2; rsp_narrow_kernel<.., WG>:
// U = dh^T x-hat | db1): slab rows [P][A * B | A] -> the parameter gradients with the per-column affine applied by linearity,
//     dW[a][b] += v0[b] * sum_p X_p[a][b] + v1[b] * sum_p d_p[a]          db[a] += sum_p d_p[a]
// pwconv2: (A, B) = (C, H), v0 = GRN scale, v1 = GRN beta (z = gelu(h) * scale + beta is never needed as a tensor);
// pwconv1: (A, B) = (H, C), v0 / v1 = LayerNorm gamma / beta (xn = x-hat * gamma + beta is never needed as a tensor).
// grid = (ceil(A B / 64), R row chunks); block = 64 elements x 4 row lanes, like reduce_partials; B >= 32
__global__ __launch_bounds__(256) void wg_fold_kernel(const float* __restrict__ part, int P, int A, int B, const float* __restrict__ v0, const float* __restrict__ v1,
                                                      float* __restrict__ dW, float* __restrict__ db) {
  constexpr int NJB = 4;                       // rows a touched by 64 consecutive elements: <= 64 / B + 2
  const int NE = A * B;
  const size_t W = (size_t)NE + A;
  __shared__ float red[4][64];
  __shared__ float dred[4][NJB];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + col, a0 = (blockIdx.x * 64) / B;
  const int chunk = (P + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * chunk, p1 = min(P, p0 + chunk);
  float s = 0.f;
  if (e < NE) {
#pragma unroll 4
    for (int p = p0 + rl; p < p1; p += 4) s += part[(size_t)p * W + e];
  }
  if (col < NJB) {
    float d = 0.f;
    if (a0 + col < A) for (int p = p0 + rl; p < p1; p += 4) d += part[(size_t)p * W + NE + a0 + col];
    dred[rl][col] = d;
  }
  red[rl][col] = s;
  __syncthreads();
  if (rl != 0 || e >= NE) return;
  s = red[0][col] + red[1][col] + red[2][col] + red[3][col];
  const int a = e / B, b = e - a * B;
  const float d = dred[0][a - a0] + dred[1][a - a0] + dred[2][a - a0] + dred[3][a - a0];
  const float g = v0[b] * s + v1[b] * d;
  if (gridDim.y == 1) { dW[e] += g; if (b == 0) db[a] += d; }
  else { atomicAdd(dW + e, g); if (b == 0) atomicAdd(db + a, d); }
}
