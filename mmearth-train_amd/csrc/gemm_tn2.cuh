// Weight-gradient GEMM, second generation (TN form): dW[n][k] = sum_m P[m][n] * Q[m][k], db[n] = sum_m P[m][n].
//
// Both operands are stored m-major ([M][width], channels-last rows), but an MFMA fragment wants 8
// consecutive reduction (m) elements per lane. gemm_tn_bf16_kernel (gemm_fast.cuh) transposes in
// registers with 4-byte loads + v_perm; this kernel copies the row-major slab to LDS unchanged with
// 16-byte loads and lets gfx950's LDS transpose-read (ds_read_b64_tr_b16) produce the fragments:
//   a 16-lane group reads a [4 m][16 col] block (lane q supplies the address of row q>>2, columns
//   4*(q&3)..+3) and lane q receives column q of it = 4 consecutive m. Two reads (m and m+16) make
//   the 8-element k-vector of mfma_f32_16x16x32_bf16; P and Q use the same m permutation, which is
//   all the contraction needs. LDS row length is an odd multiple of 16 elements so the 8 rows a
//   half-wave touches fall into 8 distinct 32-byte bank groups.
// Tiling: X = the narrow operand (C columns), Y = the wide one (4C columns). A workgroup owns
// 16*NT columns of X (all 4 waves share them) and 64*KT columns of Y (16*KT per wave) and a range of
// rows; every workgroup writes its partial tile to slab[blockIdx.z]. SWAP says X = Q (pwconv1's
// gradient, P is the wide dz): the MFMA operands are exchanged so that lanes still run along the
// contiguous k of the slab. The bias gradient comes out of one more MFMA against a ones fragment.
#pragma once
#include "common.cuh"

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

__device__ __forceinline__ bf16x8_t tn2_frag(const bf16_t* p, int half_stride) {
  // p: this lane's address for rows [0,4) of the slab half; +half_stride elements for rows [16,20)
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(const_cast<bf16_t*>(p)));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(const_cast<bf16_t*>(p + half_stride)));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

constexpr int tn2_ld(int b) { return ((b / 16) & 1) ? b : b + 16; }     // odd multiple of 16

// QGRN (pwconv2's weight gradient, !SWAP): the wide operand is given as h and becomes z = gelu(h) * qp0[k] + qp1[k] (GRN of the GELU,
// qp0 = 1 + gamma * Nx) on its way from the load registers to LDS - the forward then never stores z (100 MB per stage-0 block). Rows
// past the split and the zero rows of inactive sites need no mask: their P rows are zero.
template <int NT, int KT, bool SWAP, bool QGRN = false>
__global__ __launch_bounds__(256) void gemm_tn2_kernel(const WgradP w, int splits) {
  static_assert(!QGRN || !SWAP, "the GRN prologue is applied to the wide (Y) operand");
  constexpr int BX = 16 * NT, BY = 64 * KT, SL = 32;
  constexpr int LDX = tn2_ld(BX), LDY = tn2_ld(BY);
  constexpr int XVR = BX / 8, YVR = BY / 8;                  // 16-byte vectors per slab row
  constexpr int XV = (SL * XVR + 255) / 256, YV = (SL * YVR + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf16_t Xs[2][SL * LDX];
  __shared__ __attribute__((aligned(16))) bf16_t Ys[2][SL * LDY];
  __shared__ __attribute__((aligned(16))) float qsc[QGRN ? BY : 4], qbt[QGRN ? BY : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const bf16_t* X = reinterpret_cast<const bf16_t*>(SWAP ? w.Q : w.P);
  const bf16_t* Y = reinterpret_cast<const bf16_t*>(SWAP ? w.P : w.Q);
  const int ldx = SWAP ? w.ldq : w.ldp, ldy = SWAP ? w.ldp : w.ldq;
  const int WX = SWAP ? w.Kk : w.Nn, WY = SWAP ? w.Nn : w.Kk;
  const int x0 = blockIdx.x * BX, y0 = blockIdx.y * BY;
  const int mbeg = blockIdx.z * w.rows_per_split, mend = min(w.M, mbeg + w.rows_per_split);

  f32x4_t acc[NT][KT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of P: X side when !SWAP (wave 0 of the y-block 0), Y side when SWAP (x-block 0)
  const bool do_db = w.db != nullptr && (SWAP ? blockIdx.x == 0 : (blockIdx.y == 0 && wave == 0));
  constexpr int NB = SWAP ? KT : NT;
  f32x4_t accb[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) accb[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  uint4 xr[XV], yr[YV];
  auto gload = [&](int mb) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int v = tid + 256 * i, r = v / XVR, c = (v - r * XVR) * 8;
      const bool ok = v < SL * XVR && mb + r < mend && x0 + c < WX;
      xr[i] = ok ? *reinterpret_cast<const uint4*>(X + (size_t)(mb + r) * ldx + x0 + c) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < YV; ++i) {
      const int v = tid + 256 * i, r = v / YVR, c = (v - r * YVR) * 8;
      const bool ok = v < SL * YVR && mb + r < mend && y0 + c < WY;
      yr[i] = ok ? *reinterpret_cast<const uint4*>(Y + (size_t)(mb + r) * ldy + y0 + c) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int v = tid + 256 * i, r = v / XVR, c = (v - r * XVR) * 8;
      if (v < SL * XVR) *reinterpret_cast<uint4*>(&Xs[buf][r * LDX + c]) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < YV; ++i) {
      const int v = tid + 256 * i, r = v / YVR, c = (v - r * YVR) * 8;
      if (v < SL * YVR) {
        uint4 y = yr[i];
        if (QGRN && y0 + c < WY) {      // (columns past the operand stay zero: at C = 40 half of the 320-column tile is padding)
          float h[8], g[8];
          h[0] = __uint_as_float(y.x << 16); h[1] = __uint_as_float(y.x & 0xffff0000u);
          h[2] = __uint_as_float(y.y << 16); h[3] = __uint_as_float(y.y & 0xffff0000u);
          h[4] = __uint_as_float(y.z << 16); h[5] = __uint_as_float(y.z & 0xffff0000u);
          h[6] = __uint_as_float(y.w << 16); h[7] = __uint_as_float(y.w & 0xffff0000u);
          gelu_n<bf16_t, 8>(h, g);
          const float4 s0 = *reinterpret_cast<const float4*>(&qsc[c]), s1 = *reinterpret_cast<const float4*>(&qsc[c + 4]);
          const float4 b0 = *reinterpret_cast<const float4*>(&qbt[c]), b1 = *reinterpret_cast<const float4*>(&qbt[c + 4]);
          y.x = f2bf2(g[0] * s0.x + b0.x, g[1] * s0.y + b0.y); y.y = f2bf2(g[2] * s0.z + b0.z, g[3] * s0.w + b0.w);
          y.z = f2bf2(g[4] * s1.x + b1.x, g[5] * s1.y + b1.y); y.w = f2bf2(g[6] * s1.z + b1.z, g[7] * s1.w + b1.w);
        }
        *reinterpret_cast<uint4*>(&Ys[buf][r * LDY + c]) = y;
      }
    }
  };
  if (QGRN) {
    for (int i = tid; i < BY; i += 256) {
      const bool in = y0 + i < WY;
      qsc[i] = in ? w.qp0[y0 + i] : 0.f;
      qbt[i] = in ? w.qp1[y0 + i] : 0.f;
    }
    __syncthreads();
  }

  const int xoff = (lg * 4 + (lr >> 2)) * LDX + 4 * (lr & 3);
  const int yoff = (lg * 4 + (lr >> 2)) * LDY + wave * 16 * KT + 4 * (lr & 3);
  const int nsl = (mend - mbeg + SL - 1) / SL;
  if (nsl > 0) {
    gload(mbeg);
    lstore(0);
  }
  __syncthreads();
  for (int s = 0; s < nsl; ++s) {
    if (s + 1 < nsl) gload(mbeg + (s + 1) * SL);
    const bf16_t* xs = &Xs[s & 1][xoff];
    const bf16_t* ys = &Ys[s & 1][yoff];
    bf16x8_t xf[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) xf[i] = tn2_frag(xs + i * 16, 16 * LDX);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const bf16x8_t yf = tn2_frag(ys + j * 16, 16 * LDY);
#pragma unroll
      for (int i = 0; i < NT; ++i)
        acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, xf[i], acc[i][j], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], yf, acc[i][j], 0, 0, 0);
      if (SWAP && do_db) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, ones, accb[j], 0, 0, 0);
    }
    if (!SWAP && do_db) {
#pragma unroll
      for (int i = 0; i < NT; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], ones, accb[i], 0, 0, 0);
    }
    if (s + 1 < nsl) lstore((s + 1) & 1);
    __syncthreads();
  }

  // D layout: col = lr, row = lg*4 + r. !SWAP: row = x (n), col = y (k). SWAP: row = y (n), col = x (k).
  // slab layout: [split][Nn*Kk weights | Nn bias] so that one second-stage launch folds both
  float* slab = w.ws + (size_t)blockIdx.z * ((size_t)w.Nn * w.Kk + w.Nn);
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int xi = x0 + i * 16, yj = y0 + (wave * KT + j) * 16;
        const int n = SWAP ? yj + lg * 4 + r : xi + lg * 4 + r;
        const int k = SWAP ? xi + lr : yj + lr;
        if (n < w.Nn && k < w.Kk) slab[(size_t)n * w.Kk + k] = acc[i][j][r];
      }
  if (do_db && lr == 0) {
    float* dslab = slab + (size_t)w.Nn * w.Kk;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (SWAP ? y0 + (wave * KT + i) * 16 : x0 + i * 16) + lg * 4 + r;
        if (n < w.Nn) dslab[n] = accb[i][r];
      }
  }
}
