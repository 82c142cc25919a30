// Parameter block (RsP) and the register pack / unpack helpers of the row-streaming fused pointwise kernels (rsc.cuh).
// (The first generation of these kernels - rs_wide / rs_narrow, the whole weight matrix resident in LDS, C = 40 / 80 / 96 only,
// materialised z / dh - lived here through round 3; the chunked kernels of rsc.cuh cover the same widths with the GRN application
// and its backward in the operand prologue, and the resident-weights generation was removed in round 4.)
// Modes (kept by rsc.cuh):
// wide  : N = H outputs.  MODE 0: x-hat/rstd + h = LN(d) W1^T + b1 + sum gelu(h)^2   (LN + pw1)
//                         MODE 1: dz = dout W2 + (sum dz, sum dz*gelu(h))            (pw2 dgrad)
// narrow: N = C outputs.  MODE 0: out = x + GRN(gelu(h)) W2^T + b2                  (pw2)
//                         MODE 1: dd = LNbwd( dh W1 ), dh from (dz, h)              (pw1 dgrad + LN bwd)
#pragma once
#include "gemm.cuh"

struct RsP {
  const bf16_t* A;      // wide: d (MODE 0) / dout (MODE 1) [M,KC];  narrow: h (MODE 0) / dz (MODE 1) [M,HN]
  const bf16_t* A2;     // narrow MODE 1: h [M,HN]
  const bf16_t* W;      // staged weights [N][K] bf16, row stride ldw
  int ldw;
  const float* bias;    // [N] or nullptr
  const float* v0;      // wide MODE 0: ln gamma[KC];  narrow MODE 0: scale[HN];  narrow MODE 1: scale[HN]
  const float* v1;      // wide MODE 0: ln beta[KC];   narrow MODE 0: grn beta[HN]; narrow MODE 1: coef[HN]
  bf16_t* out;          // wide: h / dz [M,HN]; narrow: out / dd [M,KC]
  bf16_t* xhat;         // wide MODE 0: x-hat out [M,KC]; narrow MODE 1: x-hat in [M,KC]
  bf16_t* xn;           // wide MODE 0: optional LN output x-hat*gamma+beta [M,KC] (operand of the pw1 weight gradient)
  float* rstd;          // wide MODE 0: out [M]; narrow MODE 1: in [M]
  const bf16_t* R;      // wide MODE 1: h [M,HN] (statistics); narrow MODE 0: residual x [M,KC]
  const float* lng;     // narrow MODE 1: ln gamma [KC]
  float* ws;            // slab rows: wide: [gridDim.x][HN] (+ second stat); narrow MODE 1: [gridDim.x][2][KC]
  const uint8_t* act;   // row activity or nullptr
  int M;
  // rsc_narrow with LDS-staged vectors: optional folded GRN finalisation (see MpmaeRsArgs.fin_*)
  const float* fin_sum; const float* fin_sum0; const float* fin_gamma;
  float* fin_gx; float* fin_ainv; float* fin_out; float* fin_dgamma; float* fin_dbeta;
  float fin_eps;
  const bf16_t* D; const bf16_t* W2; int ldw2;     // rsc_narrow with operand recomputation: dout / xn [M][C], W2^T / W1 [H][ldw2]
  const float* hb;                                 // MODE 0: pwconv1 bias [H]
  bf16_t* dn_xhat; float* dn_rstd; bf16_t* dn_y; const float* dn_g; const float* dn_b; int dn_S;   // rsc_narrow MODE 0: the downsample LayerNorm of the stage's output, fused (see MpmaeRsArgs.dn_*)
  float* wg_ws;                                    // rsp_narrow<.., WG>: slab rows [gridDim.x][H * C + H] of the fused pwconv1 weight gradient
  float* s0a; float* s1a;                          // rsc_wide: accumulate the column statistics HERE with no-return float atomics (no slab, no fold launch)
  int perwave;                                     // rsc_wide: one LDS statistics row per wave, added in a fixed order (0: one shared row, LDS float atomics)
};

__device__ __forceinline__ bf16x8_t pack_bf16x8(const float (&v)[8]) {
  uint4 u;
  u.x = f2bf2(v[0], v[1]);
  u.y = f2bf2(v[2], v[3]);
  u.z = f2bf2(v[4], v[5]);
  u.w = f2bf2(v[6], v[7]);
  return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ void unpack8(const uint4& a, float (&o)[8]) {
  o[0] = __uint_as_float(a.x << 16); o[1] = __uint_as_float(a.x & 0xffff0000u);
  o[2] = __uint_as_float(a.y << 16); o[3] = __uint_as_float(a.y & 0xffff0000u);
  o[4] = __uint_as_float(a.z << 16); o[5] = __uint_as_float(a.z & 0xffff0000u);
  o[6] = __uint_as_float(a.w << 16); o[7] = __uint_as_float(a.w & 0xffff0000u);
}
