// Common device helpers for the MP-MAE HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mpmae_hip.h"

#define MPMAE_CHECK_LAUNCH() ((int)hipGetLastError())

typedef uint16_t bf16_t;  // raw bfloat16 bits; all arithmetic is done in fp32 registers

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---------------------------------------------------------------------------------
// scalar conversions
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even through gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR of values; the
// integer emulation this replaces cost 6-7 VALU instructions per value in every epilogue)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// 8-wide contiguous load / store (p must be 8-element aligned for the vector path)
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&o)[8]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  o[0] = __uint_as_float(a.x << 16); o[1] = __uint_as_float(a.x & 0xffff0000u);
  o[2] = __uint_as_float(a.y << 16); o[3] = __uint_as_float(a.y & 0xffff0000u);
  o[4] = __uint_as_float(a.z << 16); o[5] = __uint_as_float(a.z & 0xffff0000u);
  o[6] = __uint_as_float(a.w << 16); o[7] = __uint_as_float(a.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  uint4 a;
  a.x = f2bf2(v[0], v[1]);
  a.y = f2bf2(v[2], v[3]);
  a.z = f2bf2(v[4], v[5]);
  a.w = f2bf2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = a;
}

// ---------------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) {            // exact (erf) GELU
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Fast exact-form GELU for the bf16 path: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far
// below bf16 resolution), sharing one exp(-x^2/2) between Phi(x) and phi(x). The fp32 parity path
// keeps libm erff.
__device__ __forceinline__ void gelu_both_fast(float x, float& g, float& dg) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  const float e = __expf(-0.5f * x * x);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  const float cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  g = x * cdf;
  dg = cdf + x * e * 0.39894228040143267794f;
}
// Forward-only GELU for the bf16 path: x * sigmoid(x (c0 + c1 x^2 + c2 x^4)), a minimax fit of x Phi(x):
// |error| <= 2.6e-5 over EVERY finite bf16 input (checked exhaustively; bf16 rounding of an O(1)
// activation is 4e-3), 14 VALU issue slots (1 v_exp_f32 + 1 v_rcp_f32 + 6 fma/mul/min) instead of 20 for the
// erf form. The exponent polynomial is pre-scaled by -log2(e); x^2 is clamped at 64 because the quartic
// term turns the polynomial around at |x| ~ 11. Sites that also need GELU' keep gelu_both_fast; the fp32
// parity path keeps libm erff.
__device__ __forceinline__ float gelu_fwd_fast(float x) {
  const float x2 = fminf(x * x, 64.f);
  const float u = x * (-2.301121235f + x2 * (-1.067757234e-01f + x2 * 1.014263020e-03f));
  const float e = __builtin_amdgcn_exp2f(u);
  return x * __builtin_amdgcn_rcpf(1.f + e);
}
// same fit with its analytic derivative: g = x s, g' = s + x s (1 - s) (c0 + 3 c1 x^2 + 5 c2 x^4), s = sigmoid(x P(x^2));
// |g' - GELU'| <= 1.1e-4 over every finite bf16 input; 17 issue slots instead of 24 for the erf form
__device__ __forceinline__ void gelu_both_fwd_fast(float x, float& g, float& dg) {
  const float x2 = fminf(x * x, 64.f);
  const float u = x * (-2.301121235f + x2 * (-1.067757234e-01f + x2 * 1.014263020e-03f));
  const float e = __builtin_amdgcn_exp2f(u);
  const float s = __builtin_amdgcn_rcpf(1.f + e);
  const float v = 1.59501577f + x2 * (3.f * 7.40112921e-02f + x2 * (5.f * -7.03033580e-04f));
  g = x * s;
  dg = s + g * (1.f - s) * v;
}
// Transcendental-free GELU for the bf16 path, two values per instruction (v_pk_fma_f32):
//   GELU(x) = relu(x) - t Q(t),  t = min(|x|, 4.5),  t Q(t) ~ t Phi(-t)  (the bump between GELU and ReLU, Q of degree 9)
//   GELU'(x) = 0.5 + clamp(x, -4.5, 4.5) R(t)                            (R of degree 8)
// v_exp_f32 / v_rcp_f32 issue at quarter rate, so the logistic form above spends 8 of its 14 slots on two
// instructions; this one costs 2 v_med3 + 5 packed fma = 7 slots per value (12.5 with the derivative instead of 17).
// Minimax fits (tools/gelu_fit.py): |GELU error| <= 3.1e-5, |GELU' error| <= 1.6e-4 over every finite bf16 input,
// evaluated in fp32 exactly as here; exact 0 at 0, exact ReLU beyond +-4.5 up to 1.5e-5.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu2_bump(f32x2_t t) {      // t Q(t) without the leading t
  f32x2_t q = (f32x2_t)(-1.289379179e-05f);
  q = q * t + (f32x2_t)(3.050061932e-04f);
  q = q * t + (f32x2_t)(-2.969613997e-03f);
  q = q * t + (f32x2_t)(1.488442346e-02f);
  q = q * t + (f32x2_t)(-3.750750050e-02f);
  q = q * t + (f32x2_t)(2.816627920e-02f);
  q = q * t + (f32x2_t)(5.219671130e-02f);
  q = q * t + (f32x2_t)(1.680976129e-03f);
  q = q * t + (f32x2_t)(-3.978458941e-01f);
  q = q * t + (f32x2_t)(4.997446537e-01f);
  return q;
}
__device__ __forceinline__ f32x2_t gelu2_fwd(f32x2_t x) {
  f32x2_t t, r;
  // v_med3_f32 with an |x| source modifier; a finite upper bound keeps hipcc from rewriting it as canonicalise + max
  t.x = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.x), 0.f, 4.5f); t.y = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.y), 0.f, 4.5f);
  r.x = __builtin_amdgcn_fmed3f(x.x, 0.f, 3.4028234e38f); r.y = __builtin_amdgcn_fmed3f(x.y, 0.f, 3.4028234e38f);
  return r - t * gelu2_bump(t);
}
__device__ __forceinline__ void gelu2_both(f32x2_t x, f32x2_t& g, f32x2_t& dg) {
  f32x2_t t, r, xc;
  t.x = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.x), 0.f, 4.5f); t.y = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.y), 0.f, 4.5f);
  r.x = __builtin_amdgcn_fmed3f(x.x, 0.f, 3.4028234e38f); r.y = __builtin_amdgcn_fmed3f(x.y, 0.f, 3.4028234e38f);
  xc.x = __builtin_amdgcn_fmed3f(x.x, -4.5f, 4.5f); xc.y = __builtin_amdgcn_fmed3f(x.y, -4.5f, 4.5f);
  g = r - t * gelu2_bump(t);
  f32x2_t q = (f32x2_t)(1.184819193e-04f);
  q = q * t + (f32x2_t)(-2.549645957e-03f);
  q = q * t + (f32x2_t)(2.225454524e-02f);
  q = q * t + (f32x2_t)(-9.808389843e-02f);
  q = q * t + (f32x2_t)(2.110620588e-01f);
  q = q * t + (f32x2_t)(-1.232886910e-01f);
  q = q * t + (f32x2_t)(-2.190126926e-01f);
  q = q * t + (f32x2_t)(-4.309557844e-03f);
  q = q * t + (f32x2_t)(7.970378995e-01f);
  dg = xc * q + (f32x2_t)(0.5f);
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) {
  if (sizeof(T) == 2) { const f32x2_t v = {x, x}; return gelu2_fwd(v).x; }
  return gelu_f(x);
}
// N (even) values at once: the bf16 path pairs them up for the packed forms
template <typename T, int N> __device__ __forceinline__ void gelu_n(const float (&x)[N], float (&g)[N]) {
  static_assert(N % 2 == 0, "pairs");
  if (sizeof(T) == 2) {
#pragma unroll
    for (int e = 0; e < N; e += 2) { const f32x2_t v = {x[e], x[e + 1]}; const f32x2_t o = gelu2_fwd(v); g[e] = o.x; g[e + 1] = o.y; }
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) g[e] = gelu_f(x[e]);
  }
}
template <typename T, int N> __device__ __forceinline__ void gelu_both_n(const float (&x)[N], float (&g)[N], float (&dg)[N]) {
  static_assert(N % 2 == 0, "pairs");
  if (sizeof(T) == 2) {
#pragma unroll
    for (int e = 0; e < N; e += 2) {
      const f32x2_t v = {x[e], x[e + 1]};
      f32x2_t a, b;
      gelu2_both(v, a, b);
      g[e] = a.x; g[e + 1] = a.y; dg[e] = b.x; dg[e + 1] = b.y;
    }
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) { g[e] = gelu_f(x[e]); dg[e] = gelu_grad_f(x[e]); }
  }
}
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) {
  if (sizeof(T) == 2) { float g, dg; gelu_both_fwd_fast(x, g, dg); return dg; }
  return gelu_grad_f(x);
}
template <typename T> __device__ __forceinline__ void gelu_both_t(float x, float& g, float& dg) {
  if (sizeof(T) == 2) gelu_both_fwd_fast(x, g, dg);
  else { g = gelu_f(x); dg = gelu_grad_f(x); }
}

// sum over the 16 lanes of a DPP row (lanes sharing lane>>4), result in every lane: 4 v_add_f32_dpp
// (quad_perm xor 1, xor 2, row_ror 4, row_ror 8) instead of 4 ds_bpermute round trips
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum16(float v) {
  v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x124>(v); v += dpp_mov<0x128>(v);
  return v;
}

// sum over the 64 lanes (every lane active), result in every lane: the DPP row sum, then the four row results through
// v_readlane (SGPRs) - 4 DPP adds + 4 readlanes + 3 adds instead of 6 dependent ds_bpermute round trips
__device__ __forceinline__ float wave_sum(float v) {
  const int r = __builtin_bit_cast(int, sum16(v));
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 48));
  return (a + b) + (c + d);
}

// ---------------------------------------------------------------------------------
// sparse row geometry: rows of a stage are (n, slot k, intra-patch point q = iy*S + ix)
//   row = (n*keep + k)*S*S + q ; slot k <-> patch index vis[n*keep + k] on the grid x grid grid
//   inv[n*L + patch] = slot or -1.  inv == nullptr means "all patches visible, slot = patch".
// ---------------------------------------------------------------------------------
typedef MpmaeGeom Geom;

__device__ __forceinline__ int geom_row_of(const Geom& g, int n, int gy, int gx) {
  // (gy, gx) in stage-point coordinates on the dense (grid*S)^2 map; returns row or -1
  const int ext = g.grid * g.S;
  if (gy < 0 || gx < 0 || gy >= ext || gx >= ext) return -1;
  const int py = gy / g.S, px = gx / g.S;
  const int patch = py * g.grid + px;
  const int slot = g.inv ? g.inv[n * g.grid * g.grid + patch] : patch;
  if (slot < 0) return -1;
  return (n * g.keep + slot) * g.S * g.S + (gy - py * g.S) * g.S + (gx - px * g.S);
}

// Element offset of row m of a stage with S points per patch side inside the "grouped" matrix
// [parent rows][4][C] that the 2x2-stride-2 downsample convolution reads as a plain GEMM operand:
// parent = (patch, iy/2, ix/2), group kidx = (ix&1)*2 + (iy&1) (k = kidx*C + c, gemm.cuh down_child_row).
__device__ __forceinline__ size_t down_group_off(int m, int S, int C) {
  const int P = S * S, nk = m / P, q = m - nk * P, cy = q / S, cx = q - cy * S;
  const int Sp = S >> 1;
  const int parent = nk * (Sp * Sp) + (cy >> 1) * Sp + (cx >> 1);
  return ((size_t)parent * 4 + ((cx & 1) * 2 + (cy & 1))) * C;
}

// An all-ones / zero mask (in a VGPR) the optimizer cannot see through. `x = ptr ? load(ptr + i) : 0` (and `load(...) & (ptr ? ~0 : 0)`,
// which instcombine folds back into the select) compile to a branch around the load with an `s_waitcnt vmcnt(0)` at its join: N optional
// operands in a row become N serial round trips. With an opaque mask the load stays unconditional (from `ptr ? ptr + i : some_valid_address`)
// and only the AND depends on it.
__device__ __forceinline__ unsigned opaque_mask(bool on) {
  unsigned m = on ? 0xffffffffu : 0u;
  asm volatile("" : "+v"(m));
  return m;
}

// Workgroup -> tile mapping of the 2-D tiled kernels, XCD-aware: the hardware deals workgroup L (x fastest) to XCD L % 8, and the XCDs' L2s
// are private. With the plain mapping the column tiles of one row tile (which read the SAME rows of A, the large operand of every NT
// product here) have ids L, L + gx, L + 2 gx ... -> different XCDs and different moments: every one of them pulled its A rows over the fabric
// (pmc: 254 MB fetched per launch for 45 MB of operands at the decoder / head shapes). Here the j-th workgroup of XCD c takes row tile
// (j / gy) * 8 + c, column tile j % gy: the gy column tiles of a row tile run back to back on ONE XCD and share its L2. (The last gx % 8 row
// tiles keep the plain order.)
__device__ __forceinline__ void xcd_tile(int& mt, int& nt) {
  const int gx = gridDim.x, gy = gridDim.y, L = blockIdx.x + blockIdx.y * gx, gxm = gx & ~7;
  if (L < gxm * gy) {
    const int c = L & 7, j = L >> 3;
    mt = (j / gy) * 8 + c; nt = j - (j / gy) * gy;
  } else {
    const int t = L - gxm * gy;
    mt = gxm + t / gy; nt = t - (t / gy) * gy;
  }
}

