// Shared host/device helpers for librcb_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/rcb.h"

namespace rcb {

constexpr int kWave = 64;

// thread-local last error text (diagnostics only)
char* last_error_buf();
int fail(int code, const char* fmt, ...);

#define RCB_REQUIRE(cond, code, ...)            \
  do {                                          \
    if (!(cond)) return rcb::fail((code), __VA_ARGS__); \
  } while (0)

#define RCB_LAUNCH_CHECK()                                                         \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) return rcb::fail((int)e__, "launch failed: %s", hipGetErrorString(e__)); \
  } while (0)

// ---- elementwise math of the posterior kernels ----------------------------------------------------------------------
// The library forms (expf, log1pf, logf, IEEE division as v_div_scale / v_div_fmas / v_div_fixup) cost 30 - 150 VALU
// instructions each: softplus + sigmoid + Gaussian KL + two Adam updates came to ~680 instructions per parameter
// element and made the "HBM-bound" posterior kernels compute-bound.  These forms use the hardware v_exp / v_log / v_rcp
// / v_sqrt with one correction step each: <= 2 ulp from the correctly rounded value (the library forms are <= 1 - 2 ulp
// themselves, and torch's CPU kernels differ from either by an ulp here and there), ~10x fewer instructions.
// All of them live under the caller's `#pragma clang fp contract(off)`: the fmas below are written out.

// a / b for finite, normal operands: reciprocal + one residual correction (<= 1 ulp)
__device__ __forceinline__ float div_fast(float a, float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  const float q = a * r;
  const float e = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(e, r, q);
}

// e^x, |x| <= 80: the product x * log2(e) is carried as hi + lo so that the argument error does not grow with |x|
__device__ __forceinline__ float exp_fast(float x) {
  const float L = 1.44269502162933349609375f, L_lo = 1.92596299112661746e-8f;     // log2(e) = L + L_lo
  const float hi = x * L;
  const float lo = __builtin_fmaf(x, L, -hi) + x * L_lo;
  const float e = __builtin_amdgcn_exp2f(hi);
  return __builtin_fmaf(e, lo * 0.693147182464599609375f, e);                       // e * 2^lo ~ e (1 + lo ln 2)
}

// ln u for u away from 0 (hardware log2, <= 1 ulp, times ln 2).  PRECONDITION shared with div_fast / kl_elem_f32: normal,
// finite operands.  v_log / v_rcp flush denormal inputs, so a denormal variance ratio or prior scale would give -inf / inf
// where the library forms stay finite.  Not enforced: the scales here are softplus(log_scale) / 6 with log_scale started at
// -4 (prior_model.py:103) and moved by Adam steps of ~1e-4, and softplus(x) / 6 stays a normal float down to x ~ -85.
__device__ __forceinline__ float log_fast(float u) { return __builtin_amdgcn_logf(u) * 0.693147182464599609375f; }

// softplus(x, beta=1, threshold=20) / 6 in fp32  (prior_model.py:88)
__device__ __forceinline__ float st_f32(float x) {
  const float t = exp_fast(x);
  // log1p(t): series for t < 2^-6 (truncation error t^5 / 6 relative), otherwise ln(u) + (t - (u - 1)) / u with u = fl(1 + t):
  // the second term restores what the rounding of 1 + t lost
  const float ser = t * __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, 0.2f, -0.25f), 0.333333343f), -0.5f), 1.0f);
  const float u = 1.0f + t;
  const float c = t - (u - 1.0f);
  const float gen = __builtin_fmaf(c, __builtin_amdgcn_rcpf(u), log_fast(u));
  float sp = (t < 0.015625f) ? ser : gen;
  sp = (x > 20.0f) ? x : sp;
  // sp / 6, correctly rounded: quotient by the rounded reciprocal, one residual step
  const float q = sp * 0.16666667163372039794921875f;
  return __builtin_fmaf(__builtin_fmaf(-6.0f, q, sp), 0.16666667163372039794921875f, q);
}
// d/dx [softplus(x)/6] = sigmoid(x)/6 (1/6 above the threshold)
__device__ __forceinline__ float dst_f32(float x) {
  const float d = 1.0f + exp_fast(fminf(-x, 80.0f));      // (finite: the Newton step below must not see inf * 0)
  float r = __builtin_amdgcn_rcpf(d);
  r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);                            // Newton step: <= 1 ulp
  const float sg = (x > 20.0f) ? 1.0f : r;
  const float q = sg * 0.16666667163372039794921875f;
  return __builtin_fmaf(__builtin_fmaf(-6.0f, q, sg), 0.16666667163372039794921875f, q);
}

// torch.distributions.kl._kl_normal_normal in fp32
__device__ __forceinline__ float kl_elem_f32(float mu_q, float sig_q, float mu_p, float sig_p) {
  const float rp = __builtin_amdgcn_rcpf(sig_p);
  float ratio = sig_q * rp;
  ratio = __builtin_fmaf(__builtin_fmaf(-sig_p, ratio, sig_q), rp, ratio);          // sig_q / sig_p
  const float var_ratio = ratio * ratio;
  const float dm = mu_q - mu_p;
  float t = dm * rp;
  t = __builtin_fmaf(__builtin_fmaf(-sig_p, t, dm), rp, t);                         // (mu_q - mu_p) / sig_p
  const float t1 = t * t;
  return 0.5f * (var_ratio + t1 - 1.0f - log_fast(var_ratio));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace rcb
