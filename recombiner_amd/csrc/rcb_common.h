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

// softplus(x, beta=1, threshold=20) / 6 in fp32  (prior_model.py:88)
__device__ __forceinline__ float st_f32(float x) {
  float sp = (x > 20.0f) ? x : log1pf(expf(x));
  return sp / 6.0f;
}
// d/dx [softplus(x)/6] = sigmoid(x)/6 (1/6 above the threshold)
__device__ __forceinline__ float dst_f32(float x) {
  return (x > 20.0f) ? (1.0f / 6.0f) : (1.0f / (1.0f + expf(-x))) / 6.0f;
}

// torch.distributions.kl._kl_normal_normal in fp32
__device__ __forceinline__ float kl_elem_f32(float mu_q, float sig_q, float mu_p, float sig_p) {
  float ratio = sig_q / sig_p;
  float var_ratio = ratio * ratio;
  float t = (mu_q - mu_p) / sig_p;
  float t1 = t * t;
  return 0.5f * (var_ratio + t1 - 1.0f - logf(var_ratio));
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
