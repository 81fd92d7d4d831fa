// Elementwise / reduction kernels of the variational-posterior path (HBM-bound):
//   reparameterised sampling (K1/K10), Gaussian KL + segment sums (K5-K7), beta annealing (K8),
//   fused reparam-bwd + KL-bwd + Adam (K1'/K5'/K11), flat Adam (K11), column moments (K12).
// All arithmetic that feeds a parity check is written un-fused (mul, then add) like the torch
// CPU ops it replaces.
#include <stdarg.h>

#include <type_traits>

#include "rcb_common.h"

#pragma clang fp contract(off)

// Rounded-once arithmetic that must NOT be contracted into FMAs.  These helpers are defined here, under the pragma
// above, on purpose: HIP's __fadd_rn / __fmul_rn / ... are inline functions from headers compiled with the default
// -ffp-contract=fast, so after inlining their operations carry the `contract` flag and LLVM may fuse exactly them.
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }

namespace rcb {
char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace rcb

using namespace rcb;

// ---- order-independent reductions --------------------------------------------------------------------------------------
// Sums that many workgroups (or ranks) contribute to are accumulated in 64-bit FIXED POINT with integer atomics: integer
// addition is associative, so the result does not depend on the order in which the contributions arrive -- bitwise
// reproducible from run to run, and identical whether the rows sit on one GPU or are sharded (the all-reduce over ranks
// is an integer sum as well).  A contribution is rounded once, to the fixed-point grid, where it enters.
constexpr double KL_FX = 16777216.0;          // RCB_KL_FX_SCALE = 2^24 units per nat
// KL accumulators of a step: slots [0, RCB_KL_SLOTS - 1) hold partial sums, the LAST slot counts the contributions that
// could not be represented (NaN / Inf, or beyond 2^30 nats from one workgroup): a diverged run must surface as NaN in the
// ELBO log, as it does in the reference, not as an arbitrary finite integer (__double2ll_rn(NaN) is 0).
__device__ __forceinline__ void fx_add_kl(int64_t* slots, unsigned which, double v) {
  if (fabs(v) < 1073741824.0)
    atomicAdd(reinterpret_cast<unsigned long long*>(slots + which % (RCB_KL_SLOTS - 1)), (unsigned long long)__double2ll_rn(v * KL_FX));
  else
    atomicAdd(reinterpret_cast<unsigned long long*>(slots + (RCB_KL_SLOTS - 1)), 1ull);
}

extern "C" int rcb_version(void) { return RCB_VERSION; }
extern "C" const char* rcb_last_error_string(void) { return last_error_buf(); }
extern "C" int64_t rcb_struct_bytes(int32_t which) {
  switch (which) {
    case RCB_STRUCT_SIREN_DESC: return (int64_t)sizeof(rcb_siren_desc);
    case RCB_STRUCT_LEVEL: return (int64_t)sizeof(rcb_level);
    case RCB_STRUCT_LEVEL_BWD: return (int64_t)sizeof(rcb_level_bwd);
    case RCB_STRUCT_ADAM_CFG: return (int64_t)sizeof(rcb_adam_cfg);
    case RCB_STRUCT_ADAM_TENSOR: return (int64_t)sizeof(rcb_adam_tensor);
    case RCB_STRUCT_REC_DESC: return (int64_t)sizeof(rcb_rec_desc);
    default: return -1;
  }
}

// ------------------------------------------------------------------------------------------
// softplus/6
// ------------------------------------------------------------------------------------------
__global__ void softplus_scale_kernel(const float* __restrict__ ls, float* __restrict__ out, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = st_f32(ls[i]);
}

extern "C" int rcb_softplus_scale(const float* log_scale, float* scale, int64_t n, rcb_stream_t stream) {
  RCB_REQUIRE(log_scale && scale && n >= 0, RCB_ERR_ARG, "softplus_scale: null pointer");
  if (n == 0) return RCB_OK;
  int grid = cdiv(n, 256);
  if (grid > 4096) grid = 4096;
  softplus_scale_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(log_scale, scale, n);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

// test hook: 0 (default) = specialised kernels (flat 16-byte paths, LDS-staged gathers) where they apply;
// 1 = always the generic kernels.  Lets the tests compare both on identical shapes.
static int g_generic_only = 0;
// workgroups per CU of the PERSISTENT form of the flat posterior update; 0 (default) = one workgroup per 1024 elements, the form
// of rounds 1-3.  Measured (round 4, same-box, the forked step): 0 / 4 / 8 / 16 -> 1.087 / 1.086 / 1.091 / 1.089 ms per step,
// i.e. nothing: capping the update's footprint does not protect the library GEMMs that run beside it.  RCB_POSTERIOR_WG_PER_CU
// in the environment sets it at load time (A/B runs).
static int g_post_wg_per_cu = [] {
  const char* e = getenv("RCB_POSTERIOR_WG_PER_CU");
  return e ? atoi(e) : 0;
}();
extern "C" int rcb_debug_generic_kernels_only(int32_t on) {
  int old = g_generic_only;
  g_generic_only = on ? 1 : 0;
  return old;
}

// ------------------------------------------------------------------------------------------
// K1 / K10 reparam forward
// ------------------------------------------------------------------------------------------
struct ReparamArgs {
  rcb_level lv[3];
  int n_levels, n_inr, samples, out_cols;
  float* out;
};

// the effective (mu, sigma) of one parameter element: masks applied, softplus taken
__device__ __forceinline__ void elem_mu_sigma(const rcb_level& L, long long o, float& mu, float& sig);

__device__ __forceinline__ void level_mu_sigma(const rcb_level& L, int n, int d, float& mu, float& sig) {
  int j = L.col_map ? L.col_map[d] : d;
  int r = L.row_map ? L.row_map[n] : n;
  if (L.row_perm) r = L.row_perm[(long long)r * L.cols + j];
  long long o = (long long)r * L.cols + j;
  if (L.mu_sigma_ws) {      // packed by mu_sigma_pack_kernel: ONE 8-byte gather instead of up to four 4-byte ones
    const float2 v = reinterpret_cast<const float2*>(L.mu_sigma_ws)[o];
    mu = v.x;
    sig = v.y;
    return;
  }
  elem_mu_sigma(L, o, mu, sig);
}

__device__ __forceinline__ void elem_mu_sigma(const rcb_level& L, long long o, float& mu, float& sig) {
  float loc = L.loc[o];
  float s = L.scale_is_sigma ? L.log_scale[o] : st_f32(L.log_scale[o]);
  if (L.enc_mask) {
    float m = L.enc_mask[o];
    float z = L.enc_sample[o];
    // loc*(1-m) + z*m ; st*(1-m) + 1e-15*m   (test_model.py:289-290)
    loc = add_rn(mul_rn(loc, 1.0f - m), mul_rn(z, m));
    s = add_rn(mul_rn(s, 1.0f - m), mul_rn(1e-15f, m));
  }
  mu = loc;
  sig = s;
}

// Gathered levels (test-time layout: group-order column map, per-column row permutation): the sampling kernel reads the
// parameters of (n, d) from wherever the maps point -- scattered 4-byte reads of loc, log_scale, mask and encoded sample, a
// 64-byte sector each.  This pass forms the effective (mu, sigma) of every parameter element where the arrays are contiguous
// and stores them as ONE 8-byte record per element: the sampler's gather drops from four sectors per level to one.  Same
// operations on the same values: bit-identical samples.
__global__ void __launch_bounds__(256) mu_sigma_pack_kernel(rcb_level L, long long n_elems) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n_elems; o += stride) {
    float mu, sig;
    elem_mu_sigma(L, o, mu, sig);
    reinterpret_cast<float2*>(L.mu_sigma_ws)[o] = make_float2(mu, sig);
  }
}

// one thread per (INR, column): mu and sigma -- which may sit behind row / column permutations, i.e. scattered reads --
// are fetched once and reused for all samples of the INR
__global__ void reparam_fwd_kernel(ReparamArgs a) {
  int n = blockIdx.x;
  int d = blockIdx.y * blockDim.x + threadIdx.x;
  if (d >= a.out_cols) return;
  float mu[3], sg[3];
  bool on[3];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    on[l] = l < a.n_levels && d < a.lv[l].cols_out;
    mu[l] = 0.f;
    sg[l] = 0.f;
    if (on[l]) level_mu_sigma(a.lv[l], n, d, mu[l], sg[l]);
  }
  for (int s = 0; s < a.samples; ++s) {
    long long row = (long long)n * a.samples + s;
    float acc = 0.f;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      if (on[l]) {
        float e = a.lv[l].eps[row * a.lv[l].cols_out + d];
        float v = add_rn(mu[l], mul_rn(sg[l], e));
        acc = (l == 0) ? v : add_rn(acc, v);
      }
    }
    a.out[row * a.out_cols + d] = acc;
  }
}

// Test-time layout (parameters stored in group order, read back through a column map): mu / sigma gathers are
// scattered 4-byte reads, i.e. a 64-byte sector each.  One block per INR stages that INR's parameter rows in LDS with
// coalesced loads and does the permuted reads there.  One level, no row maps.  Same arithmetic as the generic kernel.
__global__ void __launch_bounds__(1024) reparam_staged_kernel(ReparamArgs a) {
  extern __shared__ float sm[];                 // [4][cols]: loc, log_scale, enc_mask, enc_sample
  const rcb_level& L = a.lv[0];
  const int n = blockIdx.x, cols = L.cols;
  const long long base = (long long)n * cols;
  for (int i = threadIdx.x; i < cols; i += 1024) {
    sm[i] = L.loc[base + i];
    sm[cols + i] = L.log_scale[base + i];
    if (L.enc_mask) {
      sm[2 * cols + i] = L.enc_mask[base + i];
      sm[3 * cols + i] = L.enc_sample[base + i];
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < a.out_cols; d += 1024) {
    const int j = L.col_map ? L.col_map[d] : d;
    float loc = sm[j];
    float sg = st_f32(sm[cols + j]);
    if (L.enc_mask) {
      const float m = sm[2 * cols + j], z = sm[3 * cols + j];
      loc = add_rn(mul_rn(loc, 1.0f - m), mul_rn(z, m));
      sg = add_rn(mul_rn(sg, 1.0f - m), mul_rn(1e-15f, m));
    }
    for (int sidx = 0; sidx < a.samples; ++sidx) {
      const long long row = (long long)n * a.samples + sidx;
      a.out[row * a.out_cols + d] = add_rn(loc, mul_rn(sg, L.eps[row * L.cols_out + d]));
    }
  }
}

// ---- counter-based noise (Philox4x32-10 + Box-Muller) ----------------------------------------------------------
// eps of element i depends only on (seed, stream, step, i): no generator state, nothing to capture, and a replayed
// HIP graph draws fresh noise because `step` is read from device memory.  4 normals per Philox call.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    const unsigned hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0;
    k.y += W1;
  }
  return c;
}

__device__ __forceinline__ float4 philox_normal4(unsigned long long group, unsigned stream, unsigned long long step,
                                                 unsigned long long seed) {
  const uint4 r = philox4x32_10(make_uint4((unsigned)group, (unsigned)(group >> 32), stream ^ (unsigned)(step >> 32) * 0x85EBCA6Bu,
                                           (unsigned)step),
                                make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
  constexpr float K = 1.0f / 16777216.0f;             // 24-bit uniforms: u1 in (0, 1), u2 in [0, 1)
  const float u1 = ((float)(r.x >> 8) + 0.5f) * K, u2 = (float)(r.y >> 8) * K;
  const float u3 = ((float)(r.z >> 8) + 0.5f) * K, u4 = (float)(r.w >> 8) * K;
  // hardware transcendentals (about 1 ulp: ample for noise): v_log_f32 is log2, v_sin / v_cos take revolutions
  constexpr float M2LN2 = -1.3862943611198906f;             // -2 ln 2
  const float ra = __builtin_amdgcn_sqrtf(M2LN2 * __builtin_amdgcn_logf(u1));
  const float rb = __builtin_amdgcn_sqrtf(M2LN2 * __builtin_amdgcn_logf(u3));
  return make_float4(ra * __builtin_amdgcn_cosf(u2), ra * __builtin_amdgcn_sinf(u2), rb * __builtin_amdgcn_cosf(u4),
                     rb * __builtin_amdgcn_sinf(u4));
}

__global__ void __launch_bounds__(256) philox_normal_kernel(float* __restrict__ out, long long n, unsigned long long seed,
                                                            unsigned stream, const long long* __restrict__ step_dev,
                                                            long long step_host, unsigned long long goff) {
  const unsigned long long step = step_dev ? (unsigned long long)*step_dev : (unsigned long long)step_host;
  const long long n4 = (n + 3) >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 e = philox_normal4((unsigned long long)i + goff, stream, step, seed);
    const float ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * i + k < n) out[4 * i + k] = ev[k];
  }
}

// four consecutive flat elements `first .. first + 3` of a [rows][cols] array as bf16 into [rows][ld16]: 4-byte stores where
// the four stay in one row and are aligned
__device__ __forceinline__ void store_bf16_row4(__bf16* __restrict__ out16, long long first, int cols, long long ld16, float4 o) {
  long long r = first / cols;
  int c = (int)(first - r * cols);
  const float ov[4] = {o.x, o.y, o.z, o.w};
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  if (c + 3 < cols && (ld16 & 1) == 0) {
    __bf16* d = out16 + r * ld16 + c;
    const bf16x2 p01 = {(__bf16)ov[0], (__bf16)ov[1]}, p12 = {(__bf16)ov[1], (__bf16)ov[2]}, p23 = {(__bf16)ov[2], (__bf16)ov[3]};
    if ((c & 1) == 0) {
      *reinterpret_cast<bf16x2*>(d) = p01;
      *reinterpret_cast<bf16x2*>(d + 2) = p23;
    } else {
      d[0] = (__bf16)ov[0];
      *reinterpret_cast<bf16x2*>(d + 1) = p12;
      d[3] = (__bf16)ov[3];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      out16[r * ld16 + c] = (__bf16)ov[k];
      if (++c == cols) { c = 0; ++r; }
    }
  }
}

// the (hi, lo) bf16 planes of four consecutive flat elements: hi = bf16(v), lo = bf16(v - hi) -- the split the A-transform
// kernel would form from the fp32 value, made by the producer instead (rcb_atrans_apply's plane operands)
__device__ __forceinline__ void store_planes_row4(__bf16* __restrict__ hi, __bf16* __restrict__ lo, long long first, int cols,
                                                  long long ld16, float4 o) {
  store_bf16_row4(hi, first, cols, ld16, o);
  if (lo) {
    const float4 r = make_float4(o.x - (float)(__bf16)o.x, o.y - (float)(__bf16)o.y, o.z - (float)(__bf16)o.z, o.w - (float)(__bf16)o.w);
    store_bf16_row4(lo, first, cols, ld16, r);
  }
}

// Training samples of the patched presets: two or three plain levels (level 0 one row per INR, the coarser ones behind row
// maps), one sample, every column produced.  Flat over the [n_inr * cols] arrays with 16-byte accesses like reparam_flat_kernel
// (the generic kernel's one thread per element ran at 2.3 TB/s at a rank's shard of the audio preset); the coarse levels'
// parameters are a few rows that stay cached.  Same operations in the same order as the generic kernel: bit-identical.
struct HierRng {           // eps_out[0] != NULL: the levels' noise is drawn here (and written for the posterior update)
  float* eps_out[3];
  unsigned streams[3];
  unsigned long long seed, goff;
  const long long* step_dev;
};

__global__ void __launch_bounds__(256) reparam_hier_flat_kernel(ReparamArgs a, long long n_total, HierRng rng) {
  const int D = a.out_cols;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const bool draw = rng.eps_out[0] != nullptr;
  const unsigned long long step = draw ? (unsigned long long)*rng.step_dev : 0ull;
  for (long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i4 * 4 < n_total; i4 += stride) {
    const long long b = i4 * 4;
    const float4 m4 = reinterpret_cast<const float4*>(a.lv[0].loc + b)[0];
    const float4 l4 = reinterpret_cast<const float4*>(a.lv[0].log_scale + b)[0];
    float4 e4[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      if (l < a.n_levels) {
        if (draw) {
          e4[l] = philox_normal4((unsigned long long)i4 + rng.goff, rng.streams[l], step, rng.seed);
          reinterpret_cast<float4*>(rng.eps_out[l] + b)[0] = e4[l];
        } else {
          e4[l] = reinterpret_cast<const float4*>(a.lv[l].eps + b)[0];
        }
      }
    }
    int n = (int)(b / D), d = (int)(b - (long long)n * D);
    float o[4];
    const float* mv = &m4.x;
    const float* lv = &l4.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float acc = add_rn(mv[k], mul_rn(st_f32(lv[k]), (&e4[0].x)[k]));
#pragma unroll
      for (int l = 1; l < 3; ++l) {
        if (l < a.n_levels) {
          const rcb_level& L = a.lv[l];
          const long long off = (long long)(L.row_map ? L.row_map[n] : n) * D + d;
          acc = add_rn(acc, add_rn(L.loc[off], mul_rn(st_f32(L.log_scale[off]), (&e4[l].x)[k])));
        }
      }
      o[k] = acc;
      if (++d == D) {
        d = 0;
        ++n;
      }
    }
    reinterpret_cast<float4*>(a.out + b)[0] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// reparameterised sample with the noise drawn in the kernel: out = loc + st(log_scale) * eps, eps written once for the
// posterior update.  Replaces torch.randn + the flat reparam kernel (the generated values never make a round trip
// through HBM before their first use); same arithmetic as reparam_flat_kernel on the same eps.
// out16 (nullable): a bf16 copy of out as [n / cols][ld16] rows -- the operand of the A transform's batched bf16
// weight-gradient GEMM, written while the values are in registers; lo16 (nullable, with out16): the low plane, bf16(out -
// out16), same layout -- with both the fp32 `out` may be NULL (the A transform then reads the planes); eps_out nullable
// (a consumer that re-draws the noise from the counter needs no copy of it).
__global__ void __launch_bounds__(256) reparam_rng_kernel(const float* __restrict__ loc, const float* __restrict__ ls,
                                                          float* __restrict__ eps_out, float* __restrict__ out, long long n,
                                                          unsigned long long seed, unsigned stream,
                                                          const long long* __restrict__ step_dev, __bf16* __restrict__ out16,
                                                          __bf16* __restrict__ lo16, int cols, long long ld16,
                                                          unsigned long long goff) {
  const unsigned long long step = (unsigned long long)*step_dev;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 m = reinterpret_cast<const float4*>(loc)[i];
    const float4 l = reinterpret_cast<const float4*>(ls)[i];
    const float4 e = philox_normal4((unsigned long long)i + goff, stream, step, seed);
    float4 o;
    o.x = add_rn(m.x, mul_rn(st_f32(l.x), e.x));
    o.y = add_rn(m.y, mul_rn(st_f32(l.y), e.y));
    o.z = add_rn(m.z, mul_rn(st_f32(l.z), e.z));
    o.w = add_rn(m.w, mul_rn(st_f32(l.w), e.w));
    if (eps_out) reinterpret_cast<float4*>(eps_out)[i] = e;
    if (out) reinterpret_cast<float4*>(out)[i] = o;
    if (out16) store_planes_row4(out16, lo16, 4 * i, cols, ld16, o);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 3)) {      // tail group
    const float4 e = philox_normal4((unsigned long long)n4 + goff, stream, step, seed);
    const float ev[4] = {e.x, e.y, e.z, e.w};
    for (int k = 0; k < (int)(n & 3); ++k) {
      const long long i = (n4 << 2) + k;
      if (eps_out) eps_out[i] = ev[k];
      const float o = add_rn(loc[i], mul_rn(st_f32(ls[i]), ev[k]));
      if (out) out[i] = o;
      if (out16) out16[(i / cols) * ld16 + i % cols] = (__bf16)o;
      if (out16 && lo16) lo16[(i / cols) * ld16 + i % cols] = (__bf16)(o - (float)(__bf16)o);
    }
  }
}

extern "C" int rcb_philox_normal(float* out, int64_t n, uint64_t seed, uint32_t rng_stream, const int64_t* step_dev,
                                 int64_t step_host, uint64_t group_offset, rcb_stream_t stream) {
  RCB_REQUIRE(out && n >= 0, RCB_ERR_ARG, "philox_normal: null pointer");
  if (n == 0) return RCB_OK;
  int blocks = cdiv((n + 3) >> 2, 256);
  if (blocks > 16384) blocks = 16384;
  philox_normal_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(out, (long long)n, seed, rng_stream,
                                                                (const long long*)step_dev, (long long)step_host,
                                                                (unsigned long long)group_offset);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_reparam_rng_fwd(const float* loc, const float* log_scale, int64_t n, uint64_t seed, uint32_t rng_stream,
                                   const int64_t* step_dev, float* eps_out, float* out, void* out_bf16, void* out_lo,
                                   int32_t cols, int64_t ld_bf16, uint64_t group_offset, rcb_stream_t stream) {
  RCB_REQUIRE(loc && log_scale && step_dev && n > 0, RCB_ERR_ARG, "reparam_rng_fwd: null pointer / empty");
  RCB_REQUIRE(out || (out_bf16 && out_lo), RCB_ERR_ARG, "reparam_rng_fwd: no output (fp32 `out`, or both planes out_bf16 + out_lo)");
  RCB_REQUIRE(!out_lo || out_bf16, RCB_ERR_ARG, "reparam_rng_fwd: the low plane comes with the high plane");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  RCB_REQUIRE(al16(loc) && al16(log_scale) && al16(eps_out) && al16(out), RCB_ERR_ARG, "reparam_rng_fwd: 16-byte alignment");
  RCB_REQUIRE(out_bf16 == nullptr || (cols >= 1 && n % cols == 0 && ld_bf16 >= cols), RCB_ERR_SHAPE,
              "reparam_rng_fwd: bf16 copy: n = %lld is not rows x %d columns, or row stride %lld < columns", (long long)n, cols, (long long)ld_bf16);
  int blocks = cdiv(n >> 2, 256);
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  reparam_rng_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(loc, log_scale, eps_out, out, (long long)n, seed, rng_stream,
                                                              (const long long*)step_dev, reinterpret_cast<__bf16*>(out_bf16),
                                                              reinterpret_cast<__bf16*>(out_lo), out_bf16 ? cols : 1, (long long)ld_bf16,
                                                              (unsigned long long)group_offset);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

// fast path: one level, one sample, no maps / masks -> purely elementwise over the flat [n_inr * cols] arrays
// (rows of 3267 floats are not 16-byte aligned, the flat arrays are): 16-byte accesses, same arithmetic
__global__ void __launch_bounds__(256) reparam_flat_kernel(const float* __restrict__ loc, const float* __restrict__ ls,
                                                           const float* __restrict__ eps, float* __restrict__ out,
                                                           long long n) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 m = reinterpret_cast<const float4*>(loc)[i];
    const float4 l = reinterpret_cast<const float4*>(ls)[i];
    const float4 e = reinterpret_cast<const float4*>(eps)[i];
    float4 o;
    o.x = add_rn(m.x, mul_rn(st_f32(l.x), e.x));
    o.y = add_rn(m.y, mul_rn(st_f32(l.y), e.y));
    o.z = add_rn(m.z, mul_rn(st_f32(l.z), e.z));
    o.w = add_rn(m.w, mul_rn(st_f32(l.w), e.w));
    reinterpret_cast<float4*>(out)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    out[i] = add_rn(loc[i], mul_rn(st_f32(ls[i]), eps[i]));
  }
}

extern "C" int rcb_reparam_fwd(const rcb_level* levels, int32_t n_levels, int32_t n_inr, int32_t samples,
                               int32_t out_cols, float* out, rcb_stream_t stream) {
  RCB_REQUIRE(levels && out, RCB_ERR_ARG, "reparam_fwd: null pointer");
  RCB_REQUIRE(n_levels >= 1 && n_levels <= 3, RCB_ERR_ARG, "reparam_fwd: n_levels=%d", n_levels);
  RCB_REQUIRE(n_inr > 0 && samples > 0 && out_cols > 0, RCB_ERR_SHAPE, "reparam_fwd: empty shape");
  ReparamArgs a;
  memset(&a, 0, sizeof(a));
  for (int l = 0; l < n_levels; ++l) {
    a.lv[l] = levels[l];
    RCB_REQUIRE(a.lv[l].loc && a.lv[l].log_scale && a.lv[l].eps, RCB_ERR_ARG, "reparam_fwd: level %d null", l);
    RCB_REQUIRE(a.lv[l].cols_out <= out_cols && a.lv[l].cols_out > 0, RCB_ERR_SHAPE, "reparam_fwd: level %d cols_out", l);
    RCB_REQUIRE((a.lv[l].enc_mask == nullptr) == (a.lv[l].enc_sample == nullptr), RCB_ERR_ARG, "reparam_fwd: mask/sample");
    RCB_REQUIRE(a.lv[l].col_map || a.lv[l].cols_out <= a.lv[l].cols, RCB_ERR_SHAPE, "reparam_fwd: level %d cols", l);
  }
  RCB_REQUIRE(a.lv[0].cols_out == out_cols, RCB_ERR_SHAPE, "reparam_fwd: level 0 must cover all columns");
  a.n_levels = n_levels;
  a.n_inr = n_inr;
  a.samples = samples;
  a.out_cols = out_cols;
  a.out = out;
  {
    const rcb_level& L = a.lv[0];
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!g_generic_only && n_levels == 1 && samples == 1 && !L.enc_mask && !L.row_map && !L.row_perm && !L.col_map &&
        !L.scale_is_sigma && L.cols == out_cols && L.rows == n_inr && al16(L.loc) && al16(L.log_scale) && al16(L.eps) && al16(out)) {
      const long long n = (long long)n_inr * out_cols;
      int blocks = cdiv(n >> 2, 256);
      if (blocks > 16384) blocks = 16384;
      if (blocks < 1) blocks = 1;
      reparam_flat_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(L.loc, L.log_scale, L.eps, out, n);
      RCB_LAUNCH_CHECK();
      return RCB_OK;
    }
  }
  if (!g_generic_only && n_levels == 1 && !a.lv[0].scale_is_sigma && a.lv[0].col_map && !a.lv[0].row_map && !a.lv[0].row_perm && a.lv[0].rows == n_inr &&
      a.lv[0].cols_out == out_cols && (size_t)a.lv[0].cols * 16 <= 150 * 1024) {
    // (per launch: the attribute belongs to the (function, device) pair; a process may drive several devices)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(reparam_staged_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail((int)e, "reparam_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    reparam_staged_kernel<<<n_inr, 1024, (size_t)a.lv[0].cols * 16, (hipStream_t)stream>>>(a);
    RCB_LAUNCH_CHECK();
    return RCB_OK;
  }
  {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    bool plain = !g_generic_only && n_levels >= 2 && samples == 1 && ((long long)n_inr * out_cols) % 4 == 0 && al16(out) &&
                 !a.lv[0].row_map && a.lv[0].rows == n_inr && al16(a.lv[0].loc) && al16(a.lv[0].log_scale);
    for (int l = 0; l < n_levels && plain; ++l) {
      const rcb_level& L = a.lv[l];
      plain = !L.enc_mask && !L.row_perm && !L.col_map && !L.scale_is_sigma && L.cols == out_cols && L.cols_out == out_cols && al16(L.eps);
    }
    if (plain) {
      const long long n = (long long)n_inr * out_cols;
      int blocks = cdiv(n >> 2, 256);
      if (blocks > 16384) blocks = 16384;
      HierRng no_rng;
      memset(&no_rng, 0, sizeof(no_rng));
      reparam_hier_flat_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(a, n, no_rng);
      RCB_LAUNCH_CHECK();
      return RCB_OK;
    }
  }
  for (int l = 0; l < n_levels; ++l) {            // gathered levels with scratch: pack (mu, sigma) records first
    rcb_level& L = a.lv[l];
    if (g_generic_only || !(L.col_map || L.row_perm)) L.mu_sigma_ws = nullptr;
    if (L.mu_sigma_ws) {
      const long long ne = (long long)L.rows * L.cols;
      int blocks = cdiv(ne, 256);
      if (blocks > 16384) blocks = 16384;
      mu_sigma_pack_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(L, ne);
    }
  }
  dim3 grid(n_inr, cdiv(out_cols, 256));
  RCB_REQUIRE(grid.y <= 65535, RCB_ERR_SHAPE, "reparam_fwd: too many columns");
  reparam_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_reparam_hier_rng_fwd(const rcb_level* levels, int32_t n_levels, int32_t n_inr, int32_t out_cols,
                                        float* const* eps_out, float* out, uint64_t seed, const uint32_t* rng_streams,
                                        const int64_t* step_dev, uint64_t group_offset, rcb_stream_t stream) {
  RCB_REQUIRE(levels && eps_out && out && rng_streams && step_dev, RCB_ERR_ARG, "reparam_hier_rng_fwd: null pointer");
  RCB_REQUIRE(n_levels >= 1 && n_levels <= 3 && n_inr > 0 && out_cols > 0, RCB_ERR_ARG, "reparam_hier_rng_fwd: n_levels=%d", n_levels);
  RCB_REQUIRE(((long long)n_inr * out_cols) % 4 == 0, RCB_ERR_SHAPE, "reparam_hier_rng_fwd: n_inr * out_cols must be a multiple of 4");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  ReparamArgs a;
  memset(&a, 0, sizeof(a));
  HierRng rng;
  memset(&rng, 0, sizeof(rng));
  for (int l = 0; l < n_levels; ++l) {
    a.lv[l] = levels[l];
    const rcb_level& L = a.lv[l];
    RCB_REQUIRE(L.loc && L.log_scale && eps_out[l] && al16(eps_out[l]), RCB_ERR_ARG, "reparam_hier_rng_fwd: level %d null / unaligned", l);
    RCB_REQUIRE(!L.enc_mask && !L.enc_sample && !L.row_perm && !L.col_map && !L.scale_is_sigma && L.cols == out_cols && L.cols_out == out_cols,
                RCB_ERR_UNSUPPORTED, "reparam_hier_rng_fwd: level %d is not plain (row map only, every column produced)", l);
    rng.eps_out[l] = eps_out[l];
    rng.streams[l] = rng_streams[l];
  }
  RCB_REQUIRE(!a.lv[0].row_map && a.lv[0].rows == n_inr && al16(a.lv[0].loc) && al16(a.lv[0].log_scale) && al16(out), RCB_ERR_ARG,
              "reparam_hier_rng_fwd: level 0 is one 16-byte aligned row per INR");
  a.n_levels = n_levels;
  a.n_inr = n_inr;
  a.samples = 1;
  a.out_cols = out_cols;
  a.out = out;
  rng.seed = seed;
  rng.goff = group_offset;
  rng.step_dev = reinterpret_cast<const long long*>(step_dev);
  const long long n = (long long)n_inr * out_cols;
  int blocks = cdiv(n >> 2, 256);
  if (blocks > 16384) blocks = 16384;
  reparam_hier_flat_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(a, n, rng);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

// ------------------------------------------------------------------------------------------
// K5-K7 Gaussian KL with row and (row, group) reductions
// ------------------------------------------------------------------------------------------
struct KlArgs {
  const float *loc, *ls, *p_loc, *p_scale;
  int p_is_log, rows, cols;
  const float* beta;
  const int* group_idx;
  int n_groups;
  const int *seg_start, *seg_end;
  double* kl_row;
  double* kl_group;
};

__device__ __forceinline__ double block_sum_256(double v, double* sm) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  return t;
}

__global__ void __launch_bounds__(256) kl_rows_kernel(KlArgs a) {
  __shared__ double sm[4];
  int r = blockIdx.x;
  const float* loc = a.loc + (long long)r * a.cols;
  const float* ls = a.ls + (long long)r * a.cols;
  double acc = 0.0;
  if (a.seg_start == nullptr) {
    for (int j = threadIdx.x; j < a.cols; j += 256) {
      float sp = a.p_is_log ? st_f32(a.p_scale[j]) : a.p_scale[j];
      float k = kl_elem_f32(loc[j], st_f32(ls[j]), a.p_loc[j], sp);
      if (a.beta) k = mul_rn(k, a.beta[(long long)r * a.n_groups + a.group_idx[j]]);
      acc += (double)k;
    }
  } else {
    // parameters are stored in group order: group g is the contiguous segment [start, end)
    for (int g = threadIdx.x; g < a.n_groups; g += 256) {
      int s = a.seg_start[g], e = a.seg_end[g];
      double gs = 0.0;
      for (int j = s; j < e; ++j) {
        float sp = a.p_is_log ? st_f32(a.p_scale[j]) : a.p_scale[j];
        gs += (double)kl_elem_f32(loc[j], st_f32(ls[j]), a.p_loc[j], sp);  // np.bincount order
      }
      if (a.kl_group) a.kl_group[(long long)r * a.n_groups + g] = gs;
      double w = a.beta ? (double)a.beta[(long long)r * a.n_groups + g] : 1.0;
      acc += w * gs;
    }
  }
  double tot = block_sum_256(acc, sm);
  if (threadIdx.x == 0 && a.kl_row) a.kl_row[r] = tot;
}

extern "C" int rcb_gauss_kl(const float* loc, const float* log_scale, const float* p_loc, const float* p_scale,
                            int32_t p_scale_is_log, int32_t rows, int32_t cols, const float* beta,
                            const int32_t* group_idx, int32_t n_groups, const int32_t* seg_start,
                            const int32_t* seg_end, double* kl_row, double* kl_group, rcb_stream_t stream) {
  RCB_REQUIRE(loc && log_scale && p_loc && p_scale, RCB_ERR_ARG, "gauss_kl: null input");
  RCB_REQUIRE(rows > 0 && cols > 0, RCB_ERR_SHAPE, "gauss_kl: empty shape %d x %d", rows, cols);
  RCB_REQUIRE(kl_row || kl_group, RCB_ERR_ARG, "gauss_kl: no output requested");
  RCB_REQUIRE(!beta || (group_idx && n_groups > 0), RCB_ERR_ARG, "gauss_kl: beta needs group_idx");
  RCB_REQUIRE((seg_start == nullptr) == (seg_end == nullptr), RCB_ERR_ARG, "gauss_kl: segments");
  RCB_REQUIRE(!kl_group || seg_start, RCB_ERR_ARG, "gauss_kl: kl_group needs segments");
  KlArgs a{loc, log_scale, p_loc, p_scale, p_scale_is_log, rows, cols, beta, group_idx, n_groups,
           seg_start, seg_end, kl_row, kl_group};
  kl_rows_kernel<<<rows, 256, 0, (hipStream_t)stream>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

// ------------------------------------------------------------------------------------------
// K8 beta annealing
// ------------------------------------------------------------------------------------------
__global__ void beta_update_kernel(const double* kl, float* beta, const uint8_t* done, long long n, double hi,
                                   double lo, float factor) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (done && done[i]) return;
  double bits = kl[i] / 0.6931471805599453;  // kls / np.log(2.)
  float b = beta[i];
  b = mul_rn(b, bits > hi ? factor : 1.0f);
  b = div_rn(b, bits <= lo ? factor : 1.0f);
  b = fminf(fmaxf(b, 0.0f), 10000.0f);
  beta[i] = b;
}

extern "C" int rcb_beta_update(const double* kl_group, float* beta, const uint8_t* done, int32_t rows,
                               int32_t n_groups, double bits, double upper, double lower, double step,
                               rcb_stream_t stream) {
  RCB_REQUIRE(kl_group && beta, RCB_ERR_ARG, "beta_update: null pointer");
  RCB_REQUIRE(rows > 0 && n_groups > 0, RCB_ERR_SHAPE, "beta_update: empty shape");
  long long n = (long long)rows * n_groups;
  beta_update_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(kl_group, beta, done, n, bits + upper,
                                                                   bits - lower, (float)(1.0 + step));
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

// ------------------------------------------------------------------------------------------
// fused reparam-bwd + KL-bwd (+ Adam)
// ------------------------------------------------------------------------------------------
struct AdamScalars {
  float w1;         // 1 - beta1
  float beta2;      // beta2
  float w2;         // 1 - beta2
  float step_size;  // lr / (1 - beta1^t)
  float bc2_sqrt;   // sqrt(1 - beta2^t)
  float eps;
  int enabled;
  const float* dyn;   // optional device pair {step_size, bc2_sqrt} overriding the by-value ones (graph replay)
};

static AdamScalars make_adam(const rcb_adam_cfg* c) {
  AdamScalars s;
  memset(&s, 0, sizeof(s));
  if (!c) return s;
  double bc1 = 1.0 - pow((double)c->beta1, (double)c->step);
  double bc2 = 1.0 - pow((double)c->beta2, (double)c->step);
  s.w1 = (float)(1.0 - (double)c->beta1);
  s.beta2 = c->beta2;
  s.w2 = (float)(1.0 - (double)c->beta2);
  s.step_size = (float)((double)c->lr / bc1);
  s.bc2_sqrt = (float)sqrt(bc2);
  s.eps = c->eps;
  s.enabled = 1;
  s.dyn = c->dyn_scalars;
  return s;
}

// torch.optim.Adam single step (default flags), operation order as in torch/optim/adam.py
__device__ __forceinline__ void adam_apply(float& p, float g, float& m, float& v, const AdamScalars& s) {
  const float step_size = s.dyn ? s.dyn[0] : s.step_size;
  const float bc2_sqrt = s.dyn ? s.dyn[1] : s.bc2_sqrt;
  m = add_rn(m, mul_rn(s.w1, sub_rn(g, m)));                    // exp_avg.lerp_(grad, 1-beta1)
  v = add_rn(mul_rn(v, s.beta2), mul_rn(mul_rn(s.w2, g), g));  // mul_(beta2).addcmul_(g, g, 1-beta2)
  // sqrt: hardware v_sqrt_f32 with one residual step (<= 1 ulp, 0 -> 0); the two divisions by reciprocal + residual
  // (rcb_common.h); the operation ORDER stays torch's
  float sq = __builtin_amdgcn_sqrtf(v);
  if (sq > 0.0f) sq = __builtin_fmaf(__builtin_fmaf(-sq, sq, v), 0.5f * __builtin_amdgcn_rcpf(sq), sq);
  float denom = add_rn(div_fast(sq, bc2_sqrt), s.eps);
  p = add_rn(p, mul_rn(-step_size, div_fast(m, denom)));        // addcdiv_(exp_avg, denom, -step_size)
}

struct PostBwdArgs {
  rcb_level_bwd L;
  AdamScalars adam;
};

// KL part of the gradient and the chain rule through softplus, with explicitly unfused roundings: the generic and the
// flat kernel must produce the same bits, and the reference computes these with separate torch ops anyway
__device__ __forceinline__ void kl_grad_add(float loc, float sig, float pl, float sp, float w, float& g_mu, float& g_sig) {
  const float inv_var_p = div_fast(1.0f, mul_rn(sp, sp));
  g_mu = add_rn(g_mu, mul_rn(w, mul_rn(sub_rn(loc, pl), inv_var_p)));
  g_sig = add_rn(g_sig, mul_rn(w, sub_rn(mul_rn(sig, inv_var_p), div_fast(1.0f, sig))));
}

// Levels WITHOUT members in the test-time layout (one parameter row per INR behind the reference's per-column row permutation
// and the group-order column map, S samples): the update kernel gathers d_out and eps of (n, s, d) for every sample -- 2 S
// scattered 4-byte reads per parameter, a 64-byte sector each.  This pass forms the sums over the samples where they are
// contiguous, in the update's order (g = 0; g += d_out[s]; ge += d_out[s] * eps[s], s ascending): the update then gathers two
// values.  Bit-identical: for a level without members the chain over the samples IS the whole sum.
__global__ void __launch_bounds__(256) sample_sums_kernel(const float* __restrict__ d_out, const float* __restrict__ eps, int samples,
                                                          long long row_elems, long long n_rows, float* __restrict__ ws) {
  const long long total = n_rows * row_elems, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long n = i / row_elems, d = i - n * row_elems;
    float g = 0.f, ge = 0.f;
    for (int sidx = 0; sidx < samples; ++sidx) {
      const long long e = (n * samples + sidx) * row_elems + d;
      const float go = d_out[e];
      g = add_rn(g, go);
      ge = add_rn(ge, mul_rn(go, eps[e]));
    }
    ws[i] = g;
    ws[total + i] = ge;
  }
}

__global__ void __launch_bounds__(256) posterior_bwd_kernel(PostBwdArgs a) {
  const rcb_level_bwd& L = a.L;
  int r = blockIdx.x;
  int t = blockIdx.y * blockDim.x + threadIdx.x;
  const bool act = t < L.cols;     // tail lanes stay alive (clamped) so that the wave reduction is convergent
  if (!act) t = L.cols - 1;
  // col_map set (coarse levels of the test-time layout, every column produced): the thread index is the PRODUCED column d,
  // the contiguous axis of d_out / eps, and the thread finds its parameter column j = col_map[d].  A row of a coarse level
  // gathers members x samples (80 ... 480) gradient elements per parameter: indexed by j those were scattered 4-byte reads
  // (a 64-byte sector each), indexed by d they are coalesced and only the dozen parameter accesses scatter.  Which thread
  // owns which element changes no arithmetic: bit-identical parameters.
  const bool by_d = L.col_map != nullptr;
  const int j = by_d ? L.col_map[t] : t;
  long long o = (long long)r * L.cols + j;
  float loc = L.loc[o];
  float ls = L.log_scale[o];
  float sig = st_f32(ls);
  float g_mu = 0.f, g_sig = 0.f;
  int d = by_d ? t : (L.col_inv ? L.col_inv[j] : j);
  if (L.sample_sum_ws && L.d_out && d < L.cols_out) {
    // sums over the samples formed by sample_sums_kernel (levels without members): two gathers instead of 2 S
    const long long e = (long long)(L.row_perm_inv ? L.row_perm_inv[o] : r) * L.cols_out + d;
    g_mu = L.sample_sum_ws[e];
    g_sig = L.sample_sum_ws[(long long)L.rows * L.cols_out + e];
    if (L.enc_mask) {
      float keep = 1.0f - L.enc_mask[o];
      g_mu = mul_rn(g_mu, keep);
      g_sig = mul_rn(g_sig, keep);
    }
  } else if (L.d_out && d < L.cols_out) {
    int r0 = L.row_perm_inv ? L.row_perm_inv[o] : r;
    int mb = L.member_ptr ? L.member_ptr[r0] : r0;
    int me = L.member_ptr ? L.member_ptr[r0 + 1] : r0 + 1;
    // the members' gradients, summed in member order.  One sample per member (training): loads in batches of eight -- a row
    // of a coarse level has tens of members and the launch only a few workgroups, one dependent load after the other is pure
    // latency
    if (L.samples == 1) {
      for (int q0 = mb; q0 < me; q0 += 8) {
        float go[8], ep[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int q = q0 + u < me ? q0 + u : me - 1;
          const int n = L.member_idx ? L.member_idx[q] : q;
          const long long e = (long long)n * L.cols_out + d;
          go[u] = L.d_out[e];
          ep[u] = L.eps[e];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (q0 + u < me) {
            g_mu = add_rn(g_mu, go[u]);
            g_sig = add_rn(g_sig, mul_rn(go[u], ep[u]));
          }
        }
      }
    } else {
      for (int q = mb; q < me; ++q) {
        int n = L.member_idx ? L.member_idx[q] : q;
        for (int s = 0; s < L.samples; ++s) {
          long long e = ((long long)n * L.samples + s) * L.cols_out + d;
          float go = L.d_out[e];
          g_mu = add_rn(g_mu, go);
          g_sig = add_rn(g_sig, mul_rn(go, L.eps[e]));
        }
      }
    }
    if (L.enc_mask) {
      float keep = 1.0f - L.enc_mask[o];
      g_mu = mul_rn(g_mu, keep);
      g_sig = mul_rn(g_sig, keep);
    }
  }
  float w = L.kl_scalar;
  if (L.kl_scalar_dev) w = mul_rn(w, *L.kl_scalar_dev);
  if (L.beta) w = mul_rn(w, L.beta[(long long)r * L.n_groups + L.group_idx[j]]);
  if (L.kl_accum) {   // unweighted KL of the parameters *before* this update (ELBO logging)
    __shared__ double s_kl[4];
    float spk = L.p_scale_is_log ? st_f32(L.p_scale[j]) : L.p_scale[j];
    double kv = wave_sum(act ? (double)kl_elem_f32(loc, sig, L.p_loc[j], spk) : 0.0);
    if ((threadIdx.x & 63) == 0) s_kl[threadIdx.x >> 6] = kv;
    __syncthreads();
    // one (integer, fixed-point) atomic per block, spread over RCB_KL_SLOTS addresses (a single address serialises)
    if (threadIdx.x == 0)
      fx_add_kl(L.kl_accum, blockIdx.x * 7 + blockIdx.y, (s_kl[0] + s_kl[1]) + (s_kl[2] + s_kl[3]));
  }
  if (w != 0.0f) {
    float sp = L.p_scale_is_log ? st_f32(L.p_scale[j]) : L.p_scale[j];
    kl_grad_add(loc, sig, L.p_loc[j], sp, w, g_mu, g_sig);
  }
  float g_ls = mul_rn(g_sig, dst_f32(ls));
  if (!act) return;
  if (a.adam.enabled) {
    float m1 = L.m_loc[o], v1 = L.v_loc[o], m2 = L.m_ls[o], v2 = L.v_ls[o];
    adam_apply(loc, g_mu, m1, v1, a.adam);
    adam_apply(ls, g_ls, m2, v2, a.adam);
    L.loc[o] = loc;
    L.log_scale[o] = ls;
    L.m_loc[o] = m1;
    L.v_loc[o] = v1;
    L.m_ls[o] = m2;
    L.v_ls[o] = v2;
  }
  if (L.g_loc) L.g_loc[o] = g_mu;
  if (L.g_log_scale) L.g_log_scale[o] = g_ls;
}

// Coarse levels of the patched presets in training (one sample, members behind member_ptr / member_idx, every column produced,
// no masks / permutations / per-group beta, Adam on): FOUR consecutive columns per thread.  Rows of 3201 floats are only 4-byte
// aligned, so the 16-byte accesses are declared with 4-byte alignment (global_load_dwordx4 takes any dword address); the
// generic kernel's one column per thread moved 4 bytes per lane and instruction and ran at 1.4-2 TB/s at a rank's shard of
// the audio preset.  Same per-element operations in the same member order: the updated parameters are bit-identical.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

// NT threads per workgroup: 256, or 64 when the level has so few rows that 1024-column workgroups would leave the chip empty
// (the coarsest level of two photos: 2 rows x 96 members each -- the kernel is then a chain of member loads, eight in flight)
template <int NT>
__global__ void __launch_bounds__(NT) posterior_members4_kernel(PostBwdArgs a) {
  const rcb_level_bwd& L = a.L;
  const int r = blockIdx.x;
  const int j0 = 4 * (blockIdx.y * NT + threadIdx.x);
  const int nv = L.cols - j0 >= 4 ? 4 : (L.cols - j0 > 0 ? L.cols - j0 : 0);      // valid columns of this thread
  const bool full = nv == 4;
  const int jb = nv > 0 ? j0 : 0;                   // (idle tail lanes stay alive for the wave reduction: they re-read column 0)
  auto ld = [&](const float* p, float (&v)[4]) {
    if (full) {
      const f4u t = *reinterpret_cast<const f4u*>(p);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = t[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = p[k < nv ? k : 0];
    }
  };
  auto st = [&](float* p, const float (&v)[4]) {
    if (full) {
      const f4u t = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f4u*>(p) = t;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nv) p[k] = v[k];
    }
  };
  const long long o = (long long)r * L.cols + jb;
  float loc[4], ls[4], g_mu[4], g_sig[4];
  ld(L.loc + o, loc);
  ld(L.log_scale + o, ls);
#pragma unroll
  for (int k = 0; k < 4; ++k) g_mu[k] = g_sig[k] = 0.f;
  const int mb = L.member_ptr[r], me = L.member_ptr[r + 1];
  // (the full / partial distinction is made ONCE around the whole member loop: a branch inside every load would keep the sixteen
  // loads of a batch from being in flight together)
  auto member_sums = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    for (int q0 = mb; q0 < me; q0 += 8) {           // eight members = sixteen 16-byte loads in flight
      float go[8][4], ep[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + u < me ? q0 + u : me - 1;
        const long long e = (long long)L.member_idx[q] * L.cols + jb;
        if (FULL) {
          const f4u tg = *reinterpret_cast<const f4u*>(L.d_out + e), te = *reinterpret_cast<const f4u*>(L.eps + e);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            go[u][k] = tg[k];
            ep[u][k] = te[k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            go[u][k] = L.d_out[e + (k < nv ? k : 0)];
            ep[u][k] = L.eps[e + (k < nv ? k : 0)];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (q0 + u < me) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            g_mu[k] = add_rn(g_mu[k], go[u][k]);
            g_sig[k] = add_rn(g_sig[k], mul_rn(go[u][k], ep[u][k]));
          }
        }
      }
    }
  };
  if (full) member_sums(std::true_type{});
  else member_sums(std::false_type{});
  float pl[4], psc[4];
  ld(L.p_loc + jb, pl);
  ld(L.p_scale + jb, psc);
  float w = L.kl_scalar;
  if (L.kl_scalar_dev) w = mul_rn(w, *L.kl_scalar_dev);
  float m1[4], v1[4], m2[4], v2[4];
  ld(L.m_loc + o, m1);
  ld(L.v_loc + o, v1);
  ld(L.m_ls + o, m2);
  ld(L.v_ls + o, v2);
  double kl = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float sig = st_f32(ls[k]);
    const float sp = L.p_scale_is_log ? st_f32(psc[k]) : psc[k];
    if (L.kl_accum && k < nv) kl += (double)kl_elem_f32(loc[k], sig, pl[k], sp);
    if (w != 0.0f) kl_grad_add(loc[k], sig, pl[k], sp, w, g_mu[k], g_sig[k]);
    const float g_ls = mul_rn(g_sig[k], dst_f32(ls[k]));
    adam_apply(loc[k], g_mu[k], m1[k], v1[k], a.adam);
    adam_apply(ls[k], g_ls, m2[k], v2[k], a.adam);
  }
  if (L.kl_accum) {   // unweighted KL of the parameters *before* this update (ELBO logging)
    __shared__ double s_kl[4];
    const double kv = wave_sum(kl);
    if (NT == 64) {
      if (threadIdx.x == 0) fx_add_kl(L.kl_accum, blockIdx.x * 7 + blockIdx.y, kv);
    } else {
      if ((threadIdx.x & 63) == 0) s_kl[threadIdx.x >> 6] = kv;
      __syncthreads();
      if (threadIdx.x == 0) fx_add_kl(L.kl_accum, blockIdx.x * 7 + blockIdx.y, (s_kl[0] + s_kl[1]) + (s_kl[2] + s_kl[3]));
    }
  }
  if (nv == 0) return;
  st(L.loc + o, loc);
  st(L.log_scale + o, ls);
  st(L.m_loc + o, m1);
  st(L.v_loc + o, v1);
  st(L.m_ls + o, m2);
  st(L.v_ls + o, v2);
}

// Test-time layout: the gradient / noise slabs [S][cols_out] of one INR are read through the inverse column map, i.e.
// with scattered 4-byte reads (a 64-byte sector each).  One 1024-thread block per parameter row stages both slabs in
// LDS with coalesced loads; the permuted reads then hit LDS.  No hierarchy members, no row permutation; everything
// else (encode mask, per-group beta, KL log, Adam) as in the generic kernel, same operation order: bit-identical.
__global__ void __launch_bounds__(1024) posterior_staged_kernel(PostBwdArgs a) {
  extern __shared__ float sm[];                   // [S * cols_out] d_out, [S * cols_out] eps
  const rcb_level_bwd& L = a.L;
  const int r = blockIdx.x;
  const int slab = L.samples * L.cols_out;
  {
    const float* sg = L.d_out + (long long)r * slab;
    const float* se = L.eps + (long long)r * slab;
    for (int i = threadIdx.x; i < slab; i += 1024) {
      sm[i] = sg[i];
      sm[slab + i] = se[i];
    }
  }
  __syncthreads();
  double kl = 0.0;
  for (int j = threadIdx.x; j < L.cols; j += 1024) {
    const long long o = (long long)r * L.cols + j;
    float loc = L.loc[o];
    float ls = L.log_scale[o];
    const float sig = st_f32(ls);
    float g_mu = 0.f, g_sig = 0.f;
    const int d = L.col_inv ? L.col_inv[j] : j;
    if (d < L.cols_out) {
      for (int s = 0; s < L.samples; ++s) {
        const float go = sm[s * L.cols_out + d];
        g_mu = add_rn(g_mu, go);
        g_sig = add_rn(g_sig, mul_rn(go, sm[slab + s * L.cols_out + d]));
      }
      if (L.enc_mask) {
        const float keep = 1.0f - L.enc_mask[o];
        g_mu = mul_rn(g_mu, keep);
        g_sig = mul_rn(g_sig, keep);
      }
    }
    float w = L.kl_scalar;
    if (L.kl_scalar_dev) w = mul_rn(w, *L.kl_scalar_dev);
    if (L.beta) w = mul_rn(w, L.beta[(long long)r * L.n_groups + L.group_idx[j]]);
    if (L.kl_accum) {
      const float spk = L.p_scale_is_log ? st_f32(L.p_scale[j]) : L.p_scale[j];
      kl += (double)kl_elem_f32(loc, sig, L.p_loc[j], spk);
    }
    if (w != 0.0f) {
      const float sp = L.p_scale_is_log ? st_f32(L.p_scale[j]) : L.p_scale[j];
      kl_grad_add(loc, sig, L.p_loc[j], sp, w, g_mu, g_sig);
    }
    const float g_ls = mul_rn(g_sig, dst_f32(ls));
    if (a.adam.enabled) {
      float m1 = L.m_loc[o], v1 = L.v_loc[o], m2 = L.m_ls[o], v2 = L.v_ls[o];
      adam_apply(loc, g_mu, m1, v1, a.adam);
      adam_apply(ls, g_ls, m2, v2, a.adam);
      L.loc[o] = loc;
      L.log_scale[o] = ls;
      L.m_loc[o] = m1;
      L.v_loc[o] = v1;
      L.m_ls[o] = m2;
      L.v_ls[o] = v2;
    }
    if (L.g_loc) L.g_loc[o] = g_mu;
    if (L.g_log_scale) L.g_log_scale[o] = g_ls;
  }
  if (L.kl_accum) {
    __shared__ double s_kl[16];
    const double kv = wave_sum(kl);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_kl[threadIdx.x >> 6] = kv;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int k = 0; k < 16; ++k) t += s_kl[k];
      fx_add_kl(L.kl_accum, blockIdx.x, t);
    }
  }
}

// Fast path of the kernel above for the plain case (prior training of un-patched presets): one sample, no hierarchy /
// permutation / column maps, no encode mask, no per-group beta, Adam enabled, every column produced.  Purely
// elementwise over the flat [rows * cols] arrays with 16-byte accesses (rows of 3267 floats are not 16-byte aligned,
// the flat arrays are); same per-element arithmetic, so the updated parameters are bit-identical.
__global__ void __launch_bounds__(256) posterior_flat_kernel(PostBwdArgs a, long long n_total) {
  const rcb_level_bwd& L = a.L;
  // grid-stride form: with g_post_wg_per_cu > 0 the launch is capped at a few workgroups per CU that walk the array (an
  // experiment of round 4, see g_post_wg_per_cu: no effect on the forked step); uncapped, every workgroup makes one trip
  double kl = 0.0;
  const float w = L.kl_scalar_dev ? mul_rn(L.kl_scalar, *L.kl_scalar_dev) : L.kl_scalar;
  for (long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i4 * 4 < n_total; i4 += (long long)gridDim.x * blockDim.x) {
  const long long b = i4 * 4;                                                  // n_total % 4 == 0 on this path
  // streaming accesses: everything here is read once and written once per step (900 MB against 256 MB of Infinity Cache).
  // Non-temporal loads / stores took the kernel from 93.6 to 86.3 us per launch (same box) and leave the sample written at
  // the end -- the next step's first operand -- a better chance to stay cached (with loc / log_scale non-temporal as well the
  // next step's A transform took 85.4 instead of 89.4 us).
  typedef float nt_f4 __attribute__((ext_vector_type(4)));
  auto nt_ld = [](const float* p) {
    const nt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
  };
  auto nt_st = [](float* p, float4 v) {
    const nt_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p));
  };
#define RCB_LD4(p) nt_ld(p)
#define RCB_ST4(p, v) nt_st(p, v)
#define RCB_LD4P(p) nt_ld(p)
#define RCB_ST4P(p, v) nt_st(p, v)
  float4 loc4 = RCB_LD4P(L.loc + b);
  float4 ls4 = RCB_LD4P(L.log_scale + b);
  const float4 go4 = RCB_LD4(L.d_out + b);
  // eps: the stored noise of this step's sample, or -- eps_from_rng -- re-drawn from the counter it was drawn from (a pure
  // function of (seed, stream, counter, element): the same bits), which saves its 4-byte write and 4-byte read per element
  const float4 ep4 = L.eps_from_rng
                         ? philox_normal4((unsigned long long)i4 + L.rng_group_offset, L.rng_stream,
                                          (unsigned long long)(*L.rng_step_dev + L.rng_step_add - 1), L.rng_seed)
                         : RCB_LD4(L.eps + b);
  float4 m14 = RCB_LD4(L.m_loc + b), v14 = RCB_LD4(L.v_loc + b);
  float4 m24 = RCB_LD4(L.m_ls + b), v24 = RCB_LD4(L.v_ls + b);
  float* locv = &loc4.x; float* lsv = &ls4.x;
  const float* gov = &go4.x; const float* epv = &ep4.x;
  float* m1v = &m14.x; float* v1v = &v14.x; float* m2v = &m24.x; float* v2v = &v24.x;
  int j = (int)(b % L.cols);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float loc = locv[k], ls = lsv[k];
    const float sig = st_f32(ls);
    float g_mu = gov[k];
    float g_sig = mul_rn(gov[k], epv[k]);
    const float pl = L.p_loc[j];
    const float sp = L.p_scale_is_log ? st_f32(L.p_scale[j]) : L.p_scale[j];
    if (L.kl_accum) kl += (double)kl_elem_f32(loc, sig, pl, sp);
    if (w != 0.0f) kl_grad_add(loc, sig, pl, sp, w, g_mu, g_sig);
    const float g_ls = mul_rn(g_sig, dst_f32(ls));
    adam_apply(loc, g_mu, m1v[k], v1v[k], a.adam);
    adam_apply(ls, g_ls, m2v[k], v2v[k], a.adam);
    locv[k] = loc;
    lsv[k] = ls;
    if (++j == L.cols) j = 0;
  }
  RCB_ST4P(L.loc + b, loc4);
  RCB_ST4P(L.log_scale + b, ls4);
  RCB_ST4(L.m_loc + b, m14);
  RCB_ST4(L.v_loc + b, v14);
  RCB_ST4(L.m_ls + b, m24);
  RCB_ST4(L.v_ls + b, v24);
  if (L.next_out || L.next_out_bf16) {
    // the next step's sample from the updated parameters: the arithmetic of reparam_rng_kernel on Philox group i4 at the
    // step counter the next step will see
    const unsigned long long step = (unsigned long long)(*L.rng_step_dev + L.rng_step_add);
    const float4 e = philox_normal4((unsigned long long)i4 + L.rng_group_offset, L.rng_stream, step, L.rng_seed);
    float4 o;
    o.x = add_rn(loc4.x, mul_rn(st_f32(ls4.x), e.x));
    o.y = add_rn(loc4.y, mul_rn(st_f32(ls4.y), e.y));
    o.z = add_rn(loc4.z, mul_rn(st_f32(ls4.z), e.z));
    o.w = add_rn(loc4.w, mul_rn(st_f32(ls4.w), e.w));
    if (L.next_eps) nt_st(L.next_eps + b, e);            // (read again by the NEXT step's update only: non-temporal, as above)
    if (L.next_out) reinterpret_cast<float4*>(L.next_out + b)[0] = o;
    if (L.next_out_bf16)
      store_planes_row4(reinterpret_cast<__bf16*>(L.next_out_bf16), reinterpret_cast<__bf16*>(L.next_out_lo), b, L.cols, L.next_ld_bf16, o);
  }
  }      // grid-stride loop
  if (L.kl_accum) {      // one fixed-point contribution per workgroup (fixed grid for a given size: reproducible)
    __shared__ double s_kl[4];
    const double kv = wave_sum(kl);
    if ((threadIdx.x & 63) == 0) s_kl[threadIdx.x >> 6] = kv;
    __syncthreads();
    if (threadIdx.x == 0)
      fx_add_kl(L.kl_accum, blockIdx.x, (s_kl[0] + s_kl[1]) + (s_kl[2] + s_kl[3]));
  }
#undef RCB_LD4
#undef RCB_ST4
#undef RCB_LD4P
#undef RCB_ST4P
}

extern "C" int rcb_posterior_bwd(const rcb_level_bwd* lv, const rcb_adam_cfg* adam, rcb_stream_t stream) {
  RCB_REQUIRE(lv, RCB_ERR_ARG, "posterior_bwd: null level");
  RCB_REQUIRE(lv->loc && lv->log_scale && lv->p_loc && lv->p_scale, RCB_ERR_ARG, "posterior_bwd: null tensor");
  RCB_REQUIRE(lv->rows > 0 && lv->cols > 0 && lv->cols <= 65535 * 256, RCB_ERR_SHAPE, "posterior_bwd: shape %d x %d", lv->rows, lv->cols);
  RCB_REQUIRE(!lv->d_out || (lv->samples > 0 && lv->cols_out > 0), RCB_ERR_ARG, "posterior_bwd: d_out needs samples / cols_out");
  RCB_REQUIRE((lv->member_ptr == nullptr) == (lv->member_idx == nullptr), RCB_ERR_ARG, "posterior_bwd: members");
  RCB_REQUIRE(!lv->beta || (lv->group_idx && lv->n_groups > 0), RCB_ERR_ARG, "posterior_bwd: beta needs group_idx");
  RCB_REQUIRE(adam || (lv->g_loc && lv->g_log_scale), RCB_ERR_ARG, "posterior_bwd: neither adam nor grad outputs");
  RCB_REQUIRE(!adam || (lv->m_loc && lv->v_loc && lv->m_ls && lv->v_ls && adam->step >= 1), RCB_ERR_ARG,
              "posterior_bwd: adam state missing");
  const bool want_next = lv->next_out || lv->next_out_bf16;
  RCB_REQUIRE(!want_next || (!lv->col_inv && adam), RCB_ERR_UNSUPPORTED,
              "posterior_bwd: the fused next sample needs the plain (flat) case with Adam");
  RCB_REQUIRE(!lv->next_out_lo || lv->next_out_bf16, RCB_ERR_ARG, "posterior_bwd: next_out_lo comes with next_out_bf16");
  RCB_REQUIRE(!lv->d_out || lv->eps || lv->eps_from_rng, RCB_ERR_ARG, "posterior_bwd: d_out needs eps (stored, or eps_from_rng)");
  RCB_REQUIRE(!lv->eps_from_rng || (lv->rng_step_dev && !lv->eps && !lv->col_inv), RCB_ERR_ARG,
              "posterior_bwd: eps_from_rng re-draws the noise from rng_seed / rng_stream / *rng_step_dev + rng_step_add - 1: give those, and eps = NULL");
  PostBwdArgs a;
  a.L = *lv;
  a.adam = make_adam(adam);
  {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const long long n_total = (long long)lv->rows * lv->cols;
    if (!g_generic_only && lv->d_out && lv->col_inv && !lv->member_ptr && !lv->row_perm_inv &&
        (size_t)lv->samples * lv->cols_out * 8 <= 150 * 1024) {
      // (per launch: the attribute belongs to the (function, device) pair; a process may drive several devices)
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(posterior_staged_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);   // + 128 B static
      if (e != hipSuccess) return fail((int)e, "posterior_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      posterior_staged_kernel<<<lv->rows, 1024, (size_t)lv->samples * lv->cols_out * 8, (hipStream_t)stream>>>(a);
      RCB_LAUNCH_CHECK();
      return RCB_OK;
    }
    if (!g_generic_only && adam && lv->d_out && lv->samples == 1 && lv->cols_out == lv->cols && !lv->enc_mask && !lv->beta && !lv->member_ptr &&
        !lv->row_perm_inv && !lv->col_inv && !lv->g_loc && !lv->g_log_scale && (n_total & 3) == 0 && al16(lv->loc) &&
        al16(lv->log_scale) && al16(lv->d_out) && (lv->eps_from_rng || al16(lv->eps)) && al16(lv->m_loc) && al16(lv->v_loc) && al16(lv->m_ls) &&
        al16(lv->v_ls)) {
      RCB_REQUIRE(!want_next || (lv->rng_step_dev && al16(lv->next_out) && al16(lv->next_eps) &&
                                 (lv->next_out || lv->next_out_lo) && (!lv->next_out_bf16 || lv->next_ld_bf16 >= lv->cols)),
                  RCB_ERR_ARG, "posterior_bwd: next sample: null pointer (fp32 next_out, or both planes), alignment or bf16 row stride");
      // at most RCB_POSTERIOR_WG_PER_CU workgroups per CU (hipDeviceProp multiProcessorCount is read once per device)
      static int n_cu[64] = {0};
      int dev_id = 0;
      (void)hipGetDevice(&dev_id);
      if (dev_id >= 0 && dev_id < 64 && n_cu[dev_id] == 0) {
        hipDeviceProp_t prop;
        n_cu[dev_id] = (hipGetDeviceProperties(&prop, dev_id) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
      }
      const long long want = cdiv(n_total >> 2, 256);
      const long long cap = (long long)((dev_id >= 0 && dev_id < 64) ? n_cu[dev_id] : 256) * g_post_wg_per_cu;
      posterior_flat_kernel<<<(int)(g_post_wg_per_cu > 0 && want > cap ? cap : want), 256, 0, (hipStream_t)stream>>>(a, n_total);
      RCB_LAUNCH_CHECK();
      return RCB_OK;
    }
  }
  if (!g_generic_only && adam && lv->d_out && lv->eps && lv->samples == 1 && lv->cols_out == lv->cols && !lv->enc_mask && !lv->beta &&
      lv->member_ptr && !lv->row_perm_inv && !lv->col_inv && !lv->g_loc && !lv->g_log_scale && !lv->eps_from_rng && !want_next) {
    if ((long long)lv->rows * cdiv(lv->cols, 1024) >= 1024) {
      posterior_members4_kernel<256><<<dim3(lv->rows, cdiv(lv->cols, 1024)), 256, 0, (hipStream_t)stream>>>(a);
    } else {
      posterior_members4_kernel<64><<<dim3(lv->rows, cdiv(lv->cols, 256)), 64, 0, (hipStream_t)stream>>>(a);
    }
    RCB_LAUNCH_CHECK();
    return RCB_OK;
  }
  // levels without members, several samples, gathered columns: the sums over the samples in a pass of their own (see
  // sample_sums_kernel) when the caller gave the workspace
  if (g_generic_only || lv->member_ptr || lv->samples < 2 || !lv->col_inv || !lv->d_out || !lv->eps) a.L.sample_sum_ws = nullptr;
  if (a.L.sample_sum_ws) {
    const long long tot = (long long)lv->rows * lv->cols_out;
    int blocks = cdiv(tot, 256);
    if (blocks > 16384) blocks = 16384;
    sample_sums_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(lv->d_out, lv->eps, lv->samples, lv->cols_out, lv->rows, a.L.sample_sum_ws);
  }
  // (the generic kernel indexes its threads by the produced column only where that pays: see the kernel)
  if (g_generic_only || !(lv->col_map && lv->col_inv && lv->member_ptr && lv->cols_out == lv->cols && lv->d_out)) a.L.col_map = nullptr;
  RCB_REQUIRE(!want_next && !lv->eps_from_rng, RCB_ERR_UNSUPPORTED,
              "posterior_bwd: the fused next sample / re-drawn noise need the plain (flat) case");
  dim3 grid(lv->rows, cdiv(lv->cols, 256));
  posterior_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

__global__ void adam_flat_kernel(float* p, const float* g, float* m, float* v, long long n, AdamScalars s) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_apply(pi, g[i], mi, vi, s);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// ---- per-step bookkeeping of a captured training step (device-resident step counter) -------------------------
__global__ void __launch_bounds__(1024) step_begin_kernel(const float* __restrict__ table, long long n_steps,
                                                          const long long* __restrict__ step, float* __restrict__ dyn,
                                                          long long* __restrict__ kl_slots) {
  const int t = threadIdx.x;
  if (t < 2) {
    long long s = *step;
    if (s < 0) s = 0;
    if (s >= n_steps) s = n_steps - 1;      // replayed past the table: keep the last row rather than read outside
    dyn[t] = table[2 * s + t];
  }
  if (kl_slots) kl_slots[t] = 0;
}

// fixed-order fp64 sums (thread-strided partials, then a tree over LDS): the same value eagerly and under replay; the KL
// slots are fixed-point integers (exact sum)
__global__ void __launch_bounds__(1024) step_end_kernel(const float* __restrict__ sse, int n_sse, double mse_scale,
                                                        const long long* __restrict__ kl_slots, double* __restrict__ mse_log,
                                                        double* __restrict__ kl_log, long long n_log,
                                                        long long* __restrict__ step, long long* __restrict__ aux_counter) {
  __shared__ double red[1024];
  __shared__ long long redk[1024];
  const int t = threadIdx.x;
  double a = 0.0;
  if (sse)
    for (int i = t; i < n_sse; i += 1024) a += (double)sse[i];
  red[t] = a;
  redk[t] = (kl_slots && t < RCB_KL_SLOTS - 1) ? kl_slots[t] : 0;       // (the last slot is the not-representable counter)
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (t < off) {
      red[t] += red[t + off];
      redk[t] += redk[t + off];
    }
    __syncthreads();
  }
  if (t == 0) {
    const long long s = *step;
    if (s >= 0 && s < n_log) {
      if (mse_log && sse) mse_log[s] = red[0] * mse_scale;
      if (kl_log && kl_slots)
        kl_log[s] = kl_slots[RCB_KL_SLOTS - 1] ? __builtin_nan("") : (double)redk[0] * (1.0 / KL_FX);
    }
    *step = s + 1;
    if (aux_counter) *aux_counter += 1;      // e.g. the noise counter, which is NOT reset between train() calls
  }
}

extern "C" int rcb_step_begin(const float* adam_table, int64_t n_steps, const int64_t* step, float* dyn, int64_t* kl_slots,
                              rcb_stream_t stream) {
  RCB_REQUIRE(adam_table && step && dyn && n_steps >= 1, RCB_ERR_ARG, "step_begin: null pointer / empty table");
  static_assert(RCB_KL_SLOTS == 1024, "one slot per thread");
  step_begin_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(adam_table, (long long)n_steps, (const long long*)step, dyn,
                                                          (long long*)kl_slots);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_step_end(const float* sse, int32_t n_sse, double mse_scale, const int64_t* kl_slots, double* mse_log,
                            double* kl_log, int64_t n_log, int64_t* step, int64_t* aux_counter, rcb_stream_t stream) {
  RCB_REQUIRE(step, RCB_ERR_ARG, "step_end: null step counter");
  RCB_REQUIRE(n_sse >= 0 && n_log >= 0, RCB_ERR_SHAPE, "step_end: negative size");
  step_end_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(sse, n_sse, mse_scale, (const long long*)kl_slots, mse_log, kl_log, (long long)n_log,
                                                     (long long*)step, (long long*)aux_counter);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

struct AdamMultiArgs {
  float* p[RCB_ADAM_MAX_TENSORS];
  const float* g[RCB_ADAM_MAX_TENSORS];
  float* m[RCB_ADAM_MAX_TENSORS];
  float* v[RCB_ADAM_MAX_TENSORS];
  long long n[RCB_ADAM_MAX_TENSORS];
  int blk_start[RCB_ADAM_MAX_TENSORS + 1];   // first block of each tensor (1024 elements per block)
  int count;
  AdamScalars s;
};

// one launch for a list of tensors (the shared mappings: 4 A matrices + 6 conv tensors)
__global__ void __launch_bounds__(256) adam_multi_kernel(AdamMultiArgs a) {
  int t = 0;
#pragma unroll 1
  while (t + 1 < a.count && (int)blockIdx.x >= a.blk_start[t + 1]) ++t;
  const long long base = (long long)((int)blockIdx.x - a.blk_start[t]) * 1024;
  float* p = a.p[t];
  const float* g = a.g[t];
  float* m = a.m[t];
  float* v = a.v[t];
  const long long n = a.n[t];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < n) {
      // (moments and gradient: streamed once per step, non-temporal; the parameter itself is read by the next kernel)
      float pi = p[i], mi = __builtin_nontemporal_load(m + i), vi = __builtin_nontemporal_load(v + i);
      adam_apply(pi, __builtin_nontemporal_load(g + i), mi, vi, a.s);
      p[i] = pi;
      __builtin_nontemporal_store(mi, m + i);
      __builtin_nontemporal_store(vi, v + i);
    }
  }
}

extern "C" int rcb_adam_multi(const rcb_adam_tensor* tensors, int32_t count, const rcb_adam_cfg* cfg,
                              rcb_stream_t stream) {
  RCB_REQUIRE(tensors && cfg, RCB_ERR_ARG, "adam_multi: null pointer");
  RCB_REQUIRE(count >= 1 && count <= RCB_ADAM_MAX_TENSORS, RCB_ERR_ARG, "adam_multi: %d tensors (1..%d)", count,
              RCB_ADAM_MAX_TENSORS);
  RCB_REQUIRE(cfg->step >= 1, RCB_ERR_ARG, "adam_multi: step must be >= 1");
  AdamMultiArgs a;
  memset(&a, 0, sizeof(a));
  long long blocks = 0;
  for (int t = 0; t < count; ++t) {
    const rcb_adam_tensor& x = tensors[t];
    RCB_REQUIRE(x.n >= 0 && (x.n == 0 || (x.p && x.g && x.m && x.v)), RCB_ERR_ARG, "adam_multi: tensor %d has a null pointer", t);
    a.p[t] = x.p; a.g[t] = x.g; a.m[t] = x.m; a.v[t] = x.v; a.n[t] = x.n;
    a.blk_start[t] = (int)blocks;
    blocks += (x.n + 1023) / 1024;
    RCB_REQUIRE(blocks < (1LL << 30), RCB_ERR_SHAPE, "adam_multi: too many elements");
  }
  a.blk_start[count] = (int)blocks;
  a.count = count;
  a.s = make_adam(cfg);
  if (blocks == 0) return RCB_OK;
  adam_multi_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, const rcb_adam_cfg* cfg,
                             rcb_stream_t stream) {
  RCB_REQUIRE(p && g && m && v && cfg, RCB_ERR_ARG, "adam_flat: null pointer");
  RCB_REQUIRE(cfg->step >= 1, RCB_ERR_ARG, "adam_flat: step must be >= 1");
  if (n == 0) return RCB_OK;
  int grid = cdiv(n, 256);
  if (grid > 8192) grid = 8192;
  adam_flat_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, make_adam(cfg));
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

// ------------------------------------------------------------------------------------------
// K12 column moments: exact fixed-point sums (order-independent, see fx_add).  Every term v (x, x^2, sigma^2; each exact
// in fp64) is split at 2^-30:  hi = floor(v 2^30),  lo = rint((v 2^30 - hi) 2^32)  and both parts are summed as 64-bit
// integers: resolution 2^-62; every TERM must satisfy |v| < 2^12 (so |x| < 2^6 because x^2 is a term, sigma < 2^6) for up
// to 2^20 rows over all ranks (2^12 * 2^30 * 2^20 = 2^62); terms outside that range, NaN and Inf are counted, not summed.
//   out[q][0][j] = sum_r hi,  out[q][1][j] = sum_r lo   for q = 0: x = loc[r, j], 1: x^2, 2: sigma^2 = (softplus(ls) / 6)^2
//   out[6 * cols] = number of terms that were not representable
// ------------------------------------------------------------------------------------------
constexpr int kMomRowsPerBlock = 256;
constexpr double MOM_FX = 1073741824.0, MOM_FX_LO = 4294967296.0, SQ_FX = 1073741824.0;   // 2^30, 2^32; 2^30 (KL column sums)

// terms outside the representable range (|v| >= 2^12, NaN, Inf) are not summed but COUNTED (`bad`): the consumer turns a
// non-zero count back into NaN instead of refitting a prior from garbage
__device__ __forceinline__ void fx_split(double v, long long& hi, long long& lo, long long& bad) {
  if (!(fabs(v) < 4096.0)) {
    ++bad;
    return;
  }
  const double s = v * MOM_FX;                 // exact (power of two)
  const double f = floor(s);
  hi += __double2ll_rn(f);
  lo += __double2ll_rn((s - f) * MOM_FX_LO);   // s - f in [0, 1) exactly
}

__global__ void __launch_bounds__(256) col_moments_kernel(const float* __restrict__ loc, const float* __restrict__ ls,
                                                          int rows, int cols, long long* out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  int r0 = blockIdx.y * kMomRowsPerBlock;
  int r1 = min(rows, r0 + kMomRowsPerBlock);
  long long acc[6] = {0, 0, 0, 0, 0, 0}, bad = 0;
  for (int r = r0; r < r1; ++r) {
    const double x = (double)loc[(long long)r * cols + j];
    fx_split(x, acc[0], acc[1], bad);
    fx_split(x * x, acc[2], acc[3], bad);
    const float s = st_f32(ls[(long long)r * cols + j]);
    fx_split((double)mul_rn(s, s), acc[4], acc[5], bad);
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) atomicAdd(reinterpret_cast<unsigned long long*>(out + (long long)q * cols + j), (unsigned long long)acc[q]);
  if (bad) atomicAdd(reinterpret_cast<unsigned long long*>(out + 6ll * cols), (unsigned long long)bad);
}

extern "C" int rcb_col_moments(const float* loc, const float* log_scale, int32_t rows, int32_t cols, int64_t* out_fx,
                               rcb_stream_t stream) {
  RCB_REQUIRE(loc && log_scale && out_fx, RCB_ERR_ARG, "col_moments: null pointer");
  RCB_REQUIRE(rows > 0 && cols > 0, RCB_ERR_SHAPE, "col_moments: empty shape");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out_fx, 0, sizeof(int64_t) * (6 * (size_t)cols + 1), st);
  if (e != hipSuccess) return fail((int)e, "memset");
  int row_blocks = cdiv(rows, kMomRowsPerBlock);
  RCB_REQUIRE(row_blocks <= 65535, RCB_ERR_SHAPE, "col_moments: too many rows");
  dim3 grid(cdiv(cols, 256), row_blocks);
  col_moments_kernel<<<grid, 256, 0, st>>>(loc, log_scale, rows, cols, (long long*)out_fx);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

// ------------------------------------------------------------------------------------------
// per-parameter KL summed over rows (get_grouping, prior_model.py:264-271), fp64
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kl_colsum_kernel(const float* __restrict__ loc, const float* __restrict__ sc,
                                                        int q_is_log, const float* __restrict__ p_loc,
                                                        const float* __restrict__ p_scale, int rows, int cols,
                                                        long long* out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  int r0 = blockIdx.y * kMomRowsPerBlock;
  int r1 = min(rows, r0 + kMomRowsPerBlock);
  float mp = p_loc[j], sp = p_scale[j];
  long long acc = 0, bad = 0;
  for (int r = r0; r < r1; ++r) {
    float s = sc[(long long)r * cols + j];
    if (q_is_log) s = st_f32(s);
    // EVERY element enters the integer grid on its own (2^30 units per nat; the product with the power of two is exact in
    // fp64, the rounding to the grid happens once, here): the sum is then independent of where the 256-row blocks -- and
    // the shards of a sharded run -- are cut, not only of the order in which they arrive
    const double k = (double)kl_elem_f32(loc[(long long)r * cols + j], s, mp, sp);
    if (fabs(k) < 4096.0)
      acc += __double2ll_rn(k * SQ_FX);
    else
      ++bad;                                                           // NaN / Inf / beyond 2^12 nats: counted, not summed
  }
  atomicAdd(reinterpret_cast<unsigned long long*>(out + j), (unsigned long long)acc);
  if (bad) atomicAdd(reinterpret_cast<unsigned long long*>(out + cols), (unsigned long long)bad);
}

extern "C" int rcb_gauss_kl_colsum(const float* loc, const float* q_scale, int32_t q_scale_is_log, const float* p_loc,
                                   const float* p_scale, int32_t rows, int32_t cols, int64_t* out_fx,
                                   rcb_stream_t stream) {
  RCB_REQUIRE(loc && q_scale && p_loc && p_scale && out_fx, RCB_ERR_ARG, "kl_colsum: null pointer");
  RCB_REQUIRE(rows > 0 && cols > 0, RCB_ERR_SHAPE, "kl_colsum: empty shape");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out_fx, 0, sizeof(int64_t) * ((size_t)cols + 1), st);
  if (e != hipSuccess) return fail((int)e, "memset");
  int row_blocks = cdiv(rows, kMomRowsPerBlock);
  RCB_REQUIRE(row_blocks <= 65535, RCB_ERR_SHAPE, "kl_colsum: too many rows");
  dim3 grid(cdiv(cols, 256), row_blocks);
  kl_colsum_kernel<<<grid, 256, 0, st>>>(loc, q_scale, q_scale_is_log, p_loc, p_scale, rows, cols, (long long*)out_fx);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
