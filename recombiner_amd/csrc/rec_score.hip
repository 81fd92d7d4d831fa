// K9: A* (relative entropy coding) candidate scoring, fp64, batched over (row, group) jobs.
// Restates test_model.py:501-533 (and the h_/hh_ copies :535-584) op for op in fp64:
//   z     = p_loc + p_scale * xi                                      (mul, then add)
//   logN  = -((z - loc)^2) / (2 var) - log(scale) - log(sqrt(2 pi))   var, log(scale) in fp32
//   log_w = sum_j logN_q - sum_j logN_p + gumbel ; first argmax.
// One 256-thread workgroup per job; candidates are strided over threads so table rows are
// read coalesced; block argmax keeps the lowest index on ties (torch.argmax).  fp64 vector-ALU bound.
#include "rcb_common.h"

#pragma clang fp contract(off)

// un-contracted fp64 mul / add (HIP's __dmul_rn / __dadd_rn come from headers compiled with -ffp-contract=fast: after
// inlining LLVM may fuse them; these are defined under the pragma above)
__device__ __forceinline__ double dadd_rn(double a, double b) { return a + b; }
__device__ __forceinline__ double dmul_rn(double a, double b) { return a * b; }

using namespace rcb;

constexpr int kMaxGlen = 32;

struct RecArgs {
  const float *loc, *scale, *p_loc, *p_scale;
  int cols;
  const double* tables[kMaxGlen + 1];
  int max_glen;
  const double* gumbel;
  int K;
  const int *job_row, *job_start, *job_glen;
  int* idx;
  double* z_out;
  double* best;
  double* logw0;
};

struct Top2 {
  double v1;
  int i1;
  double v2;
};

__device__ __forceinline__ void top2_push(Top2& t, double v, int i) {
  if (v > t.v1 || (v == t.v1 && i < t.i1)) {
    t.v2 = t.v1;
    t.v1 = v;
    t.i1 = i;
  } else if (v > t.v2) {
    t.v2 = v;
  }
}

__device__ __forceinline__ Top2 top2_merge(const Top2& a, const Top2& b) {
  Top2 r;
  bool a_first = (a.v1 > b.v1) || (a.v1 == b.v1 && a.i1 <= b.i1);
  if (a_first) {
    r.v1 = a.v1;
    r.i1 = a.i1;
    r.v2 = fmax(a.v2, b.v1);
  } else {
    r.v1 = b.v1;
    r.i1 = b.i1;
    r.v2 = fmax(b.v2, a.v1);
  }
  return r;
}

__global__ void __launch_bounds__(256) rec_score_kernel(RecArgs a) {
  __shared__ double s_mq[kMaxGlen], s_mp[kMaxGlen], s_sp[kMaxGlen];
  __shared__ double s_2vq[kMaxGlen], s_2vp[kMaxGlen], s_lq[kMaxGlen], s_lp[kMaxGlen];
  __shared__ Top2 s_top[4];
  const int b = blockIdx.x;
  const int row = a.job_row[b], start = a.job_start[b], g = a.job_glen[b];
  const double* __restrict__ xi = a.tables[g];
  if (threadIdx.x < g) {
    int j = threadIdx.x;
    float mq = a.loc[(long long)row * a.cols + start + j];
    float sq = a.scale[(long long)row * a.cols + start + j];
    float mp = a.p_loc[start + j];
    float sp = a.p_scale[start + j];
    s_mq[j] = (double)mq;
    s_mp[j] = (double)mp;
    s_sp[j] = (double)sp;
    s_2vq[j] = (double)(2.0f * __fmul_rn(sq, sq));  // 2 * scale**2 in fp32, then widened
    s_2vp[j] = (double)(2.0f * __fmul_rn(sp, sp));
    s_lq[j] = (double)logf(sq);
    s_lp[j] = (double)logf(sp);
  }
  __syncthreads();
  const double c = 0.91893853320467267;  // math.log(math.sqrt(2*math.pi))
  Top2 t;
  t.v1 = -INFINITY;
  t.i1 = 0x7fffffff;
  t.v2 = -INFINITY;
  for (int k = threadIdx.x; k < a.K; k += 256) {
    const double* x = xi + (long long)k * g;
    double lq = 0.0, lp = 0.0;
    for (int j = 0; j < g; ++j) {
      double z = dadd_rn(s_mp[j], dmul_rn(s_sp[j], x[j]));
      double dq = z - s_mq[j];
      double dp = z - s_mp[j];
      double tq = ((-(dq * dq)) / s_2vq[j] - s_lq[j]) - c;
      double tp = ((-(dp * dp)) / s_2vp[j] - s_lp[j]) - c;
      lq = (j == 0) ? tq : lq + tq;
      lp = (j == 0) ? tp : lp + tp;
    }
    double lw = (lq - lp) + a.gumbel[k];
    if (b == 0 && a.logw0) a.logw0[k] = lw;
    top2_push(t, lw, k);
  }
  // wave reduce
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Top2 o;
    o.v1 = __shfl_xor(t.v1, off, 64);
    o.i1 = __shfl_xor(t.i1, off, 64);
    o.v2 = __shfl_xor(t.v2, off, 64);
    t = top2_merge(t, o);
  }
  if ((threadIdx.x & 63) == 0) s_top[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    Top2 r = s_top[0];
    for (int w = 1; w < 4; ++w) r = top2_merge(r, s_top[w]);
    s_top[0] = r;
    a.idx[b] = r.i1;
    if (a.best) {
      a.best[2 * b] = r.v1;
      a.best[2 * b + 1] = r.v2;
    }
  }
  __syncthreads();
  int win = s_top[0].i1;
  if (threadIdx.x < g && a.z_out) {
    int j = threadIdx.x;
    a.z_out[(long long)b * a.max_glen + j] = dadd_rn(s_mp[j], dmul_rn(s_sp[j], xi[(long long)win * g + j]));
  }
}

extern "C" int rcb_rec_score_argmax(const float* loc, const float* scale, int32_t cols, const float* p_loc,
                                    const float* p_scale, const double* const* tables, int32_t max_glen,
                                    const double* gumbel, int32_t n_candidates, const int32_t* job_row,
                                    const int32_t* job_start, const int32_t* job_glen, int32_t n_jobs,
                                    int32_t* idx, double* z_out, double* best, double* logw_job0,
                                    rcb_stream_t stream) {
  RCB_REQUIRE(loc && scale && p_loc && p_scale && tables && gumbel && job_row && job_start && job_glen && idx,
              RCB_ERR_ARG, "rec_score: null pointer");
  RCB_REQUIRE(max_glen >= 1 && max_glen <= kMaxGlen, RCB_ERR_UNSUPPORTED, "rec_score: group length %d > %d", max_glen, kMaxGlen);
  RCB_REQUIRE(n_candidates > 0 && cols > 0, RCB_ERR_SHAPE, "rec_score: empty shape");
  if (n_jobs == 0) return RCB_OK;
  RCB_REQUIRE(n_jobs > 0, RCB_ERR_SHAPE, "rec_score: n_jobs < 0");
  RecArgs a;
  memset(&a, 0, sizeof(a));
  a.loc = loc;
  a.scale = scale;
  a.p_loc = p_loc;
  a.p_scale = p_scale;
  a.cols = cols;
  for (int g = 0; g <= max_glen; ++g) a.tables[g] = tables[g];
  a.max_glen = max_glen;
  a.gumbel = gumbel;
  a.K = n_candidates;
  a.job_row = job_row;
  a.job_start = job_start;
  a.job_glen = job_glen;
  a.idx = idx;
  a.z_out = z_out;
  a.best = best;
  a.logw0 = logw_job0;
  rec_score_kernel<<<n_jobs, 256, 0, (hipStream_t)stream>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
