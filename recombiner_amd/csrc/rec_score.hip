// K9: A* (relative entropy coding) candidate scoring, fp64, batched over (row, group) jobs.
// Reference: test_model.py:501-533 (sample_group; h_/hh_ copies :535-584) and :586-619 (compress_group).
//
//   z     = p_loc + p_scale * xi                                      (mul, then add)
//   logN  = -((z - loc)^2) / (2 var) - log(scale) - log(sqrt(2 pi))   var, log(scale) in fp32
//   log_w = sum_j logN_q - sum_j logN_p + gumbel ; first argmax.
//
// Two scorers share one job description:
//   * rec_exact_kernel -- the reference arithmetic op for op in fp64 (separate mul / add / true division), one workgroup
//     per job.  It produces the published log-weights and is the arbiter for every job the fast scorer cannot certify.
//   * rec_fast_kernel  -- the same log-weight as a quadratic in xi,  c0 + sum_j (c1_j xi_j + c2_j xi_j^2) + gumbel,  two
//     fp64 FMAs per (candidate, element) instead of two divisions and a dozen dependent operations; eight jobs of one group
//     length share every table element a workgroup loads (coefficients arrive through scalar loads), 1024 candidates per
//     workgroup so that even a single job fills the chip.
// The index is the product, so the fast scorer never decides a close call: rec_prep_kernel derives, per job, a bound E on
// |fast - exact| over ALL candidates (first-order rounding analysis of both evaluation orders, inflated 2x); if the
// gap between the best and the second-best fast score exceeds 2E the exact arg-max provably is the fast one, otherwise the
// job is flagged and re-scored by the exact kernel.  Measured gaps are ~1 (min 3e-3) against E ~ 1e-10.
//
// Candidate tables are stored TRANSPOSED as fp32 [g][K]: the reference table is fp32-precision ndtri widened to fp64
// (SURVEY A16), so fp32 storage is exact, and with the candidate index on the lane every load is coalesced.
#include "rcb_common.h"

#pragma clang fp contract(off)

// un-contracted fp64 mul / add (HIP's __dmul_rn / __dadd_rn come from headers compiled with -ffp-contract=fast: after
// inlining LLVM may fuse them; these are defined under the pragma above)
__device__ __forceinline__ double dadd_rn(double a, double b) { return a + b; }
__device__ __forceinline__ double dmul_rn(double a, double b) { return a * b; }

using namespace rcb;

namespace {

constexpr int kFastJobs = 8;       // jobs per workgroup of the fast scorer
constexpr int kFastCpt = 4;        // candidates per thread
constexpr int kFastThreads = 256;
constexpr int kFastSpan = kFastCpt * kFastThreads;   // candidates per workgroup
constexpr int kExactConsts = 7;    // mq, mp, sp, 2 var_q, 2 var_p, log s_q, log s_p
constexpr double kLogSqrt2Pi = 0.91893853320467267;   // math.log(math.sqrt(2 * math.pi))

struct Top2 {
  double v1;
  double v2;
  int i1;
};

// in memory (LDS / workspace): fields are stored and loaded one by one -- a struct copy through memory keeps a stack slot
struct Top2Mem {
  double v1;
  double v2;
  long long i1;
};

__device__ __forceinline__ void top2_store(Top2Mem* p, const Top2& t) {
  p->v1 = t.v1;
  p->v2 = t.v2;
  p->i1 = t.i1;
}

__device__ __forceinline__ void top2_load(Top2& t, const Top2Mem* p) {
  t.v1 = p->v1;
  t.v2 = p->v2;
  t.i1 = (int)p->i1;
}

__device__ __forceinline__ void top2_init(Top2& t) {
  t.v1 = -INFINITY;
  t.v2 = -INFINITY;
  t.i1 = 0x7fffffff;
}

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

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

__device__ __noinline__ Top2 top2_wave_generic(Top2 t) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Top2 o;
    o.v1 = __shfl_xor(t.v1, off, 64);
    o.i1 = __shfl_xor(t.i1, off, 64);
    o.v2 = __shfl_xor(t.v2, off, 64);
    t = top2_merge(t, o);
  }
  return t;
}

// wave-wide merge of per-lane (best, runner-up, index): max of the bests; its lane by ballot (ties between lanes -- which
// essentially never happen with fp64 scores -- take the generic butterfly); runner-up = max over the other lanes' bests and
// the winner lane's own runner-up.  ~45 instructions instead of ~180 for the butterfly of triples.
__device__ __forceinline__ Top2 top2_wave(Top2 t) {
  const double m = wave_max(t.v1);
  const unsigned long long hit = __ballot(t.v1 == m);
  if (__popcll(hit) != 1) return top2_wave_generic(t);      // wave-uniform: exact tie between lanes (or NaN everywhere)
  const int src = __ffsll((long long)hit) - 1;
  Top2 r;
  r.v1 = m;
  r.i1 = __shfl(t.i1, src, 64);
  const bool mine = (t.v1 == m);
  r.v2 = wave_max(mine ? t.v2 : t.v1);
  return r;
}

// workspace carved out of the caller's buffer
struct RecWs {
  double2* coef;   // [batches][max_glen][8 slots] {c1, c2}: job b sits in batch b / 8, slot b % 8; elements past a job's
                   // length and the slots past the last job are zero
  double* exc;     // [n_jobs * max_glen * 7] constants of the exact scorer
  double* c0;      // [n_jobs]
  double* tau;     // [n_jobs] certification threshold on the fast top-2 gap
  Top2Mem* part;   // [n_jobs * n_split]
  int32_t* glen;   // [n_jobs] validated group length (0: job rejected)
  uint8_t* flag;   // [n_jobs] 1: needs the exact scorer
};

struct RecArgs {
  const float *loc, *scale, *p_loc, *p_scale;
  int rows, cols;
  const float* const* tables;
  const double* tab_absmax;
  int max_glen;
  const double* gumbel;
  double gumbel_absmax;
  int K;
  const int *job_row, *job_start, *job_glen;
  int n_jobs;
  int n_split;
  RecWs ws;
};

inline int64_t align16(int64_t v) { return (v + 15) & ~int64_t(15); }

inline int64_t carve(RecWs& w, char* base, int n_jobs, int max_glen, int n_split) {
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align16(bytes);
    return p;
  };
  w.coef = (double2*)take(int64_t((n_jobs + kFastJobs - 1) / kFastJobs) * kFastJobs * max_glen * sizeof(double2));
  w.exc = (double*)take(int64_t(n_jobs) * max_glen * kExactConsts * sizeof(double));
  w.c0 = (double*)take(int64_t(n_jobs) * sizeof(double));
  w.tau = (double*)take(int64_t(n_jobs) * sizeof(double));
  w.part = (Top2Mem*)take(int64_t(n_jobs) * n_split * sizeof(Top2Mem));
  w.glen = (int32_t*)take(int64_t(n_jobs) * sizeof(int32_t));
  w.flag = (uint8_t*)take(int64_t(n_jobs));
  return off;
}

// ---------------------------------------------------------------------------------------------------------------------
// per-job constants of both scorers + the certification threshold.  One wave per job.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) rec_prep_kernel(RecArgs a) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  // coef(j) of this job: interleaved with the other seven jobs of its batch, so that the fast scorer fetches the eight
  // coefficient pairs of one element with two 64-byte scalar loads
  double2* coef = a.ws.coef + ((long long)(b / kFastJobs) * a.max_glen) * kFastJobs + (b % kFastJobs);
  if (b >= a.n_jobs) {   // padding slots of the last batch
    for (int j = lane; j < a.max_glen; j += 64) coef[(long long)j * kFastJobs] = make_double2(0.0, 0.0);
    return;
  }
  const int row = a.job_row[b], start = a.job_start[b];
  int g = a.job_glen[b];
  const bool ok = g >= 1 && g <= a.max_glen && row >= 0 && row < a.rows && start >= 0 && start + g <= a.cols &&
                  a.tables[g] != nullptr;
  if (!ok) g = 0;
  const double X = ok ? a.tab_absmax[g] : 0.0;
  double* exc = a.ws.exc + (long long)b * a.max_glen * kExactConsts;
  double c0 = 0.0, S = 0.0;
  for (int j = lane; j < g; j += 64) {
    const float mqf = a.loc[(long long)row * a.cols + start + j];
    const float sqf = a.scale[(long long)row * a.cols + start + j];
    const float mpf = a.p_loc[start + j];
    const float spf = a.p_scale[start + j];
    const double mq = mqf, mp = mpf, sp = spf;
    const double v2q = (double)(2.0f * __fmul_rn(sqf, sqf));   // 2 * scale**2 in fp32, then widened (torch Normal.log_prob)
    const double v2p = (double)(2.0f * __fmul_rn(spf, spf));
    const double lq = (double)logf(sqf);
    const double lp = (double)logf(spf);
    double* e = exc + (long long)j * kExactConsts;
    e[0] = mq;
    e[1] = mp;
    e[2] = sp;
    e[3] = v2q;
    e[4] = v2p;
    e[5] = lq;
    e[6] = lp;
    // fast form: -(dmq + sp x)^2 / v2q + (sp x)^2 / v2p - lq + lp
    const double rq = 1.0 / v2q, rp = 1.0 / v2p;
    const double dmq = mp - mq;
    const double sp2 = sp * sp;
    coef[(long long)j * kFastJobs] = make_double2(-2.0 * dmq * sp * rq, sp2 * rp - sp2 * rq);
    c0 += -(dmq * dmq) * rq - lq + lp;
    // magnitude bound of every intermediate of either evaluation order over |xi| <= X
    const double aq = fabs(mp) + fabs(mq) + sp * X;
    const double ap = 2.0 * fabs(mp) + sp * X;
    S += aq * aq * rq + fabs(lq) + kLogSqrt2Pi + ap * ap * rp + fabs(lp) + kLogSqrt2Pi;
  }
  for (int j = g + lane; j < a.max_glen; j += 64) coef[(long long)j * kFastJobs] = make_double2(0.0, 0.0);
  c0 = wave_sum(c0);
  S = wave_sum(S);
  if (lane == 0) {
    a.ws.c0[b] = c0;
    a.ws.glen[b] = g;
    a.ws.flag[b] = 0;
    // |exact - true| <= (g + 12) u S, |fast - true| <= (2 g + 16) u S  (u = 2^-53; S also covers the Gumbel term);
    // two candidates can swap only if their fast gap is below twice the sum; doubled once more for slack
    const double u = 1.1102230246251565e-16;
    a.ws.tau[b] = 4.0 * (3.0 * g + 28.0) * u * (S + a.gumbel_absmax);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// fast scorer: grid (job batches, candidate spans)
// ---------------------------------------------------------------------------------------------------------------------
typedef const float __attribute__((address_space(1))) * gfloat_ptr;

// coef / glen / c0 / tables / gumbel are separate `const __restrict__` kernel parameters (not members of the by-value
// struct): only then can the compiler prove that the kernel's own stores do not clobber them and fetch the wave-uniform
// coefficients with scalar loads (s_load_dwordx4 feeding v_fmac_f64 directly)
__global__ void __launch_bounds__(kFastThreads)
rec_fast_kernel(const double2* __restrict__ coef, const int32_t* __restrict__ glen, const double* __restrict__ c0s,
                const float* const* __restrict__ tables, const double* __restrict__ gumbel, Top2Mem* __restrict__ part,
                int n_jobs, int max_glen, int K_, int n_split) {
  __shared__ Top2Mem s_top[kFastJobs][kFastThreads / 64];
  const int b0 = blockIdx.x * kFastJobs;
  const int nj = min(kFastJobs, n_jobs - b0);
  const int kbase = blockIdx.y * kFastSpan + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long K = K_;
  double gum[kFastCpt];
  int kc[kFastCpt];      // clamped candidate index: loads stay unconditional, out-of-range lanes are dropped at the end
#pragma unroll
  for (int c = 0; c < kFastCpt; ++c) {
    int k = kbase + c * kFastThreads;
    kc[c] = min(k, K_ - 1);
    gum[c] = gumbel[kc[c]];
  }
  int rs = 0;
  while (rs < nj) {        // runs of equal group length (jobs arrive sorted: normally one run)
    const int g = glen[b0 + rs];
    int re = rs + 1;
    while (re < nj && glen[b0 + re] == g) ++re;
    gfloat_ptr tab = (gfloat_ptr)(uintptr_t)tables[g];
    // all eight slots are evaluated; slots outside this run (another group length, or padding) are simply not read out
    const double2* __restrict__ cf = coef + (long long)blockIdx.x * max_glen * kFastJobs;
    double acc[kFastJobs][kFastCpt];
#pragma unroll
    for (int s = 0; s < kFastJobs; ++s)
#pragma unroll
      for (int c = 0; c < kFastCpt; ++c) acc[s][c] = 0.0;
    float f[kFastCpt];
    double2 q[kFastJobs];
    if (g > 0) {
#pragma unroll
      for (int c = 0; c < kFastCpt; ++c) f[c] = tab[kc[c]];
#pragma unroll
      for (int s = 0; s < kFastJobs; ++s) q[s] = cf[s];
    }
    for (int j = 0; j < g; ++j) {
      double x[kFastCpt], x2[kFastCpt];
#pragma unroll
      for (int c = 0; c < kFastCpt; ++c) {
        x[c] = (double)f[c];
        x2[c] = x[c] * x[c];
      }
      double2 qc[kFastJobs];
#pragma unroll
      for (int s = 0; s < kFastJobs; ++s) qc[s] = q[s];
      const int jn = min(j + 1, g - 1);      // the next element's table row and coefficients are requested before this one's FMAs
#pragma unroll
      for (int c = 0; c < kFastCpt; ++c) f[c] = tab[(long long)jn * K + kc[c]];
#pragma unroll
      for (int s = 0; s < kFastJobs; ++s) q[s] = cf[(long long)jn * kFastJobs + s];
#pragma unroll
      for (int s = 0; s < kFastJobs; ++s) {
#pragma unroll
        for (int c = 0; c < kFastCpt; ++c) acc[s][c] = fma(qc[s].x, x[c], fma(qc[s].y, x2[c], acc[s][c]));
      }
    }
#pragma unroll
    for (int s = 0; s < kFastJobs; ++s) {
      if (s >= rs && s < re) {
        const double c0 = c0s[b0 + s];
        Top2 t;
        top2_init(t);
#pragma unroll
        for (int c = 0; c < kFastCpt; ++c) {
          int k = kbase + c * kFastThreads;
          if (k < K_) top2_push(t, (acc[s][c] + c0) + gum[c], k);
        }
        t = top2_wave(t);
        if (lane == 0) top2_store(&s_top[s][wave], t);
      }
    }
    __syncthreads();
    if (threadIdx.x >= rs && threadIdx.x < re) {
      Top2 r, o;
      top2_load(r, &s_top[threadIdx.x][0]);
#pragma unroll
      for (int w = 1; w < kFastThreads / 64; ++w) {
        top2_load(o, &s_top[threadIdx.x][w]);
        r = top2_merge(r, o);
      }
      top2_store(&part[(long long)(b0 + threadIdx.x) * n_split + blockIdx.y], r);
    }
    __syncthreads();
    rs = re;
  }
}

// merge the spans of a job (fixed order), certify
__global__ void __launch_bounds__(64) rec_finish_kernel(RecArgs a, int* idx, double* best) {
  const int b = blockIdx.x;
  Top2 t;
  top2_init(t);
  for (int s = threadIdx.x; s < a.n_split; s += 64) {
    Top2 o;
    top2_load(o, &a.ws.part[(long long)b * a.n_split + s]);
    t = top2_merge(t, o);
  }
  t = top2_wave(t);
  if (threadIdx.x == 0) {
    const bool valid = a.ws.glen[b] > 0;
    idx[b] = valid ? t.i1 : -1;
    if (best) {
      best[2 * b] = t.v1;
      best[2 * b + 1] = t.v2;
    }
    const bool certified = (t.v1 - t.v2) > a.ws.tau[b];     // false for NaN as well
    a.ws.flag[b] = (valid && !certified) ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// exact scorer: the reference's arithmetic op for op; one workgroup per job (only flagged jobs when `only_flagged`)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rec_exact_kernel(RecArgs a, int only_flagged, int* idx, double* best, double* logw0) {
  __shared__ Top2Mem s_top[4];
  const int b = blockIdx.x;
  if (only_flagged && !a.ws.flag[b]) return;
  const int g = a.ws.glen[b];
  if (g == 0) {
    if (threadIdx.x == 0) idx[b] = -1;
    return;
  }
  const float* __restrict__ tab = a.tables[g];
  const double* __restrict__ exc = a.ws.exc + (long long)b * a.max_glen * kExactConsts;
  const long long K = a.K;
  const double c = kLogSqrt2Pi;
  Top2 t;
  top2_init(t);
  for (int k = threadIdx.x; k < a.K; k += 256) {
    double lq = 0.0, lp = 0.0;
    for (int j = 0; j < g; ++j) {
      const double* e = exc + (long long)j * kExactConsts;
      const double x = (double)tab[(long long)j * K + k];
      double z = dadd_rn(e[1], dmul_rn(e[2], x));
      double dq = z - e[0];
      double dp = z - e[1];
      double tq = ((-(dq * dq)) / e[3] - e[5]) - c;
      double tp = ((-(dp * dp)) / e[4] - e[6]) - c;
      lq = (j == 0) ? tq : lq + tq;
      lp = (j == 0) ? tp : lp + tp;
    }
    double lw = (lq - lp) + a.gumbel[k];
    if (b == 0 && logw0) logw0[k] = lw;
    top2_push(t, lw, k);
  }
  t = top2_wave(t);
  if ((threadIdx.x & 63) == 0) top2_store(&s_top[threadIdx.x >> 6], t);
  __syncthreads();
  if (threadIdx.x == 0) {
    Top2 r, o;
    top2_load(r, &s_top[0]);
    for (int w = 1; w < 4; ++w) {
      top2_load(o, &s_top[w]);
      r = top2_merge(r, o);
    }
    idx[b] = r.i1;
    if (best) {
      best[2 * b] = r.v1;
      best[2 * b + 1] = r.v2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// commit (test_model.py:586-595): z = p_loc + p_scale * xi[idx] in fp64 (mul, then add), stored fp32; masks; beta = 0
// ---------------------------------------------------------------------------------------------------------------------
struct CommitArgs {
  const float *p_loc, *p_scale;
  int rows, cols;
  const float* const* tables;
  int max_glen, K;
  const int *job_row, *job_start, *job_glen, *job_group;
  int n_jobs, n_groups;
  const int* idx;
  double* z_out;
  float *enc_sample, *enc_mask;
  uint8_t* done;
  float* beta;
  int* idx_groupwise;
};

__global__ void __launch_bounds__(64) rec_commit_kernel(CommitArgs a) {
  const int b = blockIdx.x;
  const int row = a.job_row[b], start = a.job_start[b], g = a.job_glen[b];
  const int win = a.idx[b];
  const bool ok = g >= 1 && g <= a.max_glen && row >= 0 && row < a.rows && start >= 0 && start + g <= a.cols && win >= 0 &&
                  win < a.K && a.tables[g] != nullptr;
  if (!ok) return;
  const float* __restrict__ tab = a.tables[g];
  for (int j = threadIdx.x; j < g; j += 64) {
    double z = dadd_rn((double)a.p_loc[start + j], dmul_rn((double)a.p_scale[start + j], (double)tab[(long long)j * a.K + win]));
    if (a.z_out) a.z_out[(long long)b * a.max_glen + j] = z;
    if (a.enc_sample) a.enc_sample[(long long)row * a.cols + start + j] = (float)z;
    if (a.enc_mask) a.enc_mask[(long long)row * a.cols + start + j] = 1.0f;
  }
  if (threadIdx.x == 0 && a.job_group) {
    const int grp = a.job_group[b];
    if (grp >= 0 && grp < a.n_groups) {
      const long long o = (long long)row * a.n_groups + grp;
      if (a.done) a.done[o] = 1;
      if (a.beta) a.beta[o] = 0.0f;
      if (a.idx_groupwise) a.idx_groupwise[o] = win;
    }
  }
}

int check_desc(const rcb_rec_desc* d) {
  RCB_REQUIRE(d, RCB_ERR_ARG, "rec: null descriptor");
  RCB_REQUIRE(d->p_loc && d->p_scale && d->tables_t && d->job_row && d->job_start && d->job_glen, RCB_ERR_ARG,
              "rec: null pointer");
  RCB_REQUIRE(d->max_glen >= 1 && d->n_candidates > 0 && d->cols > 0 && d->rows > 0, RCB_ERR_SHAPE, "rec: empty shape");
  RCB_REQUIRE(d->n_jobs >= 0, RCB_ERR_SHAPE, "rec: n_jobs < 0");
  return RCB_OK;
}

}  // namespace

extern "C" int64_t rcb_rec_workspace_bytes(int32_t n_jobs, int32_t max_glen, int32_t n_candidates) {
  if (n_jobs < 0 || max_glen < 1 || n_candidates < 1) return -1;
  RecWs w;
  return carve(w, nullptr, n_jobs, max_glen, cdiv(n_candidates, kFastSpan));
}

extern "C" int rcb_rec_score_argmax(const rcb_rec_desc* d, int32_t mode, void* workspace, int64_t workspace_bytes,
                                    int32_t* idx, double* best, uint8_t* uncertified, double* logw_job0,
                                    rcb_stream_t stream) {
  int rc = check_desc(d);
  if (rc != RCB_OK) return rc;
  RCB_REQUIRE(d->loc && d->scale && d->gumbel && d->table_absmax && idx && workspace, RCB_ERR_ARG, "rec_score: null pointer");
  RCB_REQUIRE(mode == RCB_REC_EXACT || mode == RCB_REC_FAST, RCB_ERR_ARG, "rec_score: unknown mode %d", mode);
  if (d->n_jobs == 0) return RCB_OK;
  RecArgs a;
  memset(&a, 0, sizeof(a));
  a.loc = d->loc;
  a.scale = d->scale;
  a.p_loc = d->p_loc;
  a.p_scale = d->p_scale;
  a.rows = d->rows;
  a.cols = d->cols;
  a.tables = d->tables_t;
  a.tab_absmax = d->table_absmax;
  a.max_glen = d->max_glen;
  a.gumbel = d->gumbel;
  a.gumbel_absmax = d->gumbel_absmax;
  a.K = d->n_candidates;
  a.job_row = d->job_row;
  a.job_start = d->job_start;
  a.job_glen = d->job_glen;
  a.n_jobs = d->n_jobs;
  a.n_split = cdiv(a.K, kFastSpan);
  int64_t need = carve(a.ws, (char*)workspace, a.n_jobs, a.max_glen, a.n_split);
  RCB_REQUIRE(workspace_bytes >= need, RCB_ERR_SHAPE, "rec_score: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
              (long long)need);
  RCB_REQUIRE(((uintptr_t)workspace & 15) == 0, RCB_ERR_ARG, "rec_score: workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  rec_prep_kernel<<<cdiv(a.n_jobs, kFastJobs) * kFastJobs, 64, 0, s>>>(a);
  RCB_LAUNCH_CHECK();
  if (mode == RCB_REC_FAST) {
    dim3 grid(cdiv(a.n_jobs, kFastJobs), a.n_split);
    rec_fast_kernel<<<grid, kFastThreads, 0, s>>>(a.ws.coef, a.ws.glen, a.ws.c0, a.tables, a.gumbel, a.ws.part, a.n_jobs,
                                                  a.max_glen, a.K, a.n_split);
    RCB_LAUNCH_CHECK();
    rec_finish_kernel<<<a.n_jobs, 64, 0, s>>>(a, idx, best);
    RCB_LAUNCH_CHECK();
    if (uncertified) {
      hipError_t e = hipMemcpyAsync(uncertified, a.ws.flag, a.n_jobs, hipMemcpyDeviceToDevice, s);
      RCB_REQUIRE(e == hipSuccess, (int)e, "rec_score: flag copy failed: %s", hipGetErrorString(e));
    }
    // the arbiter: workgroups of certified jobs exit at once
    rec_exact_kernel<<<a.n_jobs, 256, 0, s>>>(a, 1, idx, best, nullptr);
    RCB_LAUNCH_CHECK();
    if (logw_job0) {   // published log-weights are always the exact ones
      RecArgs a0 = a;
      a0.n_jobs = 1;
      rec_exact_kernel<<<1, 256, 0, s>>>(a0, 0, idx, best, logw_job0);
      RCB_LAUNCH_CHECK();
    }
  } else {
    if (uncertified) {
      hipError_t e = hipMemsetAsync(uncertified, 0, a.n_jobs, s);
      RCB_REQUIRE(e == hipSuccess, (int)e, "rec_score: memset failed: %s", hipGetErrorString(e));
    }
    rec_exact_kernel<<<a.n_jobs, 256, 0, s>>>(a, 0, idx, best, logw_job0);
    RCB_LAUNCH_CHECK();
  }
  return RCB_OK;
}

extern "C" int rcb_rec_commit(const rcb_rec_desc* d, const int32_t* idx, const int32_t* job_group, int32_t n_groups,
                              double* z_out, float* enc_sample, float* enc_mask, uint8_t* done, float* beta,
                              int32_t* idx_groupwise, rcb_stream_t stream) {
  int rc = check_desc(d);
  if (rc != RCB_OK) return rc;
  RCB_REQUIRE(idx, RCB_ERR_ARG, "rec_commit: null index array");
  RCB_REQUIRE(job_group || !(done || beta || idx_groupwise), RCB_ERR_ARG, "rec_commit: per-group state needs job_group");
  if (d->n_jobs == 0) return RCB_OK;
  CommitArgs a;
  memset(&a, 0, sizeof(a));
  a.p_loc = d->p_loc;
  a.p_scale = d->p_scale;
  a.rows = d->rows;
  a.cols = d->cols;
  a.tables = d->tables_t;
  a.max_glen = d->max_glen;
  a.K = d->n_candidates;
  a.job_row = d->job_row;
  a.job_start = d->job_start;
  a.job_glen = d->job_glen;
  a.job_group = job_group;
  a.n_jobs = d->n_jobs;
  a.n_groups = n_groups;
  a.idx = idx;
  a.z_out = z_out;
  a.enc_sample = enc_sample;
  a.enc_mask = enc_mask;
  a.done = done;
  a.beta = beta;
  a.idx_groupwise = idx_groupwise;
  rec_commit_kernel<<<a.n_jobs, 64, 0, (hipStream_t)stream>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
