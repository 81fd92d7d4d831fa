/*
 * librcb_hip  --  C ABI of the MI355X (gfx950) kernels behind the RECOMBINER per-datapoint
 * INR training hot path.
 *
 * The reference (cambridge-mlg/RECOMBINER) is pure PyTorch and has no FFI of its own; each entry
 * point below replaces a sequence of torch ops on the reference's hot path and cites it
 * (file:line relative to the reference root).  Conventions:
 *   - plain device pointers + explicit sizes; the caller (PyTorch, or any HIP host) allocates
 *     every buffer including outputs; nothing is allocated, freed or synchronised inside;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it;
 *   - return value: 0 = ok, < 0 = argument/shape error (RCB_ERR_*), > 0 = hipError_t;
 *   - no global mutable state besides the thread-local last-error string; re-entrant per stream;
 *   - fp32 tensors are row-major and contiguous unless a stride argument says otherwise.
 */
#ifndef RCB_H_
#define RCB_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* major * 100 + minor; bumped with every change of a signature or of a structure layout (100: rounds 1-3; 400: round 4 --
 * rcb_level.scale_is_sigma, rcb_struct_bytes, the hi / lo operand planes of the A transform).  A binding compares
 * rcb_version() with the RCB_VERSION it was written against and rcb_struct_bytes() with the size of each of its mirrors. */
#define RCB_VERSION 406
#define RCB_OK 0
#define RCB_ERR_ARG (-1)
#define RCB_ERR_SHAPE (-2)
#define RCB_ERR_UNSUPPORTED (-3)
#define RCB_KL_SLOTS 1024
#define RCB_KL_FX_SCALE 16777216.0   /* 2^24: the KL slots count 2^-24 nats */

typedef void* rcb_stream_t;

int rcb_version(void);
const char* rcb_last_error_string(void);
/* sizeof() of the descriptor structures as this library was compiled -- which: 0 rcb_siren_desc, 1 rcb_level,
 * 2 rcb_level_bwd, 3 rcb_adam_cfg, 4 rcb_adam_tensor, 5 rcb_rec_desc; -1 for any other value.  A hand-written mirror of a
 * structure (ctypes, cffi ...) that has drifted from this header is caught at load time instead of corrupting a launch. */
#define RCB_STRUCT_SIREN_DESC 0
#define RCB_STRUCT_LEVEL 1
#define RCB_STRUCT_LEVEL_BWD 2
#define RCB_STRUCT_ADAM_CFG 3
#define RCB_STRUCT_ADAM_TENSOR 4
#define RCB_STRUCT_REC_DESC 5
int64_t rcb_struct_bytes(int32_t which);

/* ---------------------------------------------------------------------------------------------
 * K3 + K4: batched SIREN coordinate-MLP, one workgroup per (INR, sample).
 * Replaces prior_model.py:168-179 (+121-127) and test_model.py:347-355 (+269-280): per layer
 * `x = x @ W + b; x = sin(30 x)` with per-INR weights, and the MSE term prior_model.py:237 /
 * test_model.py:625-627.
 *
 * Input features of layer 0 are the concatenation [xf | pe] (prior_model.py:155): `xf` is the
 * Fourier embedding [*, P, F] (row stride xf_inr_stride between INRs, 0 = one grid shared by all
 * INRs), `pe` the upsampled positional encodings [G, P, E] (E may be 0 with pe = NULL).
 * `wvec` row g holds the layer vectors back to back, each `[bias(out) | W(in,out) row-major]`
 * (prior_model.py:125-126); rows are w_row_stride floats apart.  G = N*S; sample s of INR n is
 * row g = n*S + s; targets / xf are indexed by n = g / S.
 * Hidden width 32, 48 or 64 in the 16-bit modes; any width up to 64 in the fp32 mode (32 on the matrix cores, the others on a
 * plain-FMA parity kernel, reference prior_model.py:84-85); 1..4 hidden layers, out_dim <= 32, F + E <= 64.
 * precision: 0 = fp32 MFMA (exact fp32 products); 1 = bf16 operands, 2 = f16 operands (both fp32 accumulate;
 * f16 carries gradients scaled by 2^10 internally).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n_rows;        /* G = N * S workgroups                                   */
  int32_t samples;       /* S                                                       */
  int32_t n_pix;         /* P                                                       */
  int32_t fourier_dim;   /* F                                                       */
  int32_t pe_dim;        /* E                                                       */
  int32_t n_hidden;      /* number of hidden layers                                 */
  int32_t hidden;        /* hidden width (16-bit modes: 32, 48, 64; fp32: 1 .. 64)   */
  int32_t out_dim;       /* C                                                       */
  int64_t xf_inr_stride; /* floats between INRs in xf; 0 = shared grid              */
  int64_t w_row_stride;  /* floats between rows of wvec and of dwvec                */
  float   w0;            /* sine frequency (30)                                     */
  int32_t precision;     /* 0 fp32, 1 bf16, 2 f16 operands                           */
  int32_t pe_bf16;       /* 1: pe and dpe are bf16 arrays (precision >= 1, pe_dim % 8 == 0): the 16-bit
                          * kernels round pe / dpe to bf16 for their MFMA operands anyway, so storing
                          * them as bf16 gives bit-identical results with half the traffic        */
  void* dw_bf16;         /* nullable, 16-bit kernels, rcb_siren_bwd / _loss_bwd only: besides dwvec, a bf16 copy of it,
                          * [n_rows][dw_bf16_stride] with dwvec's column layout -- the operand of the batched bf16
                          * weight-gradient GEMM of the A transform (rcb_atrans_*), written while the values are in
                          * registers instead of by a cast pass over dwvec                                      */
  int32_t pixel_chunks;  /* 0 / 1: one workgroup per row of wvec.  c > 1 (16-bit kernels): the 32-pixel tiles of every row
                          * are split over c workgroups (launches with few rows -- one Kodak photo is 96 INRs -- would
                          * otherwise leave most of the 256 CUs idle).  y_out / dpe are unaffected; dwvec and sse then
                          * receive PARTIAL results: dwvec [c][n_rows][w_row_stride], sse [c][n_rows], to be summed in
                          * chunk order by rcb_siren_reduce_chunks; dw_bf16 must be NULL (the reduction emits it)   */
  const void* xf_bf16;   /* nullable, 16-bit operand modes with pe_bf16 = 1: a 16-bit copy of xf IN THE OPERAND FORMAT (bf16 for
                          * precision 1, IEEE f16 for precision 2), rows padded with zeros to a multiple of 8 features
                          * ([P][8 * ceil(F / 8)], 16-byte rows; the per-row stride is xf_row_stride / F * that; since version
                          * 406 -- before, bf16 only, unpadded, and ignored unless F % 8 == 0).  The coordinate grid is constant
                          * for a whole run and the kernel rounds it to its operand format anyway: given the copy, both input
                          * halves are loaded as 16-bit rows (no widening to fp32 / re-rounding per tile, a third of the input
                          * registers).  The rounding of the inputs is the same with and without it.                 */
  int32_t pe_grid_dims;  /* 0: pe / dpe are [G][P][E].  1..3 (16-bit kernels): the rows are the PATCHES of stitched grids, as the
                          * reference's patched presets build them (utils.py:60-116: the latent grids of a datapoint's patches
                          * are stitched, upsampled together and cut back into patches): pe / dpe are the upsampling net's own
                          * channel-last output [S * n_datapoints][G_0]..[G_d-1][E], G_i = pe_patch_nums[i] * pe_patch_size[i];
                          * row g = n * S + s is patch n % prod(patch_nums) (row-major) of datapoint n / prod(patch_nums) in
                          * image s * n_datapoints + datapoint, its pixel p the row-major position inside the patch.  Saves
                          * the cut-back copy of pe and the stitching copy of dpe; results are bit-identical          */
  int32_t pe_patch_nums[3];  /* patches per axis (first pe_grid_dims entries)                                          */
  int32_t pe_patch_size[3];  /* pixels per axis of one patch; their product is n_pix                                   */
  int64_t dw_bf16_stride;    /* elements between the rows of dw_bf16 (>= the length of a row of layer vectors)         */
  int32_t hidden_dims[4];    /* all zero: every hidden layer is `hidden` wide.  Otherwise the widths of the n_hidden hidden
                              * layers one by one (the reference builds its INR from any list, prior_model.py:84-85): fp32
                              * mode only (1 .. 64 each; the plain-FMA parity kernel), `hidden` = their maximum            */
  void* dw_lo;               /* nullable, with dw_bf16 (the HIGH plane): the LOW plane bf16(dwvec - dw_bf16), same layout and
                              * stride.  The pair is the plane operand of the A transform's data gradient (rcb_atrans_apply
                              * x_hi / x_lo): the same 4 bytes per element as the fp32 dwvec, which may then be NULL in
                              * rcb_siren_bwd / _loss_bwd (unchunked launches) and rcb_siren_reduce_chunks                */
  uint64_t* clock_probe;     /* nullable measurement aid (width-32 16-bit kernel): workgroup b < 256 writes clock_probe[4 b ..
                              * 4 b + 3] = {s_memtime, s_memrealtime at its start, s_memtime, s_memrealtime at its end}.  Both
                              * counters are read on the workgroup's own CU (s_memtime is not comparable between CUs), so
                              * (t2 - t0) / ((t3 - t1) / 1e8) is the SHADER CLOCK that workgroup ran at (s_memrealtime counts
                              * 100 MHz): the figure an issue-bound kernel's roofline is priced at -- MI355X lowers its clock
                              * under matrix / vector load, peak-clock rooflines overstate what such kernels can reach      */
} rcb_siren_desc;

/* Test hook: which kernel family runs the width-32 bf16 loss / backward launches whose inputs arrive as 16-bit rows (pe_bf16,
 * xf_bf16, [G][P][E] pe, rows of whole 32-pixel tiles): 1 (default) -> one WAVE per row (siren_mlp_wave.hip: 221 vs 236 us per
 * launch at BASELINE configs[1] size) where the rows fill the resident waves to 90 % in the last round, else -- and for
 * everything that family has no instance for -- one workgroup per row (siren_mlp_bf16.hip); 2 -> the wave family whatever
 * the number of rows (tests); 0 -> the workgroup family everywhere; < 0 only queries.  Returns the previous setting.  Both
 * families evaluate the same products with fp32 accumulation; they differ in the order in which the pixel tiles of a row are
 * summed and in how the bias enters the accumulator (fp32 value / two bf16 halves through the matrix pipe). */
int rcb_debug_siren_wave_tiles(int32_t tiles);

/* y_out[G, P, C] = MLP(x)                                                           */
int rcb_siren_fwd(const rcb_siren_desc* d, const float* xf, const void* pe, const float* wvec,
                  float* y_out, rcb_stream_t stream);

/* Given dy[G, P, C]: dwvec[G, :] (same layout/stride as wvec) and, if dpe != NULL, dpe[G, P, E].
 * The forward pass is recomputed in registers; no activations are read from memory.            */
int rcb_siren_bwd(const rcb_siren_desc* d, const float* xf, const void* pe, const float* wvec,
                  const float* dy, float* dwvec, void* dpe, rcb_stream_t stream);

/* Sum of the per-chunk partials of a pixel_chunks = c launch, in chunk order (deterministic): dwvec[g, :] = sum_k
 * dw_partial[k][g][:], sse[g] = sum_k sse_partial[k][g] (sse_partial / sse may be NULL for rcb_siren_bwd), and, if
 * d->dw_bf16 != NULL, the bf16 copy of the sums as rcb_siren_desc.dw_bf16 describes.                              */
int rcb_siren_reduce_chunks(const rcb_siren_desc* d, const float* dw_partial, const float* sse_partial, float* dwvec,
                            float* sse, rcb_stream_t stream);

/* Fused training pass: sse[g] = sum_{p,c} (y - target[g/S])^2 and the gradients of
 * dy_scale * sse[g] with respect to wvec row g and pe row g (dpe may be NULL).  With
 * dy_scale = 1/(S*P*C) this is the reference's `mean((y_hat - y)**2) * N` term.              */
int rcb_siren_loss_bwd(const rcb_siren_desc* d, const float* xf, const void* pe,
                       const float* wvec, const float* target, float dy_scale, float* sse,
                       float* dwvec, void* dpe, rcb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K1 + K10: reparameterised sampling of the latent vector of every (INR, sample) from a 1- or
 * 3-level factorised Gaussian posterior.  Replaces utils.py:142-198, prior_model.py:140-145 and
 * the mask/permutation gathers test_model.py:289-298,320-330.
 *
 *   out[n, s, d] = sum_levels  mu_L(n, d) + sigma_L(n, d) * eps_L[n, s, d]
 *   mu    = loc * (1 - m) + enc_sample * m          sigma = softplus(log_scale)/6 * (1 - m) + 1e-15 * m
 * evaluated at source element (r, j):  j = col_map ? col_map[d] : d;  r0 = row_map ? row_map[n] : n;
 * r = row_perm ? row_perm[r0 * cols + j] : r0.  A level contributes to d < cols_out only.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const float* loc;        /* [rows, cols]                                          */
  const float* log_scale;  /* [rows, cols]                                          */
  const float* enc_sample; /* nullable [rows, cols]                                 */
  const float* enc_mask;   /* nullable [rows, cols] (0/1)                           */
  const int32_t* row_map;  /* nullable [N]                                          */
  const int32_t* row_perm; /* nullable [rows, cols]                                 */
  const int32_t* col_map;  /* nullable [cols_out]                                   */
  const float* eps;        /* [N, S, cols_out]                                      */
  int32_t rows, cols, cols_out;
  int32_t scale_is_sigma;  /* 0: `log_scale` holds log-scales, sigma = softplus(.)/6 (the models' parameters).  1: it holds
                            * sigma itself -- the argument convention of utils.py:122-137, whose callers apply st() and the
                            * encoded-group masks first (prior_model.py:140-145, test_model.py:289-298); generic kernel only */
  float* mu_sigma_ws;      /* nullable scratch [2 * rows * cols]: gathered levels (col_map / row_perm) first pack the effective
                            * (mu, sigma) of every element as one 8-byte record there, contiguously; the sampler then gathers one
                            * record per level and element instead of up to four 4-byte values -- same samples bit for bit   */
} rcb_level;

int rcb_reparam_fwd(const rcb_level* levels, int32_t n_levels, int32_t n_inr, int32_t samples,
                    int32_t out_cols, float* out, rcb_stream_t stream);

/* Noise drawn in the kernel (prior_model.py:140-145 draws torch.randn_like and then forms loc + st(log_scale) * eps):
 * eps[i] ~ N(0,1) from Philox4x32-10 + Box-Muller, a pure function of (seed, rng_stream, step, i), where `step` is read
 * from device memory so that a replayed HIP graph draws fresh noise.  rcb_reparam_rng_fwd: plain case (one level, one
 * sample, no maps), flat over n = rows * cols elements; writes eps_out (for rcb_posterior_bwd; nullable: a consumer with
 * rcb_level_bwd.eps_from_rng re-draws it) and out, and, if out_bf16 != NULL, a bf16 copy of out as [n / cols][ld_bf16] rows
 * (the operand of the A transform's batched bf16 weight-gradient GEMM, written while the values are in registers); with
 * out_lo != NULL also the low plane bf16(out - out_bf16) in the same layout: (out_bf16, out_lo) are the plane operands of
 * rcb_atrans_apply, and `out` may then be NULL.
 * rcb_philox_normal materialises the same stream (step from step_dev if non-NULL, else step_host).
 * group_offset: element i draws from Philox group i / 4 + group_offset.  A launch over rows [r0, r0 + m) of a larger
 * [rows, cols] array with group_offset = r0 * cols / 4 (r0 * cols a multiple of 4) draws exactly the noise those rows get in
 * a launch over the whole array: a shard -- or a sub-batch -- sees the noise of the unsharded run.                  */
/* rcb_reparam_hier_rng_fwd: the training sample of the patched presets (utils.py:122-198: level 1 one row per INR, the coarser
 * levels behind row maps; one sample, every column produced, no masks / permutations) with the noise of every level drawn in
 * the kernel: level l uses rng stream rng_streams[l] (host array) with the element index n * cols + d of the [n_inr, cols] noise
 * array -- the indexing of rcb_reparam_rng_fwd, so level 1 on stream s draws what that call draws -- and the noise is written
 * to eps_out[l] (host array of device pointers, [n_inr * cols] each) for rcb_posterior_bwd.  rcb_level.eps is ignored.  Same
 * arithmetic and order as rcb_reparam_fwd on that noise: bit-identical.  n_inr * cols must be a multiple of 4.              */
int rcb_reparam_hier_rng_fwd(const rcb_level* levels, int32_t n_levels, int32_t n_inr, int32_t out_cols, float* const* eps_out,
                             float* out, uint64_t seed, const uint32_t* rng_streams, const int64_t* step_dev,
                             uint64_t group_offset, rcb_stream_t stream);
int rcb_philox_normal(float* out, int64_t n, uint64_t seed, uint32_t rng_stream, const int64_t* step_dev, int64_t step_host,
                      uint64_t group_offset, rcb_stream_t stream);
int rcb_reparam_rng_fwd(const float* loc, const float* log_scale, int64_t n, uint64_t seed, uint32_t rng_stream,
                        const int64_t* step_dev, float* eps_out, float* out, void* out_bf16, void* out_lo, int32_t cols,
                        int64_t ld_bf16, uint64_t group_offset, rcb_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * K5 + K6 + K7: KL( N(loc, softplus(log_scale)/6) || N(p_loc, p_scale) ) per element
 * (torch.distributions.kl._kl_normal_normal as used at prior_model.py:191-199, test_model.py:357-377,
 * 384-402), reduced per row and optionally per (row, group) segment.
 *   kl_row[r]      = sum_j w(r,j) * kl(r,j),  w = beta[r, group_idx[j]] if beta != NULL else 1
 *   kl_group[r,g]  = sum_{j in [seg_start[g], seg_end[g])} kl(r,j)        (unweighted, fp64)
 * p_scale_is_log: p_scale holds log-scales to be passed through softplus/6 (test_model.py:358).
 * ------------------------------------------------------------------------------------------- */
int rcb_gauss_kl(const float* loc, const float* log_scale, const float* p_loc, const float* p_scale,
                 int32_t p_scale_is_log, int32_t rows, int32_t cols, const float* beta,
                 const int32_t* group_idx, int32_t n_groups, const int32_t* seg_start,
                 const int32_t* seg_end, double* kl_row, double* kl_group, rcb_stream_t stream);

/* Per-parameter KL summed over rows: the statistic behind get_grouping (prior_model.py:264-271); q_scale holds sigma (or
 * log-scales if q_scale_is_log).  out_fx is int64 [cols + 1]: out_fx[j] = sum over the rows of every ELEMENT's KL rounded to
 * RCB_COLSUM_FX_SCALE units per nat (each element enters the integer grid on its own), accumulated as 64-bit integers:
 * the result does not depend on the order in which the workgroups (or, after an integer all-reduce, the ranks) contribute
 * nor on where the rows are cut into shards -- bitwise reproducible, sharded or not.  out_fx[cols] = number of elements
 * whose KL was NaN / Inf / >= 2^12 nats (counted, not summed; the caller turns a non-zero count into NaN).            */
#define RCB_COLSUM_FX_SCALE 1073741824.0        /* 2^30 */
int rcb_gauss_kl_colsum(const float* loc, const float* q_scale, int32_t q_scale_is_log, const float* p_loc,
                        const float* p_scale, int32_t rows, int32_t cols, int64_t* out_fx, rcb_stream_t stream);

/* K8: per-group beta annealing (test_model.py:404-413): groups above 16+upper bits get
 * beta *= (1+step), groups at or below 16-lower bits beta /= (1+step), clamp [0, 1e4]; encoded
 * groups (done != 0) keep their beta.                                                          */
int rcb_beta_update(const double* kl_group, float* beta, const uint8_t* done, int32_t rows,
                    int32_t n_groups, double bits, double upper, double lower, double step,
                    rcb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K1-bwd + K5-bwd + K11: one pass over a level's parameters that (a) gathers the gradient of the
 * sampled latents back to (loc, log_scale), (b) adds the analytic gradient of kl_weight * KL, and
 * (c) either writes the gradients (g_loc / g_log_scale != NULL) or applies one Adam step in place
 * (torch.optim.Adam defaults, prior_model.py:224-250, test_model.py:633-635).
 *
 * d_out[N, S, cols_out] is the gradient w.r.t. rcb_reparam_fwd's output.  Members of source row r0
 * (the INRs n with row_map[n] == r0) are member_idx[member_ptr[r0] .. member_ptr[r0+1]) (NULL =
 * identity).  row_perm_inv / col_inv are the inverse permutations of row_perm / col_map.
 * kl weight: kl_scalar * (beta ? beta[r, group_idx[j]] : 1).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  float lr, beta1, beta2, eps;
  int32_t step;            /* 1-based step count t                                  */
  const float* dyn_scalars;/* nullable device pair {lr/(1-beta1^t), sqrt(1-beta2^t)}: when set it overrides the
                              values derived from `step`, so a captured HIP graph can be replayed for every t   */
} rcb_adam_cfg;

typedef struct {
  float* loc;              /* [rows, cols] (updated in place when adam != NULL)     */
  float* log_scale;
  const float* enc_mask;   /* nullable                                              */
  const float* p_loc;      /* [cols]                                                */
  const float* p_scale;    /* [cols]                                                */
  int32_t p_scale_is_log;
  const float* beta;       /* nullable [rows, n_groups]                             */
  const int32_t* group_idx;/* nullable [cols]                                       */
  int32_t n_groups;
  float kl_scalar;
  const float* d_out;      /* [N, S, cols_out] ; nullable (KL-only)                 */
  const float* eps;        /* [N, S, cols_out]                                      */
  const int32_t* member_ptr;   /* nullable [rows+1]                                 */
  const int32_t* member_idx;   /* nullable                                          */
  const int32_t* row_perm_inv; /* nullable [rows, cols]                             */
  const int32_t* col_inv;      /* nullable [cols] : d of source column j            */
  int32_t rows, cols, cols_out, samples;
  float* g_loc;            /* nullable outputs [rows, cols]                         */
  float* g_log_scale;
  float* m_loc; float* v_loc; float* m_ls; float* v_ls;   /* Adam state            */
  int64_t* kl_accum;       /* nullable [RCB_KL_SLOTS], fixed point (RCB_KL_FX_SCALE units per nat, integer atomics: the sum does
                            * not depend on the order of the workgroups): partial sums of the unweighted elementwise KL (before the
                            * update) are atomically added to slots [0, RCB_KL_SLOTS - 1); their total is the KL.  The LAST slot
                            * counts workgroup sums that were NaN / Inf or beyond 2^30 nats (not representable): rcb_step_end
                            * logs NaN for such a step, as the reference's diverged run would                            */
  const float* kl_scalar_dev; /* nullable device scalar: the KL weight becomes kl_scalar * (*kl_scalar_dev), so a
                              captured graph can be replayed with a new beta (main_prior_training.py:144-154) */
  /* Optional (plain levels on the flat path only; next_out == NULL: off): the NEXT step's reparameterised sample drawn in
   * the same pass from the parameters just updated -- what rcb_reparam_rng_fwd would compute at step counter
   * *rng_step_dev + rng_step_add, bit for bit -- so that the sampling kernel of the next step (a second read of loc and
   * log_scale) disappears.  next_eps may alias eps: every element is read before it is written. */
  float* next_out;         /* nullable: [rows * cols] fp32 sample                                                */
  float* next_eps;         /* nullable: [rows * cols] the noise of that sample (read by the next step's call as `eps`;
                            * not needed when that call sets eps_from_rng)                                       */
  void* next_out_bf16;     /* nullable: bf16 copy of the sample as [rows][next_ld_bf16] -- the HIGH plane         */
  int64_t next_ld_bf16;
  uint64_t rng_seed;
  const int64_t* rng_step_dev;
  int64_t rng_step_add;
  uint32_t rng_stream;
  uint32_t eps_from_rng;   /* 1 (flat path, eps = NULL): this step's noise is not read from memory but re-drawn -- the noise
                            * of (rng_seed, rng_stream, counter *rng_step_dev + rng_step_add - 1), i.e. what the sampler of
                            * THIS step drew (rcb_reparam_rng_fwd at *rng_step_dev, or the previous call's next sample):
                            * the same bits, 8 bytes of traffic per element less                                  */
  uint64_t rng_group_offset; /* Philox group of element 0 (rcb_reparam_rng_fwd's group_offset), for the next sample and eps_from_rng */
  void* next_out_lo;       /* nullable, with next_out_bf16: the LOW plane bf16(sample - high plane), same layout.  With both
                            * planes next_out may be NULL: rcb_atrans_apply reads the planes (x_hi / x_lo)          */
  float* sample_sum_ws;    /* nullable scratch [2 * rows * cols_out]: levels without members, samples > 1 and col_inv (the
                            * test-time layout of level 1) first form their sums over the samples there, contiguously, and the
                            * update gathers two values per parameter instead of 2 * samples -- same results bit for bit  */
  const int32_t* col_map;  /* nullable [cols_out]: the map col_inv inverts (j = col_map[d]).  A performance hint only: with it
                            * (levels with members, every column produced) threads are indexed by d, the contiguous axis of
                            * d_out / eps, instead of by the parameter column -- same results bit for bit           */
} rcb_level_bwd;

int rcb_posterior_bwd(const rcb_level_bwd* lv, const rcb_adam_cfg* adam, rcb_stream_t stream);

/* K11: Adam on a flat fp32 array (shared mappings A / conv weights).                          */
int rcb_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, const rcb_adam_cfg* cfg,
                  rcb_stream_t stream);

/* The same update for a list of tensors in ONE launch (the optimiser step over the shared mappings,
 * main_prior_training.py / prior_model.py train(): torch.optim.Adam over linear_transform + upsample_net). */
#define RCB_ADAM_MAX_TENSORS 16
typedef struct {
  float* p;            /* parameters, updated in place */
  const float* g;      /* gradient                     */
  float* m;            /* exp_avg                      */
  float* v;            /* exp_avg_sq                   */
  int64_t n;           /* elements                     */
} rcb_adam_tensor;
int rcb_adam_multi(const rcb_adam_tensor* tensors, int32_t count, const rcb_adam_cfg* cfg, rcb_stream_t stream);

/* Test hook: on != 0 forces the generic kernels of rcb_reparam_fwd / rcb_posterior_bwd even where a specialised one
 * (flat 16-byte path, LDS-staged gather) applies; returns the previous setting.  The specialised kernels perform the
 * same operations in the same order, so both settings must give identical bits (tests/test_hip_kernels.py).     */
int rcb_debug_generic_kernels_only(int32_t on);

/* ---------------------------------------------------------------------------------------------
 * K2, hand-written: the A transform of all layer vectors in one launch per direction (atrans.hip).
 *   forward        wvec[:, lo_l:hi_l] = h_w[:, lo_l:hi_l] @ A[l]      prior_model.py:173-174, test_model.py:348-349
 *   data gradient  dh  [:, lo_l:hi_l] = dw [:, lo_l:hi_l] @ A[l]^T    (autograd of the line above)
 *   weight gradient dA[l] = h_w[:, lo_l:hi_l]^T @ dw[:, lo_l:hi_l]    (autograd; summed over the rows = INRs x samples)
 * Layer l is the square map A[l] [sizes[l], sizes[l]] (fp32, row-major); the layer vectors sit side by side in rows of
 * x / out (column offset of layer l = sizes[0] + ... + sizes[l-1], row strides ld_x / ld_out in elements, 4-byte
 * aligned rows suffice).  bf16 matrix cores with fp32 accumulation; the per-row operand is split x = hi + lo inside
 * the kernel, the mapping enters as bf16 (terms 2) or as hi + lo (terms 3: (hi + lo) A_hi + hi A_lo, ~2^-17 relative);
 * terms 1 = plain bf16 operands.  No atomics anywhere: results are bitwise reproducible.
 *
 * rcb_atrans_pack_elems : bf16 elements of the packed images of the mappings (4 planes: forward hi, gradient hi,
 *                         forward lo, gradient lo; each layer padded to a multiple of 32 both ways), < 0 on bad arguments
 * rcb_atrans_pack       : fp32 mappings -> packed images (once per step when they are trained); the lo planes only if want_lo
 * rcb_atrans_plan       : HOST: work decomposition for `rows` rows on n_cu compute units into plan[0 .. return value) (int32);
 *                         the caller keeps a 16-byte aligned device copy of all of it and the first RCB_ATRANS_PLAN_HEAD
 *                         entries on the host (plan[0] = workgroups, plan[2] = slices of the contraction: launches of few
 *                         rows cut it and add the partial sums in a fixed order) for rcb_atrans_apply
 * rcb_atrans_workspace_floats : fp32 elements of the workspace rcb_atrans_apply needs for that plan (0: none)
 *                         (plan = NULL with max_ints = 0: nothing is written, the return value is the number of ints needed;
 *                         a buffer that is too small is RCB_ERR_SHAPE, any other error RCB_ERR_ARG)
 * rcb_atrans_apply      : transpose = 0: out = x @ A (forward), 1: out = x @ A^T (data gradient).  The per-row operand comes
 *                         EITHER as fp32 rows `x` (x_hi = x_lo = NULL; split hi + lo inside the kernel) OR -- x = NULL -- as
 *                         the two bf16 PLANES its producer wrote: x_hi[r][c] = bf16(v), x_lo[r][c] = bf16(v - x_hi[r][c]),
 *                         rows ld_x16 elements apart (a multiple of 8; 32 = 64-byte rows stream best), same column layout.
 *                         The planes are the same 4 bytes per element as the fp32 row, carry exactly what the kernel would
 *                         have formed from it -- results are bit-identical -- and arrive in LDS as operand bits (no
 *                         conversion work); hi alone is the weight-gradient GEMM's operand.  Producers:
 *                         rcb_reparam_rng_fwd (out_bf16 + out_lo), rcb_level_bwd.next_out_bf16 + next_out_lo,
 *                         rcb_siren_desc.dw_bf16 + dw_lo.  x_lo may be NULL with terms = 1
 * rcb_atrans_wgrad_narrow : dA = h^T @ d for ONE narrow layer (the 99-wide output layer) in fp32, exact-product arithmetic,
 *                         fixed summation order; each operand EITHER as fp32 rows (h / d, pointing at the layer's first
 *                         column; the plane pointers NULL) OR as its (hi, lo) planes (h_hi, h_lo / d_hi, d_lo pointing at the
 *                         layer's first column, the fp32 pointer NULL; read as float(hi) + float(lo)); ld_h / ld_d = the row
 *                         stride in elements of whichever form is given; workspace: rcb_atrans_wgrad_narrow_workspace(L,
 *                         n_slabs) floats
 * ------------------------------------------------------------------------------------------- */
#define RCB_ATRANS_MAX_LAYERS 8
#define RCB_ATRANS_PLAN_HEAD 12
int64_t rcb_atrans_pack_elems(int32_t n_layers, const int32_t* sizes);
int rcb_atrans_pack(const float* const* A, int32_t n_layers, const int32_t* sizes, void* packed, int32_t want_lo,
                    rcb_stream_t stream);
int rcb_atrans_plan(int64_t rows, int32_t n_layers, const int32_t* sizes, int32_t n_cu, int32_t* plan, int32_t max_ints);
int rcb_atrans_apply(const float* x, int64_t ld_x, const void* x_hi, const void* x_lo, int64_t ld_x16, float* out,
                     int64_t ld_out, int64_t rows, int32_t n_layers, const int32_t* sizes, const void* packed, int32_t transpose,
                     int32_t terms, const int32_t* plan_dev, const int32_t* plan_head, float* workspace, rcb_stream_t stream);
int64_t rcb_atrans_workspace_floats(int64_t rows, int32_t n_layers, const int32_t* sizes, const int32_t* plan_head);
int64_t rcb_atrans_wgrad_narrow_workspace(int32_t L, int32_t n_slabs);
int rcb_atrans_wgrad_narrow(const float* h, const void* h_hi, const void* h_lo, int64_t ld_h, const float* d, const void* d_hi,
                            const void* d_lo, int64_t ld_d, int64_t rows, int32_t L, float* dA, float* workspace,
                            int32_t n_slabs, rcb_stream_t stream);

/* Bookkeeping of one optimisation step whose counter lives on the device, so that the whole step can be captured
 * once as a HIP graph and replayed (prior_model.py train() / test_model.py train() loop bodies):
 *   begin: dyn[0..1] = adam_table[step] ({lr / (1 - beta1^t), sqrt(1 - beta2^t)}, row clamped to the table);
 *          kl_slots[RCB_KL_SLOTS] = 0 (nullable)
 *   end  : mse_log[step] = mse_scale * sum(sse[0..n_sse)),  kl_log[step] = sum(kl_slots)  (each nullable, written
 *          only while step < n_log; fixed-order fp64 sums);  step += 1;  aux_counter (e.g. a noise counter that is not
 *          reset between train() calls) += 1                                                                        */
int rcb_step_begin(const float* adam_table, int64_t n_steps, const int64_t* step, float* dyn, int64_t* kl_slots,
                   rcb_stream_t stream);
int rcb_step_end(const float* sse, int32_t n_sse, double mse_scale, const int64_t* kl_slots, double* mse_log,
                 double* kl_log, int64_t n_log, int64_t* step, int64_t* aux_counter /* nullable: += 1 */,
                 rcb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K12: column moments for the closed-form prior refit (main_prior_training.py:157-172), as EXACT fixed-point sums.
 * For x = loc[rows, cols] and sigma = softplus(log_scale) / 6, every term v in {x, x^2, sigma^2} is split at 2^-30:
 * hi = floor(v 2^30), lo = rint((v 2^30 - hi) 2^32), and the parts are summed as 64-bit integers:
 * out_fx[q][0][j] = sum_r hi,  out_fx[q][1][j] = sum_r lo  for q = 0: x, 1: x^2, 2: sigma^2  (out_fx is int64
 * [3][2][cols] followed by ONE more element; value = (hi + lo 2^-32) 2^-30, resolution 2^-62).  Range: every term must
 * satisfy |v| < 2^12 -- i.e. |x| < 2^6 and sigma < 2^6, since x^2 and sigma^2 are terms -- for up to 2^20 rows over all
 * ranks (2^12 2^30 2^20 = 2^62).  Terms outside that range, NaN and Inf are not summed but counted in out_fx[6 cols]
 * (summed over ranks like the rest): a non-zero count means the refit must be NaN.  Integer addition is associative,
 * so the sums -- and the prior refit from them -- are bitwise independent of the order of the workgroups and, after an
 * integer all-reduce, of how the rows are sharded over ranks.  mean = sum x / n, M2 = sum x^2 - (sum x)^2 / n in fp64.
 * ------------------------------------------------------------------------------------------- */
#define RCB_MOM_FX_SCALE 1073741824.0            /* 2^30 */
#define RCB_MOM_FX_LO_SCALE 4294967296.0         /* 2^32 */
int rcb_col_moments(const float* loc, const float* log_scale, int32_t rows, int32_t cols, int64_t* out_fx,
                    rcb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K9: A* / relative-entropy-coding candidate scoring (test_model.py:501-533 sample_group, :535-584
 * h_/hh_ copies) and the commit of the chosen sample (test_model.py:586-619 compress_group), batched
 * over jobs.  Job b encodes columns [start[b], start[b]+glen[b]) of row row[b]:
 *   z_k    = p_loc + p_scale * xi_k                    (fp64: multiply, then add; xi = table of that group length)
 *   logw_k = sum_j logN(z_kj; loc, scale) - logN(z_kj; p_loc, p_scale) + gumbel[k]
 *   idx    = first argmax_k logw_k
 * loc/scale/p_loc/p_scale are fp32 (scale = softplus(log_scale)/6 precomputed by the caller).  Group lengths are
 * unbounded (the reference packs parameters until 16 bits of KL are reached, prior_model.py:301-316).
 * Candidate tables: the reference table (scrambled Sobol -> scipy norm.ppf, test_model.py:493-498) is fp32-precision
 * data in an fp64 container, so it is held as fp32, TRANSPOSED: tables_t[g] -> [g][K] (candidate index contiguous).
 * tables_t and table_absmax are DEVICE arrays of max_glen + 1 entries (NULL / 0 for unused lengths); the job arrays are
 * DEVICE arrays too, so an encode round needs no host round trip.  Jobs outside the matrix, with an unknown group
 * length or a missing table are rejected on the device: idx = -1, nothing committed.
 *
 * mode RCB_REC_EXACT: the reference arithmetic op for op (one workgroup per job).
 * mode RCB_REC_FAST : logw as a quadratic in xi (two fp64 FMAs per candidate and element, eight jobs share every table
 *   load), then a per-job certificate: if the gap between the two best fast scores exceeds a rigorous bound on
 *   |fast - exact| the exact arg-max is provably the same index; the remaining jobs (flagged in `uncertified`) are
 *   re-scored by the exact kernel inside the same call.  Indices are identical to RCB_REC_EXACT by construction.
 * Outputs: idx[n_jobs]; best[n_jobs, 2] (nullable) = (max, runner-up) log-weights; uncertified[n_jobs] (nullable);
 * logw_job0 (nullable, [K]) = all log-weights of job 0, always from the exact scorer.
 * `workspace`: rcb_rec_workspace_bytes(...) bytes, 16-byte aligned, caller-allocated.
 *
 * rcb_rec_commit: z = p_loc + p_scale * xi[idx] (fp64, mul then add) -> z_out[b, max_glen] (nullable, fp64),
 * enc_sample[row, start + j] = (float) z, enc_mask[row, start + j] = 1, and per (row, group = job_group[b]):
 * done = 1, beta = 0, idx_groupwise = idx  (each nullable).  The decoder rebuilds its samples with the same kernel.
 * ------------------------------------------------------------------------------------------- */
#define RCB_REC_EXACT 0
#define RCB_REC_FAST 1

typedef struct rcb_rec_desc {
  const float* loc;            /* [rows, cols] posterior means (group order)          (nullable for rcb_rec_commit) */
  const float* scale;          /* [rows, cols] posterior standard deviations           (nullable for rcb_rec_commit) */
  const float* p_loc;          /* [cols] */
  const float* p_scale;        /* [cols] */
  int32_t rows, cols;
  const float* const* tables_t; /* device array [max_glen + 1] of device pointers to fp32 [g][K] tables */
  const double* table_absmax;  /* device array [max_glen + 1]: max |xi| of each table  (nullable for rcb_rec_commit) */
  int32_t max_glen;
  const double* gumbel;        /* [K] fp64 (test_model.py:441-457)                      (nullable for rcb_rec_commit) */
  double gumbel_absmax;
  int32_t n_candidates;        /* K */
  const int32_t* job_row;      /* device arrays [n_jobs]; sorting the jobs by glen lets eight jobs share table loads */
  const int32_t* job_start;
  const int32_t* job_glen;
  int32_t n_jobs;
} rcb_rec_desc;

int64_t rcb_rec_workspace_bytes(int32_t n_jobs, int32_t max_glen, int32_t n_candidates);
int rcb_rec_score_argmax(const rcb_rec_desc* desc, int32_t mode, void* workspace, int64_t workspace_bytes, int32_t* idx,
                         double* best, uint8_t* uncertified, double* logw_job0, rcb_stream_t stream);
int rcb_rec_commit(const rcb_rec_desc* desc, const int32_t* idx, const int32_t* job_group, int32_t n_groups,
                   double* z_out, float* enc_sample, float* enc_mask, uint8_t* done, float* beta,
                   int32_t* idx_groupwise, rcb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * N1: the `nearest-upsample(2) -> conv3x3(pad 1)` stages of the upsampling net (prior_model.py:52-54:
 * up2/conv2/act2, up3/conv3) in sub-pixel (phase) form, bf16 MFMA / fp32 accumulate, channel-last images,
 * Cin = 64.  weff = kernel taps pre-summed per (phase, 2x2 window tap), fp32 [ty][tx][ci][a][b][co].
 *   fwd  : y[b, 2i+a, 2j+b', co] = bias[co] + sum weff[ty,tx,ci,a,b',co] * x[b, i+a+ty-1, j+b'+tx-1, ci]
 *          x: bf16 activations (x_is_f32_preact = 0), fp32 pre-activations (1) or bf16 pre-activations (2);
 *             LeakyReLU(0.01) is applied on load to pre-activations;
 *          y: bf16 with LeakyReLU applied (y_is_f32_linear = 0), fp32 linear output (1) or bf16 linear output (2).
 *   dgrad: dx = (conv^T dy) * LeakyReLU'(x)   (sign taken from the stored activation / pre-activation)
 *   wgrad: dweff = sum_b,i,j x (x) dy    (per-workgroup partial sums + fixed-order reduction)
 * Instantiated for (grid, cout) = (8, 64) [stage 2: fp32 or bf16 pre-activation in, bf16 out] and (16, 16)
 * [stage 3: bf16 in, fp32 or bf16 linear out; dy fp32 or bf16]; batch = number of images (INR x sample).
 * ------------------------------------------------------------------------------------------- */
int rcb_upconv_fwd(const void* x, int32_t x_is_f32_preact, const float* weff, const float* bias, void* y,
                   int32_t y_is_f32_linear, int32_t batch, int32_t grid, int32_t cout, const void* frag_pack,
                   rcb_stream_t stream);
/* dgrad, stage-2 geometry only: if dbias_partial != NULL it receives [rcb_upconv_dgrad_partial_rows(batch)][64]
 * per-workgroup channel sums of dx (= the bias gradient of the stage that produced x); elsewhere it must be NULL. */
int32_t rcb_upconv_dgrad_partial_rows(int32_t batch);
int rcb_upconv_dgrad(const void* dy, int32_t dy_is_f32, const float* weff, const void* x, int32_t x_is_f32_preact,
                     void* dx, float* dbias_partial, int32_t batch, int32_t grid, int32_t cout, const void* frag_pack,
                     rcb_stream_t stream);
/* wgrad writes (does not accumulate) dweff [2][2][64][2][2][cout] and, if non-NULL, dbias [cout] = sum of dy.
 * Each workgroup sums its INRs into its own slab of `workspace` and a second kernel adds the slabs in a fixed
 * order: no atomics, bitwise reproducible.  workspace: >= rcb_upconv_wgrad_workspace(batch, cout) floats.   */
int64_t rcb_upconv_wgrad_workspace(int32_t batch, int32_t cout);
int rcb_upconv_wgrad(const void* x, int32_t x_is_f32_preact, const void* dy, int32_t dy_is_f32, float* dweff,
                     float* dbias, int32_t batch, int32_t grid, int32_t cout, float* workspace,
                     int64_t workspace_floats, rcb_stream_t stream);
/* Stage-3 geometry (grid 16, cout 16, bf16 tensors): dgrad and wgrad of one stage in ONE pass over dy and x (each is
 * read once instead of twice).  dx as rcb_upconv_dgrad, dweff / dbias as rcb_upconv_wgrad (same workspace).          */
int rcb_upconv_bwd_fused(const void* dy, const float* weff, const void* x, void* dx, float* dweff, float* dbias,
                         int32_t batch, int32_t grid, int32_t cout, float* workspace, int64_t workspace_floats,
                         const void* frag_pack, rcb_stream_t stream);

/* The phase-form ("effective") weights of the CIFAR-geometry upsampling net from its conv weights, and the
 * transposed map for their gradients (one launch each; prior_model.py:39-57 defines the convolutions):
 *   weff1 [512][4096] (rows (s,t,i) of the 2x2x128 latent grid, columns (y,x,o) of the 8x8x64 stage-1 output) and
 *   b1rep [4096] in bf16 (bf16_out = 1) or fp32; weff2 [2][2][64][2][2][64], weff3 [2][2][64][2][2][16] fp32.
 *   grad: dW1 [64][128][5][5], dW2 [64][64][3][3], dW3 [16][64][3][3] written (not accumulated).               */
/* frag_pack (nullable, RCB_UPCONV_PACK_UINT4 x 16 bytes): the same effective weights as bf16 MFMA A-fragments in the
 * per-lane order the phase-conv kernels keep in registers, so that their prologue is a few coalesced 16-byte loads
 * instead of hundreds of strided scalar ones.  16-byte entries (8 bf16), lane = entry & 63, q = lane & 31, h = lane >> 5:
 *   [    0,  8192) stage-2 forward   [ph][mt][ty][tx][kb][lane]: weff2[ty][tx][16kb+8h+j][ph>>1][ph&1][32mt+q]
 *   [ 8192, 16384) stage-2 dgrad     [kh][mt][c][kb][lane], window combo n = 8kh+c: weff2[ty][tx][32mt+q][pa][pb][16kb+8h+j]
 *   [16384, 18432) stage-3 forward   [pa][pb][ty][tx][kb2][lane], A operands of v_mfma_f32_16x16x32_bf16:
 *                                    weff3[ty][tx][32kb2+8(lane>>4)+j][pa][pb][lane&15]; [18432, 20480) unused
 *   [20480, 22528) stage-3 dgrad     [n][mt][lane]: weff3[ty][tx][32mt+q][pa][pb][8h+j]
 * (ry = (n>>2)-1, rx = (n&3)-1, pa = ry&1, ty = ry<=0, pb = rx&1, tx = rx<=0.)  rcb_upconv_fwd / _dgrad take the pack
 * through their `frag_pack` argument; with NULL they build the fragments from `weff` themselves.                 */
#define RCB_UPCONV_PACK_UINT4 22528
int rcb_upconv_weff_build(const float* W1, const float* b1, const float* W2, const float* W3, void* weff1, void* b1rep,
                          int32_t bf16_out, float* weff2, float* weff3, void* frag_pack, rcb_stream_t stream);
int rcb_upconv_weff_grad(const void* dweff1, int32_t bf16_in, const float* dweff2, const float* dweff3, float* dW1,
                         float* dW2, float* dW3, const float* db1_partial /* nullable [n_partial][64] */,
                         int32_t n_partial, float* db1 /* nullable [64] = column sums of db1_partial */,
                         rcb_stream_t stream);

/* ---- overlapping tiles of a stitched grid (N1 for the patched presets; utils.py:71-116 stitches the patches of a
 * datapoint into one latent grid before the upsampling net, prior_model.py:52-54) ------------------------------------
 * The phase-conv kernels above work on small zero-halo images.  A large channel-last bf16 image [n][H][W][C] (C % 8 == 0)
 * is cut into tiles [n][Ty][Tx][T][T][C]; tile (ty, tx) holds pixels ty*step - off .. + T-1  x  tx*step - off .. + T-1.
 *   rcb_tile_gather: image -> tiles, zero outside the image and, with ring = 1, on the outermost row / column of each tile.
 *                    Source tiles of an upconv stage with grid G: T = G, step = G-1, off = 1, ring = 0 (tiles overlap by
 *                    one source pixel, so all but the outermost of a tile's 2G output rows see their whole window);
 *                    tiles of the upstream gradient: T = 2G, step = 2G-2, off = 2, ring = 1.
 *   rcb_tile_crop  : tiles -> image [n][H][W][C] from the inner T-2 rows / columns of every tile (step = T-2):
 *                    pixel y comes from tile (y+off) / (T-2), row (y+off) % (T-2) + 1.  Stage outputs: T = 2G, off = 1.
 *   rcb_tile_fold  : tiles -> image, the SUM of all tile elements that map to a pixel (step = T-1; adjoint of the ring-0
 *                    gather: border pixels belong to two tiles per axis); fp32 sum in a fixed order, rounded once to bf16.
 * Pure data movement (HBM-bound: bytes = image + tiles).                                                               */
int rcb_tile_gather(const void* img, void* tiles, int32_t n, int32_t H, int32_t W, int32_t C, int32_t Ty, int32_t Tx,
                    int32_t T, int32_t step, int32_t off, int32_t ring, rcb_stream_t stream);
int rcb_tile_crop(const void* tiles, void* img, int32_t n, int32_t H, int32_t W, int32_t C, int32_t Ty, int32_t Tx,
                  int32_t T, int32_t off, rcb_stream_t stream);
int rcb_tile_fold(const void* tiles, void* img, int32_t n, int32_t H, int32_t W, int32_t C, int32_t Ty, int32_t Tx,
                  int32_t T, int32_t off, rcb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * N1, grids of any dimension (nd = 1 audio / protein, 3 video; 2 is valid too): the `nearest-upsample(2) -> conv(3, pad 1)`
 * stages of the upsampling net (prior_model.py:52-54 with Conv1d / Conv3d) as direct sub-pixel convolutions on bf16 MFMA,
 * channel-last bf16 tensors, Cin = 64, Cout = 64 or 16 -- no window matrix.  x [B][g0][g1][g2][64] (axes k >= nd of
 * size 1), y [B][2 g...][Cout] (active axes doubled).
 *   rcb_phaseconv_pack : conv weight [Cout][64][3]^nd fp32 -> bf16 MFMA fragments of the pre-summed phase weights for the
 *                        forward and / or the data-gradient kernel (rcb_phaseconv_pack_uint4(nd, cout, which) 16-byte
 *                        units each; which 0 = forward, 1 = data gradient; either pointer may be NULL)
 *   rcb_phaseconv_fwd  : y = bias + sum over the 2^nd source taps of every phase (+ LeakyReLU(0.01) if leaky_out);
 *                        x holds ACTIVATIONS (post-LeakyReLU: the producer applies it)
 *   rcb_phaseconv_dgrad: dx = LeakyReLU'(x_act) * (transposed stage applied to dy [B][2 g...][Cout]); x_act = the stored
 *                        activations of the stage input (sign = derivative; NULL: no factor)
 *   rcb_phaseconv_wgrad: dW [Cout][64][3]^nd and dbias [Cout] (fp32) from the stage input activations and the gradient of
 *                        its linear output: contraction over all positions with both MFMA operands read transposed from
 *                        LDS images; per-workgroup fp32 slabs added in a fixed order (no atomics); `workspace`:
 *                        rcb_phaseconv_wgrad_workspace(nd, cout) floats, caller-allocated                              */
int64_t rcb_phaseconv_pack_uint4(int32_t nd, int32_t cout, int32_t which);
int rcb_phaseconv_pack(const float* conv_weight, int32_t nd, int32_t cout, void* fwd_frags, void* dgrad_frags,
                       rcb_stream_t stream);
int rcb_phaseconv_fwd(const void* x, const void* fwd_frags, const float* bias, void* y, int32_t B, int32_t g0, int32_t g1,
                      int32_t g2, int32_t nd, int32_t cout, int32_t leaky_out, rcb_stream_t stream);
int rcb_phaseconv_dgrad(const void* dy, const void* dgrad_frags, const void* x_act, void* dx, int32_t B, int32_t g0,
                        int32_t g1, int32_t g2, int32_t nd, int32_t cout, rcb_stream_t stream);
int64_t rcb_phaseconv_wgrad_workspace(int32_t nd, int32_t cout);
int rcb_phaseconv_wgrad(const void* x_act, const void* dy, float* dW, float* dbias, float* workspace, int64_t workspace_floats,
                        int32_t B, int32_t g0, int32_t g1, int32_t g2, int32_t nd, int32_t cout, rcb_stream_t stream);

/* Stage 1 of the 1-D upsampling net, direct (prior_model.py:23-51 with Conv1d: up1 (x4) -> conv1 (128 -> 64, k 5, pad 2) ->
 * act1; audio / protein presets): replaces the window-GEMM form (rcb_window_gather -> library GEMM -> LeakyReLU pass, and its
 * backward GEMMs / rcb_window_fold / casts) for nd = 1.  wbig = rcb_phase_bigweight's result for the stage, bf16 [3 * 128][4 * 64]
 * (row = tap * 128 + ci, column = phase * 64 + co; only two taps per phase are non-zero and only those are read).
 *   rcb_stage1_1d_fwd  : x [B][g][128] fp32 (rounded to bf16 as the operand) -> x1 [B][4 g][64] bf16 = LeakyReLU(bf16(bias + conv))
 *   rcb_stage1_1d_dgrad: dz [B][4 g][64] bf16 (gradient of the PRE-activation, as rcb_phaseconv_dgrad of stage 2 returns it)
 *                        -> dx [B][g][128] fp32
 *   rcb_stage1_1d_wgrad: -> dwbig [384][256] fp32 (zero where no phase reads; rcb_phase_bigweight_grad maps it onto the conv
 *                        weight) and dbias [64] fp32; per-workgroup slabs added in a fixed order (no atomics); `workspace`:
 *                        rcb_stage1_1d_wgrad_workspace() floats, caller-allocated.  All tensors 16-byte aligned.             */
int rcb_stage1_1d_fwd(const float* x, const void* wbig, const float* bias, void* x1, int32_t B, int32_t g, rcb_stream_t stream);
int rcb_stage1_1d_dgrad(const void* dz, const void* wbig, float* dx, int32_t B, int32_t g, rcb_stream_t stream);
int64_t rcb_stage1_1d_wgrad_workspace(void);
int rcb_stage1_1d_wgrad(const float* x, const void* dz, float* dwbig, float* dbias, float* workspace, int64_t workspace_floats,
                        int32_t B, int32_t g, rcb_stream_t stream);

/* 3^d-pixel windows of a channel-last bf16 grid x [B][g0][g1][g2][C] (nd = 1..3 windowed axes, unused trailing axes of size 1,
 * C % 8 == 0): the operand of the one-GEMM-per-stage phase form of the 1-D / 3-D upsampling nets (prior_model.py:23-59 with
 * Conv1d / Conv3d; every phase of the reference's stages reads inside the 3-pixel window around its source pixel).
 *   rcb_window_gather: cols [B*g0*g1*g2][3^nd][C], cols[b,p,k,:] = x[b, p + o(k) - 1, :] (0 outside), o(k) = base-3 digits of k
 *                      (axis 0 most significant).
 *   rcb_window_fold  : dx[b,p,:] = sum_k dcols[b, p - o(k) + 1, k, :] (the adjoint), fp32 sum in tap order, rounded to bf16. */
int rcb_window_gather(const void* x, void* cols, int32_t B, int32_t g0, int32_t g1, int32_t g2, int32_t C, int32_t nd,
                      rcb_stream_t stream);
int rcb_window_fold(const void* dcols, void* dx, int32_t B, int32_t g0, int32_t g1, int32_t g2, int32_t C, int32_t nd,
                    rcb_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The window-GEMM weight of a  nearest-upsample(f) -> conv(k, pad)  stage (reference prior_model.py:23-59: up1 + conv1 of the
 * upsampling net, evaluated on the 1-D / 3-D and stitched 2-D grids as one GEMM over 3^d-pixel windows, all phases at once):
 *   Wbig[(n_0..n_d-1, ci), (a_0..a_d-1, co)] = sum of W[co][ci][kk] over the taps kk with floor((a_i + kk_i - pad) / f_i) + 1 == n_i
 * rcb_phase_bigweight: `wt` = the conv weight in channel-last order [k^nd][cin][cout] (fp32) -> big [3^nd * cin][prod(f) * cout],
 * bf16 (out_bf16 = 1) or fp32.  rcb_phase_bigweight_grad: its adjoint, dbig (bf16 or fp32) -> dwt [k^nd][cin][cout] fp32.
 * `f` holds nd upsampling factors (<= 8); k <= 8; every phase must stay inside the 3-pixel window (all reference stages do). */
int rcb_phase_bigweight(const float* wt, void* big, int32_t out_bf16, int32_t nd, const int32_t* f, int32_t k, int32_t pad,
                        int32_t cin, int32_t cout, rcb_stream_t stream);
int rcb_phase_bigweight_grad(const void* dbig, int32_t in_bf16, float* dwt, int32_t nd, const int32_t* f, int32_t k, int32_t pad,
                             int32_t cin, int32_t cout, rcb_stream_t stream);

/* sigma = softplus(log_scale)/6 elementwise (prior_model.py:88).                                */
int rcb_softplus_scale(const float* log_scale, float* scale, int64_t n, rcb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RCB_H_ */
