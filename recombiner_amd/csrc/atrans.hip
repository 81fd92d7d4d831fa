// K2: the A transform of the latent weight vectors (prior_model.py:173-174, test_model.py:348-349)
//     wvec[:, lo_l:hi_l] = h_w[:, lo_l:hi_l] @ A[l]              (forward,  rcb_atrans_apply transpose = 0)
//     dh  [:, lo_l:hi_l] = dw [:, lo_l:hi_l] @ A[l]^T            (its data gradient, transpose = 1)
//     dA[l]              = h_w[:, lo_l:hi_l]^T @ dw[:, lo_l:hi_l] (its weight gradient, rcb_atrans_wgrad)
// for all layers of the INR in ONE launch each, on v_mfma_f32_32x32x16_bf16 at close to fp32 accuracy: the per-row (per-INR)
// operand enters as x = hi + lo (two bf16 terms, formed IN the kernel from the fp32 rows while they are staged into LDS --
// no split pass over HBM, the producers keep writing plain fp32), the shared mapping as bf16 (hi; + lo with terms = 3):
//     terms 1: hi A_hi      terms 2: (hi + lo) A_hi      terms 3: (hi + lo) A_hi + hi A_lo
// rcb_atrans_pack converts the fp32 mappings once per step into the two bf16 images the two directions read with the
// contraction index contiguous (forward: A^T, data gradient: A), zero-padded to multiples of 32.
//
// Decomposition (forward / data gradient).  The output of layer l is [rows, L_l]; rows are cut into 128-row tiles, the
// columns of all layers, flattened into 32-column blocks, into R contiguous runs per row tile of (nearly) equal cost, so
// that rows/128 * R workgroups fill the chip once (4096 rows: 32 * 8 = 256 workgroups, 12-13 blocks each) -- the 33
// blocks of a 1056-wide layer divide by nothing useful, and a tile grid per layer leaves 1/8 of a wave of tiles over.
// A run is cut at layer boundaries into segments of <= 13 blocks (host side: rcb_atrans_plan); per segment a 512-thread
// workgroup walks the contraction in 32-deep chunks: x rows fp32 -> registers -> hi / lo -> LDS, mapping rows bf16 ->
// registers -> LDS (64-byte rows, 16-byte chunk index XOR (row >> 2) & 3: conflict-free ds_read_b128 fragments), two LDS
// stages, one barrier per chunk, next chunk's global loads in flight during the MFMAs.  Wave (r, h) = row block r of the
// tile x column half h: two waves per SIMD (w and w + 4 share one) cover each other's LDS latency; <= 7 accumulator
// tiles per wave.  Workgroups sharing a row tile are mapped to one XCD (they re-read the same x rows through its L2).
#include "rcb_common.h"

#include <algorithm>
#include <vector>

using namespace rcb;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int BM = 128, BK = 32, NT = 512, MAXL = RCB_ATRANS_MAX_LAYERS;
constexpr int XT_BYTES = BM * BK * 2;              // one bf16 image of the x tile (8 KB)

__device__ __forceinline__ constexpr int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// byte offset of 16-byte chunk c (k = 8 c .. 8 c + 7) of row `row` in a [rows][32] bf16 image
__device__ __forceinline__ int swz(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

struct AtArgs {
  const float* x;
  float* out;
  long long ld_x, ld_out, rows;
  int n_layers;
  int L[MAXL], Lp[MAXL], off[MAXL];
  const __bf16* bh[MAXL];
  const __bf16* bl[MAXL];
  const int4* segs;            // {layer, row0, first column block, blocks}
  const int* seg_begin;        // [n_wg + 1]
  int n_wg;
};

template <int NCB, int TERMS>
struct Geo {
  static constexpr int NH0 = (NCB + 1) / 2, NH1 = NCB / 2;
  static constexpr int NBI = (NCB * 128 + NT - 1) / NT;        // mapping chunks (16 B) per thread and stage
  static constexpr int XL_OFF = XT_BYTES;
  static constexpr int BH_OFF = (TERMS >= 2 ? 2 : 1) * XT_BYTES;
  static constexpr int BT_BYTES = NCB * 32 * BK * 2;
  static constexpr int BL_OFF = BH_OFF + BT_BYTES;
  static constexpr int STAGE = BH_OFF + (TERMS == 3 ? 2 : 1) * BT_BYTES;
};

template <int NCB, int TERMS>
struct StageRegs {
  float xv[8];
  uint4 bh[Geo<NCB, TERMS>::NBI];
  uint4 bl[TERMS == 3 ? Geo<NCB, TERMS>::NBI : 1];
};

// RAGGED: the layer size is not a multiple of 8 (the output layer: 99 = 3 * 33): element-wise guarded loads; otherwise a
// group of 8 contraction indices lies inside the layer or outside it as a whole and the loads are two 16-byte vectors
// from a clamped (always readable) address, zeroed by selects -- no branch in the chunk loop.
template <int NCB, int TERMS, bool RAGGED>
__device__ __forceinline__ void stage_load(StageRegs<NCB, TERMS>& s, const float* __restrict__ xrow, bool row_ok, int K,
                                           const __bf16* __restrict__ bh, const __bf16* __restrict__ bl, int Lp, int brow0,
                                           int kc, int t) {
  typedef Geo<NCB, TERMS> G;
  const int c = t & 3;
  const int k0 = kc * BK + 8 * c;
  if (RAGGED) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s.xv[i] = (row_ok && k0 + i < K) ? xrow[k0 + i] : 0.f;
  } else {
    const bool in = row_ok && k0 < K;
    const float* __restrict__ p = xrow + min(k0, K - 8);
    const f32x4u v0 = *reinterpret_cast<const f32x4u*>(p);
    const f32x4u v1 = *reinterpret_cast<const f32x4u*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s.xv[i] = in ? v0[i] : 0.f;
      s.xv[4 + i] = in ? v1[i] : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < G::NBI; ++j) {
    // (threads past the end of the image repeat its last chunk: one broadcast load and a same-value store instead of a branch)
    const int item = (G::NBI * NT == NCB * 128) ? t + NT * j : min(t + NT * j, NCB * 128 - 1);
    const int grow = min(brow0 + (item >> 2), Lp - 1);          // (blocks past the end of a layer are computed, never stored)
    const long long o = (long long)grow * Lp + kc * BK + 8 * (item & 3);
    s.bh[j] = *reinterpret_cast<const uint4*>(bh + o);
    if (TERMS == 3) s.bl[j] = *reinterpret_cast<const uint4*>(bl + o);
  }
}

template <int NCB, int TERMS>
__device__ __forceinline__ void stage_write(const StageRegs<NCB, TERMS>& s, char* __restrict__ st, int t) {
  typedef Geo<NCB, TERMS> G;
  union {
    bf16x8 v;
    uint4 u;
  } hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)s.xv[i];
    hi.v[i] = h;
    lo.v[i] = (__bf16)(s.xv[i] - (float)h);
  }
  const int xo = swz(t >> 2, t & 3);
  *reinterpret_cast<uint4*>(st + xo) = hi.u;
  if (TERMS >= 2) *reinterpret_cast<uint4*>(st + G::XL_OFF + xo) = lo.u;
#pragma unroll
  for (int j = 0; j < G::NBI; ++j) {
    const int item = (G::NBI * NT == NCB * 128) ? t + NT * j : min(t + NT * j, NCB * 128 - 1);
    const int bo = swz(item >> 2, item & 3);
    *reinterpret_cast<uint4*>(st + G::BH_OFF + bo) = s.bh[j];
    if (TERMS == 3) *reinterpret_cast<uint4*>(st + G::BL_OFF + bo) = s.bl[j];
  }
}

// one segment: out[row0 .. row0 + 127, off_l + 32 cb0 .. + 32 ncb) of layer l
template <int NCB, int TERMS, bool RAGGED>
__device__ __forceinline__ void run_segment(const AtArgs& a, const int4 sg, char* __restrict__ lds) {
  typedef Geo<NCB, TERMS> G;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = wave & 3, h = wave >> 2;
  const int l = sg.x, row0 = sg.y, cb0 = sg.z, ncb = sg.w;
  const int K = a.L[l], Lp = a.Lp[l];
  const int nk = Lp / BK;
  const __bf16* __restrict__ bh = a.bh[l];
  const __bf16* __restrict__ bl = a.bl[l];
  const long long xr_i = (long long)row0 + (t >> 2);
  const bool row_ok = xr_i < a.rows;
  const float* __restrict__ xrow = a.x + (row_ok ? xr_i : 0) * a.ld_x + a.off[l];

  f32x16 acc[G::NH0];
#pragma unroll
  for (int i = 0; i < G::NH0; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

  StageRegs<NCB, TERMS> sr;
  stage_load<NCB, TERMS, RAGGED>(sr, xrow, row_ok, K, bh, bl, Lp, cb0 * 32, 0, t);
  stage_write<NCB, TERMS>(sr, lds, t);
  if (nk > 1) stage_load<NCB, TERMS, RAGGED>(sr, xrow, row_ok, K, bh, bl, Lp, cb0 * 32, 1, t);
  __syncthreads();

  const int frow = lane & 31, fh = lane >> 5;
  const int cbase = h ? G::NH0 : 0;
  // one 32-deep chunk: both k-steps of the wave's x fragments, then column block by column block (four dependent MFMAs
  // per accumulator: back-to-back issue on one accumulation chain runs at the full rate); the block only the first
  // column half owns comes last, behind the one wave-uniform branch of the chunk
  auto compute = [&](const char* __restrict__ cur) {
    const int xo0 = swz(32 * r + frow, fh), xo1 = swz(32 * r + frow, 2 + fh);
    const bf16x8 xh0 = *reinterpret_cast<const bf16x8*>(cur + xo0);
    const bf16x8 xh1 = *reinterpret_cast<const bf16x8*>(cur + xo1);
    bf16x8 xl0, xl1;
    if (TERMS >= 2) {
      xl0 = *reinterpret_cast<const bf16x8*>(cur + G::XL_OFF + xo0);
      xl1 = *reinterpret_cast<const bf16x8*>(cur + G::XL_OFF + xo1);
    }
#pragma unroll
    for (int cb = 0; cb < G::NH0; ++cb) {
      if (G::NH1 == G::NH0 || cb < G::NH1 || h == 0) {
        const int bo0 = swz((cbase + cb) * 32 + frow, fh), bo1 = swz((cbase + cb) * 32 + frow, 2 + fh);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(cur + G::BH_OFF + bo0);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(cur + G::BH_OFF + bo1);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh0, b0, acc[cb], 0, 0, 0);
        if (TERMS >= 2) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl0, b0, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh1, b1, acc[cb], 0, 0, 0);
        if (TERMS >= 2) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl1, b1, acc[cb], 0, 0, 0);
        if (TERMS == 3) {
          const bf16x8 c0 = *reinterpret_cast<const bf16x8*>(cur + G::BL_OFF + bo0);
          const bf16x8 c1 = *reinterpret_cast<const bf16x8*>(cur + G::BL_OFF + bo1);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh0, c0, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh1, c1, acc[cb], 0, 0, 0);
        }
      }
    }
  };
  // steady state without a branch inside: convert + store the chunk loaded during the previous iteration, request the one
  // after it, multiply the current one; the last two chunks are peeled
  int kc = 0;
  for (; kc + 2 < nk; ++kc) {
    stage_write<NCB, TERMS>(sr, lds + ((kc + 1) & 1) * G::STAGE, t);
    stage_load<NCB, TERMS, RAGGED>(sr, xrow, row_ok, K, bh, bl, Lp, cb0 * 32, kc + 2, t);
    compute(lds + (kc & 1) * G::STAGE);
    __syncthreads();
  }
  if (kc + 1 < nk) {
    stage_write<NCB, TERMS>(sr, lds + ((kc + 1) & 1) * G::STAGE, t);
    compute(lds + (kc & 1) * G::STAGE);
    __syncthreads();
    ++kc;
  }
  compute(lds + (kc & 1) * G::STAGE);
  __syncthreads();

  // epilogue: accumulator register q of lane (col, fh) is row rho(q, fh) of the wave's 32-row block
  const int nmine = h ? G::NH1 : G::NH0;
  float* __restrict__ ob = a.out + a.off[l];
#pragma unroll
  for (int cb = 0; cb < G::NH0; ++cb) {
    if (cb < nmine && cbase + cb < ncb) {
      const int col = (cb0 + cbase + cb) * 32 + frow;
      if (col < K) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const long long row = (long long)row0 + 32 * r + rho(q, fh);
          if (row < a.rows) ob[row * a.ld_out + col] = acc[cb][q];
        }
      }
    }
  }
}

template <int TERMS>
__global__ void __launch_bounds__(NT, 2) atrans_kernel(AtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // workgroups that share a row tile (consecutive logical ids) onto one XCD: blocks b, b + 8, ... share an XCD
  int id = blockIdx.x;
  if ((a.n_wg & 7) == 0) id = (id & 7) * (a.n_wg >> 3) + (id >> 3);
  const int s0 = a.seg_begin[id], s1 = a.seg_begin[id + 1];
  for (int s = s0; s < s1; ++s) {
    const int4 sg = a.segs[s];
    const int ncb = sg.w;
    if (a.L[sg.x] & 7) {           // (the planner cuts such layers into segments of <= 4 blocks)
      if (ncb > 2) run_segment<4, TERMS, true>(a, sg, lds);
      else run_segment<2, TERMS, true>(a, sg, lds);
    } else if (ncb > 12) run_segment<13, TERMS, false>(a, sg, lds);
    else if (ncb > 10) run_segment<12, TERMS, false>(a, sg, lds);
    else if (ncb > 8) run_segment<10, TERMS, false>(a, sg, lds);
    else if (ncb > 6) run_segment<8, TERMS, false>(a, sg, lds);
    else if (ncb > 4) run_segment<6, TERMS, false>(a, sg, lds);
    else if (ncb > 2) run_segment<4, TERMS, false>(a, sg, lds);
    else run_segment<2, TERMS, false>(a, sg, lds);
  }
}

// ---- packed images of the mappings ------------------------------------------------------------------------------------
struct PackArgs {
  const float* A[MAXL];
  int L[MAXL], Lp[MAXL], tile0[MAXL + 1];     // first 32 x 32 tile of every layer
  long long poff[MAXL];                       // element offset of the layer inside one plane
  long long plane;                            // elements of one plane
  __bf16* out;
  int n_layers, want_lo;
};

__global__ void __launch_bounds__(256) atrans_pack_kernel(PackArgs a) {
  __shared__ float tile[32][33];
  int l = 0;
  while (l + 1 < a.n_layers && (int)blockIdx.x >= a.tile0[l + 1]) ++l;
  const int L = a.L[l], Lp = a.Lp[l], nt = Lp >> 5;
  const int tl = blockIdx.x - a.tile0[l];
  const int k0 = (tl / nt) * 32, j0 = (tl % nt) * 32;
  const float* __restrict__ A = a.A[l];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  __bf16* __restrict__ fwd_hi = a.out + a.poff[l];
  __bf16* __restrict__ dg_hi = a.out + a.plane + a.poff[l];
  __bf16* __restrict__ fwd_lo = a.out + 2 * a.plane + a.poff[l];
  __bf16* __restrict__ dg_lo = a.out + 3 * a.plane + a.poff[l];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + 8 * i, j = j0 + tx;
    const float v = (k < L && j < L) ? A[(long long)k * L + j] : 0.f;
    tile[ty + 8 * i][tx] = v;
    const __bf16 hb = (__bf16)v;
    dg_hi[(long long)k * Lp + j] = hb;                                   // data gradient: rows k, contraction j contiguous
    if (a.want_lo) dg_lo[(long long)k * Lp + j] = (__bf16)(v - (float)hb);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = j0 + ty + 8 * i, k = k0 + tx;
    const float v = tile[tx][ty + 8 * i];
    const __bf16 hb = (__bf16)v;
    fwd_hi[(long long)j * Lp + k] = hb;                                  // forward: rows j (output column), contraction k contiguous
    if (a.want_lo) fwd_lo[(long long)j * Lp + k] = (__bf16)(v - (float)hb);
  }
}

inline int pad32(int v) { return (v + 31) / 32 * 32; }

// instantiated segment widths: a segment of n column blocks costs as much as the next width >= n
inline int variant_blocks(int n) { return n > 12 ? 13 : (n > 10 ? 12 : (n > 8 ? 10 : (n > 6 ? 8 : (n > 4 ? 6 : (n > 2 ? 4 : 2))))); }

struct Blk {
  int layer, cb, w, maxseg;      // column block cb of the layer; w = contraction chunks (cost of one block); longest segment
};

// cost of blocks [b0, b1) as one run: cut at layer boundaries, pieces of more than 13 blocks in near-equal parts
long long run_cost(const std::vector<Blk>& blks, int b0, int b1, std::vector<int4>* segs, int row0) {
  long long cost = 0;
  int i = b0;
  while (i < b1) {
    int j = i;
    while (j < b1 && blks[j].layer == blks[i].layer) ++j;
    const int n = j - i, ms = blks[i].maxseg, parts = (n + ms - 1) / ms;
    int done = 0;
    for (int p = 0; p < parts; ++p) {
      const int m = (n - done + (parts - p) - 1) / (parts - p);
      cost += (long long)variant_blocks(m) * blks[i].w + 8;        // + 8: prologue / epilogue of a segment, in chunk units
      if (segs) segs->push_back(make_int4(blks[i].layer, row0, blks[i + done].cb, m));
      done += m;
    }
    i = j;
  }
  return cost;
}

}  // namespace

extern "C" int64_t rcb_atrans_pack_elems(int32_t n_layers, const int32_t* sizes) {
  if (!sizes || n_layers < 1 || n_layers > MAXL) return -1;
  int64_t plane = 0;
  for (int l = 0; l < n_layers; ++l) plane += (int64_t)pad32(sizes[l]) * pad32(sizes[l]);
  return 4 * plane;
}

extern "C" int rcb_atrans_pack(const float* const* A, int32_t n_layers, const int32_t* sizes, void* packed, int32_t want_lo,
                               rcb_stream_t stream) {
  RCB_REQUIRE(A && sizes && packed && n_layers >= 1 && n_layers <= MAXL, RCB_ERR_ARG, "atrans_pack: bad arguments (%d layers)", n_layers);
  RCB_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, RCB_ERR_ARG, "atrans_pack: packed buffer must be 16-byte aligned");
  PackArgs a;
  memset(&a, 0, sizeof(a));
  long long plane = 0;
  int tiles = 0;
  for (int l = 0; l < n_layers; ++l) {
    RCB_REQUIRE(A[l] && sizes[l] >= 1, RCB_ERR_ARG, "atrans_pack: layer %d null / empty", l);
    a.A[l] = A[l];
    a.L[l] = sizes[l];
    a.Lp[l] = pad32(sizes[l]);
    a.poff[l] = plane;
    a.tile0[l] = tiles;
    plane += (long long)a.Lp[l] * a.Lp[l];
    tiles += (a.Lp[l] / 32) * (a.Lp[l] / 32);
  }
  a.tile0[n_layers] = tiles;
  a.plane = plane;
  a.out = reinterpret_cast<__bf16*>(packed);
  a.n_layers = n_layers;
  a.want_lo = want_lo;
  atrans_pack_kernel<<<tiles, 256, 0, (hipStream_t)stream>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_atrans_plan(int64_t rows, int32_t n_layers, const int32_t* sizes, int32_t n_cu, int32_t* plan,
                               int32_t max_ints) {
  RCB_REQUIRE(sizes && plan && n_layers >= 1 && n_layers <= MAXL && rows >= 1 && n_cu >= 1, RCB_ERR_ARG,
              "atrans_plan: bad arguments");
  std::vector<Blk> blks;
  for (int l = 0; l < n_layers; ++l) {
    RCB_REQUIRE(sizes[l] >= 1, RCB_ERR_ARG, "atrans_plan: layer %d empty", l);
    const int lp = pad32(sizes[l]);
    for (int cb = 0; cb < lp / 32; ++cb) blks.push_back(Blk{l, cb, lp / BK, (sizes[l] & 7) ? 4 : 13});
  }
  const int nb = (int)blks.size();
  const long long m_tiles = (rows + BM - 1) / BM;
  int R = (int)std::max<long long>(1, n_cu / m_tiles);
  R = std::min(R, (nb + 1) / 2);
  // contiguous runs minimising the largest cost: binary search on the bound, greedy maximal runs
  long long lo = 0, hi = run_cost(blks, 0, nb, nullptr, 0);
  auto fits = [&](long long bound, std::vector<int>* cuts) {
    int b = 0, runs = 0;
    while (b < nb) {
      int e = b + 1;
      if (run_cost(blks, b, e, nullptr, 0) > bound) return false;
      while (e < nb && run_cost(blks, b, e + 1, nullptr, 0) <= bound) ++e;
      if (cuts) cuts->push_back(e);
      b = e;
      if (++runs > R) return false;
    }
    return true;
  };
  while (lo < hi) {
    const long long mid = (lo + hi) / 2;
    if (fits(mid, nullptr)) hi = mid; else lo = mid + 1;
  }
  std::vector<int> cuts;
  fits(hi, &cuts);
  const int runs = (int)cuts.size();
  const long long n_wg = m_tiles * runs;
  RCB_REQUIRE(n_wg < (1 << 24), RCB_ERR_SHAPE, "atrans_plan: %lld workgroups", n_wg);
  std::vector<int4> segs;
  std::vector<int> begin;
  for (long long mt = 0; mt < m_tiles; ++mt) {
    int b = 0;
    for (int ri = 0; ri < runs; ++ri) {
      begin.push_back((int)segs.size());
      run_cost(blks, b, cuts[ri], &segs, (int)(mt * BM));
      b = cuts[ri];
    }
  }
  begin.push_back((int)segs.size());
  // layout: [n_wg, n_segs, seg_begin (n_wg + 1, padded to a multiple of 4 ints from the start), segs (4 ints each)]
  const long long head = ((2 + n_wg + 1) + 3) / 4 * 4;
  const long long need = head + 4 * (long long)segs.size();
  if (need > max_ints) return rcb::fail(RCB_ERR_SHAPE, "atrans_plan: %lld ints needed, %d given", need, max_ints);
  memset(plan, 0, sizeof(int32_t) * head);
  plan[0] = (int32_t)n_wg;
  plan[1] = (int32_t)segs.size();
  for (size_t i = 0; i < begin.size(); ++i) plan[2 + i] = begin[i];
  memcpy(plan + head, segs.data(), sizeof(int4) * segs.size());
  return (int)need;
}

extern "C" int rcb_atrans_apply(const float* x, int64_t ld_x, float* out, int64_t ld_out, int64_t rows, int32_t n_layers,
                                const int32_t* sizes, const void* packed, int32_t transpose, int32_t terms,
                                const int32_t* plan_dev, int32_t n_wg, rcb_stream_t stream) {
  RCB_REQUIRE(x && out && sizes && packed && plan_dev && n_layers >= 1 && n_layers <= MAXL && rows >= 1 && n_wg >= 1, RCB_ERR_ARG,
              "atrans_apply: bad arguments");
  RCB_REQUIRE(terms >= 1 && terms <= 3, RCB_ERR_ARG, "atrans_apply: terms = %d (1..3)", terms);
  RCB_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(plan_dev) & 15) == 0, RCB_ERR_ARG,
              "atrans_apply: packed images / plan must be 16-byte aligned");
  AtArgs a;
  memset(&a, 0, sizeof(a));
  long long plane = 0, off = 0;
  for (int l = 0; l < n_layers; ++l) plane += (long long)pad32(sizes[l]) * pad32(sizes[l]);
  const __bf16* pk = reinterpret_cast<const __bf16*>(packed);
  long long poff = 0;
  for (int l = 0; l < n_layers; ++l) {
    a.L[l] = sizes[l];
    a.Lp[l] = pad32(sizes[l]);
    a.off[l] = (int)off;
    a.bh[l] = pk + (transpose ? plane : 0) + poff;
    a.bl[l] = pk + (transpose ? 3 : 2) * plane + poff;
    off += sizes[l];
    poff += (long long)a.Lp[l] * a.Lp[l];
  }
  RCB_REQUIRE(ld_x >= off && ld_out >= off, RCB_ERR_SHAPE, "atrans_apply: row strides %lld / %lld below %lld columns", (long long)ld_x,
              (long long)ld_out, off);
  a.x = x; a.out = out; a.ld_x = ld_x; a.ld_out = ld_out; a.rows = rows; a.n_layers = n_layers;
  const long long head = ((2 + (long long)n_wg + 1) + 3) / 4 * 4;
  a.seg_begin = plan_dev + 2;
  a.segs = reinterpret_cast<const int4*>(plan_dev + head);
  a.n_wg = n_wg;
  const int lds = 2 * (terms == 1 ? Geo<13, 1>::STAGE : (terms == 2 ? Geo<13, 2>::STAGE : Geo<13, 3>::STAGE));
  hipError_t e;
  if (terms == 1) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(atrans_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) atrans_kernel<1><<<n_wg, NT, lds, (hipStream_t)stream>>>(a);
  } else if (terms == 2) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(atrans_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) atrans_kernel<2><<<n_wg, NT, lds, (hipStream_t)stream>>>(a);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(atrans_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) atrans_kernel<3><<<n_wg, NT, lds, (hipStream_t)stream>>>(a);
  }
  RCB_REQUIRE(e == hipSuccess, (int)e, "atrans_apply: hipFuncSetAttribute: %s", hipGetErrorString(e));
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
