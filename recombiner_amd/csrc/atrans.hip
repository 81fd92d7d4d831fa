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
// Decomposition (forward / data gradient).  The output of layer l is [rows, L_l]; rows are cut into 256-row tiles, the
// columns of all layers, flattened into 32-column blocks, into R contiguous runs per row tile of (nearly) equal cost, so
// that rows/256 * R workgroups fill the chip once (4096 rows: 16 * 16 = 256 workgroups of 6-7 blocks) -- the 33 blocks
// of a 1056-wide layer divide by nothing useful, and a tile grid per layer leaves 1/8 of a wave of tiles over.  A run is
// cut at layer boundaries into segments of <= 7 blocks (host side: rcb_atrans_plan).  Per segment a 512-thread workgroup
// walks the contraction in 32-deep chunks through a ring of three LDS stages filled by LDS-DMA (global_load_lds_dwordx4:
// no staging registers, no ds_write): the x rows as they are (fp32, 128-byte rows; every wave fetches the 32 rows it
// multiplies), the mapping rows as bf16 (64-byte rows); the 16-byte chunk index is XORed with row bits ON THE SOURCE
// ADDRESS (the DMA writes LDS linearly) so that the fragment reads (ds_read_b128) are conflict-free.  One barrier per
// chunk, counted vmcnt: the DMAs of the next two chunks stay in flight across it.  A wave owns one 32-row block and all
// <= 7 column blocks: it reads its x fragment as fp32, forms hi / lo in registers (each row is converted once per
// workgroup) and issues <= 28 MFMAs per chunk on <= 7 accumulator tiles; two waves per SIMD cover each other's LDS latency.
// Workgroups sharing a row tile are mapped to one XCD (they re-read the same x rows through its L2).
#include "rcb_common.h"

#include <algorithm>
#include <type_traits>
#include <vector>

using namespace rcb;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// diagnostic builds only (python -m recombiner_amd.build --variant ... -DATRANS_DIAG=n): 1 = no output stores, 2 = every
// chunk re-reads chunk 0 (cache-hot sources), 3 = no compute at all, 4 = fragment reads + conversion but no MFMAs, 6 = no DMAs
// in the chunk loop; never defined in the library
#ifndef ATRANS_DIAG
#define ATRANS_DIAG 0
#endif
#ifndef ATRANS_STAMPS
#define ATRANS_STAMPS 0      // diagnostic build: s_memtime stamps of workgroup 0 (rcb_debug_atrans_stamps)
#endif
#ifndef ATRANS_SPREAD_PIN
#define ATRANS_SPREAD_PIN 1
#endif
constexpr int BM = 256, BK = 32, NT = 512, MAXL = RCB_ATRANS_MAX_LAYERS, MAXSEG = 7;
constexpr int XT_BYTES = BM * BK * 4;              // fp32 image of the x tile (32 KB)

__device__ __forceinline__ constexpr int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// x image: [256 rows][32 fp32] = 128-byte rows of eight 16-byte chunks; chunk c of row `row` at
__device__ __forceinline__ int xswz(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
// mapping image: [rows][32 bf16] = 64-byte rows of four chunks
__device__ __forceinline__ int bswz(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void dma16(const void* g, char* l) {          // 64 lanes x 16 B -> l + 16 lane (l wave-uniform)
  __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)l, 16, 0, 0);
}
__device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) {
#if ATRANS_DIAG == 4
  asm volatile("" ::"v"(a), "v"(b));
  return c;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#if ATRANS_STAMPS
__device__ unsigned long long g_stamps[8][40][6];
#define STAMP(it, pt)                                                                 \
  do {                                                                               \
    if (blockIdx.x == 0 && (it) < 40) {                                              \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                    \
      if (lane == 0) g_stamps[wave][it][pt] = t_;                                    \
    }                                                                                \
  } while (0)
#else
#define STAMP(it, pt)
#endif

struct AtArgs {
  const float* x;
  const __bf16* xh;            // PLANES form: the per-row operand as two bf16 planes, x = hi + lo, written by its producer
  const __bf16* xl;            //   (rcb_reparam_rng_fwd / rcb_posterior_bwd next sample / rcb_siren_desc.dw_bf16 + dw_lo)
  long long ld_x16;            //   elements between the rows of a plane (a multiple of 8)
  float* out;
  long long ld_x, ld_out, rows;
  int n_layers;
  int L[MAXL], Lp[MAXL], off[MAXL];
  const __bf16* bh[MAXL];
  const __bf16* bl[MAXL];
  const int4* segs;            // two per segment: {layer, row0, first column block, blocks}, {first chunk, end chunk, slab or -1, 0}
  const int* seg_begin;        // [n_wg + 1]
  int n_wg;
  int nt_out;                  // the result is not read again soon (data gradient: consumed by the step's last kernel)
  float* ws;                   // K-split launches (few rows): partial sums [slab][rows][ld_ws], added in slab order afterwards
  long long ld_ws;
};

template <int NCB, int TERMS>
struct Geo {
  static constexpr int NBW = (2 * NCB + 7) / 8;              // mapping DMAs per wave and stage (one DMA = 16 rows x 64 B)
  static constexpr int NDMA = 4 + NBW * (TERMS == 3 ? 2 : 1);  // all DMAs per wave and stage
  static constexpr int BH_OFF = XT_BYTES;
  static constexpr int BT_BYTES = NCB * 32 * BK * 2;
  static constexpr int BL_OFF = BH_OFF + BT_BYTES;
  static constexpr int STAGE = BH_OFF + (TERMS == 3 ? 2 : 1) * BT_BYTES;
  static constexpr int NSTAGE = TERMS == 3 ? 2 : 3;
};
constexpr int LDS_MAX_BYTES = 3 * (XT_BYTES + MAXSEG * 32 * BK * 2);        // = 2 stages of the three-term form (138 KB)

// one segment: out[row0 .. row0 + 255, off_l + 32 cb0 .. + 32 ncb) of layer l
// PLANES: x arrives as the bf16 planes hi / lo instead of fp32 rows.  The x stage then holds two [256 rows][32 bf16] images
// (64-byte rows, swizzled like the mapping image: their fragment reads are the mapping's conflict-free pattern), filled by
// the same four DMAs per wave (two per plane, 16 rows each), and the fragments are READ as operand bits: the ~6 VALU
// instructions per element of the in-register split -- about as long as the matrix work of a chunk -- are gone, and with
// them the registers that spilled.  Same products in the same order as the fp32 form: bit-identical results.
template <int NCB, int TERMS, bool RAGGED, bool PLANES>
__device__ __forceinline__ void run_segment(const AtArgs& a, const int4 sg, const int4 sk, char* __restrict__ lds) {
  typedef Geo<NCB, TERMS> G;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l = sg.x, row0 = sg.y, cb0 = sg.z, ncb = sg.w;
  const int K = a.L[l], Lp = a.Lp[l];
  const int kc0 = sk.x, nk = sk.y - sk.x;          // this segment's slice of the contraction (all of it unless the launch is K-split)
  const __bf16* __restrict__ bh = a.bh[l];
  const __bf16* __restrict__ bl = a.bl[l];
  const float* __restrict__ xl0 = PLANES ? nullptr : a.x + a.off[l];

  f32x16 acc[NCB];
#pragma unroll
  for (int i = 0; i < NCB; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

  // ---- DMA sources of this lane (fixed over the chunks but for the contraction offset) ---------------------------------------
  // x: DMA i of wave w fills rows 32 w + 8 i .. + 7 of the tile (8 lanes per 128-byte row); lane -> (row, physical chunk),
  //    it fetches the logical chunk that belongs there.  Rows past the end repeat the last row (computed, never stored).
  //    PLANES: DMA i fills rows 32 w + 16 (i & 1) .. + 15 of plane i >> 1 (4 lanes per 64-byte row)
  const float* xsrc[4];
  const __bf16* psrc[4];
  int xk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (PLANES) {
      const int row = 32 * wave + 16 * (i & 1) + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      const long long gr = min((long long)row0 + row, a.rows - 1);
      psrc[i] = ((i >> 1) ? a.xl : a.xh) + gr * a.ld_x16 + a.off[l];
      xk[i] = 8 * c;
      xsrc[i] = nullptr;
    } else {
      const int row = 32 * wave + 8 * i + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      const long long gr = min((long long)row0 + row, a.rows - 1);
      xsrc[i] = xl0 + gr * a.ld_x;
      xk[i] = 4 * c;
      psrc[i] = nullptr;
    }
  }
  // mapping: DMA j of wave w fills rows 16 (w + 8 j) .. + 15 of the image (4 lanes per 64-byte row); images shorter than
  // 8 NBW DMAs: the surplus DMAs repeat the last one (same bytes to the same place)
  const __bf16* bsrc[G::NBW];
  const __bf16* bsrc2[G::NBW];
  int bdst[G::NBW];
#pragma unroll
  for (int j = 0; j < G::NBW; ++j) {
    const int d = min(wave + 8 * j, 2 * NCB - 1);
    const int row = 16 * d + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    const int grow = min(cb0 * 32 + row, Lp - 1);                 // (blocks past the end of a layer are computed, never stored)
    bsrc[j] = bh + (long long)grow * Lp + 8 * c;
    bsrc2[j] = bl + (long long)grow * Lp + 8 * c;
    bdst[j] = __builtin_amdgcn_readfirstlane(d * 1024);
  }
  // DMA i (0 .. NDMA-1) of chunk kc: the wave's four x pieces, then its mapping pieces
  auto issue_one = [&](int kc_, int i) {
    char* __restrict__ st = lds + (kc_ % G::NSTAGE) * G::STAGE;
    const int kc = ATRANS_DIAG == 2 ? 0 : kc0 + kc_;
    if (i < 4) {
      // chunks past the end of the layer (its padding to a multiple of 32) meet zero rows of the mapping: any finite
      // values do, so they repeat the layer's last chunk instead of reading past the row
      if (PLANES) {
        const int k = RAGGED ? 0 : min(kc * BK + xk[i], K - 8);
        dma16(psrc[i] + k, st + (i >> 1) * (XT_BYTES / 2) + (32 * wave + 16 * (i & 1)) * 64);
      } else {
        const int k = RAGGED ? 0 : min(kc * BK + xk[i], K - 4);
        dma16(xsrc[i] + k, st + (32 * wave + 8 * i) * 128);
      }
    } else if (TERMS == 3) {
      const int j = (i - 4) >> 1;
      if ((i - 4) & 1) dma16(bsrc2[j] + kc * BK, st + G::BL_OFF + bdst[j]);
      else dma16(bsrc[j] + kc * BK, st + G::BH_OFF + bdst[j]);
    } else {
      dma16(bsrc[i - 4] + kc * BK, st + G::BH_OFF + bdst[i - 4]);
    }
  };
  auto issue = [&](int kc) {
#pragma unroll
    for (int i = 0; i < G::NDMA; ++i) issue_one(kc, i);
  };

  const int frow = lane & 31, fh = lane >> 5;
  const int xrow = 32 * wave + frow;
  // the wave's two x fragments (k-steps 0, 1) of a chunk: fp32 from its own rows of the stage, split hi + lo in registers
  // (PLANES: they are read as they are -- raw[] then holds the NEXT chunk's four fragments until the current chunk's last
  // MFMA has been issued, and x_convert is a register move)
  bf16x8 xh[2], xlo[2];
  f32x4 raw[4];
  auto x_read = [&](const char* __restrict__ st) {
    if (PLANES) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (TERMS >= 2 || q < 2)
          raw[q] = *reinterpret_cast<const f32x4*>(st + (q >> 1) * (XT_BYTES / 2) + bswz(xrow, 2 * (q & 1) + fh));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) raw[q] = *reinterpret_cast<const f32x4*>(st + xswz(xrow, 4 * (q >> 1) + 2 * fh + (q & 1)));
    }
  };
  auto x_convert = [&](int q) {                 // quarter q: elements 4 (q & 1) .. + 3 of k-step q >> 1
    if (PLANES) {                               // q = 0, 1: hi fragments of k-steps 0, 1;  q = 2, 3: lo fragments
      if (q < 2) xh[q] = __builtin_bit_cast(bf16x8, raw[q]);
      else if (TERMS >= 2) xlo[q - 2] = __builtin_bit_cast(bf16x8, raw[q]);
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = raw[q][i];
      const __bf16 hb = (__bf16)v;
      xh[q >> 1][4 * (q & 1) + i] = hb;
      if (TERMS >= 2) xlo[q >> 1][4 * (q & 1) + i] = (__bf16)(v - (float)hb);
    }
  };
  // One 32-deep chunk on the fragments in xh / xlo.  ISSUE: the DMAs of chunk kc + AHEAD go out BETWEEN the column
  // blocks' MFMAs (a DMA costs its wave tens of cycles of issue: in one burst behind the barrier every wave of the
  // workgroup stalls at once and the matrix pipes run dry).  PREF: the x fragments of chunk kc + 1 are fetched and
  // converted in the second half of the blocks, so that the next chunk starts on its MFMAs instead of ~60 conversion
  // instructions with nothing to overlap them; the x rows of a wave are filled by its own DMAs, so its counted vmcnt
  // alone (no barrier) makes them readable.
  constexpr int NBT = G::NDMA - 4;                                 // mapping DMAs per wave and stage
  constexpr int CBR = (NCB - 1) / 2;                               // block behind which the next x fragments are requested
  auto compute = [&](const char* __restrict__ cur, const char* __restrict__ nxt, int kn, auto issue_c, auto pref_c) {
    constexpr bool ISSUE = decltype(issue_c)::value, PREF = decltype(pref_c)::value;
    bf16x8 b0 = *reinterpret_cast<const bf16x8*>(cur + G::BH_OFF + bswz(frow, fh));
    bf16x8 b1 = *reinterpret_cast<const bf16x8*>(cur + G::BH_OFF + bswz(frow, 2 + fh));
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const bf16x8 c0 = b0, c1 = b1;
      if (cb + 1 < NCB) {                                           // next block's mapping fragments: one block ahead
        b0 = *reinterpret_cast<const bf16x8*>(cur + G::BH_OFF + bswz((cb + 1) * 32 + frow, fh));
        b1 = *reinterpret_cast<const bf16x8*>(cur + G::BH_OFF + bswz((cb + 1) * 32 + frow, 2 + fh));
      }
      acc[cb] = mma(xh[0], c0, acc[cb]);
      if (TERMS >= 2) acc[cb] = mma(xlo[0], c0, acc[cb]);
      acc[cb] = mma(xh[1], c1, acc[cb]);
      if (TERMS >= 2) acc[cb] = mma(xlo[1], c1, acc[cb]);
      if (TERMS == 3) {
        const int bo0 = bswz(cb * 32 + frow, fh), bo1 = bswz(cb * 32 + frow, 2 + fh);
        const bf16x8 d0 = *reinterpret_cast<const bf16x8*>(cur + G::BL_OFF + bo0);
        const bf16x8 d1 = *reinterpret_cast<const bf16x8*>(cur + G::BL_OFF + bo1);
        acc[cb] = mma(xh[0], d0, acc[cb]);
        acc[cb] = mma(xh[1], d1, acc[cb]);
      }
      if (ISSUE && ATRANS_DIAG != 6) {
#pragma unroll
        for (int i = cb * G::NDMA / NCB; i < (cb + 1) * G::NDMA / NCB; ++i) issue_one(kn, i);
      }
      if (PREF) {
        if (cb == CBR) {
          STAMP(kn - 2, 3);
          // outstanding, oldest first: the next stage's x pieces, its mapping pieces, this iteration's DMAs so far
          wait_vm<NBT + (ISSUE ? (CBR + 1) * G::NDMA / NCB : 0)>();
          x_read(nxt);
        }
      }
#if ATRANS_SPREAD_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (PREF) {
      STAMP(kn - 2, 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) x_convert(q);
    }
  };
  auto yes = std::integral_constant<bool, true>{};
  auto no = std::integral_constant<bool, false>{};

  if (RAGGED) {
    // layer size not a multiple of 8 (the output layer: 99 = 3 * 33): a 16-byte chunk may straddle the end of the layer
    // and of the row, so x goes through registers with element-wise guards (zero fill), synchronously; the mapping by DMA.
    // A few short chunks per step: speed does not matter here.
    for (int kr = 0; kr < nk; ++kr) {
      const int kc = kc0 + kr;
      __syncthreads();                                       // every wave is done with the stage
#pragma unroll
      for (int j = 0; j < G::NBW; ++j) {
        dma16(bsrc[j] + kc * BK, lds + G::BH_OFF + bdst[j]);
        if (TERMS == 3) dma16(bsrc2[j] + kc * BK, lds + G::BL_OFF + bdst[j]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k0 = kc * BK + xk[i];
        if (PLANES) {
          const int row = 32 * wave + 16 * (i & 1) + (lane >> 2), pc = lane & 3;
          const bool rok = (long long)row0 + row < a.rows;
          bf16x8 v;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (rok && k0 + e < K) ? psrc[i][k0 + e] : (__bf16)0.f;
          *reinterpret_cast<bf16x8*>(lds + (i >> 1) * (XT_BYTES / 2) + row * 64 + pc * 16) = v;
        } else {
          const int row = 32 * wave + 8 * i + (lane >> 3), pc = lane & 7;
          const bool rok = (long long)row0 + row < a.rows;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (rok && k0 + e < K) ? xsrc[i][k0 + e] : 0.f;
          *reinterpret_cast<f32x4*>(lds + row * 128 + pc * 16) = v;
        }
      }
      wait_vm<0>();
      __syncthreads();
      x_read(lds);
#pragma unroll
      for (int q = 0; q < 4; ++q) x_convert(q);
      compute(lds, lds, -1, no, no);
    }
    __syncthreads();
  } else if (G::NSTAGE == 2) {
    // two stages (three-term form): chunk kc + 1 is requested while chunk kc is multiplied
    issue(0);
    for (int kc = 0; kc < nk; ++kc) {
      wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const char* cur = lds + (kc & 1) * G::STAGE;
      x_read(cur);
#pragma unroll
      for (int q = 0; q < 4; ++q) x_convert(q);
      if (kc + 1 < nk) compute(cur, cur, kc + 1, yes, no);
      else compute(cur, cur, -1, no, no);
    }
    __builtin_amdgcn_s_barrier();
  } else {
    // ring of three stages: chunk kc + 2 is requested during iteration kc, behind its barrier (every wave has then
    // finished reading the stage it overwrites, chunk kc - 1); a wave waits for ITS DMAs of chunk kc (all but the NDMA
    // youngest) before that barrier, so behind it the whole stage has landed.
    // ring of three stages: chunk kc + 2 is requested during iteration kc, behind its barrier (every wave has then
    // finished reading the stage it overwrites, chunk kc - 1); a wave waits for ITS DMAs of chunk kc (all but the NDMA
    // youngest) before that barrier, so behind it the whole stage has landed.
    issue(0);
    if (nk > 1) {
      issue(1);
      wait_vm<G::NDMA + NBT>();                              // the wave's x pieces of chunk 0
    } else {
      wait_vm<NBT>();
    }
    x_read(lds);
#pragma unroll
    for (int q = 0; q < 4; ++q) x_convert(q);
    int kc = 0;
    for (; kc + 2 < nk; ++kc) {
      STAMP(kc, 0);
      wait_vm<G::NDMA>();
      STAMP(kc, 1);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(kc, 2);
      if (ATRANS_DIAG != 3) compute(lds + (kc % 3) * G::STAGE, lds + ((kc + 1) % 3) * G::STAGE, kc + 2, yes, yes);
      else issue(kc + 2);
      STAMP(kc, 5);
    }
    if (kc + 1 < nk) {
      wait_vm<G::NDMA>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (ATRANS_DIAG != 3) compute(lds + (kc % 3) * G::STAGE, lds + ((kc + 1) % 3) * G::STAGE, -1, no, yes);
      ++kc;
    }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (ATRANS_DIAG != 3) compute(lds + (kc % 3) * G::STAGE, lds, -1, no, no);
    __builtin_amdgcn_s_barrier();                            // (the next segment's first DMAs overwrite stages still being read)
  }

  // epilogue: accumulator register q of lane (col, fh) is row rho(q, fh) of the wave's 32-row block
  const bool direct = sk.z < 0;
  float* __restrict__ ob = (direct ? a.out : a.ws + (long long)sk.z * a.rows * a.ld_ws) + a.off[l];
  const long long ldo = direct ? a.ld_out : a.ld_ws;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    if (cb < ncb && (ATRANS_DIAG != 1 || a.rows < 0)) {
      const int col = (cb0 + cb) * 32 + frow;
      if (col < K) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const long long row = (long long)row0 + 32 * wave + rho(q, fh);
          if (row < a.rows) {
            if (a.nt_out) __builtin_nontemporal_store(acc[cb][q], ob + row * ldo + col);
            else ob[row * ldo + col] = acc[cb][q];
          }
        }
      }
    }
  }
}

template <int TERMS, bool RAGGED, bool PLANES>
__device__ __forceinline__ void dispatch_upto4(const AtArgs& a, const int4 sg, const int4 sk, char* lds) {
  switch (sg.w) {
    case 1: run_segment<1, TERMS, RAGGED, PLANES>(a, sg, sk, lds); break;
    case 2: run_segment<2, TERMS, RAGGED, PLANES>(a, sg, sk, lds); break;
    case 3: run_segment<3, TERMS, RAGGED, PLANES>(a, sg, sk, lds); break;
    default: run_segment<4, TERMS, RAGGED, PLANES>(a, sg, sk, lds); break;
  }
}

template <int TERMS, bool RAGGED, bool PLANES>
__device__ __forceinline__ void dispatch_segment(const AtArgs& a, const int4 sg, const int4 sk, char* lds) {
  if constexpr (TERMS == 3) {
    // the three-term form holds a second set of mapping fragments: beyond four column blocks its accumulators no longer
    // fit 256 registers (148 spilled at seven), so a longer segment runs as two halves, one after the other
    if (sg.w > 4) {
      const int n1 = (sg.w + 1) / 2;
      dispatch_upto4<TERMS, RAGGED, PLANES>(a, make_int4(sg.x, sg.y, sg.z, n1), sk, lds);
      dispatch_upto4<TERMS, RAGGED, PLANES>(a, make_int4(sg.x, sg.y, sg.z + n1, sg.w - n1), sk, lds);
    } else {
      dispatch_upto4<TERMS, RAGGED, PLANES>(a, sg, sk, lds);
    }
    return;
  }
  switch (sg.w) {
    case 1: run_segment<1, TERMS, RAGGED, PLANES>(a, sg, sk, lds); break;
    case 2: run_segment<2, TERMS, RAGGED, PLANES>(a, sg, sk, lds); break;
    case 3: run_segment<3, TERMS, RAGGED, PLANES>(a, sg, sk, lds); break;
    case 4: run_segment<4, TERMS, RAGGED, PLANES>(a, sg, sk, lds); break;
    case 5: run_segment<5, TERMS, RAGGED, PLANES>(a, sg, sk, lds); break;
    case 6: run_segment<6, TERMS, RAGGED, PLANES>(a, sg, sk, lds); break;
    default: run_segment<7, TERMS, RAGGED, PLANES>(a, sg, sk, lds); break;
  }
}

template <int TERMS, bool PLANES>
__global__ void __launch_bounds__(NT, 2) atrans_kernel(AtArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // workgroups that share a row tile (consecutive logical ids) onto one XCD: blocks b, b + 8, ... share an XCD
  int id = blockIdx.x;
  if ((a.n_wg & 7) == 0) id = (id & 7) * (a.n_wg >> 3) + (id >> 3);
  const int s0 = a.seg_begin[id], s1 = a.seg_begin[id + 1];
  for (int s = s0; s < s1; ++s) {
    const int4 sg = a.segs[2 * s], sk = a.segs[2 * s + 1];
    if (a.L[sg.x] & 7) {           // (the planner cuts such layers into segments of <= 2 blocks)
      if (sg.w > 1) run_segment<2, TERMS, true, PLANES>(a, sg, sk, lds);
      else run_segment<1, TERMS, true, PLANES>(a, sg, sk, lds);
    } else {
      dispatch_segment<TERMS, false, PLANES>(a, sg, sk, lds);
    }
  }
}

// ---- K-split launches: sum of the slabs in slab order (fixed: bitwise reproducible) ---------------------------------------
struct SlabArgs {
  const float* ws;
  float* out;
  long long rows, ld_ws, ld_out;
  int n_layers;
  int off[MAXL + 1], slabs[MAXL];      // column range of every layer, its number of K slices
};

__global__ void __launch_bounds__(256) atrans_slab_sum_kernel(SlabArgs a) {
  const long long total = a.rows * a.off[a.n_layers];
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / a.off[a.n_layers];
    const int c = (int)(e - r * a.off[a.n_layers]);
    int l = 0;
    while (l + 1 < a.n_layers && c >= a.off[l + 1]) ++l;
    float v = 0.f;
    for (int k = 0; k < a.slabs[l]; ++k) v += a.ws[((long long)k * a.rows + r) * a.ld_ws + c];
    a.out[r * a.ld_out + c] = v;
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
  // one 32 x 32 tile per block; a thread takes four consecutive elements of a row: one 16-byte load (layer sizes that are
  // multiples of 4), 8-byte bf16 stores in both orientations (2-byte stores made this kernel 13 us for 40 MB)
  __shared__ float tile[32][33];
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  int l = 0;
  while (l + 1 < a.n_layers && (int)blockIdx.x >= a.tile0[l + 1]) ++l;
  const int L = a.L[l], Lp = a.Lp[l], nt = Lp >> 5;
  const int tl = blockIdx.x - a.tile0[l];
  const int k0 = (tl / nt) * 32, j0 = (tl % nt) * 32;
  const float* __restrict__ A = a.A[l];
  const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
  __bf16* __restrict__ fwd_hi = a.out + a.poff[l];
  __bf16* __restrict__ dg_hi = a.out + a.plane + a.poff[l];
  __bf16* __restrict__ fwd_lo = a.out + 2 * a.plane + a.poff[l];
  __bf16* __restrict__ dg_lo = a.out + 3 * a.plane + a.poff[l];
  auto split_store = [&](const float (&v)[4], __bf16* hi, __bf16* lo, long long at) {
    bf16x4 h, lw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h[i] = (__bf16)v[i];
      lw[i] = (__bf16)(v[i] - (float)h[i]);
    }
    *reinterpret_cast<bf16x4*>(hi + at) = h;
    if (a.want_lo) *reinterpret_cast<bf16x4*>(lo + at) = lw;
  };
  {
    const int k = k0 + r, j = j0 + c4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < L) {
      if ((L & 3) == 0 && j + 3 < L) {
        const float4 t = *reinterpret_cast<const float4*>(A + (long long)k * L + j);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (j + i < L) ? A[(long long)k * L + j + i] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[r][c4 + i] = v[i];
    split_store(v, dg_hi, dg_lo, (long long)k * Lp + j);                // data gradient: rows k, contraction j contiguous
  }
  __syncthreads();
  {
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = tile[c4 + i][r];
    split_store(v, fwd_hi, fwd_lo, (long long)(j0 + r) * Lp + k0 + c4);   // forward: rows j (output column), contraction k contiguous
  }
}

inline int pad32(int v) { return (v + 31) / 32 * 32; }


struct Blk {
  int layer, cb, w, maxseg;      // column block cb of the layer; w = contraction chunks (cost of one block); longest segment
};

// cost of blocks [b0, b1) as one run: cut at layer boundaries, pieces of more than maxseg blocks in near-equal parts
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
      cost += (long long)m * blks[i].w + 6;        // + 6: prologue / epilogue of a segment, in block-chunk units
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
  RCB_REQUIRE(sizes && (plan || max_ints == 0) && n_layers >= 1 && n_layers <= MAXL && rows >= 1 && n_cu >= 1 && max_ints >= 0,
              RCB_ERR_ARG, "atrans_plan: bad arguments");
  std::vector<Blk> blks;
  for (int l = 0; l < n_layers; ++l) {
    RCB_REQUIRE(sizes[l] >= 1, RCB_ERR_ARG, "atrans_plan: layer %d empty", l);
    const int lp = pad32(sizes[l]);
    for (int cb = 0; cb < lp / 32; ++cb) blks.push_back(Blk{l, cb, lp / BK, (sizes[l] & 7) ? 2 : MAXSEG});
  }
  const int nb = (int)blks.size();
  const long long m_tiles = (rows + BM - 1) / BM;
  int R = (int)std::max<long long>(1, n_cu / m_tiles);
  R = std::min(R, (nb + 1) / 2);
  // Few rows (a Kodak photo is 96 patches, a video clip 64): runs of one or two column blocks would fill the chip but make
  // every workgroup stream the whole x tile for a handful of MFMAs.  Keep the runs at MAXSEG blocks and cut the CONTRACTION
  // into S slices instead: a (segment, slice) per workgroup, partial sums to slabs, added in slab order afterwards.
  const int runs_full = (nb + MAXSEG - 1) / MAXSEG;
  int S = 1;
  if (m_tiles * runs_full * 2 <= n_cu) {
    S = (int)std::min<long long>(8, n_cu / (m_tiles * runs_full));
    R = runs_full;
  }
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
  std::vector<int4> segs;          // two entries per segment
  std::vector<int> begin;
  int slabs[MAXL];
  for (int l = 0; l < n_layers; ++l) slabs[l] = std::min(S, pad32(sizes[l]) / BK);
  for (long long mt = 0; mt < m_tiles; ++mt) {
    int b = 0;
    for (int ri = 0; ri < runs; ++ri) {
      std::vector<int4> run;
      run_cost(blks, b, cuts[ri], &run, (int)(mt * BM));
      b = cuts[ri];
      if (S == 1) {                       // one workgroup per run: its segments one after the other, whole contraction
        begin.push_back((int)segs.size() / 2);
        for (const int4& sg : run) {
          segs.push_back(sg);
          segs.push_back(make_int4(0, pad32(sizes[sg.x]) / BK, -1, 0));
        }
      } else {                            // one workgroup per (segment, slice)
        for (const int4& sg : run) {
          const int nc = pad32(sizes[sg.x]) / BK, sl = slabs[sg.x];
          for (int k = 0; k < sl; ++k) {
            begin.push_back((int)segs.size() / 2);
            segs.push_back(sg);
            segs.push_back(make_int4((int)((long long)k * nc / sl), (int)((long long)(k + 1) * nc / sl), k, 0));
          }
        }
      }
    }
  }
  begin.push_back((int)segs.size() / 2);
  const long long n_wg = (long long)begin.size() - 1;
  RCB_REQUIRE(n_wg < (1 << 24), RCB_ERR_SHAPE, "atrans_plan: %lld workgroups", n_wg);
  // layout: [n_wg, n_segs, S, 0, slabs[8], seg_begin (n_wg + 1; padded so that the segments start on 16 bytes), segs (2 x 4 ints each)]
  const long long head = ((RCB_ATRANS_PLAN_HEAD + n_wg + 1) + 3) / 4 * 4;
  const long long need = head + 4 * (long long)segs.size();
  RCB_REQUIRE(need < (1ll << 31), RCB_ERR_SHAPE, "atrans_plan: plan of %lld ints", need);
  if (max_ints == 0) return (int)need;            // size query: nothing written
  if (need > max_ints) return rcb::fail(RCB_ERR_SHAPE, "atrans_plan: %lld ints needed, %d given", need, max_ints);
  memset(plan, 0, sizeof(int32_t) * head);
  plan[0] = (int32_t)n_wg;
  plan[1] = (int32_t)(segs.size() / 2);
  plan[2] = S;
  for (int l = 0; l < n_layers; ++l) plan[4 + l] = S == 1 ? 0 : slabs[l];
  for (size_t i = 0; i < begin.size(); ++i) plan[RCB_ATRANS_PLAN_HEAD + i] = begin[i];
  memcpy(plan + head, segs.data(), sizeof(int4) * segs.size());
  return (int)need;
}

extern "C" int rcb_atrans_apply(const float* x, int64_t ld_x, const void* x_hi, const void* x_lo, int64_t ld_x16, float* out,
                                int64_t ld_out, int64_t rows, int32_t n_layers, const int32_t* sizes, const void* packed,
                                int32_t transpose, int32_t terms, const int32_t* plan_dev, const int32_t* plan_head,
                                float* workspace, rcb_stream_t stream) {
  RCB_REQUIRE((x || x_hi) && out && sizes && packed && plan_dev && plan_head && n_layers >= 1 && n_layers <= MAXL && rows >= 1,
              RCB_ERR_ARG, "atrans_apply: bad arguments");
  const bool planes = x_hi != nullptr;
  RCB_REQUIRE(!planes || (x == nullptr && (x_lo != nullptr || terms == 1) && (ld_x16 & 7) == 0 &&
                          ((reinterpret_cast<uintptr_t>(x_hi) | reinterpret_cast<uintptr_t>(x_lo)) & 15) == 0),
              RCB_ERR_ARG, "atrans_apply: operand planes: give x_hi and x_lo (terms >= 2) INSTEAD of x, 16-byte aligned, row stride a multiple of 8");
  const int n_wg = plan_head[0], ksplit = plan_head[2];
  RCB_REQUIRE(n_wg >= 1 && ksplit >= 1 && ksplit <= 8 && (ksplit == 1 || workspace), RCB_ERR_ARG,
              "atrans_apply: plan head (workgroups %d, slices %d) / workspace", n_wg, ksplit);
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
  RCB_REQUIRE((planes ? ld_x16 : ld_x) >= off && ld_out >= off, RCB_ERR_SHAPE, "atrans_apply: row strides %lld / %lld below %lld columns",
              (long long)(planes ? ld_x16 : ld_x), (long long)ld_out, off);
  if (planes)        // the DMAs fetch 16-byte pieces at 8-element steps from the start of every layer
    for (int l = 0; l < n_layers; ++l)
      RCB_REQUIRE((sizes[l] & 7) != 0 || (a.off[l] & 7) == 0, RCB_ERR_SHAPE, "atrans_apply: operand planes: layer %d starts at column %d (not a multiple of 8)", l, a.off[l]);
  a.x = x; a.out = out; a.ld_x = ld_x; a.ld_out = ld_out; a.rows = rows; a.n_layers = n_layers;
  a.xh = reinterpret_cast<const __bf16*>(x_hi);
  a.xl = reinterpret_cast<const __bf16*>(x_lo ? x_lo : x_hi);          // (terms = 1 never multiplies the lo fragments)
  a.ld_x16 = ld_x16;
  a.nt_out = transpose ? 1 : 0;
  const long long head = ((RCB_ATRANS_PLAN_HEAD + (long long)n_wg + 1) + 3) / 4 * 4;
  a.seg_begin = plan_dev + RCB_ATRANS_PLAN_HEAD;
  a.segs = reinterpret_cast<const int4*>(plan_dev + head);
  a.n_wg = n_wg;
  a.ws = workspace;
  a.ld_ws = (off + 3) / 4 * 4;
  const int lds = LDS_MAX_BYTES;
  hipError_t e;
  auto go = [&](auto kfn) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) kfn<<<n_wg, NT, lds, (hipStream_t)stream>>>(a);
  };
  if (terms == 1) {
    if (planes) go(atrans_kernel<1, true>); else go(atrans_kernel<1, false>);
  } else if (terms == 2) {
    if (planes) go(atrans_kernel<2, true>); else go(atrans_kernel<2, false>);
  } else {
    if (planes) go(atrans_kernel<3, true>); else go(atrans_kernel<3, false>);
  }
  RCB_REQUIRE(e == hipSuccess, (int)e, "atrans_apply: hipFuncSetAttribute: %s", hipGetErrorString(e));
  RCB_LAUNCH_CHECK();
  if (ksplit > 1) {
    SlabArgs r;
    memset(&r, 0, sizeof(r));
    r.ws = workspace; r.out = out; r.rows = rows; r.ld_ws = a.ld_ws; r.ld_out = ld_out; r.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
      r.off[l] = a.off[l];
      r.slabs[l] = plan_head[4 + l];
      RCB_REQUIRE(r.slabs[l] >= 1 && r.slabs[l] <= ksplit, RCB_ERR_ARG, "atrans_apply: plan head: %d slices for layer %d", r.slabs[l], l);
    }
    r.off[n_layers] = (int)off;
    long long blocks = (rows * off + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    atrans_slab_sum_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(r);
    RCB_LAUNCH_CHECK();
  }
  return RCB_OK;
}

extern "C" int64_t rcb_atrans_workspace_floats(int64_t rows, int32_t n_layers, const int32_t* sizes, const int32_t* plan_head) {
  if (!sizes || !plan_head || n_layers < 1 || n_layers > MAXL || rows < 1) return -1;
  if (plan_head[2] <= 1) return 0;
  long long off = 0;
  for (int l = 0; l < n_layers; ++l) off += sizes[l];
  return (int64_t)plan_head[2] * rows * ((off + 3) / 4 * 4);
}

// ---- weight gradient of a narrow layer (the output layer: 99 x 99 from 4096 rows) -------------------------------------------
// dA[k][n] = sum_m h[m][k] d[m][n] in fp32 on v_mfma_f32_32x32x2_f32 (bit for bit a chain of fmaf in row order: exact
// products, one rounding per term): 32 x 32 output tiles x row slabs, four waves per workgroup taking the slab's row
// pairs in turn; both operands are read straight from global memory in the operand layout (lane = column, 128 contiguous
// bytes per row and lane half), no LDS.  The four waves' tiles are added in wave order, a second kernel adds the slabs
// in slab order: no atomics, bitwise reproducible.  A few MFLOP per step: the library's heuristics take 29 us for this
// shape, this takes a few.
namespace {
// P16: an operand given as bf16 planes is read as float(hi) + float(lo) (exact: both terms and their sum fit fp32)
struct NarrowOp {
  const float* f;
  const __bf16* hi;
  const __bf16* lo;
  long long ld;
  __device__ __forceinline__ float at(long long m, int c) const {
    if (f) return f[m * ld + c];
    return (float)hi[m * ld + c] + (float)lo[m * ld + c];
  }
};

__global__ void __launch_bounds__(256) wgrad_narrow_kernel(NarrowOp hop, NarrowOp dop, long long rows, int L, int n_slabs,
                                                           float* __restrict__ part) {
  __shared__ float red[4][16][64];
  const int nt = (L + 31) / 32;
  const int tile = blockIdx.x % (nt * nt), slab = blockIdx.x / (nt * nt);
  const int k0 = (tile / nt) * 32, n0 = (tile % nt) * 32;
  const long long per = ((rows + n_slabs - 1) / n_slabs + 7) / 8 * 8;
  const long long m_begin = slab * per, m_end = min(rows, m_begin + per);
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, i = lane & 31, kh = lane >> 5;
  const bool kin = k0 + i < L, nin = n0 + i < L;
  const int hc = k0 + (kin ? i : 0), dc = n0 + (nin ? i : 0);
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  // wave w multiplies the row pairs (m, m + 1), m = m_begin + 2 w, + 8, + 16, ...; eight pairs' loads in flight
  for (long long m0 = m_begin + 2 * wave; m0 < m_end; m0 += 64) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long m = m0 + 8 * u + kh;
      const bool ok = m < m_end;
      const long long mc = ok ? m : m_begin;
      av[u] = hop.at(mc, hc);
      bv[u] = dop.at(mc, dc);
      av[u] = (ok && kin) ? av[u] : 0.f;
      bv[u] = (ok && nin) ? bv[u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) red[wave][q][lane] = acc[q];
  __syncthreads();
  float* o = part + ((long long)slab * nt * nt + tile) * 1024;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = t + 256 * j, q = e >> 6, ln = e & 63;
    const float v = ((red[0][q][ln] + red[1][q][ln]) + red[2][q][ln]) + red[3][q][ln];
    o[rho(q, ln >> 5) * 32 + (ln & 31)] = v;
  }
}

__global__ void __launch_bounds__(256) wgrad_narrow_sum_kernel(const float* __restrict__ part, int L, int n_slabs, float* __restrict__ dA) {
  const int nt = (L + 31) / 32;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L * L) return;
  const int k = i / L, n = i - k * L;
  const int tile = (k >> 5) * nt + (n >> 5);
  const float* p = part + (long long)tile * 1024 + (k & 31) * 32 + (n & 31);
  float s = 0.f;
  const long long st = (long long)nt * nt * 1024;
  int sl = 0;
  for (; sl + 8 <= n_slabs; sl += 8) {           // eight independent loads in flight, added in slab order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(sl + u) * st];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; sl < n_slabs; ++sl) s += p[sl * st];
  dA[i] = s;
}
}  // namespace

extern "C" int64_t rcb_atrans_wgrad_narrow_workspace(int32_t L, int32_t n_slabs) {
  if (L < 1 || n_slabs < 1) return -1;
  const int64_t nt = (L + 31) / 32;
  return (int64_t)n_slabs * nt * nt * 1024;
}

extern "C" int rcb_atrans_wgrad_narrow(const float* h, const void* h_hi, const void* h_lo, int64_t ld_h, const float* d,
                                       const void* d_hi, const void* d_lo, int64_t ld_d, int64_t rows, int32_t L, float* dA,
                                       float* workspace, int32_t n_slabs, rcb_stream_t stream) {
  RCB_REQUIRE(dA && workspace && rows >= 1 && L >= 1 && L <= 1024 && n_slabs >= 1 && n_slabs <= 4096, RCB_ERR_ARG,
              "atrans_wgrad_narrow: bad arguments (L = %d, slabs = %d)", L, n_slabs);
  RCB_REQUIRE((h != nullptr) != (h_hi != nullptr && h_lo != nullptr) && (d != nullptr) != (d_hi != nullptr && d_lo != nullptr), RCB_ERR_ARG,
              "atrans_wgrad_narrow: each operand either as fp32 rows or as a (hi, lo) pair of bf16 planes");
  RCB_REQUIRE(ld_h >= L && ld_d >= L, RCB_ERR_SHAPE, "atrans_wgrad_narrow: row strides below the layer size");
  const int nt = (L + 31) / 32;
  const NarrowOp hop{h, reinterpret_cast<const __bf16*>(h_hi), reinterpret_cast<const __bf16*>(h_lo), ld_h};
  const NarrowOp dop{d, reinterpret_cast<const __bf16*>(d_hi), reinterpret_cast<const __bf16*>(d_lo), ld_d};
  wgrad_narrow_kernel<<<nt * nt * n_slabs, 256, 0, (hipStream_t)stream>>>(hop, dop, rows, L, n_slabs, workspace);
  RCB_LAUNCH_CHECK();
  wgrad_narrow_sum_kernel<<<(L * L + 255) / 256, 256, 0, (hipStream_t)stream>>>(workspace, L, n_slabs, dA);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

#if ATRANS_STAMPS
extern "C" int rcb_debug_atrans_stamps(unsigned long long* host, int32_t n) {
  if (n > 8 * 40 * 6) n = 8 * 40 * 6;
  hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
  return (int)e;
}
#endif
