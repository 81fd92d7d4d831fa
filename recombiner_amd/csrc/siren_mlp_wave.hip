// Width-32 SIREN, fused loss + backward, ONE WAVE PER ROW of wvec (round 5).
//
// Why this form exists.  siren_mlp_bf16.hip (one 4-wave workgroup per row, two workgroups per CU) runs far below every pipe's
// limit: its waves wait on LDS -- 21 fragment reads per 32-pixel tile, each in front of its MFMA, the bias reads that
// initialise every accumulator, the write -> transposed-read round trips of the weight gradient -- and on the per-row fixed
// cost (weight staging, fragment build, the cross-wave reduction of the gradient and its barriers: 20-25 % of the kernel).
// Here a wave owns a whole row (an INR and sample), or one pixel chunk of it:
//   * what a row reuses lives in REGISTERS for the whole row: the fragments of the three sine layers in both orientations
//     (10 x 4), the biases as three more fragments (below), the 64 weight-gradient accumulators.  The five fragments a tile
//     uses once (output layer forward / backward, the map onto the input gradient) stay in the wave's own LDS;
//   * the bias enters the accumulator through the matrix pipe: one more k-step whose A fragment holds the bias as (hi, lo)
//     bf16 halves in its first two k-slots against a B operand of ones -- no accumulator initialisation, no LDS read, 12
//     registers per row instead of 48; the pipe has the room (7 of 34 MFMAs per tile);
//   * nothing of a tile is carried in registers from its forward to its backward pass except the packed output gradient: the
//     [pixel][feature] bf16 images of the layer inputs, which the weight gradient needs transposed anyway, are written in the
//     FORWARD pass and stay in LDS; the backward pass re-reads its own rows of them (a lane reads back the 8-byte pieces it
//     wrote) and RECOMPUTES each layer's pre-activation for the cosine: two more MFMAs per layer and tile, the transcendental
//     count unchanged (the forward pass evaluates sines only).  This is what frees the registers for the resident fragments;
//   * no barrier and no cross-wave reduction anywhere: the four waves of a workgroup are independent, two workgroups share a
//     CU (two waves per SIMD fill each other's gaps), the gradient of a row leaves straight from the accumulators with the
//     tiles summed in ascending order (deterministic).
// (A first version ran one wave per SIMD on 512 registers with inline-assembly MFMAs and 2 / 4 tiles in lockstep: 292 / 435 us
// against the workgroup kernel's 226 -- one wave cannot issue more than one vector instruction per 4 cycles, half of what the
// SIMD takes from two.  It is in the history of this file.)
// Matches prior_model.py:168-179,237 and test_model.py:347-355,625-627 like the kernels it stands beside.
#include <type_traits>

#include "siren_op16.h"

using namespace rcb;

namespace {
using namespace rcb::op16;

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define RCB_LDS(T) __attribute__((address_space(3))) T
#ifndef RCB_WAVE_FSB
#define RCB_WAVE_FSB 1       // scheduling barrier behind every forward layer (A/B builds)
#endif
#ifndef RCB_WAVE_PIPE_FWD
#define RCB_WAVE_PIPE_FWD 1  // forward pass software-pipelined across the layers by hand (0: layer by layer, the compiler's order)
#endif
#ifndef RCB_WAVE_FAIR
#define RCB_WAVE_FAIR 2     // issue priority between the two workgroups of a CU: 0 none (age decides), 1 time-sliced (measured: no gain), 2 swapped per row
#endif
#ifndef RCB_WAVE_FAIR_SHIFT
#define RCB_WAVE_FAIR_SHIFT 14
#endif
#ifndef RCB_WAVE_PROBE_BLOCK0
#define RCB_WAVE_PROBE_BLOCK0 0
#endif
#ifndef RCB_WAVE_PROBE_WAVE
#define RCB_WAVE_PROBE_WAVE 0
#endif
#ifndef RCB_WAVE_ZPIPE
#define RCB_WAVE_ZPIPE 1    // backward: the recomputed pre-activation one layer ahead (see compute_z)
#endif
#ifndef RCB_WAVE_BSB
#define RCB_WAVE_BSB 1       // ... behind every backward layer
#endif

template <int NH, int F, int E, int C>
struct WGeo {
  static constexpr int NL = NH + 1;
  static constexpr int IN0 = F + E;
  static constexpr int K0S = (cmax(F, E) + 7) / 8;
  static constexpr int NB0 = (IN0 + 31) / 32;
  static constexpr int NFA = K0S + 2 * NH;
  static constexpr int NFB = 2 * NH + 1;
  __host__ __device__ static constexpr int lin(int l) { return l == 0 ? IN0 : HID; }
  __host__ __device__ static constexpr int lout(int l) { return l == NL - 1 ? C : HID; }
  __host__ __device__ static constexpr int off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += lout(i) * (lin(i) + 1);
    return o;
  }
  static constexpr int DNET = off(NL);
  static constexpr int TSA = 32;                 // row stride (elements) of the dz / hidden-input images
  static constexpr int TSX = 32 * NB0;           // row stride of the layer-0 input image
  static constexpr int IMG = 32 * TSA * 2;       // bytes of a 32 x 32 image
  static constexpr int XIMG = 32 * TSX * 2;
  static constexpr int DZ_OFF = XIMG + NH * IMG;             // the inputs of layers 0 .. NH, then the dz image
  // fragments kept in LDS (read where they are used): the output layer's forward pair, its backward fragment, the pair that maps
  // dz_0 onto the input gradient
  static constexpr int NLF = 5 + 2 * (NH - 1);                   // + the data-gradient pairs of the hidden layers
  static constexpr int LFR_OFF = DZ_OFF + IMG;
  static constexpr int WAVE_LDS = LFR_OFF + NLF * 1024;
  static constexpr int LDS_BYTES = 4 * WAVE_LDS;
  __host__ __device__ static constexpr int lds_slot(int slot) {      // LDS slot of fragment `slot`, or -1: registers
    if (slot == K0S + 2 * (NH - 1)) return 0;
    if (slot == K0S + 2 * (NH - 1) + 1) return 1;
    if (slot == NFA) return 2;
    if (slot == NFA + 1 + 2 * (NH - 1)) return 3;
    if (slot == NFA + 1 + 2 * (NH - 1) + 1) return 4;
    if (slot > NFA && slot < NFA + 1 + 2 * (NH - 1)) return 5 + (slot - NFA - 1);      // data gradient of the hidden layers
    return -1;
  }
  // layer whose weights fragment `slot` is built from (the row is staged through LDS in two halves: layers 0 .. SPLIT - 1, then
  // the rest)
  __host__ __device__ static constexpr int slot_layer(int slot) {
    if (slot < K0S) return 0;
    if (slot < NFA) return 1 + (slot - K0S) / 2;
    const int b = slot - NFA;
    if (b == 0) return NL - 1;
    if (b < 1 + 2 * (NH - 1)) return (NH - 1) - (b - 1) / 2;
    return 0;
  }
  static constexpr int SPLIT = (NL + 1) / 2;
  static_assert(off(SPLIT) * 4 <= LFR_OFF && (DNET - off(SPLIT)) * 4 <= LFR_OFF, "a half of the row fits below the LDS fragments");
};

// Diagnostic build only (-DRCB_WAVE_STAMPS, tools/wave_stamps.py): wave 0 of workgroup 0 records s_memtime at the phase
// boundaries of its first row into a device array read back through rcb_debug_wave_stamps.  Nothing of this is in the library.
#ifdef RCB_WAVE_STAMPS
__device__ unsigned long long g_wave_stamps[256];
#define RCB_WSTAMP(k)                                                                                           \
  do {                                                                                                          \
    if (blockIdx.x == 0 && threadIdx.x == 0 && (k) < 256) g_wave_stamps[(k)] = __builtin_amdgcn_s_memtime();     \
  } while (0)
#else
#define RCB_WSTAMP(k) do { } while (0)
#endif

template <typename T>
__device__ __forceinline__ f32x16 mfma_i4(const i32x4& a, const i32x4& b, const f32x16& c) {
  union { i32x4 i; typename Op16<T>::v8 v; } ua, ub;
  ua.i = a;
  ub.i = b;
  return Op16<T>::mfma(ua.v, ub.v, c);
}
template <typename V>
__device__ __forceinline__ i32x4 as_i4(const V& v) {
  union { V v; i32x4 i; } u;
  u.v = v;
  return u.i;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// DPE: the gradient of the positional encodings is written.  Rows of whole 32-pixel tiles only (the launcher checks): a store
// under a per-lane condition sits in a branch, and the compiler then counts no store when it waits for the NEXT tile's inputs
// (requested before the stores were issued, vmcnt being in order) -- the wave would wait for the stores' completion, ~1.5 k
// cycles per tile by the in-kernel stamps.
// [pixel][feature] image swizzle of this kernel: the 8-byte chunk index (f >> 2) has its low 3 bits XORed with
// ((pix >> 1) & 7) ^ ((pix >> 4) & 1).  Stores go through the LDS in groups of 16 consecutive lanes on 32 banks: within a group pixel
// bit 0 picks the bank half and bits 1-3 (through the key) the chunk position -- conflict-free, as with the plain (pix >> 1) key
// of siren_op16.h's swz().  The row re-reads of the backward pass (ds_read_b64: 32-lane groups on 64 banks) also need pixels q and
// q + 16 apart: bit 4 flips the key's low bit, which keeps the stores' bijection (bit 4 is constant inside a 16-lane group) and
// separates the two.  With swz() every row re-read took two passes (PMC: 15 % of the LDS cycles were conflicts); a first attempt
// keyed on pixel bits 2-4 fixed the reads and made every STORE two-way (36 % conflicts by the counters, 32 instead of 16 cycles
// per four stores once the store grouping is modelled: tools/lds_banks.py `w64`) -- at an unchanged kernel time either way: the
// LDS is not what this kernel waits for.  The transposed reads (4 pixels x 8 chunks per 32 lanes) are conflict-free with all three.
__device__ __forceinline__ int wswz(int pix, int f, int stride) {
  const int c = f >> 2;
  return pix * stride + ((((c & 7) ^ (((pix >> 1) & 7) ^ ((pix >> 4) & 1))) | (c & ~7)) << 2) + (f & 3);
}

template <typename T, int NH, int F, int E, int C, int MODE, bool DPE>
__global__ void __launch_bounds__(256, 2) siren_wave_kernel(SirenArgs a) {
  using G = WGeo<NH, F, E, C>;
  using bf16x8 = typename Op16<T>::v8;
  using bf16x4 = typename Op16<T>::v4;
  constexpr float GS = Op16<T>::GRAD_SCALE;
  constexpr float WS = Op16<T>::W_SCALE;
  constexpr int NL = G::NL, K0S = G::K0S, NB0 = G::NB0, NFA = G::NFA, NFB = G::NFB;
  static_assert(F % 8 == 0 && E % 8 == 0 && E > 0, "16-bit input rows of whole 16-byte pieces");
  static_assert(MODE != MODE_FWD, "loss / backward only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int q = lane & 31, h = lane >> 5;
  RCB_LDS(unsigned char)* wlds = (RCB_LDS(unsigned char)*)smem_raw + wave * G::WAVE_LDS;
  // images of the tile in flight: layer 0's input at x_img, the input of layer l >= 1 (the sines of layer l - 1) at s_img(l - 1)
  RCB_LDS(T)* const x_img = (RCB_LDS(T)*)wlds;
  auto s_img = [&](int l) -> RCB_LDS(T)* { return (RCB_LDS(T)*)(wlds + G::XIMG + l * G::IMG); };
  RCB_LDS(T)* const dz_img = (RCB_LDS(T)*)(wlds + G::DZ_OFF);
  const int P = a.P;
  const int ntiles = (P + 31) >> 5;
  const int nunits = a.G * a.chunks;
  // the 8-byte pieces a lane writes into (and reads back from) an image of row stride TSA: k-step s, halves 0 / 1
  int po[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    po[s][0] = wswz(q, 16 * s + 4 * h, G::TSA);
    po[s][1] = wswz(q, 16 * s + 8 + 4 * h, G::TSA);
  }
  auto put_img = [&](RCB_LDS(T)* img, const bf16x8 (&v)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      union { bf16x8 v; bf16x4 hlf[2]; } uu;
      uu.v = v[s];
      *(RCB_LDS(bf16x4)*)(img + po[s][0]) = uu.hlf[0];
      *(RCB_LDS(bf16x4)*)(img + po[s][1]) = uu.hlf[1];
    }
  };
  auto get_img = [&](const RCB_LDS(T)* img, bf16x8 (&v)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      union { bf16x8 v; bf16x4 hlf[2]; } uu;
      uu.hlf[0] = *(const RCB_LDS(bf16x4)*)(img + po[s][0]);
      uu.hlf[1] = *(const RCB_LDS(bf16x4)*)(img + po[s][1]);
      v[s] = uu.v;
    }
  };
  // layer-0 input image: half-wave 0 holds features [0, F), half-wave 1 features [F, F + E); K0S pieces of 8 features
  const int xbase = (h == 0) ? 0 : F, xkh = (h == 0) ? F : E;
  auto put_x = [&](const bf16x8 (&v)[K0S]) {
#pragma unroll
    for (int s = 0; s < K0S; ++s) {
      union { bf16x8 v; bf16x4 hlf[2]; } uu;
      uu.v = v[s];
      if (8 * s + 8 <= xkh) {
        *(RCB_LDS(bf16x4)*)(x_img + wswz(q, xbase + 8 * s, G::TSX)) = uu.hlf[0];
        *(RCB_LDS(bf16x4)*)(x_img + wswz(q, xbase + 8 * s + 4, G::TSX)) = uu.hlf[1];
      }
    }
  };
  auto get_x = [&](bf16x8 (&v)[K0S]) {
#pragma unroll
    for (int s = 0; s < K0S; ++s) {
      union { bf16x8 v; bf16x4 hlf[2]; } uu;
#pragma unroll
      for (int j = 0; j < 8; ++j) uu.v[j] = (T)0.f;
      if (8 * s + 8 <= xkh) {
        uu.hlf[0] = *(const RCB_LDS(bf16x4)*)(x_img + wswz(q, xbase + 8 * s, G::TSX));
        uu.hlf[1] = *(const RCB_LDS(bf16x4)*)(x_img + wswz(q, xbase + 8 * s + 4, G::TSX));
      }
      v[s] = uu.v;
    }
  };
  // transposed operand of the weight gradient: 8 pixels (16 s + 8 h + 0..7) of feature column (lane & 31) + fcol
  auto read_tr3 = [&](const RCB_LDS(T)* img, int stride, int s, int fcol) -> bf16x8 {
    const int fb = (lane >> 4) & 1, i = lane & 15, q4 = i >> 2, p4 = i & 3;
    union { s16x4 v[2]; bf16x8 b; } u;
#pragma unroll
    for (int w = 0; w < 2; ++w)
      u.v[w] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RCB_LDS(s16x4)*)(img + wswz(16 * s + 8 * h + 4 * w + q4, fcol + 16 * fb + 4 * p4, stride)));
    return u.b;
  };
  // a fragment kept in LDS.  `lfr_lane` is laundered once per tile (an empty asm the compiler cannot see through): the reads
  // cannot be hoisted out of the tile loop -- where they would occupy the registers they are meant to free -- but are ordinary
  // loads inside a tile, free to be issued early
  int lfr_lane = lane * 16;
  auto LFR = [&](int ls) -> i32x4 { return *(const RCB_LDS(i32x4)*)(wlds + G::LFR_OFF + ls * 1024 + lfr_lane); };

  // measurement aid: rcb_siren_desc.clock_probe (diagnostic builds move the window: -DRCB_WAVE_PROBE_BLOCK0=256 -DRCB_WAVE_PROBE_WAVE=3)
  const int pblk = (int)blockIdx.x - RCB_WAVE_PROBE_BLOCK0;
  const bool probing = a.clock_probe != nullptr && pblk >= 0 && pblk < 256 && threadIdx.x == 64 * RCB_WAVE_PROBE_WAVE;
  if (probing) {
    a.clock_probe[4 * pblk + 0] = __builtin_amdgcn_s_memtime();
    a.clock_probe[4 * pblk + 1] = __builtin_amdgcn_s_memrealtime();
  }
#if RCB_WAVE_FAIR
  const unsigned young = (blockIdx.x >= (gridDim.x >> 1)) ? 1u : 0u;      // the second workgroup of a CU (dispatch order)
  unsigned long long fair_clk = __builtin_amdgcn_s_memtime();
  unsigned fair_row = 0;
#endif
  for (int u = blockIdx.x * 4 + wave; u < nunits; u += gridDim.x * 4) {
    RCB_WSTAMP(0);
#if RCB_WAVE_FAIR == 2
    // Mode 2 (shipped): the roles swap per ROW.  The SIMD arbitrates its two waves by priority, then age: left alone the wave of
    // the first-dispatched workgroup runs nearly unimpeded (88 us per row) and finishes its two rows at ~176 us, the other one
    // gets the leftover issue slots (106 us per row) and ends the kernel at ~212-222 us, its last 36 us alone on the SIMD
    // (rcb_siren_desc.clock_probe on both halves of the grid, tools/wave_spread.py).  Here the second workgroup of a CU takes the
    // higher priority on its odd rows: each workgroup is the pole for one row and the filler for the other, both halves finish
    // within 190-204 us and the kernel at ~210: 229.0 / 233.8 / 235.9 vs 237.6 / 237.8 / 237.5 us medians on one (slow) box.
    // (Alternating the priority in time slices of 16 k cycles, mode 1, equalises the halves too -- at 200 / 210 us -- without
    // moving the kernel's end: frequent flips cost what they gain.)
    if (young) {
      if (fair_row & 1u) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    ++fair_row;
#endif
#ifdef RCB_WAVE_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) g_wave_stamps[4] = __builtin_amdgcn_s_memrealtime();
#endif
    const int g = u % a.G, chunk = u / a.G;              // chunk-major: partial buffers are [chunk][g]
    const int n = g / a.S;
    const long long pe_row = (long long)g * a.P;          // pe / dpe as [G][P][E] (the stitched layouts stay with the workgroup kernel)
    const float* __restrict__ wsrc = a.wvec + (long long)g * a.w_stride;
    static_assert(G::off(G::SPLIT) % 4 == 0, "the second half of the row starts on a 16-byte boundary");
    const int rowcap = (int)a.w_stride;                  // floats readable from the row's start (the array holds whole strides)
    const bool dma16 = (a.w_stride & 3) == 0 && (reinterpret_cast<size_t>(wsrc) & 15) == 0 && rowcap >= ((G::DNET + 3) & ~3);

    // ---- the row's weights as MFMA A fragments ------------------------------------------------------------------------------
    // fragment k order: the chained accumulator's (siren_mlp_bf16.hip): k-slot (step s, lane half h, element j) = row fk(s,h,j).
    // The row travels through the wave's LDS in two halves by LDS-DMA (coalesced 256-byte pieces, no registers; the image
    // area is free between rows), the fragments are gathered from there.
    i32x4 FR[NFA + NFB];
    i32x4 BFR[NH];
    float bout[16];
    {
      RCB_LDS(float)* wl = (RCB_LDS(float)*)wlds;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int f0 = half == 0 ? 0 : G::off(G::SPLIT), f1 = half == 0 ? G::off(G::SPLIT) : G::DNET;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // 16-byte pieces (1 KB per instruction: 7 instead of 26 per half -- an LDS-DMA instruction costs the issuing wave 60-180
        // cycles) where the row allows them: 16-byte aligned rows whose stride covers the last piece's overrun of <= 3 floats
        // (the training step's rows sit on 128-byte lines); 4-byte pieces otherwise
        if (dma16) {
#pragma unroll
          for (int c0 = f0; c0 < f1; c0 += 256) {
            const int idx = c0 + 4 * lane + 3 < rowcap ? c0 + 4 * lane : rowcap - 4;      // (never beyond the row's stride)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + idx),
                                             (RCB_LDS(void)*)(wl + (c0 - f0)), 16, 0, 0);
          }
        } else {
#pragma unroll
          for (int c0 = f0; c0 < f1; c0 += 64) {
            const int idx = c0 + lane < G::DNET ? c0 + lane : G::DNET - 1;        // (never beyond the row)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + idx),
                                             (RCB_LDS(void)*)(wl + (c0 - f0)), 4, 0, 0);
          }
        }
        // (the builtin, not inline assembly: the compiler tracks LDS-DMA as vector-memory operations that write LDS and, unless
        // it SEES them retired, drains vmcnt in front of every LDS access of the tile loop -- including the input prefetch)
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        RCB_WSTAMP(6 + 2 * half);
        auto W = [&](int idx) -> float { return wl[idx - f0]; };
#pragma unroll
        for (int slot = 0; slot < NFA + NFB; ++slot) {
          if ((G::slot_layer(slot) < G::SPLIT) != (half == 0)) continue;
          float w8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float w = 0.f;
            if (slot < K0S) {
              const int kk = 8 * slot + j;
              const int row = (h == 0) ? (kk < F ? kk : -1) : (kk < E ? F + kk : -1);
              const float v = W(G::off(0) + HID + (row >= 0 ? row : 0) * HID + q);
              w = row >= 0 ? v : 0.f;
            } else if (slot < NFA) {
              const int l = 1 + (slot - K0S) / 2, st = (slot - K0S) & 1;
              const int no = (l == NL - 1) ? C : HID;
              // (every gather below reads unconditionally from a clamped position and drops the value by a select: a read under a
              // lane-dependent condition is an exec-masked branch per element, and the gathers then go out one behind the other)
              const float v = W(G::off(l) + no + fk(st, h, j) * no + (q < no ? q : 0));
              w = q < no ? v : 0.f;
            } else {
              const int b = slot - NFA;
              if (b == 0) {
                const int oo = fk(0, h, j);
                const float v = W(G::off(NL - 1) + C + q * C + (oo < C ? oo : 0));
                w = oo < C ? v : 0.f;
              } else if (b < 1 + 2 * (NH - 1)) {
                const int l = (NH - 1) - (b - 1) / 2, st = (b - 1) & 1;
                w = W(G::off(l) + HID + q * HID + fk(st, h, j));
              } else {
                const int st = (b - 1 - 2 * (NH - 1));
                const float v = W(G::off(0) + HID + (F + (q < E ? q : 0)) * HID + fk(st, h, j));
                w = q < E ? v : 0.f;
              }
            }
            w8[j] = w;
          }
          // forward fragments of the sine layers carry w0 / 2 pi (the accumulator feeds v_sin / v_cos in revolutions), the
          // transposed fragments that produce a hidden layer's data gradient carry w0, the others stay in the original units
          const float sc = (slot < K0S + 2 * (NH - 1)) ? WS * a.k_hi
                           : (slot >= NFA && slot < NFA + 1 + 2 * (NH - 1)) ? a.w0 : WS;
          bf16x8 fv;
#pragma unroll
          for (int j = 0; j < 8; ++j) fv[j] = (T)(w8[j] * sc);
          if (G::lds_slot(slot) >= 0) ((RCB_LDS(i32x4)*)(wlds + G::LFR_OFF))[G::lds_slot(slot) * 64 + lane] = as_i4(fv);
          else FR[slot] = as_i4(fv);
        }
        // biases of the sine layers, in revolutions, as (hi, lo) halves in the first two k-slots of a fragment (lane half 0)
#pragma unroll
        for (int l = 0; l < NH; ++l) {
          if ((l < G::SPLIT) != (half == 0)) continue;
          const float b = W(G::off(l) + q) * (a.k_hi * WS);
          const T bh = (T)b;
          const T bl = (T)(b - (float)bh);
          bf16x8 f8;
#pragma unroll
          for (int j = 0; j < 8; ++j) f8[j] = (T)0.f;
          f8[0] = (h == 0) ? bh : (T)0.f;
          f8[1] = (h == 0) ? bl : (T)0.f;
          BFR[l] = as_i4(f8);
        }
        RCB_WSTAMP(7 + 2 * half);
        if (half == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            bout[r] = 0.f;
            if (rho(r, 0) < C || rho(r, 1) < C) {
              const float v = W(G::off(NL - 1) + (rho(r, h) < C ? rho(r, h) : 0));
              bout[r] = (rho(r, h) < C) ? v : 0.f;
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    i32x4 ONES;
    {
      bf16x8 o8;
#pragma unroll
      for (int j = 0; j < 8; ++j) o8[j] = (T)1.0f;
      ONES = as_i4(o8);
    }

    f32x16 gW[NL + NB0 - 1];
#pragma unroll
    for (int i = 0; i < NL + NB0 - 1; ++i) gW[i] = zero16();
    float gb[NL];
    float sse_local = 0.f;
#pragma unroll
    for (int l = 0; l < NL; ++l) gb[l] = 0.f;

    // ---- inputs / targets of a tile: requested one tile ahead -----------------------------------------------------------------
    uint4 raw16[K0S];
    float ynext[16];
    const int t0 = (int)((long long)chunk * ntiles / a.chunks), t1 = (int)((long long)(chunk + 1) * ntiles / a.chunks);
    const float* __restrict__ yrow = a.yin + (MODE == MODE_LOSS ? (long long)n : (long long)g) * P * C;
    auto fetch = [&](int tile) {
      const int tl = tile < t1 ? tile : t1 - 1;
      const int pp = tl * 32 + q;
      const int pcl = pp < P ? pp : P - 1;
      const __bf16* s16 = (h == 0) ? (reinterpret_cast<const __bf16*>(a.xf16) + (long long)n * a.xf_stride + (long long)pcl * F)
                                   : (reinterpret_cast<const __bf16*>(a.pe) + (pe_row + pcl) * E);
#pragma unroll
      for (int s = 0; s < K0S; ++s) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (8 * s + 8 <= xkh) v = reinterpret_cast<const uint4*>(s16)[s];
        raw16[s] = v;
      }
      const int yoff = pcl * C;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ynext[r] = 0.f;
        if (rho(r, 0) < C || rho(r, 1) < C) {
          const int row = rho(r, h);
          ynext[r] = yrow[yoff + (row < C ? row : 0)];
        }
      }
    };
    fetch(t0);
    RCB_WSTAMP(1);
    // The input gradient of tile t is stored at the top of iteration t + 1, right BEHIND the request for tile t + 2's inputs:
    // vmcnt retires in order, so a wait for loads also waits for every store issued before them -- and a store takes a few
    // thousand cycles to be acknowledged.  With the stores younger than the loads, and both in the loop's one block (the first
    // iteration stores zeros where tile t0's values will land), the compiler's waits for the inputs leave the stores in flight.
    // With the stores at the end of the iteration the wave sat ~1.5 k cycles per tile waiting for its own stores (stamps).
    typename Op16<__bf16>::v4 dpe_pend[E / 8];
#pragma unroll
    for (int g4 = 0; g4 < E / 8; ++g4) dpe_pend[g4] = typename Op16<__bf16>::v4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};

    for (int t = t0; t < t1; ++t) {
#ifdef RCB_WAVE_STAMPS
      const int ts = 10 + 12 * (t - t0);          // stamps of the first tiles of the row
      RCB_WSTAMP(ts);
#endif
#if RCB_WAVE_FAIR == 1
      // The two waves of a SIMD are arbitrated by priority, then AGE: the wave of the workgroup that was dispatched first runs
      // nearly unimpeded and finishes its two rows at ~176 us, the other one gets the leftover issue slots and finishes at ~212 us,
      // the last 36 of them alone on its SIMD (rcb_siren_desc.clock_probe on both halves of the grid, tools/wave_spread.py).
      // This switch hands the high priority to the two halves of the grid in alternating time slices of the shader clock
      // (2^RCB_WAVE_FAIR_SHIFT cycles; the clock is read one tile ahead, so its wait is one that happens anyway).  Measured: the
      // halves then finish at ~200 / ~210 us -- and the kernel still ends at 218-222 us: the pole-and-filler arrangement the
      // hardware falls into is as productive as equal shares, and the end is set by the slowest XCD (their clocks differ by 5 %
      // under this kernel: 1.93 ... 2.04 GHz).  Off.
      if ((((unsigned)(fair_clk >> RCB_WAVE_FAIR_SHIFT)) & 1u) != young) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
      fair_clk = __builtin_amdgcn_s_memtime();
#endif
      bf16x8 xin[K0S];
      float yv[16];
#pragma unroll
      for (int s = 0; s < K0S; ++s) {
        union { uint4 u4; bf16x8 v; } cv;
        cv.u4 = raw16[s];
        xin[s] = cv.v;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) yv[r] = ynext[r];
      constexpr bool valid = true;          // whole tiles
      asm volatile("" : "+v"(lfr_lane));
      fetch(t + 1);
      if (DPE) {
        __bf16* d16 = reinterpret_cast<__bf16*>(a.dpe) + (pe_row + (t > t0 ? t - 1 : t) * 32 + q) * E;
#pragma unroll
        for (int g4 = 0; g4 < E / 8; ++g4) *reinterpret_cast<typename Op16<__bf16>::v4*>(d16 + 8 * g4 + 4 * h) = dpe_pend[g4];
      }
      // (the images are stored and read back through different vector types: nothing of this tile moves above the transposed
      // reads of the previous one)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

      // ================= forward ==================================================================================================
      bf16x8 S[2];                           // sines of the layer just finished: the next layer's B operand
      bf16x8 dzb[2];                         // packed gradient of the layer being processed
#if RCB_WAVE_PIPE_FWD
      // Software-pipelined across the layers: the sines of a layer are evaluated in two halves and each half's packed
      // fragment goes straight into the next layer's k-step, so that k-step 0 runs on the matrix pipe under the second half's
      // sines; the next layer's bias term (an MFMA that depends on nothing) is issued under the first half's.  Order pinned
      // with scheduling barriers: left alone the compiler puts all sixteen sines behind the three chained MFMAs and an s_nop.
      put_x(xin);
      f32x16 c = mfma_i4<T>(BFR[0], ONES, zero16());
#pragma unroll
      for (int s = 0; s < K0S; ++s) c = mfma_i4<T>(FR[s], as_i4(xin[s]), c);
      f32x16 cb = mfma_i4<T>(BFR[NH > 1 ? 1 : 0], ONES, zero16());          // bias term of layer 1 (of nothing when NH == 1)
#pragma unroll
      for (int l = 0; l < NH; ++l) {
        f32x16 cn;
        const bool last = (l + 1 == NH);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          bf16x8 pk;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) pk[jj] = (T)__builtin_amdgcn_sinf((WS == 1.0f) ? c[8 * hf + jj] : c[8 * hf + jj] * (1.0f / WS));
          S[hf] = pk;
          __builtin_amdgcn_sched_barrier(0);
          const i32x4 fr = last ? LFR(hf) : FR[K0S + 2 * l + hf];
          cn = mfma_i4<T>(fr, as_i4(pk), hf == 0 ? (last ? zero16() : cb) : cn);
          if (hf == 0 && l + 2 < NH) cb = mfma_i4<T>(BFR[l + 2 < NH ? l + 2 : 0], ONES, zero16());
          __builtin_amdgcn_sched_barrier(0);
        }
        put_img(s_img(l), S);
        c = cn;
#ifdef RCB_WAVE_STAMPS
        RCB_WSTAMP(ts + 1 + l);
#endif
      }
      {
        // output layer: loss gradient (or the upstream gradient) on the C rows that exist, packed with literal zeros
        f32x16 dzo;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.f;
          if (rho(r, 0) < C || rho(r, 1) < C) {
            const bool ok = valid && rho(r, h) < C;
            if (MODE == MODE_LOSS) {
              const float y = (WS != 1.0f) ? (c[r] * (1.0f / WS) + bout[r]) : (c[r] + bout[r]);
              const float diff = ok ? (y - yv[r]) : 0.f;
              sse_local += diff * diff;
              v = (2.0f * GS) * a.dy_scale * diff;
            } else {
              v = ok ? yv[r] * GS : 0.f;
            }
          }
          dzo[r] = v;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const int r = 8 * s + jj;
            dzb[s][jj] = (rho(r, 0) < C || rho(r, 1) < C) ? (T)dzo[r] : (T)0.0f;
          }
        __builtin_amdgcn_sched_barrier(0);
#ifdef RCB_WAVE_STAMPS
        RCB_WSTAMP(ts + 1 + NH);
#endif
      }
#else
      put_x(xin);
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        f32x16 c;
        if (l == 0) {
          c = mfma_i4<T>(BFR[0], ONES, zero16());
#pragma unroll
          for (int s = 0; s < K0S; ++s) c = mfma_i4<T>(FR[s], as_i4(xin[s]), c);
        } else if (l < NH) {
          c = mfma_i4<T>(BFR[l], ONES, zero16());
#pragma unroll
          for (int s = 0; s < 2; ++s) c = mfma_i4<T>(FR[K0S + 2 * (l - 1) + s], as_i4(S[s]), c);
        } else {
          c = mfma_i4<T>(LFR(0), as_i4(S[0]), zero16());
          c = mfma_i4<T>(LFR(1), as_i4(S[1]), c);
        }
        if (l < NH) {
          f32x16 sv;
#pragma unroll
          for (int r = 0; r < 16; ++r) sv[r] = __builtin_amdgcn_sinf((WS == 1.0f) ? c[r] : c[r] * (1.0f / WS));
          S[0] = pack8<T>(sv, 0);
          S[1] = pack8<T>(sv, 1);
          put_img(s_img(l), S);
        } else {
          // output layer: loss gradient (or the upstream gradient) on the C rows that exist, packed with literal zeros
          f32x16 dzo;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = 0.f;
            if (rho(r, 0) < C || rho(r, 1) < C) {
              const bool ok = valid && rho(r, h) < C;
              if (MODE == MODE_LOSS) {
                const float y = (WS != 1.0f) ? (c[r] * (1.0f / WS) + bout[r]) : (c[r] + bout[r]);
                const float diff = ok ? (y - yv[r]) : 0.f;
                sse_local += diff * diff;
                v = (2.0f * GS) * a.dy_scale * diff;
              } else {
                v = ok ? yv[r] * GS : 0.f;
              }
            }
            dzo[r] = v;
          }
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
              const int r = 8 * s + jj;
              dzb[s][jj] = (rho(r, 0) < C || rho(r, 1) < C) ? (T)dzo[r] : (T)0.0f;
            }
        }
#if RCB_WAVE_FSB
        __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef RCB_WAVE_STAMPS
        RCB_WSTAMP(ts + 1 + l);
#endif
      }

#endif

      // ================= backward ===============================================================================================
      // per layer: dz image written; data-gradient MFMAs; the pre-activation of layer l - 1 recomputed from its input rows (read
      // back from the tile's image) -> cosine x data gradient -> packed dz of the layer below; weight-gradient MFMAs on the
      // transposed reads of the dz image and of the layer's input image, one k-step in front of the vector work and one behind.
      // (RCB_WAVE_ZPIPE) the recomputed pre-activation is taken one layer ahead: z of sine layer l - 2 is issued at the END of layer l
      // (behind its cosine work, where d and the product are dead) and consumed by layer l - 1 -- its three MFMAs and its row
      // re-reads then sit beside vector work instead of in front of it, and the data-gradient chain d -> cos x d -> pack no
      // longer queues behind them in the matrix pipe
      auto compute_z = [&](int lm) -> f32x16 {              // lm = 0 .. NH - 1 (compile-time after unrolling)
        f32x16 zz = mfma_i4<T>(BFR[lm], ONES, zero16());
        if (lm == 0) {
          bf16x8 tx[K0S];
          get_x(tx);
#pragma unroll
          for (int s = 0; s < K0S; ++s) zz = mfma_i4<T>(FR[s], as_i4(tx[s]), zz);
        } else {
          bf16x8 ts[2];
          get_img(s_img(lm - 1), ts);
#pragma unroll
          for (int s = 0; s < 2; ++s) zz = mfma_i4<T>(FR[K0S + 2 * (lm - 1) + s], as_i4(ts[s]), zz);
        }
        return zz;
      };
      f32x16 zc = zero16();
      if (RCB_WAVE_ZPIPE) zc = compute_z(NL - 2);
#pragma unroll
      for (int l = NL - 1; l >= 0; --l) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        put_img(dz_img, dzb);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        bf16x8 rin[cmax(K0S, 2)];
        if (!RCB_WAVE_ZPIPE && l >= 1) {
          if (l - 1 == 0) {
            bf16x8 tx[K0S];
            get_x(tx);
#pragma unroll
            for (int s = 0; s < K0S; ++s) rin[s] = tx[s];
          } else {
            bf16x8 ts[2];
            get_img(s_img(l - 2), ts);
            rin[0] = ts[0];
            rin[1] = ts[1];
          }
        }
        f32x16 d = zero16();
        if (l == NL - 1) {
          d = mfma_i4<T>(LFR(2), as_i4(dzb[0]), d);
        } else if (l > 0) {
          const int base = NFA + 1 + 2 * ((NH - 1) - l);
#pragma unroll
          for (int s = 0; s < 2; ++s) d = mfma_i4<T>(G::lds_slot(base + s) >= 0 ? LFR(G::lds_slot(base + s)) : FR[base + s], as_i4(dzb[s]), d);
        } else if (DPE) {
#pragma unroll
          for (int s = 0; s < 2; ++s) d = mfma_i4<T>(LFR(3 + s), as_i4(dzb[s]), d);
        }
        f32x16 z;
        if (RCB_WAVE_ZPIPE) {
          z = zc;
        } else if (l > 0) {
          z = mfma_i4<T>(BFR[l - 1], ONES, zero16());
          if (l - 1 == 0) {
#pragma unroll
            for (int s = 0; s < K0S; ++s) z = mfma_i4<T>(FR[s], as_i4(rin[s]), z);
          } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) z = mfma_i4<T>(FR[K0S + 2 * (l - 2) + s], as_i4(rin[s]), z);
          }
        }
        // weight gradient, first k-step: dW_l += dz^T x input over pixels 0 .. 15 of the tile; bias gradient from the dz fragment
        {
          const bf16x8 av = read_tr3(dz_img, G::TSA, 0, 0);
          gb[l] = sum8_16<T>(av, gb[l]);
#pragma unroll
          for (int blk = 0; blk < ((l == 0) ? NB0 : 1); ++blk) {
            const bf16x8 bv = (l == 0) ? read_tr3(x_img, G::TSX, 0, 32 * blk) : read_tr3(s_img(l - 1), G::TSA, 0, 0);
            const int gi = (l == 0) ? blk : (l + NB0 - 1);
            gW[gi] = mfma_i4<T>(as_i4(av), as_i4(bv), gW[gi]);
          }
        }
        // data gradient: dz of the layer below (w0 is in the fragments), packed at once
        bf16x8 dzn[2];
        if (l > 0) {
          f32x16 dzf;
#pragma unroll
          for (int r = 0; r < 16; ++r) dzf[r] = d[r] * __builtin_amdgcn_cosf((WS == 1.0f) ? z[r] : z[r] * (1.0f / WS));
          dzn[0] = pack8<T>(dzf, 0);
          dzn[1] = pack8<T>(dzf, 1);
        } else if (DPE) {
          // (stored at the top of the next iteration: see dpe_pend)
#pragma unroll
          for (int g4 = 0; g4 < E / 8; ++g4)
            dpe_pend[g4] = typename Op16<__bf16>::v4{(__bf16)(d[4 * g4] * (1.0f / (GS * WS))), (__bf16)(d[4 * g4 + 1] * (1.0f / (GS * WS))),
                                                     (__bf16)(d[4 * g4 + 2] * (1.0f / (GS * WS))), (__bf16)(d[4 * g4 + 3] * (1.0f / (GS * WS)))};
        }
        // weight gradient, second k-step (pixels 16 .. 31)
        {
          const bf16x8 av = read_tr3(dz_img, G::TSA, 1, 0);
          gb[l] = sum8_16<T>(av, gb[l]);
#pragma unroll
          for (int blk = 0; blk < ((l == 0) ? NB0 : 1); ++blk) {
            const bf16x8 bv = (l == 0) ? read_tr3(x_img, G::TSX, 1, 32 * blk) : read_tr3(s_img(l - 1), G::TSA, 1, 0);
            const int gi = (l == 0) ? blk : (l + NB0 - 1);
            gW[gi] = mfma_i4<T>(as_i4(av), as_i4(bv), gW[gi]);
          }
        }
        if (l > 0) {
          dzb[0] = dzn[0];
          dzb[1] = dzn[1];
        }
        if (RCB_WAVE_ZPIPE && l >= 2) zc = compute_z(l - 2);
#if RCB_WAVE_BSB
        __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef RCB_WAVE_STAMPS
        RCB_WSTAMP(ts + 1 + NL + (NL - 1 - l));
#endif
      }
    }
    if (DPE) {
      __bf16* d16 = reinterpret_cast<__bf16*>(a.dpe) + (pe_row + (t1 - 1) * 32 + q) * E;
#pragma unroll
      for (int g4 = 0; g4 < E / 8; ++g4) *reinterpret_cast<typename Op16<__bf16>::v4*>(d16 + 8 * g4 + 4 * h) = dpe_pend[g4];
    }
    RCB_WSTAMP(2);

    // ---- the row's gradient, straight from the accumulators --------------------------------------------------------------------
    {
      const long long orow = (long long)chunk * a.G + g;
      if (MODE == MODE_LOSS) {
        const float v = wave_sum(sse_local);
        if (lane == 0) a.sse[orow] = v;
      }
      float* dst = a.dwvec ? a.dwvec + orow * a.w_stride : nullptr;
      __bf16* d16 = a.dw16 ? a.dw16 + (long long)g * a.dw16_stride : nullptr;
      __bf16* dlo = a.dwlo ? a.dwlo + (long long)g * a.dw16_stride : nullptr;
      auto put = [&](int idx, float v) {
        if (dst) dst[idx] = v;
        if (d16) {
          const __bf16 hb = (__bf16)v;
          d16[idx] = hb;
          if (dlo) dlo[idx] = (__bf16)(v - (float)hb);
        }
      };
      // 16-byte fp32 stores / 8-byte bf16 stores where the rows allow them (the training step's rows sit on 128-byte lines)
      const bool vec4 = (dst == nullptr || (((a.w_stride & 3) == 0) && ((reinterpret_cast<size_t>(a.dwvec) & 15) == 0))) &&
                        (d16 == nullptr || (((a.dw16_stride & 3) == 0) && ((reinterpret_cast<size_t>(a.dw16) & 7) == 0) &&
                                            (dlo == nullptr || (reinterpret_cast<size_t>(a.dwlo) & 7) == 0)));
      // one wave-uniform decision in front of the stores instead of a branch around every one of them (a branch per store
      // made the compiler park the accumulators in scratch and reload one per block behind a full vmcnt drain: 20 k cycles per
      // row by the stamps): STEP = the training step's form (fp32 gradient + its bf16 copy, 16-byte rows)
      auto emit = [&](auto step_form) {
        constexpr bool STEP = decltype(step_form)::value;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          const int ol = G::off(l), no = G::lout(l);
          const float bt = (gb[l] + __shfl_xor(gb[l], 32, 64)) * (1.0f / GS);
          if (h == 0 && q < no) {
            if (STEP) {
              dst[ol + q] = bt;
              d16[ol + q] = (__bf16)bt;
            } else {
              put(ol + q, bt);
            }
          }
#pragma unroll
          for (int blk = 0; blk < ((l == 0) ? NB0 : 1); ++blk) {
            const int gi = (l == 0) ? blk : (l + NB0 - 1);
            const f32x16 gv = gW[gi];
            const int i = 32 * blk + q;
            if (i < G::lin(l)) {
              if (no == HID && (ol + no) % 4 == 0 && (STEP || vec4)) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                  const int idx = ol + no + i * no + 8 * r4 + 4 * h;
                  const float v0 = gv[4 * r4] * (1.0f / GS), v1 = gv[4 * r4 + 1] * (1.0f / GS), v2 = gv[4 * r4 + 2] * (1.0f / GS),
                              v3 = gv[4 * r4 + 3] * (1.0f / GS);
                  const typename Op16<__bf16>::v4 hb = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
                  if (STEP) {
                    *reinterpret_cast<float4*>(dst + idx) = make_float4(v0, v1, v2, v3);
                    *reinterpret_cast<typename Op16<__bf16>::v4*>(d16 + idx) = hb;
                  } else {
                    if (dst) *reinterpret_cast<float4*>(dst + idx) = make_float4(v0, v1, v2, v3);
                    if (d16) {
                      *reinterpret_cast<typename Op16<__bf16>::v4*>(d16 + idx) = hb;
                      if (dlo) {
                        const typename Op16<__bf16>::v4 lb = {(__bf16)(v0 - (float)hb[0]), (__bf16)(v1 - (float)hb[1]),
                                                              (__bf16)(v2 - (float)hb[2]), (__bf16)(v3 - (float)hb[3])};
                        *reinterpret_cast<typename Op16<__bf16>::v4*>(dlo + idx) = lb;
                      }
                    }
                  }
                }
              } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  if (rho(r, 0) < no || rho(r, 1) < no) {
                    const int o = rho(r, h);
                    if (o < no) {
                      const float v = gv[r] * (1.0f / GS);
                      if (STEP) {
                        dst[ol + no + i * no + o] = v;
                        d16[ol + no + i * no + o] = (__bf16)v;
                      } else {
                        put(ol + no + i * no + o, v);
                      }
                    }
                  }
                }
              }
            }
          }
        }
      };
      if (vec4 && dst != nullptr && d16 != nullptr && dlo == nullptr) emit(std::true_type{});
      else emit(std::false_type{});
    }
    RCB_WSTAMP(3);
#ifdef RCB_WAVE_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) g_wave_stamps[5] = __builtin_amdgcn_s_memrealtime();
#endif
  }
  if (probing) {                                                                // (one wave of the workgroup: its own rows)
    a.clock_probe[4 * pblk + 2] = __builtin_amdgcn_s_memtime();
    a.clock_probe[4 * pblk + 3] = __builtin_amdgcn_s_memrealtime();
  }
}

#ifdef RCB_WAVE_STAMPS
}  // namespace
extern "C" int rcb_debug_wave_stamps(unsigned long long* dst, int n_entries) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wave_stamps), sizeof(unsigned long long) * n_entries);
}
namespace {
#endif

// compute units of the current device (cached per device: a process may drive several)
int cu_count() {
  static int n_cu[64];
  int dev_id = 0;
  if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) dev_id = 0;
  if (n_cu[dev_id] == 0) {
    hipDeviceProp_t prop;
    n_cu[dev_id] = (hipGetDeviceProperties(&prop, dev_id) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n_cu[dev_id];
}

template <typename T, int NH, int F, int E, int C, int MODE, bool DPE>
int launch_wave(const SirenArgs& a, hipStream_t st) {
  using G = WGeo<NH, F, E, C>;
  auto kfn = siren_wave_kernel<T, NH, F, E, C, MODE, DPE>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return fail((int)e, "siren(wave): hipFuncSetAttribute: %s", hipGetErrorString(e));
  const long long units = (long long)a.G * a.chunks;
  long long blocks = (units + 3) / 4;
  const int cus = cu_count();
  static const long long cap_env = getenv("RCB_WAVE_BLOCKS") ? atoll(getenv("RCB_WAVE_BLOCKS")) : 0;   // (A/B runs)
  const long long cap = cap_env > 0 ? cap_env : 2ll * cus;
  if (blocks > cap) blocks = cap;                       // two workgroups (four independent waves each) per CU
  kfn<<<(unsigned)blocks, 256, G::LDS_BYTES, st>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

}  // namespace

namespace rcb {
// RCB_SIREN_WAVE in the environment sets the initial value (A/B runs): 1 = one wave per row where this family has an instance
// (default: stand-alone 221 vs 236 us at 4096 rows x 1024 pixels, same box; inside the training step the gain is within the
// run-to-run spread), 0 = the workgroup kernel everywhere.  Raising a wave's issue priority on one of a CU's two workgroups
// (s_setprio 1 / 3) changes nothing.
int& siren_wave_tiles() {
  static int v = [] {
    const char* e = getenv("RCB_SIREN_WAVE");
    return e ? atoi(e) : 1;
  }();
  return v;
}
// *taken = true when this family has an instance for the request (the launch result is returned); otherwise the workgroup
// kernel (siren_mlp_bf16.hip) runs it
int siren_wave_dispatch(int mode, const rcb_siren_desc* d, SirenArgs& a, hipStream_t st, int variant, bool* taken) {
  *taken = false;
  if (mode == MODE_FWD || d->precision != 1 || !a.pe_bf16 || a.xf16 == nullptr || d->hidden != HID || (a.P & 31) != 0 || a.pe_nd != 0)
    return RCB_OK;
  {
    // A wave walks whole rows: the launch is as long as the wave with the most rows.  Unless the (row, chunk) units fill the
    // resident waves (two 4-wave workgroups per CU) to 90 % in the last round, the workgroup family -- four waves per row,
    // any number of rows -- is the better fit (the test-time batches: 500 images x 5 samples = 2500 rows on 2048 waves).
    const long long units = (long long)a.G * a.chunks, waves = 8ll * cu_count();
    const long long rounds = (units + waves - 1) / waves;
    if (variant != 2 && units * 10 < rounds * waves * 9) return RCB_OK;      // (2: forced, tests)
  }
#define RCB_CASE(NHv, Fv, Ev, Cv)                                                                               \
  if (d->n_hidden == NHv && d->fourier_dim == Fv && d->pe_dim == Ev && d->out_dim == Cv) {                      \
    *taken = true;                                                                                              \
    if (a.dpe != nullptr)                                                                                       \
      return mode == MODE_LOSS ? launch_wave<__bf16, NHv, Fv, Ev, Cv, MODE_LOSS, true>(a, st)                   \
                               : launch_wave<__bf16, NHv, Fv, Ev, Cv, MODE_BWD, true>(a, st);                   \
    return mode == MODE_LOSS ? launch_wave<__bf16, NHv, Fv, Ev, Cv, MODE_LOSS, false>(a, st)                    \
                             : launch_wave<__bf16, NHv, Fv, Ev, Cv, MODE_BWD, false>(a, st);                    \
  }
  RCB_CASE(3, 16, 16, 3)
#undef RCB_CASE
  return RCB_OK;
}
}  // namespace rcb

extern "C" int rcb_debug_siren_wave_tiles(int32_t tiles) {
  int& v = rcb::siren_wave_tiles();
  const int old = v;
  if (tiles >= 0) v = tiles;
  return old;
}
