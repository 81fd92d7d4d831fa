// 16-bit-operand SIREN kernel for hidden widths above 32 (48 and 64: BASELINE configs "width-48" and "width-64 fp16").
//
// Same dataflow as the width-32 kernel (siren_mlp_bf16.hip): one workgroup per (INR, sample), pixel on the lane, 32 pixels
// per wave tile, activations chained through v_mfma_f32_32x32x16 accumulators without leaving registers.  A hidden layer
// of width W is HB = ceil(W / 32) accumulator tiles of 32 features; as the next layer's B operand it is KSH = W / 16
// k-steps, k-step ks being registers 8 (ks & 1) .. + 7 of tile ks >> 1, i.e. feature
//     featk(ks, h, j) = 16 ks + 8 (j >> 2) + 4 h + (j & 3)
// for lane half h, element j.  W = 48 therefore costs 3 k-steps and 24 (not 32) sin/cos per pixel and layer: no
// transcendental is spent on padding.  The weight fragments of every (row tile, k-step) are pre-swizzled once per INR
// into LDS in that k order, in the forward (W^T, sine layers scaled by w0 / 2 pi) and the data-gradient orientation.
// Weight gradients: OB x IB accumulator tiles per layer (12, or 14 with two input blocks, at W = 64): too many for one
// wave at two waves per SIMD, so they are dealt to the four waves of the workgroup (see the kernel's comment) and each
// wave contracts its tiles over the images of all four pixel tiles of a pass: complete sums, fixed order, no atomics.
#include <type_traits>

#include "siren_op16.h"

using namespace rcb;

// Diagnostic build only (-DRCB_SIREN_STAMPS, tools/siren_stamps.py wide): see siren_mlp_bf16.hip
#ifdef RCB_SIREN_STAMPS
__device__ unsigned long long g_wide_stamps[8192 * 16];
#define RCB_WSTAMP(k)                                                                                      \
  do {                                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_wide_stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int rcb_debug_read_stamps_wide(unsigned long long* dst, int n_entries) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wide_stamps), sizeof(unsigned long long) * n_entries);
}
// barriers of the pass loop with the time wave 0 spends in them summed (stamp slot 5) and the passes' forward time (slot 4)
#define RCB_WBARRIER()                                                        \
  do {                                                                        \
    const unsigned long long tb_ = __builtin_amdgcn_s_memtime();              \
    __syncthreads();                                                          \
    bar_acc += __builtin_amdgcn_s_memtime() - tb_;                            \
  } while (0)
#define RCB_WACC_DECL unsigned long long bar_acc = 0
#define RCB_WACC_STORE()                                                      \
  do {                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_wide_stamps[blockIdx.x * 16 + 5] = bar_acc; \
  } while (0)
#else
#define RCB_WSTAMP(k) do { } while (0)
#define RCB_WBARRIER() __syncthreads()
#define RCB_WACC_DECL
#define RCB_WACC_STORE() do { } while (0)
#endif

namespace {
using namespace rcb::op16;

__device__ __forceinline__ constexpr int featk(int ks, int h, int j) { return 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3); }

template <int NH, int F, int E, int C, int W>
struct GeoW {
  static_assert(W % 16 == 0 && W >= 32 && W <= 64, "hidden width 32, 48 or 64");
  static_assert(C <= 16, "the output gradient is one k-step");
  static_assert(E <= 16, "the data-gradient fragments onto the upsampled features are stored for 16 rows");
  static constexpr int NL = NH + 1;
  static constexpr int IN0 = F + E;
  static constexpr int K0S = (cmax(F, E) + 7) / 8;
  static constexpr int NB0 = (IN0 + 31) / 32;
  static constexpr int HB = (W + 31) / 32;
  static constexpr int KSH = W / 16;
  static_assert(NB0 <= 2, "at most two 32-feature input blocks");
  static_assert(HB == 2 || NH <= 3, "width 32: layer l's gradient tiles belong to wave l");
  __host__ __device__ static constexpr int lin(int l) { return l == 0 ? IN0 : W; }
  __host__ __device__ static constexpr int lout(int l) { return l == NL - 1 ? C : W; }
  __host__ __device__ static constexpr int off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += lout(i) * (lin(i) + 1);
    return o;
  }
  static constexpr int DNET = off(NL);
  __host__ __device__ static constexpr int lsize(int l) { return lout(l) * (lin(l) + 1); }
  __host__ __device__ static constexpr int wmax() {
    int w = 0;
    for (int l = 0; l < NL; ++l) w = lsize(l) > w ? lsize(l) : w;
    return w;
  }
  static constexpr int WMAX = wmax();
  __host__ __device__ static constexpr int wide_index(int l) {
    int k = 0;
    for (int i = 0; i < l; ++i) k += (lsize(i) == WMAX) ? 1 : 0;
    return k;
  }
  // fragment slots (1 KB each; the FBX slots at the end hold 16 rows = 512 B each)
  static constexpr int FA0 = 0;                                  // + mb * K0S + s
  static constexpr int FAH = FA0 + HB * K0S;                     // + ((l - 1) * HB + mb) * KSH + ks
  static constexpr int FAO = FAH + (NH - 1) * HB * KSH;          // + ks
  static constexpr int FBO = FAO + KSH;                          // + ib
  static constexpr int FBH = FBO + HB;                           // + (((NH - 1) - l) * HB + ib) * KSH + ks
  static constexpr int NFR = FBH + (NH - 1) * HB * KSH;          // full slots
  static constexpr int NSINE = FAO;                              // slots below carry w0 / 2 pi
  // LDS map (bytes): biases | fragments | half-height FBX fragments | per-wave image pairs (= weight staging area)
  static constexpr int BIAS_STRIDE = 64;
  static constexpr int FR_OFF = NL * BIAS_STRIDE * 4;
  static constexpr int FX_OFF = FR_OFF + NFR * 1024;             // + ks * 512
  static constexpr int TILE_OFF = FX_OFF + KSH * 512;
  static constexpr int TSA = 32 * HB;                            // dZ image row stride (elements)
  static constexpr int TSBB = 32 * cmax(HB, NB0);                // input image row stride
  static constexpr int WAVE_TILE = 32 * (TSA + TSBB) * 2;
  // after the passes: bias / loss exchange at FR_OFF, then one 32 x 33 float transposition area per wave
  static constexpr int XCH_BYTES = 4096;
  static constexpr int SCR_OFF = FR_OFF + XCH_BYTES;
  static constexpr int SCR_WAVE = 32 * 33 * 4;
  static constexpr int LDS_BYTES = cmax(TILE_OFF + 4 * WAVE_TILE, SCR_OFF + 4 * SCR_WAVE);
  static constexpr int WG_PER_CU = HB == 1 ? 3 : 2;              // waves per SIMD = workgroups per CU (4 waves each)
  static_assert(WG_PER_CU * LDS_BYTES <= 160 * 1024, "workgroups per CU");
  // the fp32 weights are staged through the image area in at most two parts: layers [0, LSPLIT) and [LSPLIT, NL)
  static constexpr int CAP = 4 * WAVE_TILE / 4;                  // floats
  __host__ __device__ static constexpr int lsplit() {
    int l = 0;
    while (l < NL && off(l + 1) <= CAP) ++l;
    return l;
  }
  static constexpr int LSPLIT = lsplit();
  static_assert(LSPLIT >= 1 && DNET - off(LSPLIT) <= CAP, "weights stage in two parts");
};

// a staged weight read from a position that is valid for EVERY lane, dropped by a select where the lane has no element there.
// (`cond ? wl[idx] : 0.f` with a lane-dependent condition compiles to an exec-masked branch around each read: 150-300 of them
// in this kernel's prologue, each read waited for on its own.)
__device__ __forceinline__ float sel_ld(bool keep, float v) { return keep ? v : 0.f; }

// Two workgroups of four waves per CU (two waves per SIMD, 256 registers each).  The weight-gradient tiles are DEALT to the
// waves instead of every wave accumulating all of them: per layer each wave stores the dZ / input images of its pixel tile,
// the workgroup synchronises, and the owner of gradient tile (ob, ib) contracts it over the images of all four pixel tiles.
// A wave thus holds 3 (4 with two input blocks) accumulator tiles instead of 12 (14), the tiles come out complete (no
// cross-wave reduction in the epilogue) and are summed over the pixel tiles in ascending order.
template <typename T, int NH, int F, int E, int C, int W, int MODE, bool IN16>
__global__ void __launch_bounds__(256, (GeoW<NH, F, E, C, W>::WG_PER_CU)) siren_wide_kernel(SirenArgs a) {
  using G = GeoW<NH, F, E, C, W>;
  using bf16x8 = typename Op16<T>::v8;
  using bf16x4 = typename Op16<T>::v4;
  using bf16x2 = typename Op16<T>::v2;
  constexpr float GS = Op16<T>::GRAD_SCALE;
  constexpr float WS = Op16<T>::W_SCALE;
  constexpr int NL = G::NL, IN0 = G::IN0, K0S = G::K0S, NB0 = G::NB0, HB = G::HB, KSH = G::KSH, DNET = G::DNET;
  constexpr int BST = G::BIAS_STRIDE;
  constexpr bool FB_EARLY_C = IN16 && NB0 == 1;             // instances whose register budget allows reads far ahead of their use
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* bias = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
  const int g = blockIdx.x % a.G, chunk = blockIdx.x / a.G;     // chunk-major: partial buffers are [chunk][g]
  const int n = g / a.S;
  const int P = a.P;
  const long long pe_row = pe_row_base(a, g);      // pixel 0 of this row in pe / dpe (contiguous rows or patches of a stitched grid)

  uint4* frags = reinterpret_cast<uint4*>(smem_raw + G::FR_OFF);
  uint4* fragsx = reinterpret_cast<uint4*>(smem_raw + G::FX_OFF);
  T* images = reinterpret_cast<T*>(smem_raw + G::TILE_OFF);
  T* bufA = images + wave * (G::WAVE_TILE / 2);
  T* bufB = bufA + 32 * G::TSA;

  // ---- stage the weights (in one or two parts through the image area), build the MFMA A-fragments --------------------
  RCB_WSTAMP(0);
  {
    const float* src = a.wvec + (long long)g * a.w_stride;
    float* wst = reinterpret_cast<float*>(smem_raw + G::TILE_OFF);
    const int lane = lane0, fq = lane & 31, fh = lane >> 5;
    // all global loads of a part first, into registers (a load-store loop waits for every load before its store); the second
    // part's loads are issued before the first part's fragments are built and stay in flight behind that work
    auto load_part = [&](auto lo_c, auto hi_c, float* stage) {
      constexpr int BASE = G::off(decltype(lo_c)::value), LEN = G::off(decltype(hi_c)::value) - BASE;
#pragma unroll
      for (int k = 0; k < (LEN + 255) / 256; ++k) {
        const int i = tid + 256 * k;
        stage[k] = src[BASE + (i < LEN ? i : LEN - 1)];
      }
    };
    auto part = [&](auto lo_c, auto hi_c, const float* stage, auto&& between) {
      constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
      constexpr int BASE = G::off(LO), LEN = G::off(HI) - BASE;
#pragma unroll
      for (int k = 0; k < (LEN + 255) / 256; ++k) {
        const int i = tid + 256 * k;
        if (i < LEN) wst[i] = stage[k];
      }
      between();
      __syncthreads();
      const float* wl = wst - BASE;                               // wl[G::off(l) + ...] as in the parameter vector
#pragma unroll
      for (int l = LO; l < HI; ++l)
        if (tid < G::lout(l)) bias[l * BST + tid] = (wl[G::off(l) + tid] * (l < NH ? a.k_hi : 1.0f)) * WS;   // sine layers work in revolutions
      auto put = [&](int slot, auto&& elem) {
        union { bf16x8 v; uint4 u; } fr;
        // sine-layer forward fragments carry w0 / 2 pi, the fragments producing a hidden layer's data gradient carry w0
        const float sc = (slot < G::NSINE) ? WS * a.k_hi : (slot >= G::FBO) ? a.w0 : WS;
#pragma unroll
        for (int j = 0; j < 8; ++j) fr.v[j] = (T)(elem(j) * sc);
        frags[slot * 64 + lane] = fr.u;
      };
      int dealt = 0;                                              // fragments of this part go round the waves
      auto mine = [&]() { return ((dealt++) & 3) == wave; };
      if (LO == 0) {
        // forward, layer 0: half-wave 0 contracts the F Fourier features, half-wave 1 the E upsampled features
#pragma unroll
        for (int mb = 0; mb < HB; ++mb)
#pragma unroll
          for (int s = 0; s < K0S; ++s) {
            if (!mine()) continue;
            put(G::FA0 + mb * K0S + s, [&](int j) {
              const int kk = 8 * s + j, out = 32 * mb + fq;
              const int row = (fh == 0) ? (kk < F ? kk : -1) : (kk < E ? F + kk : -1);
              const bool ok = row >= 0 && out < W;      // (read from a clamped position, dropped by a select: see sel_ld)
              return sel_ld(ok, wl[G::off(0) + W + (ok ? row : 0) * W + (ok ? out : 0)]);
            });
          }
        // data gradient through layer 0 onto the E upsampled features: 16 rows stored
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
          if (!mine()) continue;
          union { bf16x8 v; uint4 u; } fr;
#pragma unroll
          for (int j = 0; j < 8; ++j) fr.v[j] = (T)(sel_ld(fq < E, wl[G::off(0) + W + (F + (fq < E ? fq : 0)) * W + featk(ks, fh, j)]) * WS);
          if (fq < 16) fragsx[ks * 32 + fh * 16 + fq] = fr.u;
        }
      }
#pragma unroll
      for (int l = (LO > 1 ? LO : 1); l < (HI < NH ? HI : NH); ++l) {
        // forward, hidden layer l
#pragma unroll
        for (int mb = 0; mb < HB; ++mb)
#pragma unroll
          for (int ks = 0; ks < KSH; ++ks) {
            if (!mine()) continue;
            put(G::FAH + ((l - 1) * HB + mb) * KSH + ks, [&](int j) {
              const int out = 32 * mb + fq;
              return sel_ld(out < W, wl[G::off(l) + W + featk(ks, fh, j) * W + (out < W ? out : 0)]);
            });
          }
        // data gradient through hidden layer l
#pragma unroll
        for (int ib = 0; ib < HB; ++ib)
#pragma unroll
          for (int ks = 0; ks < KSH; ++ks) {
            if (!mine()) continue;
            put(G::FBH + (((NH - 1) - l) * HB + ib) * KSH + ks, [&](int j) {
              const int m = 32 * ib + fq;
              return sel_ld(m < W, wl[G::off(l) + W + (m < W ? m : 0) * W + featk(ks, fh, j)]);
            });
          }
      }
      if (HI == NL) {
        // forward, output layer
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
          if (!mine()) continue;
          put(G::FAO + ks, [&](int j) { return sel_ld(fq < C, wl[G::off(NH) + C + featk(ks, fh, j) * C + (fq < C ? fq : 0)]); });
        }
        // data gradient through the output layer (k = output channel, one step)
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) {
          if (!mine()) continue;
          put(G::FBO + ib, [&](int j) {
            const int m = 32 * ib + fq, k = fk(0, fh, j);
            const bool ok = m < W && k < C;
            return sel_ld(ok, wl[G::off(NH) + C + (ok ? m : 0) * C + (ok ? k : 0)]);
          });
        }
      }
      __syncthreads();
    };
    using L0c = std::integral_constant<int, 0>;
    using LSc = std::integral_constant<int, G::LSPLIT>;
    using LNc = std::integral_constant<int, NL>;
    float stage_a[(G::off(G::LSPLIT) + 255) / 256];
    load_part(L0c{}, LSc{}, stage_a);
    if constexpr (G::LSPLIT == NL) {
      part(L0c{}, LNc{}, stage_a, [] {});
    } else {
      float stage_b[(DNET - G::off(G::LSPLIT) + 255) / 256];
      part(L0c{}, LSc{}, stage_a, [&] { load_part(LSc{}, LNc{}, stage_b); });
      part(LSc{}, LNc{}, stage_b, [] {});
    }
  }
  RCB_WSTAMP(2);

  // this wave's weight-gradient tiles.  Two row blocks (width 48 / 64): hidden layer l (1 .. NH-1): tile (wave >> 1, wave & 1);
  // layer 0: the same with two input blocks, (wave, 0) on waves 0 / 1 with one; output layer: (0, wave - 2) on waves 2 / 3.
  // One row block (width 32): wave l owns layer l, i.e. one tile, or two for layer 0 with two input blocks (gT).
  constexpr int NHID = NH - 1;
  constexpr int NGH = (HB == 2 && NHID > 0) ? NHID : 1;
  f32x16 gH[NGH], g0, gT[NB0];
  // output-layer tiles (waves 2 / 3).  One input block: layer 0's tiles live on waves 0 / 1 only and the output layer takes g0.
  // Two input blocks: every wave owns a layer-0 tile already, and a fourth full tile per wave is what the register file cannot
  // hold -- but only the C rows below the padding matter: each pass contracts into a scratch tile and adds those rows to gOk.
  constexpr bool SLIM_O = (HB == 2 && NB0 == 2);
  float gOk[16];
  float bH[NGH], b0 = 0.f, bO = 0.f, bT = 0.f;
  float sse_local = 0.f;
  if (MODE != MODE_FWD) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < NGH; ++i) gH[i][r] = 0.f;
      g0[r] = 0.f;
      gOk[r] = 0.f;
#pragma unroll
      for (int i = 0; i < NB0; ++i) gT[i][r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < NGH; ++i) bH[i] = 0.f;
  }
  constexpr int KH0 = F, KH1 = E;

  const int ntiles = (P + 31) >> 5;
  constexpr bool VEC4 = (F % 8 == 0) && (E % 8 == 0);
  // IN16 (as in the width-32 kernel): both input halves arrive as bf16 rows and the loaded bits are the layer-0 B operand
  float4 raw[IN16 ? 1 : 2 * K0S];
  uint4 raw16[IN16 ? K0S : 1];
  auto fetch = [&](int tile, int q, int h) {
    const int pp = tile * 32 + q;
    const int pcl = pp < P ? pp : P - 1;
    if constexpr (IN16) {
      // (16-bit elements either way: the xf copy is in T's format with rows of FP = F rounded up to 8, zero padded; pe is bf16)
      constexpr int FP = (F + 7) / 8 * 8;
      const unsigned short* s16 = (h == 0) ? (reinterpret_cast<const unsigned short*>(a.xf16) + (long long)n * (a.xf_stride / F * FP) + (long long)pcl * FP)
                                           : (reinterpret_cast<const unsigned short*>(a.pe) + (pe_row + pe_pix_off(a, pcl)) * E);
      const int kh16 = (h == 0) ? FP : KH1;
#pragma unroll
      for (int s = 0; s < K0S; ++s) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (8 * s + 8 <= kh16) u = reinterpret_cast<const uint4*>(s16)[s];
        raw16[s] = u;
      }
    } else {
    const float* src = (h == 0) ? (a.xf + (long long)n * a.xf_stride + (long long)pcl * F)
                                : (a.pe + (pe_row + pe_pix_off(a, pcl)) * E);
    const int kh = (h == 0) ? KH0 : KH1;
    if (E % 8 == 0 && a.pe_bf16 && h == 1) {   // bf16-stored pe: 16 B per 8 features, widened exactly
      const uint4* s16 = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.pe) + (pe_row + pe_pix_off(a, pcl)) * E);
#pragma unroll
      for (int s = 0; s < K0S; ++s) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (8 * s + 8 <= KH1) u = s16[s];
        raw[2 * s] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                 __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
        raw[2 * s + 1] = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                                     __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < K0S; ++s) {
      if (VEC4) {
        if (8 * s + 8 <= kh) {
          raw[2 * s] = *reinterpret_cast<const float4*>(src + 8 * s);
          raw[2 * s + 1] = *reinterpret_cast<const float4*>(src + 8 * s + 4);
        } else {
          raw[2 * s] = make_float4(0.f, 0.f, 0.f, 0.f);
          raw[2 * s + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          float2 w2 = make_float2(0.f, 0.f);
          if (8 * s + j + 1 < kh) w2 = *reinterpret_cast<const float2*>(src + 8 * s + j);
          v[j] = w2.x;
          v[j + 1] = w2.y;
        }
        raw[2 * s] = make_float4(v[0], v[1], v[2], v[3]);
        raw[2 * s + 1] = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    }
  };
  // this workgroup's share of the 32-pixel tiles (all of them unless rcb_siren_desc.pixel_chunks > 1); every wave makes
  // the same number of passes (the passes synchronise), a wave without a tile in the last pass only contracts
  const int t0 = (int)((long long)chunk * ntiles / a.chunks), t1 = (int)((long long)(chunk + 1) * ntiles / a.chunks);
  if (t0 + wave < t1) fetch(t0 + wave, lane0 & 31, lane0 >> 5);
  // fp32 inputs are converted as soon as they have arrived (end of the pass that requested them): 4 registers per k-step
  // cross the pass boundary instead of 8
  bf16x8 xnext[IN16 ? 1 : K0S];
  auto convert_inputs = [&]() {
    if constexpr (!IN16) {
#pragma unroll
      for (int s = 0; s < K0S; ++s) {
        const float4 v0 = raw[2 * s], v1 = raw[2 * s + 1];
        xnext[s][0] = (T)v0.x; xnext[s][1] = (T)v0.y; xnext[s][2] = (T)v0.z; xnext[s][3] = (T)v0.w;
        xnext[s][4] = (T)v1.x; xnext[s][5] = (T)v1.y; xnext[s][6] = (T)v1.z; xnext[s][7] = (T)v1.w;
      }
    }
  };
  if (t0 + wave < t1) convert_inputs();
  // lane-dependent LDS offsets that stay in registers for the whole kernel (everything else lane-dependent is re-derived per
  // pass, see below).  Image rows are TSA / TSBB elements with the 8-byte chunks swizzled by (pixel >> 1) & 7 (swz()):
  //   wr_off[img][2 par + e]: this lane's store of features 16 ks + 4 h + 8 e .. + 3, ks = 2 (ks >> 1) + par, + 32 (ks >> 1)
  //   rd_addr[img][wq]:       this lane's transposed 8-byte read of pixels 8 h + 4 wq + 0..3 (+ 16 s), + feature block, + tile
  constexpr bool ONE_STRIDE = (G::TSA == G::TSBB);
  constexpr int NIMG = ONE_STRIDE ? 1 : 2;
  int wr_off[NIMG][4], rd_off[NIMG][2];
#pragma unroll
  for (int im = 0; im < NIMG; ++im) {
    const int stride = im == 0 ? G::TSA : G::TSBB;
    const int q0 = lane0 & 31, h0 = lane0 >> 5, key = (q0 >> 1) & 7;
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int e = 0; e < 2; ++e) wr_off[im][2 * par + e] = q0 * stride + (((4 * par + h0 + 2 * e) ^ key) << 2);
    const int fb = (lane0 >> 4) & 1, q4 = (lane0 & 15) >> 2, p4 = lane0 & 3;
#pragma unroll
    for (int wq = 0; wq < 2; ++wq) {
      const int pix = 8 * h0 + 4 * wq + q4;
      rd_off[im][wq] = pix * stride + (((4 * fb + p4) ^ ((pix >> 1) & 7)) << 2);
    }
  }
  const uint4* fr_lane = frags + lane0;                           // + slot * 64
  const float* bias_lane = bias + 4 * (lane0 >> 5);               // + l * BST + 32 mb + (r & 3) + 8 (r >> 2)
  // transposed operand of k-step s (pixels 16 s + 8 h + 0..7) of feature block fblk of pixel tile w's image `img` (0 / 1).
  // The per-lane LDS byte addresses are kept opaque: everything added below then fits the 16-bit offset field of the
  // read (left to itself the compiler folds the image area's own offset in, exceeds the field and keeps ~40 derived bases)
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  unsigned rd_addr[NIMG][2];
#pragma unroll
  for (int im = 0; im < NIMG; ++im)
#pragma unroll
    for (int wq = 0; wq < 2; ++wq) {
      rd_addr[im][wq] = (unsigned)(size_t)((__attribute__((address_space(3))) T*)images) + 2u * (unsigned)rd_off[im][wq];
      asm volatile("" : "+v"(rd_addr[im][wq]));
    }
  auto tr_read = [&](int w, int img, int s, int fblk) -> bf16x8 {
    union { s16x4 v[2]; bf16x8 b; } u;
    const int stride = img == 0 ? G::TSA : G::TSBB;
#pragma unroll
    for (int wq = 0; wq < 2; ++wq) {
      const unsigned addr = (rd_addr[ONE_STRIDE ? 0 : img][wq] + 2u * (unsigned)fblk) +
                            2u * (unsigned)(w * (G::WAVE_TILE / 2) + img * 32 * G::TSA + s * 16 * stride);
      u.v[wq] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)addr);
    }
    return u.b;
  };
  int lane_v = lane0;
  // one pass = four pixel tiles, one per wave.  FULL: every wave has a tile (all passes but possibly the chunk's last): no
  // wave-level conditionals in the hot loop
  RCB_WACC_DECL;
  auto pass = [&](int tb, auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    // the remaining lane-dependent addresses (global rows, layer-0 image stores) are re-derived in every pass: hoisted out
    // of the loop they are ~90 live registers that the allocator spills
    asm volatile("" : "+v"(lane_v));
    const int lane = lane_v, q = lane & 31, h = lane >> 5;
    auto FA = [&](int slot) -> bf16x8 {
      union { bf16x8 v; uint4 u; } fr;
      fr.u = fr_lane[slot * 64];
      return fr.v;
    };
    auto FX = [&](int ks) -> bf16x8 {
      union { bf16x8 v; uint4 u; } fr;
      fr.u = make_uint4(0, 0, 0, 0);
      if (q < 16) fr.u = fragsx[ks * 32 + h * 16 + q];
      return fr.v;
    };
    const int t = tb + wave;
    const bool active = FULL || t < t1;                           // wave-uniform
    const int p = t * 32 + q;
    const bool valid = active && p < P;
    const int pc = p < P ? p : P - 1;
    if (tb == t0) RCB_WSTAMP(14);
    if (tb == t0 + 4) RCB_WSTAMP(15);
    bf16x8 xin[K0S];
    bf16x8 S[NH][KSH], Cs[NH][KSH];
    bf16x8 dzb[2 * HB];
    float yv[16];
    if (active) {
#pragma unroll
      for (int s = 0; s < K0S; ++s) {
        if constexpr (IN16) {
          xin[s] = rcb::op16::in16_operand<T>(raw16[s], h);
        } else {
          xin[s] = xnext[s];
        }
      }
      if (MODE != MODE_FWD) {
        const long long ybase = ((MODE == MODE_LOSS ? (long long)n : (long long)g) * P + pc) * C;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          yv[r] = 0.f;
          if (rho(r, 0) < C || rho(r, 1) < C) {
            const int row = rho(r, h);
            yv[r] = a.yin[ybase + (row < C ? row : 0)];
          }
        }
      }
      if (MODE == MODE_FWD && t + 4 < t1) {
        fetch(t + 4, q, h);
        convert_inputs();
      }
      // ---- forward: software-pipelined by hand -- the bias / fragment reads of a layer are issued before the sine / cosine
      // work of the layer before it (scheduling barriers keep them there), and the two row blocks' MFMA chains alternate
      constexpr int KMAX = cmax(K0S, KSH);
      constexpr bool EARLY_READS = FB_EARLY_C;              // (register budget: 16-bit inputs and one input block)
      bf16x8 fr[HB][KMAX];
      f32x16 acc2[HB];
      auto issue = [&](int l) {
        const float* Bl = bias_lane + l * BST;                   // Bl[rho(r, 0)] is the (pre-scaled) bias of row rho(r, h)
#pragma unroll
        for (int mb = 0; mb < HB; ++mb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[mb][r] = (32 * mb + 16 * (r >> 3) < W) ? Bl[32 * mb + rho(r, 0)] : 0.f;
#pragma unroll
          for (int k = 0; k < KMAX; ++k)
            if (k < (l == 0 ? K0S : KSH)) fr[mb][k] = FA(l == 0 ? G::FA0 + mb * K0S + k : G::FAH + ((l - 1) * HB + mb) * KSH + k);
        }
      };
      f32x16 acc;
      bf16x8 fro[KSH];
      issue(0);
#pragma unroll
      for (int l = 0; l < NH; ++l) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
          for (int mb = 0; mb < HB; ++mb)
            if (k < (l == 0 ? K0S : KSH)) acc2[mb] = Op16<T>::mfma(fr[mb][k], l == 0 ? xin[k < K0S ? k : 0] : S[l > 0 ? l - 1 : 0][k < KSH ? k : 0], acc2[mb]);
        f32x16 cur[HB];
#pragma unroll
        for (int mb = 0; mb < HB; ++mb) cur[mb] = acc2[mb];
        __builtin_amdgcn_sched_barrier(0);
        if (l + 1 < NH) {
          if (EARLY_READS) issue(l + 1);
        } else {
          const float* Bl = bias + (NL - 1) * BST;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = (rho(r, 0) < C || rho(r, 1) < C) ? ((rho(r, h) < C) ? Bl[rho(r, h) < C ? rho(r, h) : 0] : 0.f) : 0.f;
#pragma unroll
          for (int ks = 0; ks < KSH; ++ks) fro[ks] = FA(G::FAO + ks);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < HB; ++mb) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (2 * mb + s >= KSH) continue;               // rows beyond the layer width: no transcendental spent
            bf16x8 sp, cp;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float tt = (WS == 1.0f) ? cur[mb][8 * s + j] : cur[mb][8 * s + j] * (1.0f / WS);
              sp[j] = (T)__builtin_amdgcn_sinf(tt);
              if (MODE != MODE_FWD) cp[j] = (T)__builtin_amdgcn_cosf(tt);
            }
            S[l][2 * mb + s] = sp;
            if (MODE != MODE_FWD) {
              // pinned in its packed form: left alone, the compiler keeps the 8 fp32 sine arguments alive instead and
              // evaluates cos (and its rounding, value by value) in the backward pass
              union { bf16x8 v; unsigned u[4]; } pin;
              pin.v = cp;
              asm volatile("" : "+v"(pin.u[0]), "+v"(pin.u[1]), "+v"(pin.u[2]), "+v"(pin.u[3]));
              Cs[l][2 * mb + s] = pin.v;
            }
          }
        }
        if (!EARLY_READS && l + 1 < NH) issue(l + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) acc = Op16<T>::mfma(fro[ks], S[NH - 1][ks], acc);
        if (WS != 1.0f) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (rho(r, 0) < C || rho(r, 1) < C) acc[r] *= (1.0f / WS);
        }
      }
      if (MODE == MODE_FWD) {
        if (valid) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (rho(r, 0) < C || rho(r, 1) < C) {
              int row = rho(r, h);
              if (row < C) a.yout[((long long)g * P + p) * C + row] = acc[r];
            }
          }
        }
      } else {
        // ---- output gradient ----------------------------------------------------------------------------------------
        if (tb == t0) RCB_WSTAMP(8);
        f32x16 dz;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.f;
          if (rho(r, 0) < C || rho(r, 1) < C) {
            const bool ok = valid && rho(r, h) < C;
            if (MODE == MODE_LOSS) {
              float diff = ok ? (acc[r] - yv[r]) : 0.f;
              sse_local += diff * diff;
              v = (2.0f * GS) * a.dy_scale * diff;
            } else {
              v = ok ? yv[r] * GS : 0.f;
            }
          }
          dz[r] = v;
        }
        dzb[0] = pack8<T>(dz, 0);
        dzb[1] = pack8<T>(dz, 1);
      }
    }
    if (MODE == MODE_FWD) return;
    // ---- backward: per layer  images -> sync -> data gradient (own tile) + weight gradient (own gradient tiles, all
    // pixel tiles of the pass) -> sync -----------------------------------------------------------------------------------
#pragma unroll
    for (int l = NL - 1; l >= 0; --l) {
      const int KSO = (l == NL - 1) ? 2 : KSH;            // k-steps over this layer's output features
      if (tb == t0) RCB_WSTAMP(9 + (NL - 1 - l));
      if (!FULL && !active && l == NL - 1) {
        // a wave without a pixel tile in the chunk's last pass: zero images, so that the owners contract four tiles always
        uint4* z = reinterpret_cast<uint4*>(bufA);
#pragma unroll
        for (int k = 0; k < G::WAVE_TILE / 1024; ++k) z[k * 64 + lane] = make_uint4(0, 0, 0, 0);
      }
      if (active) {
        if (l == 0 && t + 4 < t1) fetch(t + 4, q, h);     // next pass's inputs: in flight behind the last layer
        union { bf16x8 v; bf16x4 hlf[2]; } u;
#pragma unroll
        for (int ks = 0; ks < 2 * HB; ++ks) {
          if (ks >= KSO) continue;
          u.v = dzb[ks];
          *reinterpret_cast<bf16x4*>(bufA + 32 * (ks >> 1) + wr_off[0][2 * (ks & 1)]) = u.hlf[0];
          *reinterpret_cast<bf16x4*>(bufA + 32 * (ks >> 1) + wr_off[0][2 * (ks & 1) + 1]) = u.hlf[1];
        }
        if (l > 0) {
#pragma unroll
          for (int ks = 0; ks < KSH; ++ks) {
            u.v = S[l - 1][ks];
            *reinterpret_cast<bf16x4*>(bufB + 32 * (ks >> 1) + wr_off[NIMG - 1][2 * (ks & 1)]) = u.hlf[0];
            *reinterpret_cast<bf16x4*>(bufB + 32 * (ks >> 1) + wr_off[NIMG - 1][2 * (ks & 1) + 1]) = u.hlf[1];
          }
        } else {
          const int base = (h == 0) ? 0 : F;
          const int kh = (h == 0) ? KH0 : KH1;
#pragma unroll
          for (int s = 0; s < K0S; ++s) {
            union { bf16x8 v; bf16x2 pr[4]; bf16x4 hlf[2]; } x;
            x.v = xin[s];
            if (F % 4 == 0 && E % 4 == 0) {
              if (8 * s + 4 <= kh) *reinterpret_cast<bf16x4*>(bufB + swz(q, base + 8 * s, G::TSBB)) = x.hlf[0];
              if (8 * s + 8 <= kh) *reinterpret_cast<bf16x4*>(bufB + swz(q, base + 8 * s + 4, G::TSBB)) = x.hlf[1];
            } else {
#pragma unroll
              for (int j = 0; j < 8; j += 2)
                if (8 * s + j + 1 < kh) *reinterpret_cast<bf16x2*>(bufB + swz(q, base + 8 * s + j, G::TSBB)) = x.pr[j >> 1];
            }
          }
          // image columns [IN0, 32 NB0) keep whatever the hidden layers / the weight staging left there: they only reach
          // accumulator columns that are never stored (likewise feature columns >= W of either image at width 48)
        }
      }
      // the data-gradient fragments do not depend on the images: their reads are in flight across the barrier
      bf16x8 fb[HB][KSH];
      auto load_fb = [&]() {
#pragma unroll
        for (int ib = 0; ib < HB; ++ib)
#pragma unroll
          for (int ks = 0; ks < KSH; ++ks) {
            if (l == NL - 1) {
              if (ks == 0) fb[ib][0] = FA(G::FBO + ib);
            } else if (l > 0) {
              fb[ib][ks] = FA(G::FBH + (((NH - 1) - l) * HB + ib) * KSH + ks);
            } else if (ib == 0 && a.dpe != nullptr) {
              fb[0][ks] = FX(ks);
            }
          }
      };
      // (with 16-bit inputs and one input block the register budget allows it; the other instances read after the barrier)
      constexpr bool FB_EARLY = FB_EARLY_C;
      if (FB_EARLY && active) load_fb();
      RCB_WBARRIER();
      // (1) data gradient of this wave's pixel tile
      if (active) {
        if (!FB_EARLY) load_fb();
        if (l > 0) {
          bf16x8 nz[2 * HB];
          f32x16 dh[HB];
#pragma unroll
          for (int ib = 0; ib < HB; ++ib)
#pragma unroll
            for (int r = 0; r < 16; ++r) dh[ib][r] = 0.f;
          if (l == NL - 1) {
#pragma unroll
            for (int ib = 0; ib < HB; ++ib) dh[ib] = Op16<T>::mfma(fb[ib][0], dzb[0], dh[ib]);
          } else {
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks)                     // the two chains alternate
#pragma unroll
              for (int ib = 0; ib < HB; ++ib) dh[ib] = Op16<T>::mfma(fb[ib][ks], dzb[ks], dh[ib]);
          }
#pragma unroll
          for (int ib = 0; ib < HB; ++ib) {
#pragma unroll
            for (int r = 0; r < 16; ++r)                         // products in place, then packed pairwise
              dh[ib][r] = (2 * ib + (r >> 3) < KSH) ? dh[ib][r] * (float)Cs[l - 1][2 * ib + (r >> 3)][r & 7] : 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s)
              if (2 * ib + s < KSH) nz[2 * ib + s] = pack8<T>(dh[ib], s);
          }
#pragma unroll
          for (int ks = 0; ks < KSH; ++ks) dzb[ks] = nz[ks];
        } else if (a.dpe != nullptr) {
          f32x16 dx;
#pragma unroll
          for (int r = 0; r < 16; ++r) dx[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KSH; ++ks) dx = Op16<T>::mfma(fb[0][ks], dzb[ks], dx);
          if (valid) {
            float* dst = a.dpe + (pe_row + pe_pix_off(a, p)) * E;
            if (E % 8 == 0 && a.pe_bf16) {
              __bf16* d16 = reinterpret_cast<__bf16*>(a.dpe) + (pe_row + pe_pix_off(a, p)) * E;
#pragma unroll
              for (int g4 = 0; g4 < E / 8; ++g4) {
                typename Op16<__bf16>::v4 ob = {(__bf16)(dx[4 * g4] * (1.0f / (GS * WS))), (__bf16)(dx[4 * g4 + 1] * (1.0f / (GS * WS))),
                                                (__bf16)(dx[4 * g4 + 2] * (1.0f / (GS * WS))), (__bf16)(dx[4 * g4 + 3] * (1.0f / (GS * WS)))};
                *reinterpret_cast<typename Op16<__bf16>::v4*>(d16 + 8 * g4 + 4 * h) = ob;
              }
            } else if (E % 8 == 0) {
#pragma unroll
              for (int g4 = 0; g4 < E / 8; ++g4)
                *reinterpret_cast<float4*>(dst + 8 * g4 + 4 * h) = make_float4(dx[4 * g4] * (1.0f / (GS * WS)), dx[4 * g4 + 1] * (1.0f / (GS * WS)),
                                                                               dx[4 * g4 + 2] * (1.0f / (GS * WS)), dx[4 * g4 + 3] * (1.0f / (GS * WS)));
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                int e = rho(r, h);
                if (e < E) dst[e] = dx[r] * (1.0f / (GS * WS));
              }
            }
          }
        }
      }
      // (2) weight gradient of this wave's tile of the layer over the images of every pixel tile of the pass
      if constexpr (HB == 1) {
        if (wave == (l & 3)) {                             // one row block: the whole layer belongs to one wave
          const int IBl = (l == 0) ? NB0 : 1;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            bf16x8 av[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) av[s2] = tr_read(w, 0, s2, 0);
            bT = sum8_16<T>(av[0], bT);
            bT = sum8_16<T>(av[1], bT);
#pragma unroll
            for (int ib = 0; ib < NB0; ++ib) {
              if (ib >= IBl) continue;
#pragma unroll
              for (int s2 = 0; s2 < 2; ++s2) gT[ib] = Op16<T>::mfma(av[s2], tr_read(w, 1, s2, 32 * ib), gT[ib]);
            }
          }
        }
      } else {
        bool own;
        int ob, ib;
        if (l == NL - 1) {
          own = wave >= 2; ob = 0; ib = wave - 2;
        } else if (l == 0 && NB0 == 1) {
          own = wave < 2; ob = wave; ib = 0;
        } else {
          own = true; ob = wave >> 1; ib = wave & 1;
        }
        if (own) {
          f32x16 scratch_o;
          if (SLIM_O && l == NL - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scratch_o[r] = 0.f;
          }
          f32x16& acc = (l == NL - 1) ? (SLIM_O ? scratch_o : g0) : (l == 0) ? g0 : gH[l > 0 && l < NL - 1 ? l - 1 : 0];
          float& bsum = (l == NL - 1) ? bO : (l == 0) ? b0 : bH[l > 0 && l < NL - 1 ? l - 1 : 0];
          // bias gradient = row sums of dZ^T: the two owners of a row block (ib = 0 / 1) sum one k-step each; layer 0 with a
          // single input block has one owner per row block, which sums both
          const bool both = (l == 0 && NB0 == 1);
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            bf16x8 av[2], bv[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) av[s2] = tr_read(w, 0, s2, 32 * ob);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) bv[s2] = tr_read(w, 1, s2, 32 * ib);
            if (both) {
              bsum = sum8_16<T>(av[0], bsum);
              bsum = sum8_16<T>(av[1], bsum);
            } else {
              union { bf16x8 v; unsigned u[4]; } a0, a1, pick;
              a0.v = av[0];
              a1.v = av[1];
#pragma unroll
              for (int k = 0; k < 4; ++k) pick.u[k] = ib ? a1.u[k] : a0.u[k];   // wave-uniform select
              bsum = sum8_16<T>(pick.v, bsum);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) acc = Op16<T>::mfma(av[s2], bv[s2], acc);
          }
          if (SLIM_O && l == NL - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (rho(r, 0) < C || rho(r, 1) < C) gOk[r] += scratch_o[r];
          }
        }
      }
      RCB_WBARRIER();
    }
    if (t + 4 < t1) convert_inputs();
  };
  int tb = t0;
  for (; tb + 4 <= t1; tb += 4) pass(tb, std::true_type{});
  if (tb < t1) pass(tb, std::false_type{});
  if (MODE == MODE_FWD) return;
  RCB_WSTAMP(3);
  RCB_WACC_STORE();

  // ---- every gradient tile is complete in its owner's registers: transpose through the wave's own image area and store --
  const int lane = lane0, q = lane & 31, h = lane >> 5;
  {
    // bias gradients (two row blocks): the two owners of a row block hold one k-step's sums each; the ib = 1 owner hands its
    // part over through LDS (the fragment area is no longer read) and the ib = 0 owner stores own + partner
    float* xch = reinterpret_cast<float*>(smem_raw + G::FR_OFF);  // [slot][wave][32], then the four waves' loss sums
    constexpr int NSLOT = NHID + 2;
    static_assert((NSLOT * 4 * 32 + 4) * 4 <= G::XCH_BYTES, "exchange area");
    auto half_sum = [&](float v) { return v + __shfl_xor(v, 32, 64); };
    float bsumH[NGH];
#pragma unroll
    for (int i = 0; i < NGH; ++i) bsumH[i] = half_sum(bH[i]);
    float bsum0 = half_sum(b0), bsumO = half_sum(bO);
    const float bsumT = half_sum(bT);
    if (HB == 2 && h == 0) {
#pragma unroll
      for (int i = 0; i < NGH; ++i) xch[(i * 4 + wave) * 32 + q] = bsumH[i];
      xch[(NHID * 4 + wave) * 32 + q] = bsum0;
      xch[((NHID + 1) * 4 + wave) * 32 + q] = bsumO;
    }
    float sse_wave = 0.f;
    if (MODE == MODE_LOSS) {
      sse_wave = wave_sum(sse_local);
      if (lane == 0) xch[NSLOT * 4 * 32 + wave] = sse_wave;
    }
    __syncthreads();
    if (MODE == MODE_LOSS && tid == 0)
      a.sse[(long long)chunk * a.G + g] = ((xch[NSLOT * 4 * 32] + xch[NSLOT * 4 * 32 + 1]) + xch[NSLOT * 4 * 32 + 2]) + xch[NSLOT * 4 * 32 + 3];
    if (HB == 2) {
      const int partner = (wave | 1) * 32 + q;                    // read by the ib = 0 owners only
#pragma unroll
      for (int i = 0; i < NGH; ++i) bsumH[i] += xch[i * 4 * 32 + partner];
      if (NB0 == 2) bsum0 += xch[NHID * 4 * 32 + partner];
      bsumO += xch[(NHID + 1) * 4 * 32 + partner];
    }

    float* scr = reinterpret_cast<float*>(smem_raw + G::SCR_OFF + wave * G::SCR_WAVE);   // (the images' area is dead)
    float* dst = a.dwvec + ((long long)chunk * a.G + g) * a.w_stride;
    auto store_tile = [&](auto l_c, int ob, int ib, const f32x16& acc, float bt, bool with_bias) {
      constexpr int l = decltype(l_c)::value;
      constexpr int no = G::lout(l), ni = G::lin(l), ol = G::off(l);
      __bf16* sp = a.dw16 != nullptr ? a.dw16 + (long long)g * a.dw16_stride + ol : nullptr;
      __bf16* sl = a.dwlo != nullptr ? a.dwlo + (long long)g * a.dw16_stride + ol : nullptr;
      auto emit = [&](int e, float v) {
        if (a.dwvec != nullptr) dst[ol + e] = v;
        if (sp != nullptr) {                         // high plane: operand of the weight-gradient GEMM; + low plane: the pair is
          const __bf16 hb = (__bf16)v;               // the A transform's data-gradient operand
          sp[e] = hb;
          if (sl != nullptr) sl[e] = (__bf16)(v - (float)hb);
        }
      };
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int r = 0; r < 16; ++r) scr[q * 33 + rho(r, h)] = acc[r] * (1.0f / GS);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const int o = 32 * ob + q;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int il = 2 * k + h, i = 32 * ib + il;
        if (i < ni && o < no) emit(no + i * no + o, scr[il * 33 + q]);
      }
      if (with_bias && h == 0 && o < no) emit(o, bt * (1.0f / GS));
    };
    if constexpr (HB == 2) {
      if constexpr (NHID >= 1) store_tile(std::integral_constant<int, 1>{}, wave >> 1, wave & 1, gH[0], bsumH[0], (wave & 1) == 0);
      if constexpr (NHID >= 2) store_tile(std::integral_constant<int, 2>{}, wave >> 1, wave & 1, gH[NHID >= 2 ? 1 : 0], bsumH[NHID >= 2 ? 1 : 0], (wave & 1) == 0);
      if constexpr (NHID >= 3) store_tile(std::integral_constant<int, 3>{}, wave >> 1, wave & 1, gH[NHID >= 3 ? 2 : 0], bsumH[NHID >= 3 ? 2 : 0], (wave & 1) == 0);
      static_assert(NHID <= 3, "hidden-layer tiles are stored by the three calls above");
      if (NB0 == 1) {
        if (wave < 2) store_tile(std::integral_constant<int, 0>{}, wave, 0, g0, bsum0, true);
      } else {
        store_tile(std::integral_constant<int, 0>{}, wave >> 1, wave & 1, g0, bsum0, (wave & 1) == 0);
      }
      if (wave >= 2) {
        if (SLIM_O) {
          f32x16 go;
#pragma unroll
          for (int r = 0; r < 16; ++r) go[r] = (rho(r, 0) < C || rho(r, 1) < C) ? gOk[r] : 0.f;
          store_tile(std::integral_constant<int, NL - 1>{}, 0, wave - 2, go, bsumO, wave == 2);
        } else {
          store_tile(std::integral_constant<int, NL - 1>{}, 0, wave - 2, g0, bsumO, wave == 2);
        }
      }
    } else {
      if (wave == 0) {
#pragma unroll
        for (int ib = 0; ib < NB0; ++ib) store_tile(std::integral_constant<int, 0>{}, 0, ib, gT[ib], bsumT, ib == 0);
      }
      if constexpr (NL > 1) { if (wave == 1) store_tile(std::integral_constant<int, 1>{}, 0, 0, gT[0], bsumT, true); }
      if constexpr (NL > 2) { if (wave == 2) store_tile(std::integral_constant<int, (NL > 2 ? 2 : 0)>{}, 0, 0, gT[0], bsumT, true); }
      if constexpr (NL > 3) { if (wave == 3) store_tile(std::integral_constant<int, (NL > 3 ? 3 : 0)>{}, 0, 0, gT[0], bsumT, true); }
    }
  }
  RCB_WSTAMP(6);
  RCB_WSTAMP(7);
}

template <typename T, int NH, int F, int E, int C, int W, int MODE, bool IN16>
int launch_one(const SirenArgs& a, hipStream_t st) {
  using G = GeoW<NH, F, E, C, W>;
  auto kfn = siren_wide_kernel<T, NH, F, E, C, W, MODE, IN16>;
  // (per launch: the attribute belongs to the (function, device) pair; a process may drive several devices)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     G::LDS_BYTES);
  if (e != hipSuccess) return fail((int)e, "siren(wide): hipFuncSetAttribute: %s", hipGetErrorString(e));
  kfn<<<a.G * a.chunks, 256, G::LDS_BYTES, st>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

template <typename T, int NH, int F, int E, int C, int W>
int launch_mode(int mode, const SirenArgs& a, hipStream_t st) {
  constexpr bool can16 = (E % 8 == 0) && E > 0;
  if (can16 && a.pe_bf16 && a.xf16 != nullptr && mode == MODE_LOSS) return launch_one<T, NH, F, E, C, W, MODE_LOSS, can16>(a, st);
  if (mode == MODE_FWD) return launch_one<T, NH, F, E, C, W, MODE_FWD, false>(a, st);
  if (mode == MODE_BWD) return launch_one<T, NH, F, E, C, W, MODE_BWD, false>(a, st);
  return launch_one<T, NH, F, E, C, W, MODE_LOSS, false>(a, st);
}

}  // namespace

namespace rcb {
int siren_wide_dispatch(int mode, const rcb_siren_desc* d, SirenArgs& a, hipStream_t st) {
#define RCB_CASE(NHv, Fv, Ev, Cv, Wv)                                                                              \
  if (d->n_hidden == NHv && d->fourier_dim == Fv && d->pe_dim == Ev && d->out_dim == Cv && d->hidden == Wv)          \
    return d->precision == 2 ? launch_mode<_Float16, NHv, Fv, Ev, Cv, Wv>(mode, a, st)                             \
                             : launch_mode<__bf16, NHv, Fv, Ev, Cv, Wv>(mode, a, st);
#ifdef RCB_SIREN_DEALT32
  // width 32 on this kernel (wave l owns layer l's gradient tile, three workgroups per CU): measured 0.284 ms against the
  // 0.25 ms of siren_mlp_bf16.hip in the same back-to-back loop (4096 x 1024 px) -- eight barriers per pass are too many for
  // its short layers -- so these instances are only built on request (-DRCB_SIREN_DEALT32, selected by RCB_SIREN_W32_DEALT=1)
  RCB_CASE(3, 16, 16, 3, 32)
  RCB_CASE(3, 16, 16, 1, 32)
  RCB_CASE(3, 18, 16, 3, 32)
  RCB_CASE(2, 16, 16, 3, 32)
#endif
  RCB_CASE(3, 16, 16, 3, 48)   // kodak / cifar / protein geometry at width 48
  RCB_CASE(3, 16, 16, 3, 64)
  RCB_CASE(3, 18, 16, 3, 64)   // video geometry at width 64
  RCB_CASE(3, 16, 16, 1, 64)   // audio geometry at width 64
#undef RCB_CASE
  return fail(RCB_ERR_UNSUPPORTED, "siren(wide): geometry n_hidden=%d F=%d E=%d C=%d width=%d is not instantiated",
              d->n_hidden, d->fourier_dim, d->pe_dim, d->out_dim, d->hidden);
}
}  // namespace rcb
