// bf16-operand SIREN kernel: v_mfma_f32_32x32x16_bf16, fp32 accumulate, hardware sin/cos.
//
// Same dataflow as the fp32 kernel (siren_mlp.hip): one workgroup per (INR, sample), pixel on the
// lane, chained 32x32 accumulators so activations never leave registers in the forward and
// data-gradient chains.  Specifics of this path:
//   * geometry (F, E, C, hidden layers) is a template parameter: all layer offsets, predicates and
//     LDS addresses are compile-time, the tile loop has no divergent branch except the tail store;
//   * one 32x32x16 MFMA consumes 16 k-values; for an accumulator tile used as the next B operand,
//     k-slot (step s, lane half h, element j) is tile row fk(s,h,j) = 16 s + 8 (j>>2) + 4 h + (j&3),
//     i.e. registers 8s..8s+7 of the lane, converted pairwise to bf16.  The weight (A) fragments
//     are pre-swizzled once per INR into LDS in exactly that k order (one ds_read_b128 per MFMA),
//     for the forward (W^T) and the data-gradient (W) orientation;
//   * weight gradient (contraction over pixels = lanes): the [pixel][feature] bf16 image of a tile is
//     written with four 8-byte stores per lane (64-byte rows, XOR-swizzled chunks: conflict-free) and read back
//     transposed with ds_read_b64_tr_b16 (4 per operand), 2 MFMAs per layer per 32 pixels;
//   * bias gradients: v_dot2c_f32_bf16 of the transposed dZ fragment against ones;
//   * sin and cos are kept as packed bf16 (8 VGPRs per layer each).
#include "siren_op16.h"

using namespace rcb;

// Diagnostic build only (-DRCB_SIREN_STAMPS, tools/siren_stamps.py): wave 0 of every workgroup records s_memtime at its
// phase boundaries into a device array read back through rcb_debug_read_stamps.  Nothing of this exists in the library.
#ifdef RCB_SIREN_STAMPS
__device__ unsigned long long g_siren_stamps[8192 * 16];
#define RCB_STAMP(k)                                                                                      \
  do {                                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_siren_stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int rcb_debug_read_stamps(unsigned long long* dst, int n_entries) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_siren_stamps), sizeof(unsigned long long) * n_entries);
}
#else
#define RCB_STAMP(k) do { } while (0)
#endif
#ifndef RCB_SIREN_DEFER_WGRAD
#define RCB_SIREN_DEFER_WGRAD 1     // 0: write and read the weight-gradient images of a layer back to back (rounds 1-3; A/B builds)
#endif
#ifndef RCB_W32_DIRECT_FRAGS
#define RCB_W32_DIRECT_FRAGS 0      // 1: gather the fragments straight from global memory (measured slower: 0.265 vs 0.254 ms; A/B builds)
#endif

namespace {
using namespace rcb::op16;

template <int NH, int F, int E, int C>
struct Geo {
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
  __host__ __device__ static constexpr int lsize(int l) { return lout(l) * (lin(l) + 1); }
  __host__ __device__ static constexpr int wmax() {
    int w = 0;
    for (int l = 0; l < NL; ++l) w = lsize(l) > w ? lsize(l) : w;
    return w;
  }
  static constexpr int WMAX = wmax();                        // length of the widest layer vector(s)
  __host__ __device__ static constexpr int wide_index(int l) {   // rank of layer l among the layers of length WMAX
    int k = 0;
    for (int i = 0; i < l; ++i) k += (lsize(i) == WMAX) ? 1 : 0;
    return k;
  }
  // LDS map (bytes)
  static constexpr int FR_OFF = ((DNET * 4 + 15) / 16) * 16;
  static constexpr int TILE_OFF = FR_OFF + (NFA + NFB) * 1024;
  static constexpr int TSA = 32;                   // bufA row stride (bf16 elements), XOR-swizzled 8-byte chunks
  static constexpr int TSBB = 32 * NB0;            // bufB row stride
  static constexpr int WAVE_TILE = 32 * (TSA + TSBB) * 2;   // bytes per wave and image set
  // TWO image sets per wave (RCB_SIREN_DEFER_WGRAD): layer l's [pixel][feature] images are written while layer l + 1's are read
  // back transposed for its weight gradient -- the write -> read round trip through LDS then has a whole layer of other work
  // in front of it instead of stalling the (in-order) wave twice per layer
  // (one input block only: with two -- 34 inputs, the video geometry -- the extra fragments spill)
  static constexpr int NSET = (RCB_SIREN_DEFER_WGRAD && NB0 == 1) ? 2 : 1;
  static constexpr int LDS_MAIN = TILE_OFF + 4 * NSET * WAVE_TILE;
  // cross-wave reduction scratch: per wave, layer l stored [out][in] with row stride 33 (+ bias row)
  __host__ __device__ static constexpr int roff(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += lout(i) * (32 * (i == 0 ? NB0 : 1) + 1) + 32;
    return o;
  }
  static constexpr int RED_WAVE = roff(NL);
  static constexpr int LDS_BYTES = cmax(LDS_MAIN, 4 * RED_WAVE * 4 + 16);   // + the four waves' loss sums
};

// Waves per SIMD the register allocation aims at.  The loss / backward instances need their 231 registers: at three waves (<= 168)
// they spill 142 of them and run 3.4 x slower (0.92 vs 0.27 ms; -DRCB_SIREN_WAVES=3 builds that variant).  The forward-only
// instances (prediction, decoding) use ~130 and are 12 % faster when the allocator is TOLD three waves fit (0.136 -> 0.119 ms).
#ifndef RCB_SIREN_WAVES
#define RCB_SIREN_WAVES 2
#endif
template <typename T, int NH, int F, int E, int C, int MODE, bool IN16>
__global__ void __launch_bounds__(256, MODE == MODE_FWD ? 3 : RCB_SIREN_WAVES) siren_bf16_kernel(SirenArgs a) {
  using G = Geo<NH, F, E, C>;
  using bf16x8 = typename Op16<T>::v8;
  using bf16x4 = typename Op16<T>::v4;
  using bf16x2 = typename Op16<T>::v2;
  constexpr float GS = Op16<T>::GRAD_SCALE;
  constexpr float WS = Op16<T>::W_SCALE;
  constexpr int NL = G::NL, IN0 = G::IN0, K0S = G::K0S, NB0 = G::NB0, NFA = G::NFA, NFB = G::NFB, DNET = G::DNET;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = lane & 31, h = lane >> 5;
  const int g = blockIdx.x % a.G, chunk = blockIdx.x / a.G;     // chunk-major: partial buffers are [chunk][g]
  const int n = g / a.S;
  const int P = a.P;
  const long long pe_row = pe_row_base(a, g);      // pixel 0 of this row in pe / dpe (contiguous rows or patches of a stitched grid)

  float* wl = smem;
  uint4* frags = reinterpret_cast<uint4*>(smem_raw + G::FR_OFF);
  T* bufA0 = reinterpret_cast<T*>(smem_raw + G::TILE_OFF + wave * G::NSET * G::WAVE_TILE);
  T* bufB0 = bufA0 + 32 * G::TSA;
  // image set of layer l: sets alternate between consecutive layers
  auto bufA_of = [&](int l) -> T* { return bufA0 + (G::NSET == 2 ? (l & 1) * (G::WAVE_TILE / (int)sizeof(T)) : 0); };
  auto bufB_of = [&](int l) -> T* { return bufB0 + (G::NSET == 2 ? (l & 1) * (G::WAVE_TILE / (int)sizeof(T)) : 0); };

  if (a.clock_probe != nullptr && blockIdx.x < 256 && tid == 0) {      // measurement aid: rcb_siren_desc.clock_probe
    a.clock_probe[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime();
    a.clock_probe[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
  RCB_STAMP(0);
  // ---- stage weights, build MFMA A-fragments, clear the zero padding of bufB ---------------------
  {
    const float* src = a.wvec + (long long)g * a.w_stride;
#if RCB_W32_DIRECT_FRAGS
    // The fragments are gathered straight from the row of wvec in global memory: every wave issues the 8 x (3 or 4) loads of
    // its slots back to back -- ONE memory round trip -- instead of staging the row in LDS (a round trip and a barrier) and
    // gathering from there (another 32 dependent LDS reads per lane).  The lines of the 13 KB row are fetched from L2 once and
    // hit in L1 afterwards.  Only the biases live in LDS (wl keeps the layout of the parameter vector; nothing else of it
    // is read).  The sine layers work in revolutions: w0 / 2 pi is folded into their forward fragments and biases.
    const float* wsrc = src;
#pragma unroll
    for (int l = 0; l < NL; ++l)
      if (tid < G::lout(l)) wl[G::off(l) + tid] = src[G::off(l) + tid] * (l < NH ? a.k_hi : 1.0f);
    if (32 * NB0 > IN0) {
      for (int p2 = 0; p2 < G::NSET; ++p2)
        for (int i = lane; i < 32 * G::TSBB; i += 64) bufB_of(p2)[i] = (T)0.f;
    }
    RCB_STAMP(1);
#else
    const float* wsrc = wl;
    {
      // all loads first, then the LDS stores: written as `wl[i] = src[i]` in a loop the compiler waits for every load before
      // its store -- 13 serialized HBM round trips, 17 % of the whole kernel by the in-kernel stamps (tools/siren_stamps.py)
      constexpr int NLD = (DNET + 255) / 256;
      float stage[NLD];
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        const int i = tid + 256 * k;
        stage[k] = src[i < DNET ? i : DNET - 1];
      }
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        const int i = tid + 256 * k;
        if (i < DNET) wl[i] = stage[k];
      }
    }
    if (32 * NB0 > IN0) {
      for (int p2 = 0; p2 < G::NSET; ++p2)
        for (int i = lane; i < 32 * G::TSBB; i += 64) bufB_of(p2)[i] = (T)0.f;
    }
    __syncthreads();
    RCB_STAMP(1);
    // the sine layers work in revolutions: w0 / 2 pi is folded into their forward fragments (below) and biases (here),
    // so the accumulator feeds v_sin / v_cos directly -- no scaling multiply per activation
    if (tid < HID) {
#pragma unroll
      for (int l = 0; l < NH; ++l) wl[G::off(l) + tid] *= a.k_hi;
    }
#endif
    // one fragment slot per wave and pass; `slot` is a compile-time constant inside the unrolled loop, so the layer
    // offsets and shapes below fold away and only the lane-dependent part of the gather address remains
#pragma unroll
    for (int slot = 0; slot < NFA + NFB; ++slot) {
      if ((slot & 3) != wave) continue;
      const int fq = lane & 31, fh = lane >> 5;
      union { bf16x8 v; uint4 u; } fr;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float w = 0.f;
        if (slot < K0S) {
          int kk = 8 * slot + j;
          int row = (fh == 0) ? (kk < F ? kk : -1) : (kk < E ? F + kk : -1);
          // (every read below comes from a position that is valid for all lanes and is dropped by a select: under a
          // lane-dependent condition each read is an exec-masked branch of its own, 125 of them in this prologue)
          const float v = wsrc[G::off(0) + HID + (row >= 0 ? row : 0) * HID + fq];
          w = row >= 0 ? v : 0.f;
        } else if (slot < NFA) {
          int l = 1 + (slot - K0S) / 2, st = (slot - K0S) & 1;
          int no = (l == NL - 1) ? C : HID;
          int o = 0;
          for (int i = 0; i < l; ++i) o += G::lout(i) * (G::lin(i) + 1);
          const float v = wsrc[o + no + fk(st, fh, j) * no + (fq < no ? fq : 0)];
          w = fq < no ? v : 0.f;
        } else {
          int b = slot - NFA;
          if (b == 0) {
            int oo = fk(0, fh, j);
            const float v = wsrc[G::off(NL - 1) + C + fq * C + (oo < C ? oo : 0)];
            w = oo < C ? v : 0.f;
          } else if (b < 1 + 2 * (NH - 1)) {
            int l = (NH - 1) - (b - 1) / 2, st = (b - 1) & 1;
            int o = 0;
            for (int i = 0; i < l; ++i) o += G::lout(i) * (G::lin(i) + 1);
            w = wsrc[o + HID + fq * HID + fk(st, fh, j)];
          } else {
            int st = (b - 1 - 2 * (NH - 1));
            const float v = wsrc[G::off(0) + HID + (F + (fq < E ? fq : 0)) * HID + fk(st, fh, j)];
            w = fq < E ? v : 0.f;
          }
        }
        // forward fragments of the sine layers carry w0 / 2 pi; the transposed fragments that produce a hidden layer's
        // data gradient carry w0 (d sin(w0 z) / dz = w0 cos(w0 z): the multiply by w0 happens inside the MFMA); the output
        // layer and the fragments of the input (pe) gradient stay in the original units
        const float sc = (slot < K0S + 2 * (NH - 1)) ? WS * a.k_hi
                         : (slot >= NFA && slot < NFA + 1 + 2 * (NH - 1)) ? a.w0 : WS;
        fr.v[j] = (T)(w * sc);
      }
      frags[slot * 64 + lane] = fr.u;
    }
    __syncthreads();
  }
  RCB_STAMP(2);
  auto FA = [&](int slot) -> bf16x8 {
    union { bf16x8 v; uint4 u; } fr;
    fr.u = frags[slot * 64 + lane];
    return fr.v;
  };

  f32x16 gW[NL + NB0 - 1];
  float gb[NL];
  float sse_local = 0.f;
  if (MODE != MODE_FWD) {
#pragma unroll
    for (int i = 0; i < NL + NB0 - 1; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) gW[i][r] = 0.f;
#pragma unroll
    for (int l = 0; l < NL; ++l) gb[l] = 0.f;
  }
  constexpr int KH0 = F, KH1 = E;

  // ---- weight-gradient helpers (backward, per layer) --------------------------------------------------------------------
  // images of layer l: dz (packed) into bufA, the layer's input (previous activations, or the tile's input features) into bufB
  auto wg_write = [&](int l, const bf16x8 (&dzb)[2], const bf16x8 (*Sprev)[2], const bf16x8* xin_) __attribute__((always_inline)) {
    T* bufA = bufA_of(l);
    T* bufB = bufB_of(l);
    const int q_ = lane & 31, h_ = lane >> 5;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    union { bf16x8 v; bf16x4 hlf[2]; } u;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u.v = dzb[s];
      *reinterpret_cast<bf16x4*>(bufA + swz(q_, 16 * s + 4 * h_, G::TSA)) = u.hlf[0];
      *reinterpret_cast<bf16x4*>(bufA + swz(q_, 16 * s + 8 + 4 * h_, G::TSA)) = u.hlf[1];
    }
    if (l > 0) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u.v = Sprev[l - 1][s];
        *reinterpret_cast<bf16x4*>(bufB + swz(q_, 16 * s + 4 * h_, G::TSBB)) = u.hlf[0];
        *reinterpret_cast<bf16x4*>(bufB + swz(q_, 16 * s + 8 + 4 * h_, G::TSBB)) = u.hlf[1];
      }
    } else {
      // input image: half-wave 0 holds features [0,F), half-wave 1 features [F, F+E)
      const int base = (h_ == 0) ? 0 : F;
      const int kh = (h_ == 0) ? F : E;
#pragma unroll
      for (int s = 0; s < K0S; ++s) {
        union { bf16x8 v; bf16x2 pr[4]; bf16x4 hlf[2]; } x;
        x.v = xin_[s];
        if (F % 4 == 0 && E % 4 == 0) {
          if (8 * s + 4 <= kh) *reinterpret_cast<bf16x4*>(bufB + swz(q_, base + 8 * s, G::TSBB)) = x.hlf[0];
          if (8 * s + 8 <= kh) *reinterpret_cast<bf16x4*>(bufB + swz(q_, base + 8 * s + 4, G::TSBB)) = x.hlf[1];
        } else {
#pragma unroll
          for (int j = 0; j < 8; j += 2)
            if (8 * s + j + 1 < kh) *reinterpret_cast<bf16x2*>(bufB + swz(q_, base + 8 * s + j, G::TSBB)) = x.pr[j >> 1];
        }
      }
    }
  };
  // transposed fragments of layer l's images
  auto frag_read = [&](int l, bf16x8 (&av)[2], bf16x8 (&bv)[NB0][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s) av[s] = read_tr<T>(bufA_of(l), G::TSA, s, lane, 0);
#pragma unroll
    for (int blk = 0; blk < ((l == 0) ? NB0 : 1); ++blk)
#pragma unroll
      for (int s = 0; s < 2; ++s) bv[blk][s] = read_tr<T>(bufB_of(l), G::TSBB, s, lane, 32 * blk);
  };
  // dW_l += dz^T x input (contraction over the tile's 32 pixels: two k-steps), bias gradient from the transposed dz fragments
  auto wg_mfma = [&](int l, const bf16x8 (&av)[2], const bf16x8 (&bv)[NB0][2]) __attribute__((always_inline)) {
    gb[l] = sum8_16<T>(av[0], gb[l]);
    gb[l] = sum8_16<T>(av[1], gb[l]);
#pragma unroll
    for (int blk = 0; blk < ((l == 0) ? NB0 : 1); ++blk) {
      const int gi = (l == 0) ? blk : (l + NB0 - 1);
#pragma unroll
      for (int s = 0; s < 2; ++s) gW[gi] = Op16<T>::mfma(av[s], bv[blk][s], gW[gi]);
    }
  };

  const int ntiles = (P + 31) >> 5;
  constexpr bool VEC4 = (F % 4 == 0) && (E % 4 == 0) && (F % 8 == 0) && (E % 8 == 0);
  // raw fp32 input rows of the NEXT tile are fetched while the current tile computes (HBM/L2 latency
  // is otherwise exposed at only 2 waves per SIMD)
  // IN16 (chosen by the launcher: pe stored as bf16, a 16-bit copy of xf in the operand format supplied, 16-byte rows): both input
  // halves arrive as 16-bit rows and the loaded bits ARE the tile's B operand -- no widening to fp32 and re-rounding per
  // tile (24 VALU instructions and 8 registers less; same-box A/B: -15 % kernel time)
  float4 raw[IN16 ? 1 : 2 * K0S];
  uint4 raw16[IN16 ? K0S : 1];
  auto fetch = [&](int tile) {
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
    if (E % 8 == 0 && a.pe_bf16 && h == 1) {   // bf16-stored pe: 16 B per 8 features, widened exactly (whatever F is: the
                                                // fp32 path below would read the bf16 array as floats, out of bounds)
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
  float ynext[16];
  auto fetch_targets = [&](int tile) {
    if (MODE == MODE_FWD) return;
    const int pp = tile * 32 + q;
    const int pcl = pp < P ? pp : P - 1;
    // wave-uniform row base (scalar registers) + a 32-bit per-lane offset: the loads take the saddr + voffset form instead
    // of a 64-bit address computed per lane and tile (P * C < 2^31 elements per row is checked by the launcher)
    const float* __restrict__ yrow = a.yin + (MODE == MODE_LOSS ? (long long)n : (long long)g) * P * C;
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
  // this workgroup's share of the 32-pixel tiles (all of them unless rcb_siren_desc.pixel_chunks > 1)
  const int t0 = (int)((long long)chunk * ntiles / a.chunks), t1 = (int)((long long)(chunk + 1) * ntiles / a.chunks);
  if (t0 + wave < t1) {
    fetch(t0 + wave);
    fetch_targets(t0 + wave);
  }
  for (int t = t0 + wave; t < t1; t += 4) {
#ifdef RCB_SIREN_STAMPS
    if (t == t0 + wave) RCB_STAMP(14);
    if (t == t0 + wave + 4) RCB_STAMP(15);
#endif
    const int p = t * 32 + q;
    const bool valid = p < P;
    const int pc = valid ? p : P - 1;                       // clamped: loads never leave the arrays
    // ---- layer-0 input fragments (from the prefetched rows), then prefetch the next tile -------------
    bf16x8 xin[K0S];
#pragma unroll
    for (int s = 0; s < K0S; ++s) {
      if constexpr (IN16) {
        xin[s] = rcb::op16::in16_operand<T>(raw16[s], h);
      } else {
        const float4 v0 = raw[2 * s], v1 = raw[2 * s + 1];
        xin[s][0] = (T)v0.x; xin[s][1] = (T)v0.y; xin[s][2] = (T)v0.z; xin[s][3] = (T)v0.w;
        xin[s][4] = (T)v1.x; xin[s][5] = (T)v1.y; xin[s][6] = (T)v1.z; xin[s][7] = (T)v1.w;
      }
    }
    // targets / upstream gradient of this tile: requested ONE TILE AHEAD, with the inputs (they are read once per step from
    // HBM; requested at the top of their own tile they arrive ~1 us later, i.e. after the forward pass has already reached the loss)
    float yv[16];
    if (MODE != MODE_FWD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) yv[r] = ynext[r];
    }
    fetch(t + 4 < t1 ? t + 4 : t);
    fetch_targets(t + 4 < t1 ? t + 4 : t);
    // ---- forward ----------------------------------------------------------------------------------
    // cosines: packed bf16 (8 registers per layer; unpacked again for dz = dh * cos) or, in the IN16 variant whose register
    // budget allows it, the fp32 values themselves (16 per layer, no pack / unpack: -72 VALU instructions per tile; the
    // product then uses the unrounded cosine)
    constexpr bool COS32 = IN16 && NB0 == 1;      // (two input blocks, F = 18: a second layer-0 gradient tile takes those registers)
    bf16x8 S[NH][2], Cs[COS32 ? 1 : NH][2];
    f32x16 Cf[COS32 ? NH : 1];
    f32x16 acc;
#pragma unroll
    for (int l = 0; l < NH; ++l) {
      const float* Bl = wl + G::off(l);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = Bl[rho(r, h)] * WS;
      if (l == 0) {
#pragma unroll
        for (int s = 0; s < K0S; ++s) acc = Op16<T>::mfma(FA(s), xin[s], acc);
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) acc = Op16<T>::mfma(FA(K0S + 2 * (l - 1) + s), S[l - 1][s], acc);
      }
      f32x16 sv, cv;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float tt = (WS == 1.0f) ? acc[r] : acc[r] * (1.0f / WS);   // already revolutions; the hardware reduces |tt| < 256 itself
        sv[r] = __builtin_amdgcn_sinf(tt);
        cv[r] = __builtin_amdgcn_cosf(tt);
      }
      S[l][0] = pack8<T>(sv, 0);
      S[l][1] = pack8<T>(sv, 1);
      if (MODE != MODE_FWD) {
        if constexpr (COS32) {
          Cf[l] = cv;
        } else {
          Cs[l][0] = pack8<T>(cv, 0);
          Cs[l][1] = pack8<T>(cv, 1);
        }
      }
    }
    {
      // output layer: the accumulator starts from the inline constant 0 and the bias is added to the C rows that exist
      // afterwards (a bias-initialised accumulator is a 16-register tuple of which 13 are zeros that have to be materialised
      // and moved into place every tile: ~20 instructions for three useful values)
      const float* Bl = wl + G::off(NL - 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) acc = Op16<T>::mfma(FA(K0S + 2 * (NH - 1) + s), S[NH - 1][s], acc);
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (rho(r, 0) < C || rho(r, 1) < C) {
          const float b = (rho(r, h) < C) ? Bl[rho(r, h) < C ? rho(r, h) : 0] : 0.f;
          acc[r] = (WS != 1.0f) ? (acc[r] * (1.0f / WS) + b) : (acc[r] + b);
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
      continue;
    }
#ifdef RCB_SIREN_STAMPS
    if (t == t0 + wave) RCB_STAMP(8);
#endif
    // ---- output gradient (branch-free: selects on the prefetched targets) ---------------------------------
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
    // ---- backward ----------------------------------------------------------------------------------
    constexpr bool DEFER = G::NSET == 2;
    bf16x8 av_d[2], bv_d[NB0][2];          // deferred form: the fragments of the layer ABOVE, read at the top of an iteration
#pragma unroll
    for (int l = NL - 1; l >= 0; --l) {
#ifdef RCB_SIREN_STAMPS
      if (t == t0 + wave) RCB_STAMP(9 + (NL - 1 - l));
#endif
      // the output layer's dz has C of its 32 rows set (registers r with rho(r, .) < C); packed straight from those, the other
      // entries as literal zeros: the generic pack8 of the tile made the compiler materialise the 13 zero registers of a
      // C = 3 tile and move them into place every tile (20 moves + 6 conversions of zeros per tile)
      bf16x8 dzb[2];
      if (l == NL - 1) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * s + j;
            dzb[s][j] = (rho(r, 0) < C || rho(r, 1) < C) ? (T)dz[r] : (T)0.0f;
          }
      } else {
        dzb[0] = pack8<T>(dz, 0);
        dzb[1] = pack8<T>(dz, 1);
      }
      if (DEFER && l < NL - 1) {
        // layer l + 1's images were written an iteration ago: their transposed reads go out NOW, in front of this layer's
        // data-gradient chain, and are consumed behind it
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        frag_read(l + 1, av_d, bv_d);
      }
      // (1) data gradient FIRST: dz -> dh -> dz of the next layer is the serial chain of the backward pass; the weight
      // gradient below (LDS transpose + MFMAs nothing waits for) then fills the latency of these MFMAs instead of delaying them
      if (l > 0) {
        f32x16 dh;
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = 0.f;
        if (l == NL - 1) {
          dh = Op16<T>::mfma(FA(NFA + 0), dzb[0], dh);
        } else {
          constexpr int dummy = 0;
          const int base = NFA + 1 + 2 * ((NH - 1) - l);
#pragma unroll
          for (int s = 0; s < 2; ++s) dh = Op16<T>::mfma(FA(base + s), dzb[s], dh);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {     // w0 is in the fragments
          if constexpr (COS32) dz[r] = dh[r] * Cf[l - 1][r];
          else dz[r] = dh[r] * (float)Cs[l - 1][r >> 3][r & 7];
        }
      } else if (a.dpe != nullptr) {
        f32x16 dx;
#pragma unroll
        for (int r = 0; r < 16; ++r) dx[r] = 0.f;
        const int base = NFA + 1 + 2 * (NH - 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) dx = Op16<T>::mfma(FA(base + s), dzb[s], dx);
        if (valid) {
          // (the lane half enters the address through an opaque copy: the row base then stays a scalar pair -- hoisted as a
          // per-lane 64-bit pointer it was spilled and reloaded once per tile behind a full vmcnt drain)
          int ln = threadIdx.x;
          asm volatile("" : "+v"(ln));
          const int hs = (ln >> 5) & 1;
          float* dst = a.dpe + (pe_row + pe_pix_off(a, p)) * E;
          if (E % 8 == 0 && a.pe_bf16) {
            __bf16* d16 = reinterpret_cast<__bf16*>(a.dpe) + pe_row * E;
            const int off16 = pe_pix_off(a, p) * E + 4 * hs;
#pragma unroll
            for (int g4 = 0; g4 < E / 8; ++g4) {
              typename Op16<__bf16>::v4 ob = {(__bf16)(dx[4 * g4] * (1.0f / (GS * WS))), (__bf16)(dx[4 * g4 + 1] * (1.0f / (GS * WS))),
                                              (__bf16)(dx[4 * g4 + 2] * (1.0f / (GS * WS))), (__bf16)(dx[4 * g4 + 3] * (1.0f / (GS * WS)))};
              *reinterpret_cast<typename Op16<__bf16>::v4*>(d16 + off16 + 8 * g4) = ob;
            }
          } else if (E % 8 == 0) {
            float* d32 = a.dpe + pe_row * E;
            const int off32 = pe_pix_off(a, p) * E + 4 * hs;
#pragma unroll
            for (int g4 = 0; g4 < E / 8; ++g4)
              *reinterpret_cast<float4*>(d32 + off32 + 8 * g4) = make_float4(dx[4 * g4] * (1.0f / (GS * WS)), dx[4 * g4 + 1] * (1.0f / (GS * WS)),
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
      // (2) weight gradient: [pixel][feature] images of dz and of the layer's input -> transposed reads -> 2 MFMAs per block.
      // Deferred form: the images of layer l are WRITTEN here and read back one layer later (frag_read(l + 1) at the top of
      // this iteration, its MFMAs behind the data-gradient chain), from the other image set.
      wg_write(l, dzb, S, xin);
      if (!DEFER) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        bf16x8 av[2], bv[NB0][2];
        frag_read(l, av, bv);
        wg_mfma(l, av, bv);
      } else if (l < NL - 1) {
        wg_mfma(l + 1, av_d, bv_d);
      }
    }
    if (DEFER) {                                        // layer 0's images: nothing left to overlap them with
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      frag_read(0, av_d, bv_d);
      wg_mfma(0, av_d, bv_d);
    }
  }
  if (MODE == MODE_FWD) return;
  RCB_STAMP(3);

  // ---- deterministic cross-wave reduction of the weight gradients -------------------------------------
  // per wave scratch: layer l as [out o][in i] rows of stride (32*NBl + 1) (conflict-free for the
  // lane = i writes and for the o-fastest reads below), followed by 32 bias slots
  __syncthreads();
  RCB_STAMP(4);
  float* part = smem + wave * G::RED_WAVE;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int no = G::lout(l), ro = G::roff(l);
    const int rs = 32 * ((l == 0) ? NB0 : 1) + 1;
    float bt = gb[l] + __shfl_xor(gb[l], 32, 64);
    if (h == 0) part[ro + no * rs + q] = bt;
#pragma unroll
    for (int blk = 0; blk < ((l == 0) ? NB0 : 1); ++blk) {
      const int gi = (l == 0) ? blk : (l + NB0 - 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (rho(r, 0) < no || rho(r, 1) < no) {
          int o = rho(r, h);
          if (o < no) part[ro + o * rs + 32 * blk + q] = gW[gi][r];
        }
      }
    }
  }
  if (MODE == MODE_LOSS) {                                 // the loss sums travel with the same barrier
    const float v = wave_sum(sse_local);
    if (lane == 0) smem[4 * G::RED_WAVE + wave] = v;
  }
  __syncthreads();
  RCB_STAMP(5);
  if (MODE == MODE_LOSS && tid == 0)
    a.sse[(long long)chunk * a.G + g] = ((smem[4 * G::RED_WAVE] + smem[4 * G::RED_WAVE + 1]) + smem[4 * G::RED_WAVE + 2]) + smem[4 * G::RED_WAVE + 3];
  {
    float* dst = a.dwvec + ((long long)chunk * a.G + g) * a.w_stride;
    // layer by layer with compile-time shapes (constant divisors, no layer search per element)
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int ol = G::off(l), no = G::lout(l), ro = G::roff(l), rs = 32 * ((l == 0) ? NB0 : 1) + 1;
      const int size = no * (G::lin(l) + 1);
      for (int e = tid; e < size; e += 256) {
        int src;
        if (e < no) {
          src = ro + no * rs + e;                 // bias
        } else {
          const int i = (e - no) / no, o = (e - no) - i * no;
          src = ro + o * rs + i;
        }
        const float v = (((smem[src] + smem[G::RED_WAVE + src]) + smem[2 * G::RED_WAVE + src]) + smem[3 * G::RED_WAVE + src]) * (1.0f / GS);
        if (a.dwvec != nullptr) dst[ol + e] = v;
        if (a.dw16 != nullptr) {                  // high plane: operand of the weight-gradient GEMM; with the low plane the
          const __bf16 hb = (__bf16)v;            // pair is the A transform's data-gradient operand (rcb_atrans_apply x_hi / x_lo)
          a.dw16[(long long)g * a.dw16_stride + ol + e] = hb;
          if (a.dwlo != nullptr) a.dwlo[(long long)g * a.dw16_stride + ol + e] = (__bf16)(v - (float)hb);
        }
      }
    }
  }
  RCB_STAMP(6);
  RCB_STAMP(7);
  if (a.clock_probe != nullptr && blockIdx.x < 256 && tid == 0) {
    a.clock_probe[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
    a.clock_probe[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
  }
}

template <typename T, int NH, int F, int E, int C, int MODE, bool IN16>
int launch_one(const SirenArgs& a, hipStream_t st) {
  using G = Geo<NH, F, E, C>;
    auto kfn = siren_bf16_kernel<T, NH, F, E, C, MODE, IN16>;
  // (per launch: the attribute belongs to the (function, device) pair; a process may drive several devices)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024);
  if (e != hipSuccess) return fail((int)e, "siren(bf16): hipFuncSetAttribute: %s", hipGetErrorString(e));
  kfn<<<a.G * a.chunks, 256, G::LDS_BYTES, st>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

template <typename T, int NH, int F, int E, int C>
int launch_mode(int mode, const SirenArgs& a, hipStream_t st) {
  // both input halves as 16-bit rows: pe stored as bf16, a 16-bit copy of xf in the operand format supplied (rows padded to 8 features)
  constexpr bool can16 = (E % 8 == 0) && E > 0;
  if (can16 && a.pe_bf16 && a.xf16 != nullptr && mode == MODE_LOSS) return launch_one<T, NH, F, E, C, MODE_LOSS, can16>(a, st);
  if (mode == MODE_FWD) return launch_one<T, NH, F, E, C, MODE_FWD, false>(a, st);
  if (mode == MODE_BWD) return launch_one<T, NH, F, E, C, MODE_BWD, false>(a, st);
  return launch_one<T, NH, F, E, C, MODE_LOSS, false>(a, st);
}

}  // namespace

namespace rcb {
int siren_bf16_dispatch(int mode, const rcb_siren_desc* d, SirenArgs& a, hipStream_t st) {
#define RCB_CASE(NHv, Fv, Ev, Cv)                                                                   \
  if (d->n_hidden == NHv && d->fourier_dim == Fv && d->pe_dim == Ev && d->out_dim == Cv)              \
    return d->precision == 2 ? launch_mode<_Float16, NHv, Fv, Ev, Cv>(mode, a, st)                  \
                             : launch_mode<__bf16, NHv, Fv, Ev, Cv>(mode, a, st);
  RCB_CASE(3, 16, 16, 3)   // cifar / kodak / protein
  RCB_CASE(3, 16, 16, 1)   // audio
  RCB_CASE(3, 18, 16, 3)   // video
  RCB_CASE(2, 16, 16, 3)   // 2 hidden layers
#undef RCB_CASE
  return fail(RCB_ERR_UNSUPPORTED, "siren(bf16): geometry n_hidden=%d F=%d E=%d C=%d is not instantiated", d->n_hidden,
              d->fourier_dim, d->pe_dim, d->out_dim);
}
}  // namespace rcb
