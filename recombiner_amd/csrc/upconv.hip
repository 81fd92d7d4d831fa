// N1: the upsampling net's  nearest-upsample(2) -> conv3x3(pad 1)  stages in sub-pixel ("phase") form,
// bf16 MFMA with fp32 accumulation.  Replaces prior_model.py:52-54 (up2/conv2/act2/up3/conv3) without
// ever materialising the up-sampled intermediates.
//
// For output pixel (2i+a, 2j+b) only a 2x2 window of SOURCE pixels is touched:
//     y[2i+a, 2j+b, co] = bias[co] + sum_{ty,tx in {0,1}} sum_ci  Weff[ty,tx,ci,a,b,co] * x[i+a+ty-1, j+b+tx-1, ci]
// with Weff the kernel taps pre-summed per (phase, window tap) (host: upsample_fast.PhaseStage.eff_weight,
// layout [ty][tx][ci][a][b][co] fp32).  All images are channel-last; Cin = 64.
//
// Three kernel families, all persistent (one workgroup per CU walking its INRs, the next INR's images prefetched into
// registers), with the *position on the MFMA lane* (the B / D column) and channels on rows:
//   forward : D[co, pos] += A[co, (tap,ci)] * B[(tap,ci), pos]    A = weight fragments resident in registers, B from the LDS image of x
//   dgrad   : D[ci, pos] += A[ci, (combo,co)] * B[(combo,co), pos] A = transposed weight fragments, B from the LDS image of dy
//   wgrad   : D[ci, co]  += A[ci, pos] * B[pos, co]                per-INR images staged in LDS, transposed reads
// LeakyReLU(0.01) is fused: forward applies it on load (fp32 pre-activation input) or in the epilogue;
// dgrad multiplies by its derivative taken from the sign of the stored activation.
#include "rcb_common.h"

using namespace rcb;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CIN = 64;
#ifndef RCB_UP_HOT
#define RCB_UP_HOT 0       // ablation builds (wrong results): the stage-2 forward / data gradient read one cache-resident INR
#endif
#ifndef RCB_UP_NOSTORE
#define RCB_UP_NOSTORE 0   // ablation builds (wrong results): 1 stage-2 forward, 2 stage-2 data gradient, 3 stage-3 forward without their stores
#endif
#ifndef RCB_B3_PF
#define RCB_B3_PF 2          // taps whose LDS reads are in flight ahead of their MFMAs in the stage-3 backward's data gradient
#endif
#ifndef RCB_UP_PF
#define RCB_UP_PF 3          // image gathers in flight ahead of their MFMAs in the register-fragment kernels
#endif
constexpr int XRS = 72;   // row stride (elements) of a staged 64-channel image: 144-byte rows spread 128-byte-strided
                          // gathers (consecutive positions, one channel chunk) over all LDS banks
constexpr float SLOPE = 0.01f;

__device__ __forceinline__ constexpr int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : v * SLOPE; }

__host__ __device__ constexpr long long weff_index(int ty, int tx, int ci, int a, int b, int co, int cout) {
  return ((((long long)(ty * 2 + tx) * CIN + ci) * 2 + a) * 2 + b) * cout + co;
}

union Frag {
  bf16x8 v;
  uint4 u;
};

// the value must exist in registers here: stops the compiler from sinking a prologue's conversions into the main loop
// (where their operands would stay live, and spill, across it)
__device__ __forceinline__ void pin(uint4& u) { asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w)); }

#ifndef RCB_WC_STAMPS
#define RCB_WC_STAMPS 0    // diagnostic build: s_memrealtime (100 MHz) of workgroup 0 at kernel entry / loop start / loop end / kernel end
#endif
#if RCB_WC_STAMPS
__device__ unsigned long long g_wc_stamps[2][4];      // [0]: upconv_wgrad_kernel, [1]: upconv_fwd3_lds_kernel (last launch of each)
#define WC_T(which, k)                                                                            \
  do {                                                                                            \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_wc_stamps[which][k] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define WC_T(which, k) do { } while (0)
#endif

// raw (unconverted) 8-channel loads: the next INR's image is requested into registers while the current one is
// consumed, and converted (mode 1/3: with LeakyReLU) only when it is committed to LDS
template <int MODE> struct Raw8;
template <> struct Raw8<0> { uint4 u; };
template <> struct Raw8<1> { float4 a, b; };
template <> struct Raw8<2> { float4 a, b; };
template <> struct Raw8<3> { uint4 u; };

template <int MODE>
__device__ __forceinline__ Raw8<MODE> raw_load(const void* base, long long elem_off) {
  Raw8<MODE> r;
  if constexpr (MODE == 0 || MODE == 3) {
    r.u = *reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(base) + elem_off);
  } else {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem_off);
    r.a = p[0];
    r.b = p[1];
  }
  return r;
}

template <int MODE>
__device__ __forceinline__ bf16x8 raw_frag(const Raw8<MODE>& r, bool ok) {
  Frag f;
  if constexpr (MODE == 0 || MODE == 3) {
    f.u = ok ? r.u : make_uint4(0, 0, 0, 0);
    if constexpr (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)lrelu((float)f.v[j]);
    }
  } else {
    float t[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = ok ? t[j] : 0.f;
      if (MODE == 1) x = lrelu(x);
      f.v[j] = (__bf16)x;
    }
  }
  return f.v;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
struct FwdArgs {
  const void* x;
  const float* weff;
  const float* bias;
  void* y;
  int batch;
  const uint4* pack;   // nullable: pre-packed fragments (rcb_upconv_weff_build), else built from weff in the prologue
};

// ------------------------------------------------------------------------------------------------
// data gradient
// ------------------------------------------------------------------------------------------------
struct DgradArgs {
  const void* dy;
  const float* weff;
  const void* x;   // stored activation (bf16 post-LeakyReLU) or fp32 pre-activation: sign source
  void* dx;
  int batch;
  const uint4* pack;   // nullable, as in FwdArgs
};

// ------------------------------------------------------------------------------------------------
// stage-3 variants (G = 16: one INR = 8 tiles = one 8-wave workgroup pass): the INR's source image is
// staged once in LDS (zero halo, padded rows: conflict-free 16-byte gathers), so every gather is an LDS
// read instead of an L2 round trip; the next INR's image is prefetched into registers during compute.
// ------------------------------------------------------------------------------------------------
// Forward: wave w owns output-row phase pa = w & 1 and the source tiles 2 * (w >> 1), + 1 (32 positions each), both
// column phases.  Its 32 weight fragments ([pb][ty][tx][kb]) stay in registers for the whole kernel, so LDS only
// serves the image: 24 fragment reads feed 32 MFMAs per tile.  In the epilogue the two lane halves swap
// (v_permlane32_swap) so that each lane owns all 16 channels of one output pixel and stores them contiguously.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16x32(bf16x8 a_, bf16x8 b_, f32x4v c_) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0);
}

// One wave, one output-row phase pa, two 32-position tiles of the INR: 16 output channels are ONE row block of
// v_mfma_f32_16x16x32_bf16 (the 32 x 32 x 16 form spends half of every MFMA on padding rows), so a tile is two position halves
// (image rows) x two column phases = four 16 x 16 accumulators, K = 64 input channels = two MFMAs per (phase, tap).
//   A (weights)  [16 co x 32 ci]: lane l = row co = l & 15, k-group kg = l >> 4: ci = 32 kb2 + 8 kg + j   -> fr[pb][ty][tx][kb2]
//   B (image)    [32 ci x 16 pos]: lane l = position l & 15 of the image row, the 16-byte chunk 4 kb2 + kg of its pixel
//   D            [16 co x 16 pos]: lane l = position l & 15, channels 4 kg + r
// 24 gathers feed 32 MFMAs per tile, issued PF ahead of their use (ring of PF + 1 registers, order pinned by scheduling
// barriers).  Epilogue: v_permlane16_swap between neighbouring k-groups turns (4 channels of pixel 2j, 4 of pixel 2j + 1) into
// 8 consecutive channels of ONE pixel per lane: 16-byte stores, 1 KB contiguous per instruction.
template <int COUT, int OUT_BF16>
__device__ __forceinline__ void fwd3_body(const __bf16* img, const uint4 (&fr)[2][2][2][2], const FwdArgs& a, int b,
                                          int pa, int tp, int lane, const float (&bia)[4]) {
  constexpr int G = 16, HG = 18, RS = 64;
  const int j = lane & 15, kg = lane >> 4;
#pragma unroll 1
  for (int tt = 0; tt < 2; ++tt) {
    const int i0 = 2 * (2 * tp + tt);              // the tile's two image rows: i0, i0 + 1
    f32x4v acc[2][2];   // [position half][pb]
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ph][pb][r] = 0.f;
    const __bf16* base = img + ((i0 + pa) * HG + j) * RS;
    constexpr int NG = 24, PF = RCB_UP_PF;
    uint4 ring[PF + 1];
    auto gather = [&](int g) {                    // g = ((ty * 3 + dxi) * 2 + kb2) * 2 + ph
      const int ph = g & 1, kb2 = (g >> 1) & 1, td = g >> 2, ty = td / 3, dxi = td - 3 * ty;
      const int sw = ((j + dxi) >> 1) & 7;        // chunk swizzle of the image (see the kernel)
      return *reinterpret_cast<const uint4*>(base + ((ph + ty) * HG + dxi) * RS + 8 * ((4 * kb2 + kg) ^ sw));
    };
#pragma unroll
    for (int g = 0; g < PF; ++g) ring[g] = gather(g);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + PF < NG) ring[(g + PF) % (PF + 1)] = gather(g + PF);
      __builtin_amdgcn_sched_barrier(0);
      const int ph = g & 1, kb2 = (g >> 1) & 1, td = g >> 2, ty = td / 3, dxi = td - 3 * ty;
      Frag bf, fa;
      bf.u = ring[g % (PF + 1)];
      if (dxi <= 1) {
        fa.u = fr[0][ty][dxi][kb2];
        acc[ph][0] = mfma16x32(fa.v, bf.v, acc[ph][0]);
      }
      if (dxi >= 1) {
        fa.u = fr[1][ty][dxi - 1][kb2];
        acc[ph][1] = mfma16x32(fa.v, bf.v, acc[ph][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      // lane (j, kg) holds channels 4 kg + r of pixels 2j [pb 0] and 2j + 1 [pb 1]; after the swap: channels 8 (kg >> 1) .. + 7 of
      // pixel 2j + (kg & 1)
      float lo[4], hi[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x0 = acc[ph][0][r] + bia[r], x1 = acc[ph][1][r] + bia[r];
        auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0), __float_as_uint(x1), false, false);
        lo[r] = __uint_as_float(sw[0]);
        hi[r] = __uint_as_float(sw[1]);
      }
      const long long opix = ((long long)b * (2 * G) + 2 * (i0 + ph) + pa) * (2 * G) + 2 * j + (kg & 1);
      if (OUT_BF16) {
        Frag o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o.v[k] = (__bf16)lo[k];
          o.v[4 + k] = (__bf16)hi[k];
        }
        if (RCB_UP_NOSTORE != 3 || a.batch < 0)
          *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(a.y) + opix * COUT + 8 * (kg >> 1)) = o.u;
      } else {
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + opix * COUT + 8 * (kg >> 1));
        dst[0] = make_float4(lo[0], lo[1], lo[2], lo[3]);
        dst[1] = make_float4(hi[0], hi[1], hi[2], hi[3]);
      }
    }
  }
}

// NW waves per workgroup: 8 (one workgroup per CU), or 4 with two workgroups per CU -- each on its own INR, so one stages its
// image and waits for its loads while the other computes (a single workgroup runs stage -> barrier -> compute in lockstep and
// leaves the matrix cores idle during the first and HBM idle during most of the second); a wave then owns two tile pairs.
template <int COUT, int OUT_BF16, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 3 : 1) upconv_fwd3_lds_kernel(FwdArgs a) {
  static_assert(COUT == 16, "epilogue lane swap is written for 16 output channels");
  WC_T(1, 0);
  static_assert(NW == 8 || NW == 4, "8 or 4 waves");
  // image [18][18][64] in LDS, the 16-byte chunk c of pixel column x stored at c ^ ((x >> 1) & 7): the gathers of fwd3_body
  // (lane = source position, consecutive pixels, one chunk per instruction) cover all 64 banks once per ds_read_b128 lane
  // group (tools/lds_banks.py; 144-byte pixel rows without the swizzle cost 2 cycles per group).  Same-box A/B: 66.3 -> 63.7 us.
  // The same treatment of the other kernels' images removed their conflicts too (SQ_LDS_BANK_CONFLICT 40-55 % -> 0-1 % of the
  // LDS cycles) but not their time -- stage-2 forward 46.9 -> 48.0 us, stage-2 weight gradient 44.6 -> 47.8, stage-3 backward
  // 117.0 -> 121.5: the XOR per read address costs them more than the conflicts did -- and was not kept there.
  constexpr int G = 16, HG = 18, RS = 64;
  constexpr int NT = 64 * NW, NPRE = 2048 / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* img = reinterpret_cast<__bf16*>(smem_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int pa = wave & 1, tp = wave >> 1;
  uint4 fr[2][2][2][2];   // [pb][ty][tx][kb2]: A operands of v_mfma_f32_16x16x32_bf16 (row = co = lane & 15, ci = 32 kb2 + 8 (lane >> 4) + j)
#pragma unroll
  for (int pb = 0; pb < 2; ++pb)
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int tx = 0; tx < 2; ++tx)
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
          Frag f;
          if (a.pack) {                     // one coalesced 16-byte load per fragment
            f.u = a.pack[16384 + ((((pa * 2 + pb) * 2 + ty) * 2 + tx) * 2 + kb2) * 64 + lane];
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              f.v[j] = (__bf16)a.weff[weff_index(ty, tx, 32 * kb2 + 8 * (lane >> 4) + j, pa, pb, lane & 15, COUT)];
          }
          fr[pb][ty][tx][kb2] = f.u;
        }
  // (pinned in a loop of their own: an empty asm that takes a loaded value waits for THAT load, so pinning inside the load loop
  // serialised the loads -- 32 dependent L2 / HBM round trips, 12.9 us of the stage-2 forward's 54 us by its own stamps)
#pragma unroll
  for (int pb = 0; pb < 2; ++pb)
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int tx = 0; tx < 2; ++tx)
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) pin(fr[pb][ty][tx][kb2]);
  float bia[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bia[r] = a.bias[4 * (lane >> 4) + r];
  for (int e = tid; e < HG * HG * RS / 8; e += NT) reinterpret_cast<uint4*>(img)[e] = make_uint4(0, 0, 0, 0);
  // each thread stages NPRE x 16 B of the 32 KB image (element e -> pixel e / 8, chunk e % 8); the next INR's image is
  // in flight while the current one is consumed
  uint4 pre[NPRE];
#pragma unroll
  for (int k = 0; k < NPRE; ++k) pre[k] = make_uint4(0, 0, 0, 0);
#define RCB_FETCH3(bb)                                                                                          \
  {                                                                                                            \
    const uint4* src_ = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.x) + (long long)(bb) * G * G * CIN); \
    _Pragma("unroll") for (int k = 0; k < NPRE; ++k) pre[k] = src_[tid + NT * k];                               \
  }
  const int gs = gridDim.x;
  int b = blockIdx.x;
  if (b < a.batch) RCB_FETCH3(b)
  WC_T(1, 1);
  for (; b < a.batch; b += gs) {
    __syncthreads();          // everyone is done with the previous image (and with the halo setup)
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int e_ = tid + NT * k, pix_ = e_ >> 3, c8_ = e_ & 7, x_ = (pix_ & 15) + 1;
      *reinterpret_cast<uint4*>(img + (((pix_ >> 4) + 1) * HG + x_) * RS + 8 * (c8_ ^ ((x_ >> 1) & 7))) = pre[k];
    }
    __syncthreads();
    if (b + gs < a.batch) RCB_FETCH3(b + gs)
#pragma unroll 1
    for (int t2 = 0; t2 < 8 / NW; ++t2) fwd3_body<COUT, OUT_BF16>(img, fr, a, b, pa, tp + (NW / 2) * t2, lane, bia);
  }
  WC_T(1, 2);
  WC_T(1, 3);
#undef RCB_FETCH3
}

// one INR's 16x16x64 input gradient from the LDS dy image; the sign source is requested before the MFMA loop
template <int COUT>
__device__ __forceinline__ void dgrad3_body(const __bf16* img, const uint4* frags, const DgradArgs& a, int b, int u, int v,
                                            int h, int lane, int wave, int q) {
  constexpr int G = 16, HO = 34, RS = 24;
  const long long xpix = ((long long)b * G * G + wave * 32 + q) * CIN;
  // (lane (q, h) ends up with the 16 consecutive channels 32 mt + 16 h .. + 15 of its pixel -- the lane halves swap below --
  // so the sign source is two 16-byte loads per block and the result two 16-byte stores, not four 8-byte pieces each)
  Frag xs[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
      xs[mt][hf].u = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.x) + xpix + 32 * mt + 16 * h)[hf];
  __builtin_amdgcn_sched_barrier(0);
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    const int ry = (n >> 2) - 1, rx = (n & 3) - 1;
    Frag bf;
    bf.u = *reinterpret_cast<const uint4*>(img + ((2 * u + ry + 1) * HO + (2 * v + rx + 1)) * RS + 8 * h);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      Frag fa;
      fa.u = frags[(n * 2 + mt) * 64 + lane];
      acc[mt] = mfma16(fa.v, bf.v, acc[mt]);
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float o[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mt][4 * hf + k]), __float_as_uint(acc[mt][8 + 4 * hf + k]),
                                                   false, false);
        o[k] = __uint_as_float(sw[0]);
        o[4 + k] = __uint_as_float(sw[1]);
      }
      Frag ob;
#pragma unroll
      for (int k = 0; k < 8; ++k) ob.v[k] = (__bf16)(o[k] * ((float)xs[mt][hf].v[k] > 0.f ? 1.0f : SLOPE));
      reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(a.dx) + xpix + 32 * mt + 16 * h)[hf] = ob.u;
    }
  }
}

template <int COUT, int DY_BF16>
__global__ void __launch_bounds__(512) upconv_dgrad3_lds_kernel(DgradArgs a) {
  constexpr int G = 16, OG = 32, HO = 34, RS = 24;   // dy image: [34][34] pixels x 16 channels (+8 pad = 48 B rows)
  constexpr int NF = 16 * 2;                          // KB = 1, [combo][mt]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* frags = reinterpret_cast<uint4*>(smem_raw);
  __bf16* img = reinterpret_cast<__bf16*>(smem_raw + NF * 1024);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane & 31, h = lane >> 5;
  for (int e = tid; e < NF * 64; e += 512) {
    if (a.pack) {                      // [combo][mt][lane] is exactly the layout of the LDS fragment table
      frags[e] = a.pack[20480 + e];
      continue;
    }
    const int ln = e & 63, slot = e >> 6;
    const int mt = slot & 1, combo = slot >> 1;
    const int ry = (combo >> 2) - 1, rx = (combo & 3) - 1;
    const int pa = (ry & 1), ty = (ry <= 0) ? 1 : 0;
    const int pb = (rx & 1), tx = (rx <= 0) ? 1 : 0;
    const int fq = ln & 31, fh = ln >> 5, ci = 32 * mt + fq;
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)a.weff[weff_index(ty, tx, ci, pa, pb, 8 * fh + j, COUT)];
    frags[e] = f.u;
  }
  for (int e = tid; e < HO * HO * RS / 8; e += 512) reinterpret_cast<uint4*>(img)[e] = make_uint4(0, 0, 0, 0);
  // dy image of one INR: 32*32 pixels x 16 channels: fp32 = 4096 float4 (8 per thread, converted to bf16 on
  // commit), bf16 = 2048 x 16 B (4 per thread, committed as they are).  Two register sets: two INRs in flight.
  constexpr int NPRE = DY_BF16 ? 4 : 8;
  float4 preA[NPRE], preB[NPRE];
#pragma unroll
  for (int k = 0; k < NPRE; ++k) preA[k] = preB[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#define RCB_FETCHD(set, bb)                                                                                          \
  {                                                                                                                 \
    const float4* src_ = DY_BF16                                                                                    \
        ? reinterpret_cast<const float4*>(reinterpret_cast<const __bf16*>(a.dy) + (long long)(bb) * OG * OG * COUT) \
        : reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.dy) + (long long)(bb) * OG * OG * COUT); \
    _Pragma("unroll") for (int k = 0; k < NPRE; ++k) set[k] = src_[tid + 512 * k];                                  \
  }
#define RCB_COMMITD(set)                                                                                             \
  _Pragma("unroll") for (int k = 0; k < NPRE; ++k) {                                                                \
    if (DY_BF16) {                                                                                                  \
      const int e_ = tid + 512 * k, pix_ = e_ >> 1, c8_ = e_ & 1;                                                   \
      *reinterpret_cast<float4*>(img + (((pix_ >> 5) + 1) * HO + ((pix_ & 31) + 1)) * RS + 8 * c8_) = set[k];       \
    } else {                                                                                                        \
      const int e_ = tid + 512 * k, pix_ = e_ >> 2, c4_ = e_ & 3;                                                   \
      bf16x4 v_ = {(__bf16)set[k].x, (__bf16)set[k].y, (__bf16)set[k].z, (__bf16)set[k].w};                         \
      *reinterpret_cast<bf16x4*>(img + (((pix_ >> 5) + 1) * HO + ((pix_ & 31) + 1)) * RS + 4 * c4_) = v_;           \
    }                                                                                                               \
  }
  const int gs = gridDim.x;
  int b = blockIdx.x;
  if (b < a.batch) RCB_FETCHD(preA, b)
  if (b + gs < a.batch) RCB_FETCHD(preB, b + gs)
  const int u = (wave * 32 + q) >> 4, v = (wave * 32 + q) & 15;
  for (; b < a.batch; b += 2 * gs) {
    __syncthreads();
    RCB_COMMITD(preA)
    __syncthreads();
    if (b + 2 * gs < a.batch) RCB_FETCHD(preA, b + 2 * gs)
    dgrad3_body<COUT>(img, frags, a, b, u, v, h, lane, wave, q);
    if (b + gs < a.batch) {
      __syncthreads();
      RCB_COMMITD(preB)
      __syncthreads();
      if (b + 3 * gs < a.batch) RCB_FETCHD(preB, b + 3 * gs)
      dgrad3_body<COUT>(img, frags, a, b + gs, u, v, h, lane, wave, q);
    }
  }
#undef RCB_FETCHD
#undef RCB_COMMITD
}

// ------------------------------------------------------------------------------------------------
// stage-2 variants (G = 8, 64 -> 64 channels) with register-resident weight fragments.
//
// forward: wave w = (INR of the pair w >> 2, output phase w & 3); it owns both 32-position tiles of its INR and both
// halves of the 64 output channels: 32 fragments in registers, 16 image reads feed 32 MFMAs per tile.  Two INRs per
// pass halve the barriers per INR; LeakyReLU of the pre-activation input is applied once, when the image is staged.
// ------------------------------------------------------------------------------------------------
#ifndef RCB_F2_STAMPS
#define RCB_F2_STAMPS 0    // diagnostic build: ticks per phase of workgroup 0's waves (rcb_debug_f2_stamps, tools/f2_stamps.py)
#endif
#if RCB_F2_STAMPS
__device__ unsigned long long g_f2_stamps[8 * 8];
#define F2_T(k)                                                     \
  do {                                                              \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
    ph_[k] += now_ - last_;                                         \
    last_ = now_;                                                   \
  } while (0)
#else
#define F2_T(k) do { } while (0)
#endif
template <int IN_MODE>   // 1: fp32 pre-activation, 3: bf16 pre-activation
__global__ void __launch_bounds__(512) upconv_fwd2_reg_kernel(FwdArgs a) {
  constexpr int G = 8, HG = 10, RS = 72, COUT = 64, IMG = HG * HG * RS;
#if RCB_F2_STAMPS
  const unsigned long long t_entry_ = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int NI = 4;                                               // INRs per pass
  __bf16* img = reinterpret_cast<__bf16*>(smem_raw);                  // [NI][IMG]
  float* bs = reinterpret_cast<float*>(smem_raw + NI * IMG * 2);     // [64]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane & 31, h = lane >> 5;
  const int ph = wave & 3, pa = ph >> 1, pb = ph & 1, half = wave >> 2;   // the wave owns INRs 2 half, 2 half + 1 of a pass
  uint4 fr[2][2][2][4];   // [mt][ty][tx][kb]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int tx = 0; tx < 2; ++tx)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          Frag f;
          if (a.pack) {
            f.u = a.pack[((((ph * 2 + mt) * 2 + ty) * 2 + tx) * 4 + kb) * 64 + lane];
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)a.weff[weff_index(ty, tx, 16 * kb + 8 * h + j, pa, pb, 32 * mt + q, COUT)];
          }
          fr[mt][ty][tx][kb] = f.u;
        }
  // (pinned in a loop of their own: an empty asm that takes a loaded value waits for THAT load, so pinning inside the load loop
  // serialised the loads -- 32 dependent L2 / HBM round trips, 12.9 us of the stage-2 forward's 54 us by its own stamps)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int tx = 0; tx < 2; ++tx)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) pin(fr[mt][ty][tx][kb]);
  for (int e = tid; e < NI * IMG / 8; e += 512) reinterpret_cast<uint4*>(img)[e] = make_uint4(0, 0, 0, 0);
  if (tid < COUT) bs[tid] = a.bias[tid];
  // NI INRs = NI x 64 pixels x 8 chunks of 8 channels = 2048 chunks, 4 per thread: 32 KB (bf16) in flight per CU
  Raw8<IN_MODE> pre[NI];
  const int npair = (a.batch + NI - 1) / NI;
#define RCB_FETCH2(pp)                                                                      \
  _Pragma("unroll") for (int k = 0; k < NI; ++k) {                                         \
    const int e_ = tid + 512 * k;                                                          \
    int b_ = NI * (pp) + (e_ >> 9);                                                        \
    if (b_ >= a.batch) b_ = a.batch - 1;                                                   \
    pre[k] = raw_load<IN_MODE>(a.x, (long long)(RCB_UP_HOT ? (int)(blockIdx.x & 1023) : b_) * G * G * CIN + 8 * (e_ & 511));         \
  }
  const int gs = gridDim.x;
  int p = blockIdx.x;
  if (p < npair) RCB_FETCH2(p)
#if RCB_F2_STAMPS
  unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
  ph_[5] = __builtin_amdgcn_s_memrealtime();      // 100 MHz wall clock: loop start / end in slots 5, 6, kernel entry in 7
#endif
  for (; p < npair; p += gs) {
    __syncthreads();
    F2_T(0);
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int e_ = tid + 512 * k, pix = (e_ >> 3) & 63, c8 = e_ & 7;
      Frag f;
      f.v = raw_frag<IN_MODE>(pre[k], true);
      *reinterpret_cast<uint4*>(img + (e_ >> 9) * IMG + (((pix >> 3) + 1) * HG + ((pix & 7) + 1)) * RS + 8 * c8) = f.u;
    }
    F2_T(1);
    __syncthreads();
    F2_T(2);
    if (p + gs < npair) RCB_FETCH2(p + gs)
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {         // (INR of this wave, tile) pairs, one after the other
      const int inr = 2 * half + (it >> 1), tt = it & 1;
      const int b = NI * p + inr;
      if (b < a.batch) {
        const __bf16* im = img + inr * IMG;
        const int pos = tt * 32 + q, i = pos >> 3, j = pos & 7;
        f32x16 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        const __bf16* base = im + ((i + pa) * HG + (j + pb)) * RS + 8 * h;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
          for (int tx = 0; tx < 2; ++tx)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
              Frag bf, fa;
              bf.u = *reinterpret_cast<const uint4*>(base + (ty * HG + tx) * RS + 16 * kb);
#pragma unroll
              for (int mt = 0; mt < 2; ++mt) {
                fa.u = fr[mt][ty][tx][kb];
                acc[mt] = mfma16(fa.v, bf.v, acc[mt]);
              }
            }
        __builtin_amdgcn_sched_barrier(0);
        F2_T(3);
        // bias + LeakyReLU on the original channels, then the lane halves swap so that lane (q, h) owns the 16
        // consecutive channels 32 mt + 16 h .. + 15 of its pixel
        const long long opix = ((long long)b * (2 * G) + 2 * i + pa) * (2 * G) + 2 * j + pb;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          float v[16];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bb = *reinterpret_cast<const float4*>(bs + 32 * mt + 8 * g4 + 4 * h);
            v[4 * g4 + 0] = lrelu(acc[mt][4 * g4 + 0] + bb.x);
            v[4 * g4 + 1] = lrelu(acc[mt][4 * g4 + 1] + bb.y);
            v[4 * g4 + 2] = lrelu(acc[mt][4 * g4 + 2] + bb.z);
            v[4 * g4 + 3] = lrelu(acc[mt][4 * g4 + 3] + bb.w);
          }
          float xa[8], ya[8];
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[m]), __float_as_uint(v[8 + m]), false, false);
            xa[m] = __uint_as_float(sw[0]);   // channel 32 mt + 16 h + (m & 3) + 8 (m >> 2)
            ya[m] = __uint_as_float(sw[1]);   //   ... + 4
          }
          Frag o0, o1;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            o0.v[k] = (__bf16)xa[k];     o0.v[4 + k] = (__bf16)ya[k];
            o1.v[k] = (__bf16)xa[4 + k]; o1.v[4 + k] = (__bf16)ya[4 + k];
          }
          uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(a.y) + opix * COUT + 32 * mt + 16 * h);
          if (RCB_UP_NOSTORE != 1 || a.batch < 0) {
            dst[0] = o0.u;
            dst[1] = o1.u;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        F2_T(4);
      }
    }
  }
#if RCB_F2_STAMPS
  ph_[6] = __builtin_amdgcn_s_memrealtime();
  ph_[7] = t_entry_;
  if (blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 8; ++k) g_f2_stamps[wave * 8 + k] = ph_[k];
#endif
#undef RCB_FETCH2
}

// data gradient of stage 2: D[ci, pos] = sum over the 16 window combos and 64 output channels.  The 128 fragments do
// not fit one wave, so wave w = (ci half w & 1, tile (w >> 1) & 1, combo half w >> 2) keeps 32 of them and the two
// combo halves are added through LDS.  The epilogue (combo-half 0 waves) multiplies by LeakyReLU'(x), swaps lane halves
// for 32-byte stores and accumulates the per-channel sums of dx (the bias gradient of the stage before) per workgroup.
#ifndef RCB_D2_STAMPS
#define RCB_D2_STAMPS 0    // diagnostic build: cycles per phase of workgroup 0's waves (rcb_debug_d2_stamps, tools/d2_stamps.py)
#endif
#if RCB_D2_STAMPS
__device__ unsigned long long g_d2_stamps[8 * 8];
#define D2_T(k)                                                     \
  do {                                                              \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
    ph_[k] += now_ - last_;                                         \
    last_ = now_;                                                   \
  } while (0)
#else
#define D2_T(k) do { } while (0)
#endif
template <int X_F32>   // sign source / dx type: 1 fp32, 0 bf16
__global__ void __launch_bounds__(512) upconv_dgrad2_reg_kernel(DgradArgs a, float* __restrict__ dbias_partial) {
  // dy image [18][18][64] in LDS: 128-byte pixels, rows padded by 64 B, and the 16-byte chunk c of pixel column x stored at
  // c ^ ((x >> 1) & 7): every gather below (lane = output position, pixels two apart, one chunk per instruction) then covers
  // all 64 banks once per ds_read_b128 lane group (tools/lds_banks.py; the plain padded image cost 4 cycles per group)
  constexpr int G = 8, OG = 16, HO = 18, RS = 64, ROWS = HO * RS + 32, COUT = 64, IMG = HO * ROWS;
#if RCB_D2_STAMPS
  const unsigned long long t_entry_ = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // Two image buffers and two exchange buffers (118 KB): INR n + 1 is staged into the other image while INR n is contracted,
  // and ONE barrier per INR remains -- between the exchange write of the kh = 1 waves and its read by the kh = 0 waves, which
  // then finish INR n (add, LeakyReLU', stores) while the kh = 1 waves -- the younger wave of every SIMD, which needs 42 k ticks
  // for the MFMA loops the kh = 0 waves do in 31 k (tools/d2_stamps.py) -- are already in the MFMA loop of INR n + 1.  With one
  // image the loop had three barriers per INR.  Measured: 76 k -> 66 k ticks per wave in the loop.  Splitting the epilogue between the two halves as well made it slower again (72 k ticks): the
  // kh = 1 waves are the slow ones already.
  constexpr int IMGB = ((IMG * 2 + 15) / 16) * 16;                                 // bytes of one image buffer
  __bf16* img = reinterpret_cast<__bf16*>(smem_raw);                               // [2][IMG]
  float* red = reinterpret_cast<float*>(smem_raw + 2 * IMGB);                      // [2][4 waves][16][64]
  // (the wave index as a scalar: branches on mt / tile / kh are then scalar branches, not exec masks and select chains)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q = lane & 31, h = lane >> 5;
  const int mt = wave & 1, tile = (wave >> 1) & 1, kh = wave >> 2;
  uint4 fr[8][4];   // [combo of this half][kb]
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int n = 8 * kh + c, ry = (n >> 2) - 1, rx = (n & 3) - 1;
    const int pa = ry & 1, ty = (ry <= 0) ? 1 : 0, pb = rx & 1, tx = (rx <= 0) ? 1 : 0;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      Frag f;
      if (a.pack) {
        f.u = a.pack[8192 + (((kh * 2 + mt) * 8 + c) * 4 + kb) * 64 + lane];
      } else {                             // the 8 elements are consecutive output channels: two 16-byte loads
        const float4* wp = reinterpret_cast<const float4*>(a.weff + weff_index(ty, tx, 32 * mt + q, pa, pb, 16 * kb + 8 * h, COUT));
        const float4 w0 = wp[0], w1 = wp[1];
        f.v[0] = (__bf16)w0.x; f.v[1] = (__bf16)w0.y; f.v[2] = (__bf16)w0.z; f.v[3] = (__bf16)w0.w;
        f.v[4] = (__bf16)w1.x; f.v[5] = (__bf16)w1.y; f.v[6] = (__bf16)w1.z; f.v[7] = (__bf16)w1.w;
      }
      fr[c][kb] = f.u;
    }
  }
  // (pinned in a loop of their own: an empty asm that takes a loaded value waits for THAT load, so pinning inside the load loop
  // serialised the loads -- 32 dependent L2 / HBM round trips, 12.9 us of the stage-2 forward's 54 us by its own stamps)
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) pin(fr[c][kb]);
  for (int e = tid; e < 2 * IMGB / 16; e += 512) reinterpret_cast<uint4*>(img)[e] = make_uint4(0, 0, 0, 0);
  float dbsum[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) dbsum[k] = 0.f;
  // dy image of one INR: 16 x 16 pixels x 64 channels bf16 = 2048 x 16 B -> 4 per thread
  uint4 pre[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pre[k] = make_uint4(0, 0, 0, 0);
#define RCB_FETCHD2(bb)                                                                                           \
  {                                                                                                               \
    const uint4* src_ = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.dy) + (long long)(RCB_UP_HOT ? (int)blockIdx.x : (bb)) * OG * OG * COUT); \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) pre[k] = src_[tid + 512 * k];                                   \
  }
  const int gs = gridDim.x;
  int b = blockIdx.x;
  __syncthreads();   // also a scheduling boundary: the prefetch registers must not overlap the prologue's register peak
  if (b < a.batch) RCB_FETCHD2(b)
  const int pos = tile * 32 + q, u = pos >> 3, v = pos & 7;
#if RCB_D2_STAMPS
  unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
  const unsigned long long t_loop_ = __builtin_amdgcn_s_memrealtime();
#endif
#define RCB_STAGED2(dst)                                                                                                   \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                          \
    const int e_ = tid + 512 * k, pix = e_ >> 3, c8 = e_ & 7, xx = (pix & 15) + 1;                                          \
    *reinterpret_cast<uint4*>((dst) + ((pix >> 4) + 1) * ROWS + xx * RS + 8 * (c8 ^ ((xx >> 1) & 7))) = pre[k];            \
  }
  if (b < a.batch) {                 // the first INR's image; the second one's rows are requested behind it
    RCB_STAGED2(img)
    if (b + gs < a.batch) RCB_FETCHD2(b + gs)
  }
  __syncthreads();
  int cur = 0;
  for (; b < a.batch; b += gs, cur ^= 1) {
    const __bf16* im = img + cur * (IMGB / 2);
    D2_T(0);
    // INR n + 1 into the other buffer (every wave is past the barrier of INR n - 1, i.e. past its reads of that buffer), then
    // the rows of INR n + 2 are requested: in flight during the MFMA loop
    if (b + gs < a.batch) {
      __bf16* nx = img + (cur ^ 1) * (IMGB / 2);
      RCB_STAGED2(nx)
      if (b + 2 * gs < a.batch) RCB_FETCHD2(b + 2 * gs)
    }
    D2_T(1);
    // sign source of this lane's 16 output channels, requested before the MFMA loop
    const long long xoff = ((long long)b * G * G + pos) * CIN + 32 * mt + 16 * h;
    float xf[16];     // fp32 sign source
    Frag xb[2];       // bf16 sign source
    if (kh == 0) {
      if (X_F32) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 t = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.x) + xoff)[k];
          xf[4 * k] = t.x; xf[4 * k + 1] = t.y; xf[4 * k + 2] = t.z; xf[4 * k + 3] = t.w;
        }
      } else {
        xb[0].u = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.x) + xoff)[0];
        xb[1].u = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.x) + xoff)[1];
      }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // combos one at a time: the scheduler would otherwise hoist all 32 fragment reads and push the prefetch
    // registers into scratch
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int n = 8 * kh + c, ry = (n >> 2) - 1, rx = (n & 3) - 1;
      const int xx = 2 * v + rx + 1, sw = (xx >> 1) & 7;
      const __bf16* px = im + (2 * u + ry + 1) * ROWS + xx * RS;
      uint4 curf[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) curf[kb] = *reinterpret_cast<const uint4*>(px + 8 * ((2 * kb + h) ^ sw));
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        Frag bf, fa;
        bf.u = curf[kb];
        fa.u = fr[c][kb];
        acc = mfma16(fa.v, bf.v, acc);
      }
      __builtin_amdgcn_sched_barrier(0);     // one combo at a time (the other wave of the SIMD covers the LDS latency)
    }
    D2_T(3);
    // add the two combo halves (register-major layout: conflict-free)
    float* rw = red + (cur * 4 + (wave & 3)) * 16 * 64;
    if (kh == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) rw[r * 64 + lane] = acc[r];
    }
    D2_T(4);
    __syncthreads();
    D2_T(5);
    if (kh == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += rw[r * 64 + lane];
      // lane halves swap so that lane (q, h) owns channels 32 mt + 16 h + {0..15}; two halves of 8 channels keep the
      // live temporaries small (the prefetch registers must not spill)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[4 * hf + k]), __float_as_uint(acc[8 + 4 * hf + k]),
                                                     false, false);
          o[k] = __uint_as_float(sw[0]);
          o[4 + k] = __uint_as_float(sw[1]);
        }
        if (X_F32) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            o[k] *= (xf[8 * hf + k] > 0.f ? 1.0f : SLOPE);
            dbsum[8 * hf + k] += o[k];
          }
          float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(a.dx) + xoff + 8 * hf);
          dst[0] = make_float4(o[0], o[1], o[2], o[3]);
          dst[1] = make_float4(o[4], o[5], o[6], o[7]);
        } else {
          Frag oo;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            oo.v[k] = (__bf16)(o[k] * ((float)xb[hf].v[k] > 0.f ? 1.0f : SLOPE));
            dbsum[8 * hf + k] += (float)oo.v[k];     // sums of the values as stored
          }
          if (RCB_UP_NOSTORE != 2 || a.batch < 0) reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(a.dx) + xoff)[hf] = oo.u;
        }
      }
    }
    D2_T(6);
  }
#if RCB_D2_STAMPS
  ph_[7] = __builtin_amdgcn_s_memrealtime() - t_loop_;      // wall clock (100 MHz) of the INR loop; slot 2 (unused): the prologue
  ph_[2] = t_loop_ - t_entry_;
  if (blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 8; ++k) g_d2_stamps[wave * 8 + k] = ph_[k];
#endif
#undef RCB_FETCHD2
#undef RCB_STAGED2
  if (dbias_partial) {   // fixed-order reduction over the 32 pixels of a lane half, then over the two tiles
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) dbsum[k] += __shfl_xor(dbsum[k], off, 64);
    }
    float* r2 = red;   // [tile][64]
    if (kh == 0 && q == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) r2[tile * 64 + 32 * mt + 16 * h + k] = dbsum[k];
    }
    __syncthreads();
    if (tid < 64) dbias_partial[(long long)blockIdx.x * 64 + tid] = r2[tid] + r2[64 + tid];
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const void* x;
  const void* dy;
  float* partial;   // [gridDim.x][1024 * COUT + COUT]: every workgroup's own sums (weights, then bias)
  int batch;
};

// transposed gather read: the 16-lane group supplies addresses of 4 "rows" (positions) x 4 chunks of 4 channels
__device__ __forceinline__ s16x4 tr_read(const __bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// Bias gradient of a persistent weight-gradient kernel: the 512 threads' private partial sums (8 channels each; thread t holds
// channels 8 (t % (COUT / 8)) .. + 7) are joined in a FIXED order -- 512 / COUT groups of eight threads' values per channel
// (each group summed by one thread, ascending), then the groups ascending: deterministic, and 8 + 512 / COUT dependent LDS reads
// instead of the 512 * 8 / COUT of a single thread per channel (stamps: 10.5 us of epilogue per launch in the fused stage-3
// backward, most of it this chain of 256 reads).  `red`: at least 512 * 8 + 512 floats of LDS.
template <int COUT>
__device__ __forceinline__ void bias_partials_join(const float (&dbsum)[8], float* red, int tid, float* dst) {
  constexpr int C8 = COUT / 8, NG = 512 / COUT;           // NG groups of eight values per channel
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid * 8 + j] = dbsum[j];
  __syncthreads();
  {
    const int ch = tid % COUT, grp = tid / COUT;           // every thread: one (channel, group)
    const int c8 = ch >> 3, j = ch & 7;
    float sacc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sacc += red[(c8 + C8 * (8 * grp + k)) * 8 + j];
    red[512 * 8 + grp * COUT + ch] = sacc;
  }
  __syncthreads();
  if (tid < COUT) {
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) tot += red[512 * 8 + g * COUT + tid];
    dst[tid] = tot;
  }
}

// One persistent workgroup per CU walks its INRs: the x / dy images of the NEXT INR are fetched into registers
// while the matrix cores consume the current ones from LDS (a CU needs tens of KB in flight to keep its share of
// HBM busy), and the workgroup's sums go to its own slab: no atomics, bitwise reproducible after the fixed-order
// reduction below.
template <int COUT, int G, int X_MODE, int DY_F32>
__global__ void __launch_bounds__(512) upconv_wgrad_kernel(WgradArgs a) {
  constexpr int NT = (COUT + 31) / 32;
  WC_T(0, 0);
  constexpr int HG = G + 2;             // halo grid
  constexpr int OG = 2 * G;             // output grid
  constexpr int DM = DY_F32 ? 2 : 0;
  constexpr int NX = G * G * (CIN / 8) / 512, ND = OG * OG * (COUT / 8) / 512;
  static_assert(NX * 512 == G * G * (CIN / 8) && ND * 512 == OG * OG * (COUT / 8) && 512 % (COUT / 8) == 0, "tiling");
  constexpr int WSZ = 1024 * COUT, ROW = WSZ + COUT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* ximg = reinterpret_cast<__bf16*>(smem_raw);               // [HG*HG][XRS]
  constexpr int DRS = COUT == 64 ? XRS : COUT;                      // dy image row stride: 128-byte rows are padded like x's
  __bf16* dimg = ximg + HG * HG * XRS;                              // [OG*OG][DRS]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5, fb = (lane >> 4) & 1, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
  for (int e = tid; e < HG * HG * XRS / 8; e += 512) reinterpret_cast<uint4*>(ximg)[e] = make_uint4(0, 0, 0, 0);
  f32x16 acc[2][2][NT];   // [combo slot][mt][nt]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][mt][nt][r] = 0.f;
  float dbsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbsum[j] = 0.f;
  Raw8<X_MODE> px[NX];
  Raw8<DM> pd[ND];
#define RCB_WG_FETCH(bb)                                                                                   \
  {                                                                                                        \
    _Pragma("unroll") for (int k = 0; k < NX; ++k)                                                         \
        px[k] = raw_load<X_MODE>(a.x, (long long)(bb) * G * G * CIN + 8 * (tid + 512 * k));                \
    _Pragma("unroll") for (int k = 0; k < ND; ++k)                                                         \
        pd[k] = raw_load<DM>(a.dy, (long long)(bb) * OG * OG * COUT + 8 * (tid + 512 * k));                \
  }
  int b = blockIdx.x;
  if (b < a.batch) RCB_WG_FETCH(b)
  WC_T(0, 1);
  for (; b < a.batch; b += gridDim.x) {
    __syncthreads();   // previous INR fully consumed (also orders the halo clear)
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int e = tid + 512 * k, pix = e >> 3, c8 = e & 7;
      const int i = pix / G, j = pix - i * G;
      Frag f;
      f.v = raw_frag<X_MODE>(px[k], true);
      *reinterpret_cast<uint4*>(ximg + ((i + 1) * HG + (j + 1)) * XRS + 8 * c8) = f.u;
    }
    // 512 % (COUT / 8) == 0: a thread always handles the same 8 channels -> private bias-gradient partials
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      Frag f;
      f.v = raw_frag<DM>(pd[k], true);
      const int e = tid + 512 * k;
      *reinterpret_cast<uint4*>(dimg + (e / (COUT / 8)) * DRS + 8 * (e % (COUT / 8))) = f.u;
#pragma unroll
      for (int j = 0; j < 8; ++j) dbsum[j] += (float)f.v[j];
    }
    __syncthreads();
    if (b + (int)gridDim.x < a.batch) RCB_WG_FETCH(b + gridDim.x)
    {
      // 16 combos (phase p, tap t) over 8 waves: wave w takes t = 2 (w & 1) + c, c = 0 / 1, of phase p = w >> 1.  The dy
      // operand depends on the phase only: one set of transposed reads feeds both combos
      const int p = wave >> 1, pa = p >> 1, pb = p & 1, ty = wave & 1;
      for (int pt = 0; pt < G * G / 32; ++pt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          union { s16x4 s[2]; bf16x8 v; } av[2][2], bv[NT];
#pragma unroll
          for (int w2 = 0; w2 < 2; ++w2) {
            const int pos = 32 * pt + 16 * ks + 8 * h + 4 * w2 + q4;
            const int i = pos / G, j = pos - i * G;
            const __bf16* xr = ximg + ((i + pa + ty) * HG + (j + pb)) * XRS + 16 * fb + 4 * p4;
#pragma unroll
            for (int c = 0; c < 2; ++c)               // tx = c: the neighbouring column
#pragma unroll
              for (int mt = 0; mt < 2; ++mt) av[c][mt].s[w2] = tr_read(xr + c * XRS + 32 * mt);
            const __bf16* dr = dimg + ((2 * i + pa) * OG + (2 * j + pb)) * DRS + 4 * p4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt].s[w2] = tr_read(dr + ((COUT >= 32) ? (32 * nt + 16 * fb) : 0));
          }
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[c][mt][nt] = mfma16(av[c][mt].v, bv[nt].v, acc[c][mt][nt]);
        }
      }
    }
  }
#undef RCB_WG_FETCH
  WC_T(0, 2);
  float* slab = a.partial + (long long)blockIdx.x * ROW;
  bias_partials_join<COUT>(dbsum, reinterpret_cast<float*>(smem_raw), tid, slab + WSZ);
  // D[m = ci, n = co]: rows in registers, column on the lane
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int combo = 2 * wave + c;
    const int p = combo >> 2, t = combo & 3;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = 32 * nt + (lane & 31);
        if (co < COUT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ci = 32 * mt + rho(r, h);
            slab[weff_index(t >> 1, t & 1, ci, p >> 1, p & 1, co, COUT)] = acc[c][mt][nt][r];
          }
        }
      }
  }
  WC_T(0, 3);
}

// ------------------------------------------------------------------------------------------------
// Stage-3 backward, fused: the data gradient and the weight gradient both need dy (dpe) and x (h2) of every INR; as two
// kernels each tensor is read twice (268 MB of the 670 MB the pair moves).  Here one staging of the INR's two images
// in LDS feeds both MFMA loops:
//   data gradient  : as upconv_dgrad3_lds_kernel (fragments in LDS, dy gathered from the zero-halo image, two channel
//                    halves per wave); the LeakyReLU' sign now comes from the staged x image instead of a global re-read
//   weight gradient: as upconv_wgrad_kernel, with the dy operand read (ds_read_b64_tr_b16) from the same halo image
//                    (row stride 24 elements instead of 16) and the x operand from the staged zero-halo x image
// bf16 dy / x / dx, COUT = 16.  Sums go to per-workgroup slabs (upconv_wgrad_reduce_kernel finishes them).
// ------------------------------------------------------------------------------------------------
// Diagnostic build only (-DRCB_B3_STAMPS=1, tools/b3_stamps.py): cycles each wave of workgroup 0 spends per phase, summed
// over its INRs: 0 first barrier, 1 staging, 2 second barrier, 3 data gradient (incl. prefetch issue), 4 weight gradient
#ifndef RCB_B3_STAMPS
#define RCB_B3_STAMPS 0
#endif
#ifndef RCB_B3_WG16
#define RCB_B3_WG16 1        // weight-gradient part on v_mfma_f32_16x16x32_bf16 (0: the 32 x 32 x 16 form of the separate kernel)
#endif
#ifndef RCB_B3_DIAG
#define RCB_B3_DIAG 0      // ablations (wrong results): 1 no dx stores, 2 one fragment read per tap pair, 3 one dy read for all taps,
                           // 4 dx stores land in the workgroup's first INR (L2-resident target)
#endif
#if RCB_B3_STAMPS
__device__ unsigned long long g_b3_stamps[8 * 8];
#define B3_T(k)                                                     \
  do {                                                              \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
    ph_[k] += now_ - last_;                                         \
    last_ = now_;                                                   \
  } while (0)
#else
#define B3_T(k) do { } while (0)
#endif

struct Bwd3Args {
  const void* dy;
  const float* weff;
  const void* x;
  void* dx;
  float* partial;
  int batch;
  const uint4* pack;
};

__global__ void __launch_bounds__(512) upconv_bwd3_fused_kernel(Bwd3Args a) {
  constexpr int COUT = 16, G = 16, OG = 32, HO = 34, RS = 24, HG = 18, NF = 32;
#if RCB_B3_STAMPS
  const unsigned long long t_entry_ = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int WSZ = 1024 * COUT, ROW = WSZ + COUT;
#if RCB_B3_WG16
  // x image of THIS kernel: 128-byte pixel rows, the 32-byte block cb of 16 channels stored at block cb ^ xkey(column) with
  // xkey = column bit 1 | column bit 3 << 1.  The transposed reads of the weight gradient take, per 32 lanes, four adjacent
  // columns and the four columns eight further on: with this key (and the bank half that the column parity selects) their 32
  // 8-byte pieces fall on 32 distinct bank pairs.  On the 144-byte rows of the other kernels every one of those reads took two
  // passes, and with the matrix work halved (16 x 16 x 32) they are what the phase waits for (tools/lds_banks.py: 4.0 -> 2.0
  // cycles per read; staging stores 8 -> 6, the data gradient's eight sign reads per INR 4 -> 8)
  constexpr int XR3 = 64;
#define B3_XKEY(col) ((((col) >> 1) & 1) | ((((col) >> 3) & 1) << 1))
#else
  constexpr int XR3 = XRS;
#define B3_XKEY(col) 0
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* frags = reinterpret_cast<uint4*>(smem_raw);                              // [16 combos][2 mt][64 lanes]
  __bf16* dyimg = reinterpret_cast<__bf16*>(smem_raw + NF * 1024);                // [34][34][24]
  __bf16* ximg = dyimg + HO * HO * RS;                                            // [18][18][XRS]
  constexpr int SCR_RS = 40;                                                      // dx transposition tile: 80-byte rows
  __bf16* dxscr = ximg + HG * HG * XR3;                                           // [8 waves][32][SCR_RS]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane & 31, h = lane >> 5;
  const int fb = (lane >> 4) & 1, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
  if (a.pack) {      // four loads in flight, then four stores (a load-store loop waits for every load before its store)
    uint4 t4[NF * 64 / 512];
#pragma unroll
    for (int k = 0; k < NF * 64 / 512; ++k) t4[k] = a.pack[20480 + tid + 512 * k];
#pragma unroll
    for (int k = 0; k < NF * 64 / 512; ++k) frags[tid + 512 * k] = t4[k];
  }
  for (int e = tid; e < (a.pack ? 0 : NF * 64); e += 512) {
    const int ln = e & 63, slot = e >> 6;
    const int mt = slot & 1, combo = slot >> 1;
    const int ry = (combo >> 2) - 1, rx = (combo & 3) - 1;
    const int pa = (ry & 1), ty = (ry <= 0) ? 1 : 0;
    const int pb = (rx & 1), tx = (rx <= 0) ? 1 : 0;
    const int fq = ln & 31, fh = ln >> 5, ci = 32 * mt + fq;
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)a.weff[weff_index(ty, tx, ci, pa, pb, 8 * fh + j, COUT)];
    frags[e] = f.u;
  }
  for (int e = tid; e < (HO * HO * RS + HG * HG * XR3) / 8; e += 512) reinterpret_cast<uint4*>(dyimg)[e] = make_uint4(0, 0, 0, 0);
#if RCB_B3_WG16
  // weight gradient on v_mfma_f32_16x16x32_bf16: [combo slot][16-channel block of cin] tiles of 16 cin x 16 cout (the 32 x 32 x 16
  // form has 32 columns for the 16 output channels: half of every MFMA, and this phase ran at 90 % of the matrix pipe)
  f32x4v wacc[2][4];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) wacc[c][cb][r] = 0.f;
#else
  f32x16 wacc[2][2];   // weight gradient: [combo slot][mt]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) wacc[c][mt][r] = 0.f;
#endif
  float dbsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbsum[j] = 0.f;
  // per INR: dy 32 x 32 x 16 bf16 = 2048 x 16 B, x 16 x 16 x 64 bf16 = 2048 x 16 B: 4 + 4 per thread, 64 KB in flight
  uint4 pdy[4], px[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pdy[k] = px[k] = make_uint4(0, 0, 0, 0);
#define RCB_FETCH_B3(bb)                                                                                               \
  {                                                                                                                   \
    const uint4* sd_ = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.dy) + (long long)(bb) * OG * OG * COUT); \
    const uint4* sx_ = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.x) + (long long)(bb) * G * G * CIN);    \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) { pdy[k] = sd_[tid + 512 * k]; px[k] = sx_[tid + 512 * k]; }         \
  }
  const int gs = gridDim.x;
  int b = blockIdx.x;
  __syncthreads();
  if (b < a.batch) RCB_FETCH_B3(b)
  const int u = (wave * 32 + q) >> 4, v = (wave * 32 + q) & 15;      // data gradient: this lane's output position
#if RCB_B3_STAMPS
  unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
  ph_[5] = __builtin_amdgcn_s_memrealtime();      // 100 MHz wall clock at the loop's start / end: slots 5, 6
#endif
  for (; b < a.batch; b += gs) {
    __syncthreads();
    B3_T(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = tid + 512 * k;
      {   // dy: pixel e >> 1, chunk of 8 channels e & 1 (the same chunk for every k: private bias partials)
        const int pix = e >> 1, c8 = e & 1;
        *reinterpret_cast<uint4*>(dyimg + (((pix >> 5) + 1) * HO + ((pix & 31) + 1)) * RS + 8 * c8) = pdy[k];
        Frag f;
        f.u = pdy[k];
#pragma unroll
        for (int j = 0; j < 8; ++j) dbsum[j] += (float)f.v[j];
      }
      {   // x: pixel e >> 3, chunk e & 7
        const int pix = e >> 3, c8 = e & 7;
        *reinterpret_cast<uint4*>(ximg + (((pix >> 4) + 1) * HG + ((pix & 15) + 1)) * XR3 + 16 * ((c8 >> 1) ^ B3_XKEY((pix & 15) + 1)) + 8 * (c8 & 1)) = px[k];
      }
    }
    B3_T(1);
    __syncthreads();
    B3_T(2);
    if (b + gs < a.batch) RCB_FETCH_B3(b + gs)
    // ---- data gradient --------------------------------------------------------------------------------------
    {
      f32x16 acc[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
      // per tap: one dy gather and two weight fragments from LDS feed two MFMAs.  The reads of tap n + B3_PF are issued before
      // the MFMAs of tap n (register ring, order pinned by scheduling barriers): the compiler's own schedule re-used one
      // register set and waited for every read in front of its MFMA, and this phase -- a third of the kernel's matrix work --
      // took 45 % of its time
      constexpr int B3_PF = RCB_B3_PF;
      uint4 rg[B3_PF + 1][3];
      auto tap_reads = [&](int n, uint4 (&dst)[3]) {
        const int ry = (n >> 2) - 1, rx = (n & 3) - 1;
        dst[0] = *reinterpret_cast<const uint4*>(dyimg + ((2 * u + ((RCB_B3_DIAG == 3) ? 0 : ry) + 1) * HO + (2 * v + ((RCB_B3_DIAG == 3) ? 0 : rx) + 1)) * RS + 8 * h);
        dst[1] = frags[((RCB_B3_DIAG == 2 ? 0 : n) * 2 + 0) * 64 + lane];
        dst[2] = frags[((RCB_B3_DIAG == 2 ? 0 : n) * 2 + 1) * 64 + lane];
      };
#pragma unroll
      for (int n = 0; n < B3_PF; ++n) tap_reads(n, rg[n]);
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        if (n + B3_PF < 16) tap_reads(n + B3_PF, rg[(n + B3_PF) % (B3_PF + 1)]);
        __builtin_amdgcn_sched_barrier(0);
        Frag bf, fa;
        bf.u = rg[n % (B3_PF + 1)][0];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          fa.u = rg[n % (B3_PF + 1)][1 + mt];
          acc[mt] = mfma16(fa.v, bf.v, acc[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // dx rows are 128 B per position; a lane holds 4 x 4 channels of each 32-channel half: stored from the registers that
      // is eight 8-byte pieces per lane at 128-byte stride (measured: a quarter of the kernel's time).  Instead each half goes
      // through a per-wave LDS tile [32 positions][32 channels] and leaves as 16-byte pieces, four consecutive lanes per
      // position: 64 contiguous bytes per position and instruction.
      const __bf16* xs = ximg + ((u + 1) * HG + (v + 1)) * XR3;
      const int xk16 = 16 * B3_XKEY(v + 1);
      __bf16* scr = dxscr + wave * (32 * SCR_RS);
      __bf16* dxw = reinterpret_cast<__bf16*>(a.dx) + ((long long)(RCB_B3_DIAG == 4 ? (int)blockIdx.x : b) * G * G + wave * 32) * CIN;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int ci = 32 * mt + 8 * g4 + 4 * h;
          const bf16x4 t = *reinterpret_cast<const bf16x4*>(xs + (((32 * mt + 8 * g4) & ~15) ^ xk16) + ((8 * g4) & 15) + 4 * h);
          bf16x4 ob;
#pragma unroll
          for (int k = 0; k < 4; ++k) ob[k] = (__bf16)(acc[mt][4 * g4 + k] * ((float)t[k] > 0.f ? 1.0f : SLOPE));
          *reinterpret_cast<bf16x4*>(scr + q * SCR_RS + 8 * g4 + 4 * h) = ob;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int pl = 16 * ps + (lane >> 2), ch = lane & 3;
          const uint4 o = *reinterpret_cast<const uint4*>(scr + pl * SCR_RS + 8 * ch);
          if (RCB_B3_DIAG != 1 || a.batch < 0) *reinterpret_cast<uint4*>(dxw + pl * CIN + 32 * mt + 8 * ch) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    // ---- weight gradient --------------------------------------------------------------------------------------
    // 16 combos (phase p, tap t) over 8 waves: wave w takes t = 2 (w & 1) + c, c = 0 / 1, of phase p = w >> 1.  The dy operand
    // depends on the phase only: one transposed read feeds both combos (10 reads per 4 MFMAs instead of 12)
    __builtin_amdgcn_sched_barrier(0);
    B3_T(3);
#if RCB_B3_WG16
    {
      // per 32 positions (two rows of the 16 x 16 grid) and combo: four MFMAs of K = 32, one per 16-channel block of cin; a lane
      // group g = lane >> 4 supplies positions 8 g .. 8 g + 7 of both operands (two transposed 8-byte reads each)
      const int p = wave >> 1, pa = p >> 1, pb = p & 1, ty = wave & 1, kg = lane >> 4;
      // this lane's 18 read addresses of position block 0 (the blocks that follow are two image rows further on: one
      // wave-uniform offset per trip of the loop)
      // (LDS byte addresses, opaque to the compiler: left to itself it re-derives every address in every trip)
      typedef __attribute__((address_space(3))) s16x4* lds_tr_t;
      unsigned xa[2][2][4], da[2];
#pragma unroll
      for (int w2 = 0; w2 < 2; ++w2) {
        const int pos = 8 * kg + 4 * w2 + q4;
        const int i = pos >> 4, j = pos & 15;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int col = j + pb + c;
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            xa[w2][c][cb] = (unsigned)(size_t)(lds_tr_t)(ximg + ((i + pa + ty) * HG + col) * XR3 + 16 * (cb ^ B3_XKEY(col)) + 4 * p4);
            asm volatile("" : "+v"(xa[w2][c][cb]));
          }
        }
        da[w2] = (unsigned)(size_t)(lds_tr_t)(dyimg + ((2 * i + pa + 1) * HO + (2 * j + pb + 1)) * RS + 4 * p4);
        asm volatile("" : "+v"(da[w2]));
      }
      constexpr unsigned XSTEP = 2 * HG * XR3 * 2, DSTEP = 4 * HO * RS * 2;       // bytes per position block
#pragma unroll 1
      for (int trip = 0; trip < G * G / 64; ++trip) {     // two position blocks per trip (the prefetch registers must stay in registers)
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {
          union { s16x4 s[2]; bf16x8 v; } av[2][4], bv;
#pragma unroll
          for (int w2 = 0; w2 < 2; ++w2) {
#pragma unroll
            for (int c = 0; c < 2; ++c)               // tx = c: the neighbouring column
#pragma unroll
              for (int cb = 0; cb < 4; ++cb)
                av[c][cb].s[w2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(size_t)(xa[w2][c][cb] + (2 * trip + u2) * XSTEP));
            bv.s[w2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(size_t)(da[w2] + (2 * trip + u2) * DSTEP));
          }
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) wacc[c][cb] = mfma16x32(av[c][cb].v, bv.v, wacc[c][cb]);
        }
      }
    }
#else
    {
      const int p = wave >> 1, pa = p >> 1, pb = p & 1, ty = wave & 1;
#pragma unroll 2
      for (int pt = 0; pt < G * G / 32; ++pt) {       // partly unrolled: the prefetch registers must stay in registers
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          union { s16x4 s[2]; bf16x8 v; } av[2][2], bv;
#pragma unroll
          for (int w2 = 0; w2 < 2; ++w2) {
            const int pos = 32 * pt + 16 * ks + 8 * h + 4 * w2 + q4;
            const int i = pos >> 4, j = pos & 15;
            const __bf16* xr = ximg + ((i + pa + ty) * HG + (j + pb)) * XRS + 16 * fb + 4 * p4;
#pragma unroll
            for (int c = 0; c < 2; ++c)               // tx = c: the neighbouring column
#pragma unroll
              for (int mt = 0; mt < 2; ++mt) av[c][mt].s[w2] = tr_read(xr + c * XRS + 32 * mt);
            bv.s[w2] = tr_read(dyimg + ((2 * i + pa + 1) * HO + (2 * j + pb + 1)) * RS + 4 * p4);
          }
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) wacc[c][mt] = mfma16(av[c][mt].v, bv.v, wacc[c][mt]);
        }
      }
    }
#endif
    B3_T(4);
  }
#if RCB_B3_STAMPS
  ph_[6] = __builtin_amdgcn_s_memrealtime();
  ph_[7] = t_entry_;
  if (blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 8; ++k) g_b3_stamps[wave * 8 + k] = ph_[k];
#endif
#undef RCB_FETCH_B3
#undef B3_XKEY
  float* slab = a.partial + (long long)blockIdx.x * ROW;
  bias_partials_join<COUT>(dbsum, reinterpret_cast<float*>(smem_raw), tid, slab + WSZ);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int combo = 2 * wave + c;
    const int p = combo >> 2, t = combo & 3;
#if RCB_B3_WG16
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r)       // 16 x 16 accumulator: column (cout) = lane & 15, rows (cin) 4 (lane >> 4) + r
        slab[weff_index(t >> 1, t & 1, 16 * cb + 4 * (lane >> 4) + r, p >> 1, p & 1, lane & 15, COUT)] = wacc[c][cb][r];
#else
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int co = lane & 31;
      if (co < COUT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = 32 * mt + rho(r, h);
          slab[weff_index(t >> 1, t & 1, ci, p >> 1, p & 1, co, COUT)] = wacc[c][mt][r];
        }
      }
    }
#endif
  }
#if RCB_B3_STAMPS
  if (blockIdx.x == 0 && lane == 0) g_b3_stamps[wave * 8 + 4 + 60 - 60] = g_b3_stamps[wave * 8 + 4];   // (keeps slot 4)
  if (blockIdx.x == 0 && lane == 0 && wave == 0) g_b3_stamps[63] = __builtin_amdgcn_s_memrealtime();   // kernel end of workgroup 0 (slot 7 of wave 7)
#endif
}

#if RCB_B3_STAMPS
}  // namespace
#if RCB_D2_STAMPS
extern "C" int rcb_debug_d2_stamps(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_d2_stamps), sizeof(unsigned long long) * n);
}
#endif
#if RCB_WC_STAMPS
extern "C" int rcb_debug_wc_stamps(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wc_stamps), sizeof(unsigned long long) * n);
}
#endif
#if RCB_F2_STAMPS
extern "C" int rcb_debug_f2_stamps(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_f2_stamps), sizeof(unsigned long long) * n);
}
#endif
extern "C" int rcb_debug_b3_stamps(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_b3_stamps), sizeof(unsigned long long) * n);
}
namespace {
#endif

// fixed-order sum of the workgroup slabs: out[j] = sum_w partial[w][j]
__global__ void __launch_bounds__(256) upconv_wgrad_reduce_kernel(const float* __restrict__ partial, int nblk, int row,
                                                                  int wsz, float* __restrict__ dweff,
                                                                  float* __restrict__ dbias) {
  // 64 outputs per block; sixteen partial sums per output -- partial k adds the slabs k, k + 16, ... (coalesced 256-byte rows,
  // eight loads in flight each) -- joined by a fixed tree: a fixed association (bitwise reproducible).  Four waves of a
  // 256-thread block form four of the sixteen partial sums each.  (As ONE 1024-thread block of sixteen waves -- the same sums --
  // the kernel took 9 us alone and 145 us inside the three-stream step: a block that needs sixteen free wave slots on one CU
  // at once is not placed while the neighbouring streams' 256-thread blocks keep refilling the CUs, and the chain behind it
  // -- the last weight-gradient GEMM, the map onto the conv weights -- ended the forked region, tools/step_timeline.py.)
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  float s[4][8];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int u = 0; u < 8; ++u) s[v][u] = 0.f;
  if (j < row) {
    const float* p = partial + j;
    int w = 4 * wave;                      // partial sums 4 wave + v, v = 0 .. 3, side by side (32 loads in flight)
    for (; w + 3 + 7 * 16 < nblk; w += 8 * 16) {
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) s[v][u] += p[(long long)(w + v + 16 * u) * row];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {          // the ragged rest, per partial sum exactly as its own loop would run
      int wv = w + v;
      for (; wv + 7 * 16 < nblk; wv += 8 * 16) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[v][u] += p[(long long)(wv + 16 * u) * row];
      }
      for (; wv < nblk; wv += 16) s[v][0] += p[(long long)wv * row];
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v)
    part[4 * wave + v][lane] = ((s[v][0] + s[v][1]) + (s[v][2] + s[v][3])) + ((s[v][4] + s[v][5]) + (s[v][6] + s[v][7]));
  __syncthreads();
  if (wave == 0 && j < row) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = part[2 * k][lane] + part[2 * k + 1][lane];
    const float tot = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    if (j < wsz) dweff[j] = tot;
    else if (dbias) dbias[j - wsz] = tot;
  }
}

template <typename K, typename A>
int launch(K kfn, const A& args, int grid, size_t smem, hipStream_t st, bool& /*unused: the attribute is set per launch*/,
           int threads = 512) {
  // (the attribute belongs to the (function, device) pair and a process may drive several devices: set it every time)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return fail((int)e, "upconv: hipFuncSetAttribute: %s", hipGetErrorString(e));
  kfn<<<grid, threads, smem, st>>>(args);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

// two half-size workgroups per CU where the kernel has the variant and the batch fills them (RCB_UPCONV_HALF_WG=1: opt-in)
static bool half_wg(int batch) {
  static const bool on = [] { const char* e = getenv("RCB_UPCONV_HALF_WG"); return e && e[0] == '1'; }();   // off by default: no gain measured, and the 4-wave instances are at the register limit with the gather ring
  return on && batch >= 768;
}


}  // namespace

extern "C" int rcb_upconv_fwd(const void* x, int32_t x_is_f32_preact, const float* weff, const float* bias, void* y,
                              int32_t y_is_f32_linear, int32_t batch, int32_t grid, int32_t cout, const void* frag_pack,
                              rcb_stream_t stream) {
  RCB_REQUIRE(x && weff && bias && y, RCB_ERR_ARG, "upconv_fwd: null pointer");
  RCB_REQUIRE(batch > 0, RCB_ERR_SHAPE, "upconv_fwd: empty batch");
  RCB_REQUIRE((reinterpret_cast<uintptr_t>(frag_pack) & 15) == 0, RCB_ERR_ARG, "upconv_fwd: frag_pack must be 16-byte aligned");
  FwdArgs a{x, weff, bias, y, batch, reinterpret_cast<const uint4*>(frag_pack)};
  hipStream_t st = (hipStream_t)stream;
  constexpr int kFwd2Smem = 4 * 10 * 10 * 72 * 2 + 64 * 4;
  const int npair = (batch + 3) / 4;
  if (grid == 8 && cout == 64 && x_is_f32_preact == 1 && !y_is_f32_linear) {
    static bool done = false;
    return launch(upconv_fwd2_reg_kernel<1>, a, npair < 256 ? npair : 256, kFwd2Smem, st, done);
  }
  if (grid == 16 && cout == 16 && !x_is_f32_preact && y_is_f32_linear == 1) {
    static bool done = false;
    if (half_wg(batch)) return launch(upconv_fwd3_lds_kernel<16, 0, 4>, a, 768, 18 * 18 * 64 * 2, st, done, 256);
    return launch(upconv_fwd3_lds_kernel<16, 0, 8>, a, batch < 256 ? batch : 256, 18 * 18 * 64 * 2, st, done);
  }
  if (grid == 16 && cout == 16 && !x_is_f32_preact && y_is_f32_linear == 2) {   // bf16 output, no activation
    static bool done = false;
    if (half_wg(batch)) return launch(upconv_fwd3_lds_kernel<16, 1, 4>, a, 768, 18 * 18 * 64 * 2, st, done, 256);
    return launch(upconv_fwd3_lds_kernel<16, 1, 8>, a, batch < 256 ? batch : 256, 18 * 18 * 64 * 2, st, done);
  }
  if (grid == 8 && cout == 64 && x_is_f32_preact == 2 && !y_is_f32_linear) {   // bf16 pre-activation input
    static bool done = false;
    return launch(upconv_fwd2_reg_kernel<3>, a, npair < 256 ? npair : 256, kFwd2Smem, st, done);
  }
  return fail(RCB_ERR_UNSUPPORTED, "upconv_fwd: grid=%d cout=%d in_mode=%d out_f32=%d not instantiated", grid, cout,
              x_is_f32_preact, y_is_f32_linear);
}

static inline int dgrad_blocks(int batch) { return batch < 256 ? batch : 256; }

extern "C" int32_t rcb_upconv_dgrad_partial_rows(int32_t batch) { return batch > 0 ? dgrad_blocks(batch) : 0; }

template <int X_F32>
static int launch_dgrad2(const DgradArgs& a, float* dbias_partial, hipStream_t st) {
  constexpr int kSmem = 2 * (((18 * (18 * 64 + 32) * 2 + 15) / 16) * 16) + 2 * 4 * 16 * 64 * 4;      // two images, two exchange buffers
  auto kfn = upconv_dgrad2_reg_kernel<X_F32>;
  // (per launch: the attribute belongs to the (function, device) pair)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return fail((int)e, "upconv: hipFuncSetAttribute: %s", hipGetErrorString(e));
  kfn<<<dgrad_blocks(a.batch), 512, kSmem, st>>>(a, dbias_partial);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_upconv_dgrad(const void* dy, int32_t dy_is_f32, const float* weff, const void* x,
                                int32_t x_is_f32_preact, void* dx, float* dbias_partial, int32_t batch, int32_t grid,
                                int32_t cout, const void* frag_pack, rcb_stream_t stream) {
  RCB_REQUIRE(dy && weff && x && dx, RCB_ERR_ARG, "upconv_dgrad: null pointer");
  RCB_REQUIRE(batch > 0, RCB_ERR_SHAPE, "upconv_dgrad: empty batch");
  RCB_REQUIRE((reinterpret_cast<uintptr_t>(frag_pack) & 15) == 0, RCB_ERR_ARG, "upconv_dgrad: frag_pack must be 16-byte aligned");
  DgradArgs a{dy, weff, x, dx, batch, reinterpret_cast<const uint4*>(frag_pack)};
  hipStream_t st = (hipStream_t)stream;
  if (grid == 8 && cout == 64 && !dy_is_f32 && x_is_f32_preact == 1) return launch_dgrad2<1>(a, dbias_partial, st);
  if (grid == 8 && cout == 64 && !dy_is_f32 && x_is_f32_preact == 2) return launch_dgrad2<0>(a, dbias_partial, st);
  RCB_REQUIRE(dbias_partial == nullptr, RCB_ERR_UNSUPPORTED, "upconv_dgrad: channel sums of dx only for the stage-2 kernels");
  if (grid == 16 && cout == 16 && dy_is_f32 && !x_is_f32_preact) {
    static bool done = false;
    return launch(upconv_dgrad3_lds_kernel<16, 0>, a, batch < 256 ? batch : 256, 32 * 1024 + 34 * 34 * 24 * 2, st, done);
  }
  if (grid == 16 && cout == 16 && !dy_is_f32 && !x_is_f32_preact) {
    static bool done = false;
    return launch(upconv_dgrad3_lds_kernel<16, 1>, a, batch < 256 ? batch : 256, 32 * 1024 + 34 * 34 * 24 * 2, st, done);
  }
  return fail(RCB_ERR_UNSUPPORTED, "upconv_dgrad: grid=%d cout=%d not instantiated", grid, cout);
}

static inline int wgrad_blocks(int batch) { return batch < 256 ? batch : 256; }

extern "C" int64_t rcb_upconv_wgrad_workspace(int32_t batch, int32_t cout) {
  if (batch <= 0 || cout <= 0) return 0;
  return (int64_t)wgrad_blocks(batch) * (1024LL * cout + cout);
}

extern "C" int rcb_upconv_wgrad(const void* x, int32_t x_is_f32_preact, const void* dy, int32_t dy_is_f32, float* dweff,
                                float* dbias, int32_t batch, int32_t grid, int32_t cout, float* workspace,
                                int64_t workspace_floats, rcb_stream_t stream) {
  RCB_REQUIRE(x && dy && dweff && workspace, RCB_ERR_ARG, "upconv_wgrad: null pointer");
  RCB_REQUIRE(batch > 0, RCB_ERR_SHAPE, "upconv_wgrad: empty batch");
  RCB_REQUIRE(workspace_floats >= rcb_upconv_wgrad_workspace(batch, cout), RCB_ERR_SHAPE,
              "upconv_wgrad: workspace of %lld floats, %lld needed", (long long)workspace_floats,
              (long long)rcb_upconv_wgrad_workspace(batch, cout));
  WgradArgs a{x, dy, workspace, batch};
  hipStream_t st = (hipStream_t)stream;
  const int g = wgrad_blocks(batch);
  int rc = RCB_ERR_UNSUPPORTED;
  bool hit = false;
#define RCB_WGRAD_CASE(cond, COUTv, Gv, XM, DF, SMEM)                                        \
  if (!hit && (cond)) {                                                                      \
    static bool done = false;                                                                \
    hit = true;                                                                              \
    rc = launch(upconv_wgrad_kernel<COUTv, Gv, XM, DF>, a, g, SMEM, st, done);               \
  }
  RCB_WGRAD_CASE(grid == 16 && cout == 16 && !x_is_f32_preact && dy_is_f32, 16, 16, 0, 1, (18 * 18 * XRS + 32 * 32 * 16) * 2)
  RCB_WGRAD_CASE(grid == 16 && cout == 16 && !x_is_f32_preact && !dy_is_f32, 16, 16, 0, 0, (18 * 18 * XRS + 32 * 32 * 16) * 2)
  RCB_WGRAD_CASE(grid == 8 && cout == 64 && x_is_f32_preact == 1 && !dy_is_f32, 64, 8, 1, 0, (10 * 10 * XRS + 16 * 16 * XRS) * 2)
  RCB_WGRAD_CASE(grid == 8 && cout == 64 && x_is_f32_preact == 2 && !dy_is_f32, 64, 8, 3, 0, (10 * 10 * XRS + 16 * 16 * XRS) * 2)
#undef RCB_WGRAD_CASE
  if (!hit) return fail(RCB_ERR_UNSUPPORTED, "upconv_wgrad: grid=%d cout=%d not instantiated", grid, cout);
  if (rc) return rc;
  const int row = 1024 * cout + cout;
  upconv_wgrad_reduce_kernel<<<(row + 63) / 64, 256, 0, st>>>(workspace, g, row, 1024 * cout, dweff, dbias);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

// Stage-3 backward in one pass over dy and x (see upconv_bwd3_fused_kernel).  Same workspace contract as rcb_upconv_wgrad.
extern "C" int rcb_upconv_bwd_fused(const void* dy, const float* weff, const void* x, void* dx, float* dweff, float* dbias,
                                    int32_t batch, int32_t grid, int32_t cout, float* workspace, int64_t workspace_floats,
                                    const void* frag_pack, rcb_stream_t stream) {
  RCB_REQUIRE(dy && weff && x && dx && dweff && workspace, RCB_ERR_ARG, "upconv_bwd_fused: null pointer");
  RCB_REQUIRE(batch > 0, RCB_ERR_SHAPE, "upconv_bwd_fused: empty batch");
  RCB_REQUIRE(grid == 16 && cout == 16, RCB_ERR_UNSUPPORTED, "upconv_bwd_fused: grid=%d cout=%d not instantiated", grid, cout);
  RCB_REQUIRE(workspace_floats >= rcb_upconv_wgrad_workspace(batch, cout), RCB_ERR_SHAPE,
              "upconv_bwd_fused: workspace of %lld floats, %lld needed", (long long)workspace_floats,
              (long long)rcb_upconv_wgrad_workspace(batch, cout));
  RCB_REQUIRE((reinterpret_cast<uintptr_t>(frag_pack) & 15) == 0, RCB_ERR_ARG, "upconv_bwd_fused: frag_pack must be 16-byte aligned");
  Bwd3Args a{dy, weff, x, dx, workspace, batch, reinterpret_cast<const uint4*>(frag_pack)};
  hipStream_t st = (hipStream_t)stream;
  const int g = wgrad_blocks(batch);
  static bool done = false;
  int rc = launch(upconv_bwd3_fused_kernel, a, g, 32 * 1024 + (34 * 34 * 24 + 18 * 18 * (RCB_B3_WG16 ? 64 : XRS)) * 2 + 8 * 32 * 40 * 2, st, done);
  if (rc) return rc;
  const int row = 1024 * cout + cout;
  upconv_wgrad_reduce_kernel<<<(row + 63) / 64, 256, 0, st>>>(workspace, g, row, 1024 * cout, dweff, dbias);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
