// Width-32 SIREN, fused loss + backward, ONE WAVE PER ROW of wvec with K pixel tiles in flight (round 5).
//
// Why this form exists.  siren_mlp_bf16.hip (one 4-wave workgroup per row, two workgroups per CU, one 32-pixel tile per wave
// at a time) runs at 0.42 of its vector-issue floor: its 231 registers allow two waves per SIMD, and two in-order waves cannot
// cover the dependent MFMA -> sin / cos -> convert -> MFMA links, the LDS round trips of the weight-gradient transposes and the
// per-tile fragment reads (21 ds_read_b128 per tile, each in front of its MFMA).  Asking the allocator for a third wave spills
// (142 registers).  Here the latency is covered INSIDE a wave instead:
//   * one wave owns a whole row (an INR and sample) -- or one pixel chunk of it -- and walks its 32-pixel tiles K at a time,
//     in lockstep and skewed by one step: while the matrix pipe works on tile k of a layer, the vector ALU does the sines /
//     cosine products / conversions of tile k - 1.  Same arithmetic per tile as the workgroup kernel; 1 wave per SIMD, the
//     whole 512-entry register file;
//   * everything a row reuses lives in REGISTERS for the whole row: the 15 weight fragments and the 64 weight-gradient
//     accumulators in the accumulation half of the file (AGPRs: MFMA operands only), the biases as read-only MFMA C operands
//     (no accumulator initialisation).  No fragment is read from LDS per tile;
//   * nothing of a tile is carried in registers from its forward to its backward pass except the packed output gradient: the
//     [pixel][feature] bf16 images of the layer inputs, which the weight gradient needs transposed anyway, are written in the
//     FORWARD pass and stay in LDS; the backward pass re-reads its own rows of them (the lane reads back the 8-byte pieces it
//     wrote) and RECOMPUTES each layer's pre-activation for the cosine: two more MFMAs per layer and tile on a matrix pipe that
//     is ~15 % busy, the transcendental count unchanged (the forward pass evaluates sines only);
//   * no barrier and no cross-wave reduction anywhere: the four waves of a workgroup are independent, the gradient of a row
//     leaves straight from the accumulators, tiles summed in ascending order (deterministic).
// The MFMAs are inline assembly: with more than 256 registers per wave hipcc selects the AGPR-destination form for every MFMA
// builtin and copies each chained accumulator back with 16 v_accvgpr_read (and -amdgpu-mfma-vgpr-form crashes its rewrite
// pass on this kernel).  What the compiler therefore does not see and the structure guarantees: (i) a chained accumulator is
// read by vector instructions one whole STEP after its MFMAs were issued (>= 100 vector instructions, a scheduling barrier
// between: the hazard needs 11 wait states); (ii) the weight-gradient accumulators are touched by MFMAs only until the s_nops
// in front of the epilogue.
// Matches prior_model.py:168-179,237 and test_model.py:347-355,625-627 like the kernels it stands beside.
#include "siren_op16.h"

using namespace rcb;

namespace {
using namespace rcb::op16;

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <int NH, int F, int E, int C, int K>
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
  static constexpr int TILE = XIMG + NH * IMG;   // per tile in flight: the inputs of layers 0 .. NH
  static constexpr int DZ_OFF = K * TILE;        // two dz images (consecutive steps alternate)
  // K == 1 (two waves per SIMD, 256 registers): the four fragments a tile uses once -- the output layer's forward pair and the
  // pair that maps dz_0 onto the input gradient -- stay in LDS (4 KB per wave) and are read where they are used
  static constexpr int NLF = (K == 1) ? 4 : 0;
  static constexpr int LFR_OFF = DZ_OFF + 2 * IMG;
  static constexpr int WAVE_LDS = cmax(LFR_OFF + NLF * 1024, (NFA + NFB) * 1024);   // (the fragments are staged through the same bytes)
  // LDS slot of fragment `slot`, or -1 when it lives in registers
  __host__ __device__ static constexpr int lds_slot(int slot) {
    if (K != 1) return -1;
    if (slot == K0S + 2 * (NH - 1)) return 0;
    if (slot == K0S + 2 * (NH - 1) + 1) return 1;
    if (slot == NFA + 1 + 2 * (NH - 1)) return 2;
    if (slot == NFA + 1 + 2 * (NH - 1) + 1) return 3;
    return -1;
  }
  static constexpr int LDS_BYTES = 4 * WAVE_LDS;
};

// ---- MFMA as inline assembly (see the header) -------------------------------------------------------------------------------
#ifndef RCB_WAVE_DBG_NOP
#define RCB_WAVE_DBG_NOP 0
#endif
// diagnostic builds: the MFMAs of a class retire before the next instruction issues (bit 0: new, 1: zero, 2: acc, 3: gw)
#ifndef RCB_WAVE_NOPN
#define RCB_WAVE_NOPN "s_nop 15\n\ts_nop 15"
#endif
#define RCB_TAIL_ON "\n\t" RCB_WAVE_NOPN
#if RCB_WAVE_DBG_NOP & 1
#define RCB_TAIL_NEW RCB_TAIL_ON
#else
#define RCB_TAIL_NEW ""
#endif
#if RCB_WAVE_DBG_NOP & 2
#define RCB_TAIL_ZERO RCB_TAIL_ON
#else
#define RCB_TAIL_ZERO ""
#endif
#if RCB_WAVE_DBG_NOP & 4
#define RCB_TAIL_ACC RCB_TAIL_ON
#else
#define RCB_TAIL_ACC ""
#endif
#if RCB_WAVE_DBG_NOP & 8
#define RCB_TAIL_GW RCB_TAIL_ON
#else
#define RCB_TAIL_GW ""
#endif
#ifdef RCB_WAVE_DBG_VM0     // diagnostic builds: no memory operation outstanding at the top of a round
#define RCB_DBG_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define RCB_DBG_DRAIN() do { } while (0)
#endif
#define RCB_MFMA_BF16 "v_mfma_f32_32x32x16_bf16"
#define RCB_MFMA_F16 "v_mfma_f32_32x32x16_f16"
// chained form: D (vector registers) = A (fragment, accumulation registers) x B (vector registers) + C
template <typename T>
__device__ __forceinline__ f32x16 mfma_builtin(const i32x4& a, const i32x4& b, const f32x16& c) {
  union { i32x4 i; typename Op16<T>::v8 v; } ua, ub;
  ua.i = a;
  ub.i = b;
  return Op16<T>::mfma(ua.v, ub.v, c);
}
template <typename T, bool ASM = true>
__device__ __forceinline__ f32x16 mfma_new(const i32x4& a, const i32x4& b, const f32x16& c) {
  if constexpr (!ASM) return mfma_builtin<T>(a, b, c);
  f32x16 d;
  if constexpr (Op16<T>::IS_BF16) asm volatile(RCB_MFMA_BF16 " %0, %1, %2, %3" RCB_TAIL_NEW : "=&v"(d) : "a"(a), "v"(b), "v"(c));
  else asm volatile(RCB_MFMA_F16 " %0, %1, %2, %3" RCB_TAIL_NEW : "=&v"(d) : "a"(a), "v"(b), "v"(c));
  return d;
}
template <typename T, bool ASM = true>
__device__ __forceinline__ f32x16 mfma_zero(const i32x4& a, const i32x4& b) {
  if constexpr (!ASM) {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return mfma_builtin<T>(a, b, z);
  }
  f32x16 d;
  if constexpr (Op16<T>::IS_BF16) asm volatile(RCB_MFMA_BF16 " %0, %1, %2, 0" RCB_TAIL_ZERO : "=&v"(d) : "a"(a), "v"(b));
  else asm volatile(RCB_MFMA_F16 " %0, %1, %2, 0" RCB_TAIL_ZERO : "=&v"(d) : "a"(a), "v"(b));
  return d;
}
template <typename T, int WHERE = 0, bool ASM = true>
__device__ __forceinline__ void mfma_acc(f32x16& d, const i32x4& a, const i32x4& b) {
  if constexpr (!ASM) {
    d = mfma_builtin<T>(a, b, d);
    return;
  }
#ifdef RCB_WAVE_DBG_WHERE
  if constexpr (((RCB_WAVE_DBG_WHERE) >> WHERE) & 1) {
    asm volatile(RCB_MFMA_BF16 " %0, %1, %2, %0\n\ts_nop 1" : "+v"(d) : "a"(a), "v"(b));
    return;
  }
#endif
  if constexpr (Op16<T>::IS_BF16) asm volatile(RCB_MFMA_BF16 " %0, %1, %2, %0" RCB_TAIL_ACC : "+v"(d) : "a"(a), "v"(b));
  else asm volatile(RCB_MFMA_F16 " %0, %1, %2, %0" RCB_TAIL_ACC : "+v"(d) : "a"(a), "v"(b));
}
// weight-gradient form: accumulator in the accumulation registers, both operands vector registers
template <typename T, bool ASM = true>
__device__ __forceinline__ void mfma_gw(f32x16& g, const i32x4& a, const i32x4& b) {
  if constexpr (!ASM) {
    g = mfma_builtin<T>(a, b, g);
    return;
  }
  if constexpr (Op16<T>::IS_BF16) asm volatile(RCB_MFMA_BF16 " %0, %1, %2, %0" RCB_TAIL_GW : "+a"(g) : "v"(a), "v"(b));
  else asm volatile(RCB_MFMA_F16 " %0, %1, %2, %0" RCB_TAIL_GW : "+a"(g) : "v"(a), "v"(b));
}
template <typename T, bool ASM = true>
__device__ __forceinline__ f32x16 mfma_gw_zero(const i32x4& z) {     // an all-zero accumulator (z: a zero fragment)
  if constexpr (!ASM) {
    f32x16 g0;
#pragma unroll
    for (int r = 0; r < 16; ++r) g0[r] = 0.f;
    return g0;
  }
  f32x16 g;
  if constexpr (Op16<T>::IS_BF16) asm volatile(RCB_MFMA_BF16 " %0, %1, %1, 0" RCB_TAIL_GW : "=&a"(g) : "v"(z));
  else asm volatile(RCB_MFMA_F16 " %0, %1, %1, 0" RCB_TAIL_GW : "=&a"(g) : "v"(z));
  return g;
}

// Diagnostic build only (-DRCB_WAVE_STAMPS, tools/wave_stamps.py): wave 0 of workgroup 0 records s_memtime at its phase
// boundaries (first row, first and second round) into a device array read back through rcb_debug_wave_stamps.
#ifdef RCB_WAVE_STAMPS
__device__ unsigned long long g_wave_stamps[256];
#define RCB_WSTAMP(k)                                                                                           \
  do {                                                                                                          \
    if (blockIdx.x == 0 && threadIdx.x == 0 && (k) < 256) g_wave_stamps[(k)] = __builtin_amdgcn_s_memtime();     \
  } while (0)
#else
#define RCB_WSTAMP(k) do { } while (0)
#endif
// an MFMA result may be read by vector instructions 11 wait states after the MFMA was issued; the compiler does not see the
// MFMAs, so wherever the step structure does not put a whole step between the two this pads the distance
__device__ __forceinline__ void mfma_settle() { asm volatile("s_nop 7\n\ts_nop 4" ::: "memory"); }

template <typename V>
__device__ __forceinline__ i32x4 as_i4(const V& v) {
  union { V v; i32x4 i; } u;
  u.v = v;
  return u.i;
}

template <typename T, int NH, int F, int E, int C, int MODE, int K>
__global__ void __launch_bounds__(256, K >= 2 ? 1 : 2) siren_wave_kernel(SirenArgs a) {
  using G = WGeo<NH, F, E, C, K>;
  // K >= 2: one wave per SIMD, inline-assembly MFMAs, steps skewed by one.  K == 1: two waves per SIMD (256 registers each: the
  // compiler's own MFMA forms and hazard handling), every step finished before the next is issued -- the other wave of the
  // SIMD fills the gaps
  constexpr bool ASM = K >= 2, SKEW = K >= 2;
  // BIASM: the bias enters the accumulator through the matrix pipe -- one more k-step whose A fragment holds the bias as
  // (hi, lo) bf16 halves in its first two k-slots against a B operand of ones: 3 x 4 registers per row instead of the 3 x 16 of
  // read-only C tuples, no vector instruction, no LDS read; the pipe has the room
  constexpr bool BIASM = K == 1;
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
  unsigned char* wlds = smem_raw + wave * G::WAVE_LDS;
  // images of tile k: layer 0's input at x_img(k), the input of layer l >= 1 (the sines of layer l - 1) at s_img(k, l - 1)
  auto x_img = [&](int k) -> T* { return reinterpret_cast<T*>(wlds + k * G::TILE); };
  auto s_img = [&](int k, int l) -> T* { return reinterpret_cast<T*>(wlds + k * G::TILE + G::XIMG + l * G::IMG); };
  auto dz_img = [&](int set) -> T* { return reinterpret_cast<T*>(wlds + G::DZ_OFF + set * G::IMG); };
  const int P = a.P;
  const int ntiles = (P + 31) >> 5;
  const int nunits = a.G * a.chunks;
  // the 8-byte pieces a lane writes into (and reads back from) an image of row stride TSA: k-step s, halves 0 / 1
  int po[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    po[s][0] = swz(q, 16 * s + 4 * h, G::TSA);
    po[s][1] = swz(q, 16 * s + 8 + 4 * h, G::TSA);
  }
  auto put_img = [&](T* img, const bf16x8 (&v)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      union { bf16x8 v; bf16x4 hlf[2]; } uu;
      uu.v = v[s];
      *reinterpret_cast<bf16x4*>(img + po[s][0]) = uu.hlf[0];
      *reinterpret_cast<bf16x4*>(img + po[s][1]) = uu.hlf[1];
    }
  };
  auto get_img = [&](const T* img, bf16x8 (&v)[2]) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      union { bf16x8 v; bf16x4 hlf[2]; } uu;
      uu.hlf[0] = *reinterpret_cast<const bf16x4*>(img + po[s][0]);
      uu.hlf[1] = *reinterpret_cast<const bf16x4*>(img + po[s][1]);
      v[s] = uu.v;
    }
  };
  // layer-0 input image: half-wave 0 holds features [0, F), half-wave 1 features [F, F + E); K0S pieces of 8 features
  const int xbase = (h == 0) ? 0 : F, xkh = (h == 0) ? F : E;
  auto put_x = [&](T* img, const bf16x8 (&v)[K0S]) {
#pragma unroll
    for (int s = 0; s < K0S; ++s) {
      union { bf16x8 v; bf16x4 hlf[2]; } uu;
      uu.v = v[s];
      if (8 * s + 8 <= xkh) {
        *reinterpret_cast<bf16x4*>(img + swz(q, xbase + 8 * s, G::TSX)) = uu.hlf[0];
        *reinterpret_cast<bf16x4*>(img + swz(q, xbase + 8 * s + 4, G::TSX)) = uu.hlf[1];
      }
    }
  };
  auto get_x = [&](const T* img, bf16x8 (&v)[K0S]) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int s = 0; s < K0S; ++s) {
      union { bf16x8 v; bf16x4 hlf[2]; } uu;
#pragma unroll
      for (int j = 0; j < 8; ++j) uu.v[j] = (T)0.f;
      if (8 * s + 8 <= xkh) {
        uu.hlf[0] = *reinterpret_cast<const bf16x4*>(img + swz(q, xbase + 8 * s, G::TSX));
        uu.hlf[1] = *reinterpret_cast<const bf16x4*>(img + swz(q, xbase + 8 * s + 4, G::TSX));
      }
      v[s] = uu.v;
    }
  };

  for (int u = blockIdx.x * 4 + wave; u < nunits; u += gridDim.x * 4) {
    RCB_WSTAMP(0);
    const int g = u % a.G, chunk = u / a.G;              // chunk-major: partial buffers are [chunk][g]
    const int n = g / a.S;
    const long long pe_row = pe_row_base(a, g);
    const float* __restrict__ wsrc = a.wvec + (long long)g * a.w_stride;

    // ---- the row's weights as MFMA A fragments, in accumulation registers for the whole row -----------------------------------
    // fragment k order: the chained accumulator's (siren_mlp_bf16.hip): k-slot (step s, lane half h, element j) = row fk(s,h,j).
    // Gathered from global memory into vector registers, staged through LDS (a 16-byte store and load per lane and fragment:
    // the only way to define a four-register accumulation tuple from inline assembly), 5 fragments per wait.
    i32x4 FR[NFA + NFB];
    {
      i32x4* stage = reinterpret_cast<i32x4*>(wlds);
#pragma unroll
      for (int slot = 0; slot < NFA + NFB; ++slot) {
        float w8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float w = 0.f;
          if (slot < K0S) {
            const int kk = 8 * slot + j;
            const int row = (h == 0) ? (kk < F ? kk : -1) : (kk < E ? F + kk : -1);
            if (row >= 0) w = wsrc[G::off(0) + HID + row * HID + q];
          } else if (slot < NFA) {
            const int l = 1 + (slot - K0S) / 2, st = (slot - K0S) & 1;
            const int no = (l == NL - 1) ? C : HID;
            if (q < no) w = wsrc[G::off(l) + no + fk(st, h, j) * no + q];
          } else {
            const int b = slot - NFA;
            if (b == 0) {
              const int oo = fk(0, h, j);
              if (oo < C) w = wsrc[G::off(NL - 1) + C + q * C + oo];
            } else if (b < 1 + 2 * (NH - 1)) {
              const int l = (NH - 1) - (b - 1) / 2, st = (b - 1) & 1;
              w = wsrc[G::off(l) + HID + q * HID + fk(st, h, j)];
            } else {
              const int st = (b - 1 - 2 * (NH - 1));
              if (q < E) w = wsrc[G::off(0) + HID + (F + q) * HID + fk(st, h, j)];
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
        if constexpr (ASM) stage[slot * 64 + lane] = as_i4(fv);
        else if (G::lds_slot(slot) >= 0) reinterpret_cast<i32x4*>(wlds + G::LFR_OFF)[G::lds_slot(slot) * 64 + lane] = as_i4(fv);
        else FR[slot] = as_i4(fv);
      }
      if constexpr (ASM) {
      // (wlds is a generic pointer into LDS: its low 32 bits are the LDS byte address)
      const unsigned sbase = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)wlds) + lane * 16;
      // five fragments per wait: the destination registers of an un-waited read must not be touched (moved, spilled) by
      // code the compiler places after the statement, so every statement ends with its own wait
      constexpr int NFR = NFA + NFB;
#pragma unroll
      for (int s0 = 0; s0 + 5 <= NFR; s0 += 5) {
        const unsigned ad = sbase + s0 * 1024;
        asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:1024\n\tds_read_b128 %2, %5 offset:2048\n\t"
                     "ds_read_b128 %3, %5 offset:3072\n\tds_read_b128 %4, %5 offset:4096\n\ts_waitcnt lgkmcnt(0)"
                     : "=a"(FR[s0]), "=a"(FR[s0 + 1]), "=a"(FR[s0 + 2]), "=a"(FR[s0 + 3]), "=a"(FR[s0 + 4])
                     : "v"(ad)
                     : "memory");
      }
#pragma unroll
      for (int slot = NFR / 5 * 5; slot < NFR; ++slot) {
        const unsigned ad = sbase + slot * 1024;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=a"(FR[slot]) : "v"(ad) : "memory");
      }
      }
    }
    auto FRG = [&](int slot) -> i32x4 {
      // (volatile: the read stays where it is used -- hoisted out of the tile loop it would occupy the registers it is meant to free)
      if (G::lds_slot(slot) >= 0) return reinterpret_cast<const volatile i32x4*>(wlds + G::LFR_OFF)[G::lds_slot(slot) * 64 + lane];
      return FR[slot];
    };
    // biases of the sine layers in accumulator layout (register r <-> row rho(r, h)), in revolutions: read-only C operands
    f32x16 BIAS[BIASM ? 1 : NH];
    i32x4 BFR[BIASM ? NH : 1];
    i32x4 ONES;
    if constexpr (BIASM) {
      bf16x8 o8;
#pragma unroll
      for (int j = 0; j < 8; ++j) o8[j] = (T)1.0f;
      ONES = as_i4(o8);
#pragma unroll
      for (int l = 0; l < NH; ++l) {
        const float b = wsrc[G::off(l) + q] * (a.k_hi * WS);
        const T bh = (T)b;
        const T bl = (T)(b - (float)bh);
        bf16x8 f8;
#pragma unroll
        for (int j = 0; j < 8; ++j) f8[j] = (T)0.f;
        f8[0] = (h == 0) ? bh : (T)0.f;
        f8[1] = (h == 0) ? bl : (T)0.f;
        BFR[l] = as_i4(f8);
      }
    } else {
#pragma unroll
      for (int l = 0; l < NH; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) BIAS[l][r] = wsrc[G::off(l) + rho(r, h)] * (a.k_hi * WS);
    }
    float bout[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      bout[r] = 0.f;
      if (rho(r, 0) < C || rho(r, 1) < C) bout[r] = (rho(r, h) < C) ? wsrc[G::off(NL - 1) + (rho(r, h) < C ? rho(r, h) : 0)] : 0.f;
    }

    f32x16 gW[NL + NB0 - 1];
    {
      const i32x4 z4 = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < NL + NB0 - 1; ++i) gW[i] = mfma_gw_zero<T, ASM>(z4);
    }
    float gb[NL];
    float sse_local = 0.f;
#pragma unroll
    for (int l = 0; l < NL; ++l) gb[l] = 0.f;

    // ---- inputs / targets of a tile: requested one round ahead ---------------------------------------------------------------
    uint4 raw16[K][K0S];
    float ynext[K][16];
    const int t0 = (int)((long long)chunk * ntiles / a.chunks), t1 = (int)((long long)(chunk + 1) * ntiles / a.chunks);
    const float* __restrict__ yrow = a.yin + (MODE == MODE_LOSS ? (long long)n : (long long)g) * P * C;
    auto fetch = [&](int k, int tile) {
      const int tl = tile < t1 ? tile : t1 - 1;
      const int pp = tl * 32 + q;
      const int pcl = pp < P ? pp : P - 1;
      const __bf16* s16 = (h == 0) ? (reinterpret_cast<const __bf16*>(a.xf16) + (long long)n * a.xf_stride + (long long)pcl * F)
                                   : (reinterpret_cast<const __bf16*>(a.pe) + (pe_row + pe_pix_off(a, pcl)) * E);
#pragma unroll
      for (int s = 0; s < K0S; ++s) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (8 * s + 8 <= xkh) v = reinterpret_cast<const uint4*>(s16)[s];
        raw16[k][s] = v;
      }
      const int yoff = pcl * C;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ynext[k][r] = 0.f;
        if (rho(r, 0) < C || rho(r, 1) < C) {
          const int row = rho(r, h);
          ynext[k][r] = yrow[yoff + (row < C ? row : 0)];
        }
      }
    };
#pragma unroll
    for (int k = 0; k < K; ++k) fetch(k, t0 + k);

    RCB_WSTAMP(1);
    for (int tb = t0; tb < t1; tb += K) {
      RCB_DBG_DRAIN();
      const int rnd = (tb - t0) / K;
      // (the images are stored and read back through different vector types: nothing of this round moves above the
      // transposed reads of the previous one)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      bf16x8 xin[K][K0S];
      float yv[K][16];
      bool valid[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int s = 0; s < K0S; ++s) {
          union { uint4 u4; bf16x8 v; } cv;
          cv.u4 = raw16[k][s];
          xin[k][s] = cv.v;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[k][r] = ynext[k][r];
        valid[k] = (tb + k < t1) && ((tb + k) * 32 + q < P);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) fetch(k, tb + K + k);

      // ================= forward: NL * K steps, step j = (layer j / K, tile j % K).  The matrix instructions of step j are
      // issued first; the vector work on step j - 1's accumulator (sines, conversion, image stores) runs behind them =========
      bf16x8 Scur[K][2];                      // sines of the layer being finished, per tile: the next layer's B operand
      bf16x8 dzb[K][2];                       // packed gradient of the layer being processed, per tile
      f32x16 acc[2];
#pragma unroll
      for (int j = 0; j <= NL * K; ++j) {
        if (j < NL * K) {
          const int l = j / K, k = j % K;
          f32x16 c;
          if (l == 0) {
            put_x(x_img(k), xin[k]);
            if constexpr (BIASM) {
              c = mfma_zero<T, ASM>(BFR[0], ONES);
              mfma_acc<T, 1, ASM>(c, FR[0], as_i4(xin[k][0]));
            } else {
              c = mfma_new<T, ASM>(FR[0], as_i4(xin[k][0]), BIAS[0]);
            }
#pragma unroll
            for (int s = 1; s < K0S; ++s) mfma_acc<T, 1, ASM>(c, FR[s], as_i4(xin[k][s]));
          } else if (l < NH) {
            if constexpr (BIASM) {
              c = mfma_zero<T, ASM>(BFR[l], ONES);
              mfma_acc<T, 2, ASM>(c, FR[K0S + 2 * (l - 1)], as_i4(Scur[k][0]));
            } else {
              c = mfma_new<T, ASM>(FR[K0S + 2 * (l - 1)], as_i4(Scur[k][0]), BIAS[l]);
            }
            mfma_acc<T, 2, ASM>(c, FR[K0S + 2 * (l - 1) + 1], as_i4(Scur[k][1]));
          } else {
            c = mfma_zero<T, ASM>(FRG(K0S + 2 * (NH - 1)), as_i4(Scur[k][0]));
            mfma_acc<T, 3, ASM>(c, FRG(K0S + 2 * (NH - 1) + 1), as_i4(Scur[k][1]));
          }
          acc[j & 1] = c;
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef RCB_WAVE_STAMPS
        if (rnd < 2) RCB_WSTAMP(2 + rnd * 40 + j);
#endif
        if (SKEW ? j > 0 : j < NL * K) {
          constexpr int DJ = SKEW ? 1 : 0;
          const int l = (j - DJ) / K, k = (j - DJ) % K;
          // step j - 1's MFMAs have a whole step of vector work behind them, except the first step of a round and where the
          // step before was an output-layer step (a dozen instructions)
          if (ASM && (j == 1 || l == NH)) {
            mfma_settle();
            __builtin_amdgcn_sched_barrier(0);
          }
          const f32x16 c = acc[(j - DJ) & 1];
          if (l < NH) {
            f32x16 sv;
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = __builtin_amdgcn_sinf((WS == 1.0f) ? c[r] : c[r] * (1.0f / WS));
            Scur[k][0] = pack8<T>(sv, 0);
            Scur[k][1] = pack8<T>(sv, 1);
            put_img(s_img(k, l), Scur[k]);
          } else {
            // output layer: loss gradient (or the upstream gradient) on the C rows that exist, packed with literal zeros
            f32x16 dzo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = 0.f;
              if (rho(r, 0) < C || rho(r, 1) < C) {
                const bool ok = valid[k] && rho(r, h) < C;
                if (MODE == MODE_LOSS) {
                  const float y = (WS != 1.0f) ? (c[r] * (1.0f / WS) + bout[r]) : (c[r] + bout[r]);
                  const float diff = ok ? (y - yv[k][r]) : 0.f;
                  sse_local += diff * diff;
                  v = (2.0f * GS) * a.dy_scale * diff;
                } else {
                  v = ok ? yv[k][r] * GS : 0.f;
                }
              }
              dzo[r] = v;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
              for (int jj = 0; jj < 8; ++jj) {
                const int r = 8 * s + jj;
                dzb[k][s][jj] = (rho(r, 0) < C || rho(r, 1) < C) ? (T)dzo[r] : (T)0.0f;
              }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }

      // ================= backward: step j = (layer NL - 1 - j / K, tile j % K) ===================================================
      // issue part of step j: dz image of (l, k) -> set j & 1; data-gradient MFMAs of (l, k); the pre-activation of layer l - 1
      // recomputed from its input rows (read back from the tile's image one step ahead).  Behind them, on step j - 1: the
      // transposed reads of its images, cosine x data gradient -> packed dz of the layer below, weight-gradient MFMAs.
      f32x16 dh[2], zc[2];
      bf16x8 rin[2][cmax(K0S, 2)];             // input rows of the layer whose pre-activation step j recomputes
      auto read_rows = [&](int j) {
        const int l = NL - 1 - j / K, k = j % K;
        if (l >= 1) {
          if (l - 1 == 0) {
            bf16x8 t[K0S];
            get_x(x_img(k), t);
#pragma unroll
            for (int s = 0; s < K0S; ++s) rin[j & 1][s] = t[s];
          } else {
            bf16x8 t[2];
            get_img(s_img(k, l - 2), t);
            rin[j & 1][0] = t[0];
            rin[j & 1][1] = t[1];
          }
        }
      };
      read_rows(0);
#pragma unroll
      for (int j = 0; j <= NL * K; ++j) {
        if (j < NL * K) {
          const int l = NL - 1 - j / K, k = j % K, set = j & 1;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          put_img(dz_img(set), dzb[k]);
          f32x16 d;
          if (l == NL - 1) {
            d = mfma_zero<T, ASM>(FR[NFA + 0], as_i4(dzb[k][0]));
          } else if (l > 0) {
            const int base = NFA + 1 + 2 * ((NH - 1) - l);
            d = mfma_zero<T, ASM>(FR[base], as_i4(dzb[k][0]));
            mfma_acc<T, 4, ASM>(d, FR[base + 1], as_i4(dzb[k][1]));
          } else {
            const int base = NFA + 1 + 2 * (NH - 1);
            d = mfma_zero<T, ASM>(FRG(base), as_i4(dzb[k][0]));
            mfma_acc<T, 5, ASM>(d, FRG(base + 1), as_i4(dzb[k][1]));
          }
          dh[j & 1] = d;
          if (l > 0) {
            f32x16 z;
            if (l - 1 == 0) {
              if constexpr (BIASM) {
                z = mfma_zero<T, ASM>(BFR[0], ONES);
                mfma_acc<T, 6, ASM>(z, FR[0], as_i4(rin[j & 1][0]));
              } else {
                z = mfma_new<T, ASM>(FR[0], as_i4(rin[j & 1][0]), BIAS[0]);
              }
#pragma unroll
              for (int s = 1; s < K0S; ++s) mfma_acc<T, 6, ASM>(z, FR[s], as_i4(rin[j & 1][s]));
            } else {
              if constexpr (BIASM) {
                z = mfma_zero<T, ASM>(BFR[l - 1], ONES);
                mfma_acc<T, 7, ASM>(z, FR[K0S + 2 * (l - 2)], as_i4(rin[j & 1][0]));
              } else {
                z = mfma_new<T, ASM>(FR[K0S + 2 * (l - 2)], as_i4(rin[j & 1][0]), BIAS[l - 1]);
              }
              mfma_acc<T, 7, ASM>(z, FR[K0S + 2 * (l - 2) + 1], as_i4(rin[j & 1][1]));
            }
            zc[j & 1] = z;
          }
          if (j + 1 < NL * K) read_rows(j + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef RCB_WAVE_STAMPS
        if (rnd < 2) RCB_WSTAMP(2 + rnd * 40 + 20 + j);
#endif
        if (SKEW ? j > 0 : j < NL * K) {
          constexpr int DJ = SKEW ? 1 : 0;
          const int l = NL - 1 - (j - DJ) / K, k = (j - DJ) % K, set = (j - DJ) & 1;
          if (ASM && (j == 1 || j == NL * K)) {
            mfma_settle();
            __builtin_amdgcn_sched_barrier(0);
          }
          // transposed fragments of step j - 1's images
          bf16x8 av[2], bv[NB0][2];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (images are stored and read back through different types)
          auto tr_reads = [&](int s) {
            av[s] = read_tr<T>(dz_img(set), G::TSA, s, lane, 0);
#pragma unroll
            for (int blk = 0; blk < ((l == 0) ? NB0 : 1); ++blk)
              bv[blk][s] = (l == 0) ? read_tr<T>(x_img(k), G::TSX, s, lane, 32 * blk) : read_tr<T>(s_img(k, l - 1), G::TSA, s, lane, 0);
          };
          tr_reads(0);
          if (SKEW) tr_reads(1);        // (two waves per SIMD: the second k-step's fragments are read behind the first's MFMA)
          // data gradient of step j - 1: dz of the layer below (w0 is in the fragments), packed at once
          const f32x16 d = dh[(j - DJ) & 1];
          if (l > 0) {
            const f32x16 z = zc[(j - DJ) & 1];
            f32x16 dzn;
#pragma unroll
            for (int r = 0; r < 16; ++r) dzn[r] = d[r] * __builtin_amdgcn_cosf((WS == 1.0f) ? z[r] : z[r] * (1.0f / WS));
            dzb[k][0] = pack8<T>(dzn, 0);
            dzb[k][1] = pack8<T>(dzn, 1);
          } else if (a.dpe != nullptr) {
            const int p = (tb + k) * 32 + q;
            if (valid[k]) {
              if (a.pe_bf16) {
                __bf16* d16 = reinterpret_cast<__bf16*>(a.dpe) + (pe_row + pe_pix_off(a, p)) * E;
#pragma unroll
                for (int g4 = 0; g4 < E / 8; ++g4) {
                  typename Op16<__bf16>::v4 ob = {(__bf16)(d[4 * g4] * (1.0f / (GS * WS))), (__bf16)(d[4 * g4 + 1] * (1.0f / (GS * WS))),
                                                  (__bf16)(d[4 * g4 + 2] * (1.0f / (GS * WS))), (__bf16)(d[4 * g4 + 3] * (1.0f / (GS * WS)))};
                  *reinterpret_cast<typename Op16<__bf16>::v4*>(d16 + 8 * g4 + 4 * h) = ob;
                }
              } else {
                float* dst = a.dpe + (pe_row + pe_pix_off(a, p)) * E;
#pragma unroll
                for (int g4 = 0; g4 < E / 8; ++g4)
                  *reinterpret_cast<float4*>(dst + 8 * g4 + 4 * h) =
                      make_float4(d[4 * g4] * (1.0f / (GS * WS)), d[4 * g4 + 1] * (1.0f / (GS * WS)), d[4 * g4 + 2] * (1.0f / (GS * WS)),
                                  d[4 * g4 + 3] * (1.0f / (GS * WS)));
              }
            }
          }
          // weight gradient of step j - 1: dW_l += dz^T x input over the tile's 32 pixels, bias gradient from the dz fragments
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (!SKEW && s == 1) {
              __builtin_amdgcn_sched_barrier(0);
              tr_reads(1);
            }
            gb[l] = sum8_16<T>(av[s], gb[l]);
#pragma unroll
            for (int blk = 0; blk < ((l == 0) ? NB0 : 1); ++blk) {
              const int gi = (l == 0) ? blk : (l + NB0 - 1);
              mfma_gw<T, ASM>(gW[gi], as_i4(av[s]), as_i4(bv[blk][s]));
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    RCB_WSTAMP(100);
    // ---- the row's gradient, straight from the accumulators --------------------------------------------------------------------
    if constexpr (ASM) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last weight-gradient MFMAs retire before their registers are read
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
      const bool vec4 = dst != nullptr && d16 == nullptr && ((a.w_stride & 3) == 0) && ((reinterpret_cast<size_t>(a.dwvec) & 15) == 0);
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const int ol = G::off(l), no = G::lout(l);
        const float bt = (gb[l] + __shfl_xor(gb[l], 32, 64)) * (1.0f / GS);
        if (h == 0 && q < no) put(ol + q, bt);
#pragma unroll
        for (int blk = 0; blk < ((l == 0) ? NB0 : 1); ++blk) {
          const int gi = (l == 0) ? blk : (l + NB0 - 1);
          const f32x16 gv = gW[gi];
          const int i = 32 * blk + q;
          if (i < G::lin(l)) {
            if (no == HID && (ol + no) % 4 == 0 && vec4) {
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4*>(dst + ol + no + i * no + 8 * r4 + 4 * h) =
                    make_float4(gv[4 * r4] * (1.0f / GS), gv[4 * r4 + 1] * (1.0f / GS), gv[4 * r4 + 2] * (1.0f / GS),
                                gv[4 * r4 + 3] * (1.0f / GS));
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                if (rho(r, 0) < no || rho(r, 1) < no) {
                  const int o = rho(r, h);
                  if (o < no) put(ol + no + i * no + o, gv[r] * (1.0f / GS));
                }
              }
            }
          }
        }
      }
    }
    RCB_WSTAMP(101);
  }
}

template <typename T, int NH, int F, int E, int C, int MODE, int K>
int launch_wave(const SirenArgs& a, hipStream_t st) {
  using G = WGeo<NH, F, E, C, K>;
  auto kfn = siren_wave_kernel<T, NH, F, E, C, MODE, K>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return fail((int)e, "siren(wave): hipFuncSetAttribute: %s", hipGetErrorString(e));
  const long long units = (long long)a.G * a.chunks;
  long long blocks = (units + 3) / 4;
  const long long cap = K >= 2 ? 256 : 512;             // one workgroup (four independent waves) per CU at 512 registers, two at 256
  if (blocks > cap) blocks = cap;
  kfn<<<(unsigned)blocks, 256, G::LDS_BYTES, st>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

template <typename T, int NH, int F, int E, int C, int MODE>
int launch_variant(int variant, const SirenArgs& a, hipStream_t st) {
#ifdef RCB_WAVE_ONLY          // experiment builds: one instance only
  return launch_wave<T, NH, F, E, C, MODE, RCB_WAVE_ONLY>(a, st);
#else
  if (variant == 1) return launch_wave<T, NH, F, E, C, MODE, 1>(a, st);
  if (variant == 2) return launch_wave<T, NH, F, E, C, MODE, 2>(a, st);
  return launch_wave<T, NH, F, E, C, MODE, 4>(a, st);
#endif
}

}  // namespace

namespace rcb {
// RCB_SIREN_WAVE in the environment sets the initial value (A/B runs): 0 = the workgroup kernel everywhere, 2 / 4 = tiles in flight
int& siren_wave_tiles() {
  static int v = [] {
    const char* e = getenv("RCB_SIREN_WAVE");
    return e ? atoi(e) : 4;
  }();
  return v;
}
// *taken = true when this family has an instance for the request (the launch result is returned); otherwise the workgroup
// kernel (siren_mlp_bf16.hip) runs it
int siren_wave_dispatch(int mode, const rcb_siren_desc* d, SirenArgs& a, hipStream_t st, int variant, bool* taken) {
  *taken = false;
  if (mode == MODE_FWD || d->precision != 1 || !a.pe_bf16 || a.xf16 == nullptr || d->hidden != HID) return RCB_OK;
#define RCB_CASE(NHv, Fv, Ev, Cv)                                                                               \
  if (d->n_hidden == NHv && d->fourier_dim == Fv && d->pe_dim == Ev && d->out_dim == Cv) {                      \
    *taken = true;                                                                                              \
    return mode == MODE_LOSS ? launch_variant<__bf16, NHv, Fv, Ev, Cv, MODE_LOSS>(variant, a, st)               \
                             : launch_variant<__bf16, NHv, Fv, Ev, Cv, MODE_BWD>(variant, a, st);               \
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

#ifdef RCB_WAVE_STAMPS
extern "C" int rcb_debug_wave_stamps(unsigned long long* dst, int n_entries) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wave_stamps), sizeof(unsigned long long) * n_entries);
}
#endif
