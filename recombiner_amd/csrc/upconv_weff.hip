// N1 (host-side algebra of the phase form, moved onto the device): effective weights of the three upsampling stages
// from the conv weights, and the transposed map for their gradients.  Replaces the einsum chains of
// upsample_fast._UpsampleCifarFn (about twenty tiny launches per training step) by one launch each way.
//
//   stage 1 (nearest x4 of the 2x2 latent grid -> conv 5x5 pad 2 -> 8x8): every output pixel only sees the 2x2
//     source pixels, so the stage is one dense map  z1[b, (y,x,o)] = b1[o] + sum_{(s,t,i)} lpe[b, (s,t,i)] Weff1[(s,t,i), (y,x,o)]
//       Weff1[s,t,i,y,x,o] = sum_{k,l : (y+k-2) in rows of s, (x+l-2) in cols of t} W1[o,i,k,l]
//   stages 2, 3 (nearest x2 -> conv 3x3 pad 1): Weff[ty,tx,ci,a,b,co] = sum_{k in K(a,ty), l in K(b,tx)} W[co,ci,k,l]
//       with K(0,0) = {0}, K(0,1) = {1,2}, K(1,0) = {0,1}, K(1,1) = {2}      (layout of upconv.hip's weff_index)
#include "rcb_common.h"

using namespace rcb;

namespace {

__device__ __forceinline__ int tap_of(int a, int k) { return a == 0 ? (k == 0 ? 0 : 1) : (k == 2 ? 1 : 0); }

// one effective weight straight from the 3x3 taps: Weff[ty][tx][ci][a][b][co] = sum_{k in K(a,ty), l in K(b,tx)} W[co][ci][k][l]
__device__ __forceinline__ float weff_of(const float* __restrict__ W, int ty, int tx, int ci, int a, int b, int co) {
  const float* w = W + (co * 64 + ci) * 9;
  // (all nine taps loaded, the ones outside (a, ty) x (b, tx) dropped by a select: a branch per tap serialises the loads)
  float wv[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wv[k] = w[k];
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int l = 0; l < 3; ++l) acc += (tap_of(a, k) == ty && tap_of(b, l) == tx) ? wv[k * 3 + l] : 0.f;
  return acc;
}

struct WeffArgs {
  const float* W1;   // [64][128][5][5]
  const float* b1;   // [64]
  const float* W2;   // [64][64][3][3]
  const float* W3;   // [16][64][3][3]
  void* weff1;       // [512][4096] bf16 or fp32
  void* b1rep;       // [4096] same type: b1 tiled over the 64 output pixels
  float* weff2;      // [2][2][64][2][2][64]
  float* weff3;      // [2][2][64][2][2][16]
  int bf16_out;
  uint4* pack;       // nullable: MFMA fragment images for the phase-conv kernels (layout: include/rcb.h)
};

template <int COUT>
__device__ __forceinline__ void build_stage(const float* __restrict__ W, float* __restrict__ out, int ci, float* w, int tid) {
  {   // (loads first, then stores: see the stage-1 branch)
    constexpr int NT = (COUT * 9 + 255) / 256;
    float t3[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const int e = tid + 256 * k;
      t3[k] = e < COUT * 9 ? W[((e / 9) * 64 + ci) * 9 + e % 9] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NT; ++k)
      if (tid + 256 * k < COUT * 9) w[tid + 256 * k] = t3[k];
  }
  __syncthreads();
  for (int e = tid; e < 16 * COUT; e += 256) {
    const int co = e % COUT, ab = (e / COUT) & 3, tt = e / (4 * COUT);
    const int a = ab >> 1, b = ab & 1, ty = tt >> 1, tx = tt & 1;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = 0; l < 3; ++l)
        if (tap_of(a, k) == ty && tap_of(b, l) == tx) acc += w[co * 9 + k * 3 + l];
    out[((tt * 64 + ci) * 4 + ab) * COUT + co] = acc;
  }
}

// blocks [0, 512): stage 1, one (input channel i, source pixel st) each; [512, 576): stage 2 per ci;
// [576, 640): stage 3 per ci; 640: the tiled bias; [641, 641 + 88): the packed MFMA fragments (optional)
__global__ void __launch_bounds__(256) weff_build_kernel(WeffArgs a) {
  __shared__ float w[64 * 25];
  const int blk = blockIdx.x, tid = threadIdx.x;
  if (blk < 512) {
    const int i = blk >> 2, st = blk & 3, s = st >> 1, t = st & 1;
    {   // all loads first, then the stores (a load-store loop waits one memory round trip per trip: seven here)
      float t7[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int e = tid + 256 * k;
        t7[k] = e < 64 * 25 ? a.W1[((e / 25) * 128 + i) * 25 + e % 25] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 7; ++k)
        if (tid + 256 * k < 64 * 25) w[tid + 256 * k] = t7[k];
    }
    __syncthreads();
    for (int rem = 2 * tid; rem < 4096; rem += 512) {         // two neighbouring output channels per thread: 4-byte bf16 stores
      const int y = rem >> 9, x = (rem >> 6) & 7, o = rem & 63;
      // taps whose source row y + k - 2 lies in rows [4s, 4s + 3] of the up-sampled grid (and likewise for columns)
      const int k0 = max(0, 4 * s + 2 - y), k1 = min(4, 4 * s + 5 - y);
      const int l0 = max(0, 4 * t + 2 - x), l1 = min(4, 4 * t + 5 - x);
      float acc0 = 0.f, acc1 = 0.f;
      for (int k = k0; k <= k1; ++k)
        for (int l = l0; l <= l1; ++l) {
          acc0 += w[o * 25 + k * 5 + l];
          acc1 += w[(o + 1) * 25 + k * 5 + l];
        }
      const long long idx = ((long long)st * 128 + i) * 4096 + rem;
      if (a.bf16_out) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 pr = {(__bf16)acc0, (__bf16)acc1};
        *reinterpret_cast<bf16x2*>(reinterpret_cast<__bf16*>(a.weff1) + idx) = pr;
      } else {
        *reinterpret_cast<float2*>(reinterpret_cast<float*>(a.weff1) + idx) = make_float2(acc0, acc1);
      }
    }
  } else if (blk < 576) {
    build_stage<64>(a.W2, a.weff2, blk - 512, w, tid);
  } else if (blk < 640) {
    build_stage<16>(a.W3, a.weff3, blk - 576, w, tid);
  } else if (blk == 640) {
    for (int e = tid; e < 4096; e += 256) {
      const float v = a.b1[e & 63];
      if (a.bf16_out) reinterpret_cast<__bf16*>(a.b1rep)[e] = (__bf16)v;
      else reinterpret_cast<float*>(a.b1rep)[e] = v;
    }
  } else if (a.pack) {
    // one 16-byte fragment (8 bf16) per thread, each element summed straight from the conv taps
    const int e = (blk - 641) * 256 + tid;          // < RCB_UPCONV_PACK_UINT4
    const int lane = e & 63, q = lane & 31, h = lane >> 5;
    union { __bf16 v[8]; uint4 u; } f;
    if (e < 8192) {                                  // F2: stage-2 forward, [ph][mt][ty][tx][kb][lane]
      const int s5 = e >> 6, kb = s5 & 3, tx = (s5 >> 2) & 1, ty = (s5 >> 3) & 1, mt = (s5 >> 4) & 1, ph = s5 >> 5;
#pragma unroll
      for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)weff_of(a.W2, ty, tx, 16 * kb + 8 * h + j, ph >> 1, ph & 1, 32 * mt + q);
    } else if (e < 16384) {                          // D2: stage-2 data gradient, [kh][mt][c][kb][lane]
      const int s5 = (e - 8192) >> 6, kb = s5 & 3, c = (s5 >> 2) & 7, mt = (s5 >> 5) & 1, kh = s5 >> 6;
      const int n = 8 * kh + c, ry = (n >> 2) - 1, rx = (n & 3) - 1;
      const int pa = ry & 1, ty = (ry <= 0) ? 1 : 0, pb = rx & 1, tx = (rx <= 0) ? 1 : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)weff_of(a.W2, ty, tx, 32 * mt + q, pa, pb, 16 * kb + 8 * h + j);
    } else if (e < 20480) {                          // F3: stage-3 forward, [pa][pb][ty][tx][kb2][lane]: A operands of
      //                                                 v_mfma_f32_16x16x32_bf16 (row co = lane & 15, ci = 32 kb2 + 8 (lane >> 4) + j);
      //                                                 the second half of the region is unused
      if (e >= 16384 + 2048) return;
      const int s5 = (e - 16384) >> 6, kb2 = s5 & 1, tx = (s5 >> 1) & 1, ty = (s5 >> 2) & 1, pb = (s5 >> 3) & 1, pa = s5 >> 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)weff_of(a.W3, ty, tx, 32 * kb2 + 8 * (lane >> 4) + j, pa, pb, lane & 15);
    } else {                                         // D3: stage-3 data gradient, [combo][mt][lane]
      const int s5 = (e - 20480) >> 6, mt = s5 & 1, n = s5 >> 1;
      const int ry = (n >> 2) - 1, rx = (n & 3) - 1;
      const int pa = ry & 1, ty = (ry <= 0) ? 1 : 0, pb = rx & 1, tx = (rx <= 0) ? 1 : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)weff_of(a.W3, ty, tx, 32 * mt + q, pa, pb, 8 * h + j);
    }
    a.pack[e] = f.u;
  }
}

struct WeffGradArgs {
  const void* dweff1;   // [512][4096] bf16 or fp32
  const float* dweff2;
  const float* dweff3;
  float* dW1;
  float* dW2;
  float* dW3;
  int bf16_in;
  const float* db1_partial;   // nullable [n_partial][64]: per-workgroup channel sums of dz1 (rcb_upconv_dgrad)
  int n_partial;
  float* db1;                 // [64]
};

template <int COUT>
__device__ __forceinline__ void grad_stage(const float* __restrict__ dweff, float* __restrict__ dW, int ci, int tid) {
  for (int e = tid; e < COUT * 9; e += 256) {
    const int co = e % COUT, kl = e / COUT, k = kl / 3, l = kl % 3;
    float acc = 0.f;
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      const int tt = tap_of(ab >> 1, k) * 2 + tap_of(ab & 1, l);
      acc += dweff[((tt * 64 + ci) * 4 + ab) * COUT + co];
    }
    dW[(co * 64 + ci) * 9 + kl] = acc;
  }
}

// blocks [0, 800): stage 1, one thread per dW1[o, i, k, l] (o fastest: the 64 lanes of a wave read 64 consecutive
// channels of dWeff1); [800, 864): stage 2 per ci; [864, 928): stage 3 per ci
__global__ void __launch_bounds__(256) weff_grad_kernel(WeffGradArgs a) {
  const int blk = blockIdx.x, tid = threadIdx.x;
  if (blk < 800) {
    if (a.bf16_in) {
      // bf16 dWeff1: a thread takes EIGHT consecutive output channels of one (i, k, l): 16-byte loads, up to 64 in flight
      // (blocks [0, 100) do the work of all 800; the others return).  Same terms in the same order as the scalar form.
      if (blk >= 100) return;
      const int e = blk * 256 + tid;                // < 128 * 25 * 8
      const int o8 = e & 7, kl = (e >> 3) % 25, i = (e >> 3) / 25, k = kl / 5, l = kl % 5;
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll 2
      for (int y = 0; y < 8; ++y) {             // 16 loads of 16 bytes in flight per thread
        const int u = y + k - 2;
        const bool oky = u >= 0 && u < 8;
        const int s = oky ? (u >> 2) : 0;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const int v = x + l - 2;
          const bool ok = oky && v >= 0 && v < 8;
          const int t = ok ? (v >> 2) : 0;
          const long long idx = ((long long)(s * 2 + t) * 128 + i) * 4096 + y * 512 + x * 64 + 8 * o8;
          union { uint4 u4; __bf16 h[8]; } val;
          val.u4 = *reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.dweff1) + idx);
          if (!ok) val.u4 = zero;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += (float)val.h[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) a.dW1[((8 * o8 + j) * 128 + i) * 25 + kl] = acc[j];
      return;
    }
    const int e = blk * 256 + tid;                  // < 128 * 25 * 64
    const int o = e & 63, kl = (e >> 6) % 25, i = (e >> 6) / 25, k = kl / 5, l = kl % 5;
    // fixed-trip, branch-free: all (up to 64) loads of a thread are in flight together
    float acc = 0.f;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const int u = y + k - 2;
      const bool oky = u >= 0 && u < 8;
      const int s = oky ? (u >> 2) : 0;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int v = x + l - 2;
        const bool ok = oky && v >= 0 && v < 8;
        const int t = ok ? (v >> 2) : 0;
        const long long idx = ((long long)(s * 2 + t) * 128 + i) * 4096 + y * 512 + x * 64 + o;
        const float val = reinterpret_cast<const float*>(a.dweff1)[idx];
        acc += ok ? val : 0.f;
      }
    }
    a.dW1[(o * 128 + i) * 25 + kl] = acc;
  } else if (blk < 864) {
    grad_stage<64>(a.dweff2, a.dW2, blk - 800, tid);
  } else if (blk < 928) {
    grad_stage<16>(a.dweff3, a.dW3, blk - 864, tid);
  } else {   // stage-1 bias gradient: fixed-order sum of the per-workgroup partials (4 row groups x 16 accumulators:
    //          this single block is the kernel's critical path, so its loads go out sixteen at a time)
    __shared__ float red[4][64];
    const int ch = tid & 63, part = tid >> 6;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    int w = part;
    for (; w + 60 < a.n_partial; w += 64) {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] += a.db1_partial[(w + 4 * k) * 64 + ch];
    }
    for (; w < a.n_partial; w += 4) acc[0] += a.db1_partial[w * 64 + ch];
#pragma unroll
    for (int st = 8; st > 0; st >>= 1)
#pragma unroll
      for (int k = 0; k < st; ++k) acc[k] += acc[k + st];
    red[part][ch] = acc[0];
    __syncthreads();
    if (tid < 64) a.db1[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

}  // namespace

extern "C" int rcb_upconv_weff_build(const float* W1, const float* b1, const float* W2, const float* W3, void* weff1,
                                     void* b1rep, int32_t bf16_out, float* weff2, float* weff3, void* frag_pack,
                                     rcb_stream_t stream) {
  RCB_REQUIRE(W1 && b1 && W2 && W3 && weff1 && b1rep && weff2 && weff3, RCB_ERR_ARG, "upconv_weff_build: null pointer");
  WeffArgs a{W1, b1, W2, W3, weff1, b1rep, weff2, weff3, bf16_out ? 1 : 0, reinterpret_cast<uint4*>(frag_pack)};
  weff_build_kernel<<<frag_pack ? 641 + RCB_UPCONV_PACK_UINT4 / 256 : 641, 256, 0, (hipStream_t)stream>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_upconv_weff_grad(const void* dweff1, int32_t bf16_in, const float* dweff2, const float* dweff3,
                                    float* dW1, float* dW2, float* dW3, const float* db1_partial, int32_t n_partial,
                                    float* db1, rcb_stream_t stream) {
  RCB_REQUIRE(dweff1 && dweff2 && dweff3 && dW1 && dW2 && dW3, RCB_ERR_ARG, "upconv_weff_grad: null pointer");
  RCB_REQUIRE((db1_partial == nullptr) == (db1 == nullptr) && n_partial >= 0, RCB_ERR_ARG,
              "upconv_weff_grad: db1 and its partials go together");
  WeffGradArgs a{dweff1, dweff2, dweff3, dW1, dW2, dW3, bf16_in ? 1 : 0, db1_partial, n_partial, db1};
  weff_grad_kernel<<<db1 ? 929 : 928, 256, 0, (hipStream_t)stream>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
