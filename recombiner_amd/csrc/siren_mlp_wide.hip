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
// Weight gradients: OB x IB accumulator tiles per layer (224 accumulator registers at W = 64, three hidden layers), so the
// kernel runs one wave per SIMD with the unified 512-register file; the cross-wave reduction goes through LDS one layer
// at a time (deterministic, no atomics).
#include "siren_op16.h"

using namespace rcb;

namespace {
using namespace rcb::op16;

__device__ __forceinline__ constexpr int featk(int ks, int h, int j) { return 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3); }

template <int NH, int F, int E, int C, int W>
struct GeoW {
  static_assert(W % 16 == 0 && W > 32 && W <= 64, "hidden width 48 or 64");
  static_assert(C <= 16, "the output gradient is one k-step");
  static constexpr int NL = NH + 1;
  static constexpr int IN0 = F + E;
  static constexpr int K0S = (cmax(F, E) + 7) / 8;
  static constexpr int NB0 = (IN0 + 31) / 32;
  static constexpr int HB = (W + 31) / 32;
  static constexpr int KSH = W / 16;
  __host__ __device__ static constexpr int lin(int l) { return l == 0 ? IN0 : W; }
  __host__ __device__ static constexpr int lout(int l) { return l == NL - 1 ? C : W; }
  __host__ __device__ static constexpr int ib(int l) { return l == 0 ? NB0 : HB; }     // 32-feature input blocks
  __host__ __device__ static constexpr int ob(int l) { return l == NL - 1 ? 1 : HB; }  // 32-feature output blocks
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
  // weight-gradient accumulator tiles: layer l owns ob(l) x ib(l) tiles starting at gbase(l)
  __host__ __device__ static constexpr int gbase(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += ob(i) * ib(i);
    return o;
  }
  static constexpr int NT = gbase(NL);
  // fragment slots (1 KB each)
  static constexpr int FA0 = 0;                                  // + mb * K0S + s
  static constexpr int FAH = FA0 + HB * K0S;                     // + ((l - 1) * HB + mb) * KSH + ks
  static constexpr int FAO = FAH + (NH - 1) * HB * KSH;          // + ks
  static constexpr int FBO = FAO + KSH;                          // + ib
  static constexpr int FBH = FBO + HB;                           // + (((NH - 1) - l) * HB + ib) * KSH + ks
  static constexpr int FBX = FBH + (NH - 1) * HB * KSH;          // + ks
  static constexpr int NFR = FBX + KSH;
  static constexpr int NSINE = FAO;                              // slots below carry w0 / 2 pi
  // LDS map (bytes)
  static constexpr int FR_OFF = ((DNET * 4 + 15) / 16) * 16;
  static constexpr int TILE_OFF = FR_OFF + NFR * 1024;
  static constexpr int TSA = 32 * HB;                            // dZ image row stride (elements)
  static constexpr int TSBB = 32 * cmax(HB, NB0);                // input image row stride
  static constexpr int WAVE_TILE = 32 * (TSA + TSBB) * 2;
  static constexpr int LDS_MAIN = TILE_OFF + 4 * WAVE_TILE;
  // cross-wave reduction scratch of ONE layer: [out][in] rows of stride 32 ib + 1, then 32 ob bias slots
  __host__ __device__ static constexpr int rsize(int l) { return lout(l) * (32 * ib(l) + 1) + 32 * ob(l); }
  __host__ __device__ static constexpr int rmax() {
    int w = 0;
    for (int l = 0; l < NL; ++l) w = rsize(l) > w ? rsize(l) : w;
    return w;
  }
  static constexpr int RED_WAVE = rmax();
  static constexpr int LDS_BYTES = cmax(LDS_MAIN, 4 * RED_WAVE * 4);
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <typename T, int NH, int F, int E, int C, int W, int MODE, bool IN16>
__global__ void __launch_bounds__(256, 1) siren_wide_kernel(SirenArgs a) {
  using G = GeoW<NH, F, E, C, W>;
  using bf16x8 = typename Op16<T>::v8;
  using bf16x4 = typename Op16<T>::v4;
  using bf16x2 = typename Op16<T>::v2;
  constexpr float GS = Op16<T>::GRAD_SCALE;
  constexpr float WS = Op16<T>::W_SCALE;
  constexpr int NL = G::NL, IN0 = G::IN0, K0S = G::K0S, NB0 = G::NB0, HB = G::HB, KSH = G::KSH, DNET = G::DNET;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = lane & 31, h = lane >> 5;
  const int g = blockIdx.x % a.G, chunk = blockIdx.x / a.G;     // chunk-major: partial buffers are [chunk][g]
  const int n = g / a.S;
  const int P = a.P;

  float* wl = smem;
  uint4* frags = reinterpret_cast<uint4*>(smem_raw + G::FR_OFF);
  T* bufA = reinterpret_cast<T*>(smem_raw + G::TILE_OFF + wave * G::WAVE_TILE);
  T* bufB = bufA + 32 * G::TSA;

  // ---- stage weights, build the MFMA A-fragments, clear the image buffers ---------------------------------------------
  {
    const float* src = a.wvec + (long long)g * a.w_stride;
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
    for (int i = lane; i < 32 * (G::TSA + G::TSBB); i += 64) bufA[i] = (T)0.f;
    __syncthreads();
    if (tid < W) {
#pragma unroll
      for (int l = 0; l < NH; ++l) wl[G::off(l) + tid] *= a.k_hi;    // sine layers work in revolutions
    }
    const int fq = lane & 31, fh = lane >> 5;
    auto put = [&](int slot, auto&& elem) {
      union { bf16x8 v; uint4 u; } fr;
      // sine-layer forward fragments carry w0 / 2 pi, the fragments producing a hidden layer's data gradient carry w0
      const float sc = (slot < G::NSINE) ? WS * a.k_hi : (slot >= G::FBO && slot < G::FBX) ? a.w0 : WS;
#pragma unroll
      for (int j = 0; j < 8; ++j) fr.v[j] = (T)(elem(j) * sc);
      frags[slot * 64 + lane] = fr.u;
    };
    // forward, layer 0: half-wave 0 contracts the F Fourier features, half-wave 1 the E upsampled features
#pragma unroll
    for (int mb = 0; mb < HB; ++mb)
#pragma unroll
      for (int s = 0; s < K0S; ++s) {
        const int slot = G::FA0 + mb * K0S + s;
        if ((slot & 3) != wave) continue;
        put(slot, [&](int j) {
          const int kk = 8 * s + j, out = 32 * mb + fq;
          const int row = (fh == 0) ? (kk < F ? kk : -1) : (kk < E ? F + kk : -1);
          return (row >= 0 && out < W) ? wl[G::off(0) + W + row * W + out] : 0.f;
        });
      }
    // forward, hidden layers 1 .. NH-1
#pragma unroll
    for (int l = 1; l < NH; ++l)
#pragma unroll
      for (int mb = 0; mb < HB; ++mb)
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
          const int slot = G::FAH + ((l - 1) * HB + mb) * KSH + ks;
          if ((slot & 3) != wave) continue;
          put(slot, [&](int j) {
            const int out = 32 * mb + fq;
            return out < W ? wl[G::off(l) + W + featk(ks, fh, j) * W + out] : 0.f;
          });
        }
    // forward, output layer
#pragma unroll
    for (int ks = 0; ks < KSH; ++ks) {
      const int slot = G::FAO + ks;
      if ((slot & 3) != wave) continue;
      put(slot, [&](int j) { return fq < C ? wl[G::off(NH) + C + featk(ks, fh, j) * C + fq] : 0.f; });
    }
    // data gradient through the output layer (k = output channel, one step)
#pragma unroll
    for (int ib = 0; ib < HB; ++ib) {
      const int slot = G::FBO + ib;
      if ((slot & 3) != wave) continue;
      put(slot, [&](int j) {
        const int m = 32 * ib + fq, k = fk(0, fh, j);
        return (m < W && k < C) ? wl[G::off(NH) + C + m * C + k] : 0.f;
      });
    }
    // data gradient through hidden layers NH-1 .. 1
#pragma unroll
    for (int l = NH - 1; l >= 1; --l)
#pragma unroll
      for (int ib = 0; ib < HB; ++ib)
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
          const int slot = G::FBH + (((NH - 1) - l) * HB + ib) * KSH + ks;
          if ((slot & 3) != wave) continue;
          put(slot, [&](int j) {
            const int m = 32 * ib + fq;
            return m < W ? wl[G::off(l) + W + m * W + featk(ks, fh, j)] : 0.f;
          });
        }
    // data gradient through layer 0 onto the E upsampled features
#pragma unroll
    for (int ks = 0; ks < KSH; ++ks) {
      const int slot = G::FBX + ks;
      if ((slot & 3) != wave) continue;
      put(slot, [&](int j) { return fq < E ? wl[G::off(0) + W + (F + fq) * W + featk(ks, fh, j)] : 0.f; });
    }
    __syncthreads();
  }
  auto FA = [&](int slot) -> bf16x8 {
    union { bf16x8 v; uint4 u; } fr;
    fr.u = frags[slot * 64 + lane];
    return fr.v;
  };

  f32x16 gW[G::NT];
  float gb[NL][HB];
  float sse_local = 0.f;
  if (MODE != MODE_FWD) {
#pragma unroll
    for (int i = 0; i < G::NT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) gW[i][r] = 0.f;
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int b = 0; b < HB; ++b) gb[l][b] = 0.f;
  }
  constexpr int KH0 = F, KH1 = E;

  const int ntiles = (P + 31) >> 5;
  constexpr bool VEC4 = (F % 8 == 0) && (E % 8 == 0);
  // IN16 (as in the width-32 kernel): both input halves arrive as bf16 rows and the loaded bits are the layer-0 B operand
  float4 raw[IN16 ? 1 : 2 * K0S];
  uint4 raw16[IN16 ? K0S : 1];
  auto fetch = [&](int tile) {
    const int pp = tile * 32 + q;
    const int pcl = pp < P ? pp : P - 1;
    if constexpr (IN16) {
      const __bf16* s16 = (h == 0) ? (reinterpret_cast<const __bf16*>(a.xf16) + (long long)n * a.xf_stride + (long long)pcl * F)
                                   : (reinterpret_cast<const __bf16*>(a.pe) + ((long long)g * P + pcl) * E);
      const int kh16 = (h == 0) ? KH0 : KH1;
#pragma unroll
      for (int s = 0; s < K0S; ++s) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (8 * s + 8 <= kh16) u = reinterpret_cast<const uint4*>(s16)[s];
        raw16[s] = u;
      }
    } else {
    const float* src = (h == 0) ? (a.xf + (long long)n * a.xf_stride + (long long)pcl * F)
                                : (a.pe + ((long long)g * P + pcl) * E);
    const int kh = (h == 0) ? KH0 : KH1;
    if (E % 8 == 0 && a.pe_bf16 && h == 1) {   // bf16-stored pe: 16 B per 8 features, widened exactly
      const uint4* s16 = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(a.pe) + ((long long)g * P + pcl) * E);
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
  // this workgroup's share of the 32-pixel tiles (all of them unless rcb_siren_desc.pixel_chunks > 1)
  const int t0 = (int)((long long)chunk * ntiles / a.chunks), t1 = (int)((long long)(chunk + 1) * ntiles / a.chunks);
  if (t0 + wave < t1) fetch(t0 + wave);
  for (int t = t0 + wave; t < t1; t += 4) {
    const int p = t * 32 + q;
    const bool valid = p < P;
    const int pc = valid ? p : P - 1;
    bf16x8 xin[K0S];
#pragma unroll
    for (int s = 0; s < K0S; ++s) {
      if constexpr (IN16) {
        union { uint4 u; bf16x8 v; } cv;
        cv.u = raw16[s];
        xin[s] = cv.v;
      } else {
        const float4 v0 = raw[2 * s], v1 = raw[2 * s + 1];
        xin[s][0] = (T)v0.x; xin[s][1] = (T)v0.y; xin[s][2] = (T)v0.z; xin[s][3] = (T)v0.w;
        xin[s][4] = (T)v1.x; xin[s][5] = (T)v1.y; xin[s][6] = (T)v1.z; xin[s][7] = (T)v1.w;
      }
    }
    float yv[16];
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
    fetch(t + 4 < t1 ? t + 4 : t);
    // ---- forward ----------------------------------------------------------------------------------------------------
    bf16x8 S[NH][KSH], Cs[NH][KSH];
#pragma unroll
    for (int l = 0; l < NH; ++l) {
      const float* Bl = wl + G::off(l);
#pragma unroll
      for (int mb = 0; mb < HB; ++mb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = (32 * mb + 16 * (r >> 3) < W) ? Bl[32 * mb + rho(r, h)] * WS : 0.f;
        if (l == 0) {
#pragma unroll
          for (int s = 0; s < K0S; ++s) acc = Op16<T>::mfma(FA(G::FA0 + mb * K0S + s), xin[s], acc);
        } else {
#pragma unroll
          for (int ks = 0; ks < KSH; ++ks) acc = Op16<T>::mfma(FA(G::FAH + ((l - 1) * HB + mb) * KSH + ks), S[l - 1][ks], acc);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (2 * mb + s >= KSH) continue;               // rows beyond the layer width: no transcendental spent
          bf16x8 sp, cp;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float tt = (WS == 1.0f) ? acc[8 * s + j] : acc[8 * s + j] * (1.0f / WS);
            sp[j] = (T)__builtin_amdgcn_sinf(tt);
            if (MODE != MODE_FWD) cp[j] = (T)__builtin_amdgcn_cosf(tt);
          }
          S[l][2 * mb + s] = sp;
          if (MODE != MODE_FWD) Cs[l][2 * mb + s] = cp;
        }
      }
    }
    f32x16 acc;
    {
      const float* Bl = wl + G::off(NL - 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = (rho(r, 0) < C || rho(r, 1) < C) ? ((rho(r, h) < C) ? Bl[rho(r, h) < C ? rho(r, h) : 0] * WS : 0.f) : 0.f;
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks) acc = Op16<T>::mfma(FA(G::FAO + ks), S[NH - 1][ks], acc);
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
      continue;
    }
    // ---- output gradient --------------------------------------------------------------------------------------------
    f32x16 dz[HB];
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
      dz[0][r] = v;
    }
    // ---- backward ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int l = NL - 1; l >= 0; --l) {
      const int OBl = (l == NL - 1) ? 1 : HB;
      const int KSO = (l == NL - 1) ? 2 : KSH;            // k-steps over this layer's output features
      const int IBl = (l == 0) ? NB0 : HB;
      bf16x8 dzb[2 * HB];
#pragma unroll
      for (int ks = 0; ks < 2 * HB; ++ks)
        if (ks < KSO) dzb[ks] = pack8<T>(dz[ks >> 1], ks & 1);
      // (1) data gradient FIRST (the serial chain of the backward pass; the weight gradient below fills its MFMA latency --
      // at one wave per SIMD nothing else would)
      if (l > 0) {
#pragma unroll
        for (int ib = 0; ib < HB; ++ib) {
          f32x16 dh;
#pragma unroll
          for (int r = 0; r < 16; ++r) dh[r] = 0.f;
          if (l == NL - 1) {
            dh = Op16<T>::mfma(FA(G::FBO + ib), dzb[0], dh);
          } else {
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) dh = Op16<T>::mfma(FA(G::FBH + (((NH - 1) - l) * HB + ib) * KSH + ks), dzb[ks], dh);
          }
          // dz of layer l-1, tile ib (written after every use of the old dz tiles: dzb holds their packed copies)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dz[ib][r] = (2 * ib + (r >> 3) < KSH) ? dh[r] * (float)Cs[l - 1][2 * ib + (r >> 3)][r & 7] : 0.f;
        }
      } else if (a.dpe != nullptr) {
        f32x16 dx;
#pragma unroll
        for (int r = 0; r < 16; ++r) dx[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) dx = Op16<T>::mfma(FA(G::FBX + ks), dzb[ks], dx);
        if (valid) {
          float* dst = a.dpe + ((long long)g * P + p) * E;
          if (E % 8 == 0 && a.pe_bf16) {
            __bf16* d16 = reinterpret_cast<__bf16*>(a.dpe) + ((long long)g * P + p) * E;
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
      // (2) weight gradient: [pixel][feature] images of dZ and of the layer input -> transposed reads
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      {
        union { bf16x8 v; bf16x4 hlf[2]; } u;
#pragma unroll
        for (int ks = 0; ks < 2 * HB; ++ks) {
          if (ks >= KSO) continue;
          u.v = dzb[ks];
          *reinterpret_cast<bf16x4*>(bufA + swz(q, 16 * ks + 4 * h, G::TSA)) = u.hlf[0];
          *reinterpret_cast<bf16x4*>(bufA + swz(q, 16 * ks + 8 + 4 * h, G::TSA)) = u.hlf[1];
        }
        if (l > 0) {
#pragma unroll
          for (int ks = 0; ks < KSH; ++ks) {
            u.v = S[l - 1][ks];
            *reinterpret_cast<bf16x4*>(bufB + swz(q, 16 * ks + 4 * h, G::TSBB)) = u.hlf[0];
            *reinterpret_cast<bf16x4*>(bufB + swz(q, 16 * ks + 8 + 4 * h, G::TSBB)) = u.hlf[1];
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
          if (32 * NB0 > IN0) {
            // the hidden layers' input images overwrote the padding columns [IN0, 32 NB0): finite values that only reach
            // accumulator columns which are never stored
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      {
        bf16x8 av[HB][2];
#pragma unroll
        for (int ob = 0; ob < HB; ++ob) {
          if (ob >= OBl) continue;
#pragma unroll
          for (int s = 0; s < 2; ++s) av[ob][s] = read_tr<T>(bufA, G::TSA, s, lane, 32 * ob);
          gb[l][ob] = sum8_16<T>(av[ob][0], gb[l][ob]);
          gb[l][ob] = sum8_16<T>(av[ob][1], gb[l][ob]);
        }
#pragma unroll
        for (int ib = 0; ib < cmax(HB, NB0); ++ib) {
          if (ib >= IBl) continue;
          bf16x8 bv[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) bv[s] = read_tr<T>(bufB, G::TSBB, s, lane, 32 * ib);
#pragma unroll
          for (int ob = 0; ob < HB; ++ob) {
            if (ob >= OBl) continue;
            const int gi = G::gbase(l) + ob * IBl + ib;
#pragma unroll
            for (int s = 0; s < 2; ++s) gW[gi] = Op16<T>::mfma(av[ob][s], bv[s], gW[gi]);
          }
        }
      }
    }
  }
  if (MODE == MODE_FWD) return;

  // ---- deterministic cross-wave reduction of the weight gradients, one layer at a time -----------------------------------
  float* part = smem + wave * G::RED_WAVE;
  float* dst = a.dwvec + ((long long)chunk * a.G + g) * a.w_stride;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int no = G::lout(l), IBl = G::ib(l), OBl = G::ob(l);
    const int rs = 32 * IBl + 1;
    __syncthreads();                                   // the tile loop / the previous layer's sums are done with LDS
#pragma unroll
    for (int ob = 0; ob < HB; ++ob) {
      if (ob >= OBl) continue;
      const float bt = gb[l][ob] + __shfl_xor(gb[l][ob], 32, 64);
      if (h == 0) part[no * rs + 32 * ob + q] = bt;
#pragma unroll
      for (int ib = 0; ib < cmax(HB, NB0); ++ib) {
        if (ib >= IBl) continue;
        const int gi = G::gbase(l) + ob * IBl + ib;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (32 * ob + rho(r, 0) < no || 32 * ob + rho(r, 1) < no) {
            const int o = 32 * ob + rho(r, h);
            if (o < no) part[o * rs + 32 * ib + q] = gW[gi][r];
          }
        }
      }
    }
    __syncthreads();
    const int ol = G::off(l);
    const int size = no * (G::lin(l) + 1);
    for (int e = tid; e < size; e += 256) {
      int src;
      if (e < no) {
        src = no * rs + e;                 // bias
      } else {
        const int i = (e - no) / no, o = (e - no) - i * no;
        src = o * rs + i;
      }
      const float v = (((smem[src] + smem[G::RED_WAVE + src]) + smem[2 * G::RED_WAVE + src]) + smem[3 * G::RED_WAVE + src]) * (1.0f / GS);
      dst[ol + e] = v;
      if (size == G::WMAX && (G::WMAX & 1) == 0 && a.dw_split != nullptr) {
        const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
        __bf16* o = reinterpret_cast<__bf16*>(a.dw_split) + ((long long)G::wide_index(l) * a.G + g) * (3 * G::WMAX) + e;
        o[0] = hi;
        o[G::WMAX] = lo;
        o[2 * G::WMAX] = hi;
      }
    }
  }
  if (MODE == MODE_LOSS) {
    float v = wave_sum(sse_local);
    __syncthreads();
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    if (tid == 0) a.sse[(long long)chunk * a.G + g] = ((smem[0] + smem[1]) + smem[2]) + smem[3];
  }
}

template <typename T, int NH, int F, int E, int C, int W, int MODE, bool IN16>
int launch_one(const SirenArgs& a, hipStream_t st) {
  using G = GeoW<NH, F, E, C, W>;
  static bool attr_done = false;
  auto kfn = siren_wide_kernel<T, NH, F, E, C, W, MODE, IN16>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return fail((int)e, "siren(wide): hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  kfn<<<a.G * a.chunks, 256, G::LDS_BYTES, st>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

template <typename T, int NH, int F, int E, int C, int W>
int launch_mode(int mode, const SirenArgs& a, hipStream_t st) {
  constexpr bool can16 = Op16<T>::IS_BF16 && (E % 8 == 0) && (F % 8 == 0) && E > 0;
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
  RCB_CASE(3, 16, 16, 3, 48)   // kodak / cifar / protein geometry at width 48
  RCB_CASE(3, 16, 16, 3, 64)
  RCB_CASE(3, 18, 16, 3, 64)   // video geometry at width 64
  RCB_CASE(3, 16, 16, 1, 64)   // audio geometry at width 64
#undef RCB_CASE
  return fail(RCB_ERR_UNSUPPORTED, "siren(wide): geometry n_hidden=%d F=%d E=%d C=%d width=%d is not instantiated",
              d->n_hidden, d->fourier_dim, d->pe_dim, d->out_dim, d->hidden);
}
}  // namespace rcb
