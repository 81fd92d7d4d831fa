// Stage 1 of the 1-D upsampling net, direct: `nearest-upsample(4) -> Conv1d(128 -> 64, k 5, pad 2) -> LeakyReLU`
// (prior_model.py:23-51 with Conv1d: up1 / conv1 / act1; audio and protein presets) on the latent grid x [B][g][128].
//
// In phase form output pixel 4 i + a reads the source pixels i + t - 1, t in {0, 1, 2}, with the pre-summed weights
//     Wbig[t * 128 + ci][a * 64 + co]            (rcb_phase_bigweight; the operand of the window-GEMM form of the stage)
// and only TWO taps per phase are non-zero: a in {0, 1}: t in {0, 1};  a in {2, 3}: t in {1, 2}.  The window-GEMM form
// (rcb_window_gather -> library GEMM -> LeakyReLU pass; backward: two more GEMMs, rcb_window_fold, casts) moves the
// 3-pixel window matrix through HBM five times per step: at a rank's shard of the audio preset (1024 clips, g = 3000:
// 3.07 M latent pixels) that was 6 ms of a 30 ms step.  The three kernels here read the latent grid and the stage's
// output gradient once each:
//   forward : x1[b, 4 i + a, co] = LeakyReLU(bf16(bias[co] + sum_{t, ci} bf16(x[b, i + t - 1, ci]) Wbig[t][ci][a][co]))
//   dgrad   : dx[b, i, ci]       = sum_{u = -2 .. 5} sum_co dz[b, 4 i + u, co] Wbig[t(u)][ci][a(u)][co]   (fp32)
//             with a(u) = u mod 4, t(u) = 1 - floor(u / 4)
//   wgrad   : dWbig[t][ci][a][co] = sum_{b, i} bf16(x[b, i + t - 1, ci]) dz[b, 4 i + a, co],  dbias[co] = sum dz   (fp32)
// dz is the gradient of the PRE-activation (rcb_phaseconv_dgrad of stage 2 applies LeakyReLU' from the sign of x1).
// All three are implicit GEMMs on v_mfma_f32_32x32x16_bf16 with the position on the lane, operands staged through LDS
// images (x converted to bf16 on the way in: the rounding of the window path's cast pass), several tiles per barrier,
// the next pass's loads in flight during the MFMAs.
#include "rcb_common.h"

using namespace rcb;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int CIN = 128, COUT = 64, NPH = 4, WLD = NPH * COUT;      // Wbig [3 * CIN][WLD]
constexpr float SLOPE = 0.01f;

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// row of accumulator register r for lane half h (32x32 MFMA C/D layout)
__device__ __forceinline__ constexpr int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

union Frag {
  bf16x8 v;
  uint4 u;
};

__device__ __forceinline__ uint4 to_bf16x8(float4 lo, float4 hi) {
  Frag f;
  f.v[0] = (__bf16)lo.x; f.v[1] = (__bf16)lo.y; f.v[2] = (__bf16)lo.z; f.v[3] = (__bf16)lo.w;
  f.v[4] = (__bf16)hi.x; f.v[5] = (__bf16)hi.y; f.v[6] = (__bf16)hi.z; f.v[7] = (__bf16)hi.w;
  return f.u;
}

struct S1Args {
  const float* x;        // [B][g][128] fp32 latent grid
  const __bf16* wbig;    // [384][256]
  const float* bias;     // [64]
  __bf16* y;             // fwd: x1 [B][4 g][64]
  const __bf16* dz;      // dgrad / wgrad: [B][4 g][64]
  float* dx;             // dgrad: [B][g][128]
  float* partial;        // wgrad: [gridDim.x][384][256] slabs, then [gridDim.x][64] bias partials
  float* bias_part;
  int B, g, tiles_per_row;
};

// ---- forward ---------------------------------------------------------------------------------------------------------------
// One workgroup = 8 waves = (phase a, 32-channel output block); a pass = TB consecutive 32-pixel tiles of one row, whose
// (32 TB + 2) source pixels are staged once (fp32 -> bf16, 256-byte pixels, the 16-byte chunk index XORed with pixel & 15:
// conflict-free for the 16-lane groups of ds_read_b128, tools/lds_banks.py).  A wave keeps its 2 taps x 8 k-steps of weight
// fragments in registers for the whole kernel: 16 MFMAs per tile.
template <int TB>
__global__ void __launch_bounds__(512) s1_fwd1d_kernel(S1Args p) {
  constexpr int PW = 32 * TB + 2, NCH = PW * 16, NIT = (NCH + 511) / 512;
  extern __shared__ uint4 s1_smem[];
  uint4* img = s1_smem;                         // [PW][16]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane & 31, h = lane >> 5;
  const int a = wave >> 1, mb = wave & 1, t0 = a < 2 ? 0 : 1;
  bf16x8 wf[2][8];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        wf[tt][c][j] = p.wbig[(long long)((t0 + tt) * CIN + 16 * c + 8 * h + j) * WLD + a * COUT + 32 * mb + q];
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = p.bias[32 * mb + rho(r, h)];
  const int g = p.g, gpr = (p.tiles_per_row + TB - 1) / TB, n_groups = p.B * gpr;
  float4 stg[NIT][2];
  auto fetch = [&](int grp) {
    const int b = grp / gpr, l0 = 32 * TB * (grp - b * gpr);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = threadIdx.x + 512 * it;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (e < NCH) {
        const int sl = l0 - 1 + (e >> 4);
        if (sl >= 0 && sl < g) {
          const float4* src = reinterpret_cast<const float4*>(p.x + ((long long)b * g + sl) * CIN + 8 * (e & 15));
          v0 = src[0];
          v1 = src[1];
        }
      }
      stg[it][0] = v0;
      stg[it][1] = v1;
    }
  };
  if ((int)blockIdx.x < n_groups) fetch(blockIdx.x);
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int b = grp / gpr, tg = grp - b * gpr;
    __syncthreads();                               // every wave is done reading the previous pass's image
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = threadIdx.x + 512 * it;
      if (e < NCH) {
        const int pp = e >> 4;
        img[pp * 16 + ((e & 15) ^ (pp & 15))] = to_bf16x8(stg[it][0], stg[it][1]);
      }
    }
    __syncthreads();
    if (grp + (int)gridDim.x < n_groups) fetch(grp + gridDim.x);
#pragma unroll
    for (int ts = 0; ts < TB; ++ts) {
      const int tile = tg * TB + ts;
      if (tile >= p.tiles_per_row) break;          // (uniform)
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bv[r];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int pp = 32 * ts + q + t0 + tt;      // image pixel 0 = source pixel l0 - 1
        const uint4* src = img + pp * 16;
        const int sw = pp & 15;
        Frag f[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c].u = src[(2 * c + h) ^ sw];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc = mfma16(wf[tt][c], f[c].v, acc);
      }
      // the pre-activation rounded to bf16 (what the window path stored), LeakyReLU, and the lane halves swapped so that lane
      // (q, h) owns the 16 consecutive channels 32 mb + 16 h .. + 15 of its output pixel: two 16-byte stores
      const int il = 32 * tile + q;
      uint4* dst = reinterpret_cast<uint4*>(p.y + (((long long)b * g + il) * NPH + a) * COUT + 32 * mb + 16 * h);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        Frag ob;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v0 = (float)(__bf16)acc[4 * hf + k], v1 = (float)(__bf16)acc[8 + 4 * hf + k];
          v0 = v0 > 0.f ? v0 : v0 * SLOPE;
          v1 = v1 > 0.f ? v1 : v1 * SLOPE;
          auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
          ob.v[k] = (__bf16)__uint_as_float(sw2[0]);
          ob.v[4 + k] = (__bf16)__uint_as_float(sw2[1]);
        }
        if (il < g) dst[hf] = ob.u;
      }
    }
  }
}

// ---- data gradient -----------------------------------------------------------------------------------------------------------
// dx[i] gathers the eight output pixels 4 i - 2 .. 4 i + 5: K = 8 x 64 = 32 k-steps.  One workgroup = 8 waves = (32-channel
// input block, tile of the pair); a wave keeps its 32 weight fragments (16-byte row pieces of Wbig) in registers.  The dz
// segment of a pass (128 TB + 8 pixels x 128 B) is staged with adjacent pixels swapped where bit 2 of the pixel index is
// set and the chunk index XORed with (pixel >> 3) & 7: the lanes' stride-4 gather then covers all 64 banks once per
// 16-lane group (tools/lds_banks.py model).
__device__ __forceinline__ int dz_slot(int P, int chunk) { return (P ^ ((P >> 2) & 1)) * 8 + (chunk ^ ((P >> 3) & 7)); }

template <int TB>
__global__ void __launch_bounds__(512) s1_dgrad1d_kernel(S1Args p) {
  static_assert(TB == 2, "eight waves = four input blocks x two tiles");
  constexpr int PWD = 128 * TB + 8, NCH = PWD * 8, NIT = (NCH + 511) / 512;
  extern __shared__ uint4 s1_smem[];
  uint4* img = s1_smem;                         // [PWD][8]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane & 31, h = lane >> 5;
  const int mb = wave >> 1, ts = wave & 1;
  bf16x8 wf[8][4];                               // [u + 2][16-channel block of co]
#pragma unroll
  for (int uu = 0; uu < 8; ++uu) {
    const int u = uu - 2, a = u & 3, t = 1 - ((u - a) >> 2);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      Frag f;
      f.u = *reinterpret_cast<const uint4*>(p.wbig + (long long)(t * CIN + 32 * mb + q) * WLD + a * COUT + 16 * c + 8 * h);
      wf[uu][c] = f.v;
    }
  }
  const int g = p.g, gpr = (p.tiles_per_row + TB - 1) / TB, n_groups = p.B * gpr;
  uint4 stg[NIT];
  auto fetch = [&](int grp) {
    const int b = grp / gpr, l0 = 32 * TB * (grp - b * gpr);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = threadIdx.x + 512 * it;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < NCH) {
        const int sl = 4 * l0 - 2 + (e >> 3);
        if (sl >= 0 && sl < 4 * g) v = reinterpret_cast<const uint4*>(p.dz + ((long long)b * 4 * g + sl) * COUT)[e & 7];
      }
      stg[it] = v;
    }
  };
  if ((int)blockIdx.x < n_groups) fetch(blockIdx.x);
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int b = grp / gpr, tg = grp - b * gpr;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = threadIdx.x + 512 * it;
      if (e < NCH) img[dz_slot(e >> 3, e & 7)] = stg[it];
    }
    __syncthreads();
    if (grp + (int)gridDim.x < n_groups) fetch(grp + gridDim.x);
    const int tile = tg * TB + ts;
    if (tile >= p.tiles_per_row) continue;         // (uniform; no barrier below)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      const int P = 4 * (32 * ts + q) + uu;        // image pixel 0 = dz pixel 4 l0 - 2
      Frag f[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) f[c].u = img[dz_slot(P, 2 * c + h)];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc = mfma16(wf[uu][c], f[c].v, acc);
    }
    // lane halves swap: lane (q, h) owns the 16 consecutive channels 32 mb + 16 h .. + 15 of its pixel: 64 contiguous bytes
    const int il = 32 * tile + q;
    float4* dst = reinterpret_cast<float4*>(p.dx + ((long long)b * g + il) * CIN + 32 * mb + 16 * h);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float o[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[4 * hf + k]), __float_as_uint(acc[8 + 4 * hf + k]), false, false);
        o[k] = __uint_as_float(sw[0]);
        o[4 + k] = __uint_as_float(sw[1]);
      }
      if (il < g) {
        dst[2 * hf] = make_float4(o[0], o[1], o[2], o[3]);
        dst[2 * hf + 1] = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
  }
}

// ---- weight gradient ----------------------------------------------------------------------------------------------------------
// The contraction runs over positions: both operands are read TRANSPOSED from [pixel][32 channels] LDS images (64-byte rows,
// ds_read_b64_tr_b16: see pc_wgrad_kernel).  One workgroup owns all of dWbig: wave = (phase a, tap of its pair), its
// accumulators the 4 input blocks x 2 output blocks of that (tap, phase) panel (8 tiles of 32 x 32); x and dz are read from
// memory once.  A pass = TB tiles of one row, two image sets; sums stay in registers across the walk and leave as one slab
// per workgroup in the layout of dWbig; s1_wgrad_sum_kernel adds the slabs in a fixed order (no atomics).
__device__ __forceinline__ bf16x8 tr_read_plain(const __bf16* img, int ks, int lane) {
  const int h = lane >> 5, fb = (lane >> 4) & 1, i = lane & 15, q4 = i >> 2, p4 = i & 3;
  union { s16x4 v[2]; bf16x8 b; } u;
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const __bf16* ptr = img + (16 * ks + 8 * h + 4 * w + q4) * 32 + 16 * fb + 4 * p4;
    u.v[w] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
  }
  return u.b;
}

template <int TB>
__global__ void __launch_bounds__(512) s1_wgrad1d_kernel(S1Args p) {
  constexpr int PWX = 34;
  constexpr int XT = 4 * PWX * 4, DT = NPH * 2 * 32 * 4;      // uint4 per tile: x [mb][34][4], dz [phase][nb][32][4]
  constexpr int XPT = PWX * 16, DPT = 128 * 8;                 // chunks fetched per tile (x: 8 floats -> one 16-byte chunk)
  constexpr int NXC = TB * XPT, NDC = TB * DPT;
  constexpr int NITX = (NXC + 511) / 512, NITD = (NDC + 511) / 512;
  extern __shared__ uint4 s1_smem[];
  uint4* ximg = s1_smem;                         // [2][TB][XT]
  uint4* dimg = s1_smem + 2 * TB * XT;           // [2][TB][DT]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5;
  const int a = wave >> 1, tt = wave & 1, t = (a < 2 ? 0 : 1) + tt;
  const int g = p.g, n_st = p.B * ((p.tiles_per_row + TB - 1) / TB), spr = (p.tiles_per_row + TB - 1) / TB;
  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  float dbsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbsum[j] = 0.f;
  float4 sx[NITX][2];
  uint4 sd[NITD];
  auto fetch = [&](int st) {
    const int b = st / spr, l0 = 32 * TB * (st - b * spr);
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int e = threadIdx.x + 512 * it;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (e < NXC) {
        const int tl = e / XPT, r = e - tl * XPT, sl = l0 + 32 * tl - 1 + (r >> 4);
        if (sl >= 0 && sl < g && l0 + 32 * tl < g) {
          const float4* src = reinterpret_cast<const float4*>(p.x + ((long long)b * g + sl) * CIN + 8 * (r & 15));
          v0 = src[0];
          v1 = src[1];
        }
      }
      sx[it][0] = v0;
      sx[it][1] = v1;
    }
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int e = threadIdx.x + 512 * it;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < NDC) {
        const int tl = e / DPT, r = e - tl * DPT, sl = 4 * (l0 + 32 * tl) + (r >> 3);
        if (sl < 4 * g) v = reinterpret_cast<const uint4*>(p.dz + ((long long)b * 4 * g + sl) * COUT)[r & 7];
      }
      sd[it] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int e = threadIdx.x + 512 * it;
      if (e < NXC) {
        const int tl = e / XPT, r = e - tl * XPT, pp = r >> 4, c16 = r & 15;
        ximg[(buf * TB + tl) * XT + ((c16 >> 2) * PWX + pp) * 4 + (c16 & 3)] = to_bf16x8(sx[it][0], sx[it][1]);
      }
    }
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int e = threadIdx.x + 512 * it;
      if (e < NDC) {
        const int tl = e / DPT, r = e - tl * DPT, d = r >> 3, c = r & 7;       // d = dz pixel of the tile: 4 i + a
        dimg[(buf * TB + tl) * DT + (((d & 3) * 2 + (c >> 2)) * 32 + (d >> 2)) * 4 + (c & 3)] = sd[it];
        Frag f;
        f.u = sd[it];
#pragma unroll
        for (int j = 0; j < 8; ++j) dbsum[j] += (float)f.v[j];
      }
    }
  };
  int st = blockIdx.x, buf = 0;
  if (st < n_st) {
    fetch(st);
    stash(0);
  }
  __syncthreads();
  for (; st < n_st; st += gridDim.x, buf ^= 1) {
    const int next = st + gridDim.x;
    if (next < n_st) fetch(next);
#pragma unroll
    for (int tl = 0; tl < TB; ++tl) {
      const __bf16* di = reinterpret_cast<const __bf16*>(dimg + (buf * TB + tl) * DT + a * 2 * 128);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 b0 = tr_read_plain(di, ks, lane), b1 = tr_read_plain(di + 1024, ks, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const __bf16* xi = reinterpret_cast<const __bf16*>(ximg + (buf * TB + tl) * XT + (m * PWX + t) * 4);
          const bf16x8 aop = tr_read_plain(xi, ks, lane);
          acc[m][0] = mfma16(aop, b0, acc[m][0]);
          acc[m][1] = mfma16(aop, b1, acc[m][1]);
        }
      }
    }
    if (next < n_st) stash(buf ^ 1);               // (buffer buf ^ 1 was last read before the previous barrier)
    __syncthreads();
  }
  float* slab = p.partial + (long long)blockIdx.x * (3 * CIN * WLD);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[(long long)(t * CIN + 32 * m + rho(r, h)) * WLD + a * COUT + 32 * n + (lane & 31)] = acc[m][n][r];
  // bias partials: thread e stages the channels 8 (e % 8) .. + 7 in every chunk of dz it handles
  float* red_sm = reinterpret_cast<float*>(s1_smem);
#pragma unroll
  for (int j = 0; j < 8; ++j) red_sm[threadIdx.x * 8 + j] = dbsum[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = threadIdx.x / 8, j = threadIdx.x % 8;
    float sacc = 0.f;
    for (int th = c; th < 512; th += 8) sacc += red_sm[th * 8 + j];            // fixed order: deterministic
    p.bias_part[(long long)blockIdx.x * COUT + 8 * c + j] = sacc;
  }
}

// sum of the slabs (one thread per element of dWbig, sixteen slabs in flight; the panels no phase reads are zero) and of
// the bias partials, in a fixed association
__global__ void __launch_bounds__(256) s1_wgrad_sum_kernel(const float* __restrict__ partial, const float* __restrict__ bias_part,
                                                           int n_slabs, float* __restrict__ dwbig, float* __restrict__ dbias) {
  constexpr int TOTAL = 3 * CIN * WLD;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < TOTAL) {
    const int t = e / (CIN * WLD), a = (e % WLD) / COUT;
    float s = 0.f;
    if (a < 2 ? t <= 1 : t >= 1) {
      float sa[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) sa[k] = 0.f;
      int gI = 0;
      for (; gI + 15 < n_slabs; gI += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) sa[k] += partial[(long long)(gI + k) * TOTAL + e];
      }
      for (int k = 0; gI < n_slabs; ++gI, ++k) sa[k] += partial[(long long)gI * TOTAL + e];
#pragma unroll
      for (int stp = 8; stp >= 1; stp >>= 1)
#pragma unroll
        for (int k = 0; k < stp; ++k) sa[k] += sa[k + stp];
      s = sa[0];
    }
    dwbig[e] = s;
  }
  __shared__ float bias_sm[256];
  if (blockIdx.x == 0) {
    const int part = threadIdx.x / COUT, ch = threadIdx.x % COUT;           // 4 threads per channel
    float sb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sb[k] = 0.f;
    for (int g0 = part; g0 < n_slabs; g0 += 32) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int gI = g0 + 4 * k;
        if (gI < n_slabs) sb[k] += bias_part[(long long)gI * COUT + ch];
      }
    }
#pragma unroll
    for (int stp = 4; stp >= 1; stp >>= 1)
#pragma unroll
      for (int k = 0; k < stp; ++k) sb[k] += sb[k + stp];
    bias_sm[threadIdx.x] = sb[0];
    __syncthreads();
    if (threadIdx.x < COUT) dbias[threadIdx.x] = (bias_sm[threadIdx.x] + bias_sm[COUT + threadIdx.x]) +
                                                 (bias_sm[2 * COUT + threadIdx.x] + bias_sm[3 * COUT + threadIdx.x]);
  }
}

constexpr int kFwdTiles = 4, kDgradTiles = 2, kWgradTiles = 2, kWgradSlotsMax = 256;

int check_s1(const char* who, int B, int g, S1Args& p) {
  RCB_REQUIRE(B > 0 && g > 0 && (long long)B * g * 4 * COUT < (1ll << 40) && (long long)B * ((g + 31) / 32) < (1ll << 30), RCB_ERR_SHAPE,
              "%s: B=%d g=%d", who, B, g);
  p.B = B;
  p.g = g;
  p.tiles_per_row = (g + 31) / 32;
  return RCB_OK;
}

template <typename K>
int set_lds(K kfn, int bytes, const char* who) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  RCB_REQUIRE(e == hipSuccess, (int)e, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
  return RCB_OK;
}

}  // namespace

extern "C" int rcb_stage1_1d_fwd(const float* x, const void* wbig, const float* bias, void* x1, int32_t B, int32_t g,
                                 rcb_stream_t stream) {
  RCB_REQUIRE(x && wbig && bias && x1, RCB_ERR_ARG, "stage1_1d_fwd: null pointer");
  RCB_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(x1) & 15) == 0, RCB_ERR_ARG,
              "stage1_1d_fwd: 16-byte aligned tensors expected");
  S1Args p;
  memset(&p, 0, sizeof(p));
  int rc = check_s1("stage1_1d_fwd", B, g, p);
  if (rc) return rc;
  p.x = x;
  p.wbig = static_cast<const __bf16*>(wbig);
  p.bias = bias;
  p.y = static_cast<__bf16*>(x1);
  constexpr int lds = (32 * kFwdTiles + 2) * 256;
  rc = set_lds(s1_fwd1d_kernel<kFwdTiles>, lds, "stage1_1d_fwd");
  if (rc) return rc;
  const long long groups = (long long)B * cdiv(p.tiles_per_row, kFwdTiles);
  s1_fwd1d_kernel<kFwdTiles><<<(int)(groups < 512 ? groups : 512), 512, lds, (hipStream_t)stream>>>(p);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_stage1_1d_dgrad(const void* dz, const void* wbig, float* dx, int32_t B, int32_t g, rcb_stream_t stream) {
  RCB_REQUIRE(dz && wbig && dx, RCB_ERR_ARG, "stage1_1d_dgrad: null pointer");
  RCB_REQUIRE((reinterpret_cast<uintptr_t>(dz) & 15) == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(wbig) & 15) == 0,
              RCB_ERR_ARG, "stage1_1d_dgrad: 16-byte aligned tensors expected");
  S1Args p;
  memset(&p, 0, sizeof(p));
  int rc = check_s1("stage1_1d_dgrad", B, g, p);
  if (rc) return rc;
  p.dz = static_cast<const __bf16*>(dz);
  p.wbig = static_cast<const __bf16*>(wbig);
  p.dx = dx;
  constexpr int lds = (128 * kDgradTiles + 8) * 128;
  rc = set_lds(s1_dgrad1d_kernel<kDgradTiles>, lds, "stage1_1d_dgrad");
  if (rc) return rc;
  const long long groups = (long long)B * cdiv(p.tiles_per_row, kDgradTiles);
  s1_dgrad1d_kernel<kDgradTiles><<<(int)(groups < 512 ? groups : 512), 512, lds, (hipStream_t)stream>>>(p);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int64_t rcb_stage1_1d_wgrad_workspace(void) { return (int64_t)kWgradSlotsMax * (3 * CIN * WLD + COUT); }

extern "C" int rcb_stage1_1d_wgrad(const float* x, const void* dz, float* dwbig, float* dbias, float* workspace,
                                   int64_t workspace_floats, int32_t B, int32_t g, rcb_stream_t stream) {
  RCB_REQUIRE(x && dz && dwbig && dbias && workspace, RCB_ERR_ARG, "stage1_1d_wgrad: null pointer");
  RCB_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0, RCB_ERR_ARG,
              "stage1_1d_wgrad: 16-byte aligned tensors expected");
  RCB_REQUIRE(workspace_floats >= rcb_stage1_1d_wgrad_workspace(), RCB_ERR_SHAPE, "stage1_1d_wgrad: workspace too small");
  S1Args p;
  memset(&p, 0, sizeof(p));
  int rc = check_s1("stage1_1d_wgrad", B, g, p);
  if (rc) return rc;
  const long long units = (long long)B * cdiv(p.tiles_per_row, kWgradTiles);
  long long gx = units < kWgradSlotsMax ? units : kWgradSlotsMax;
  if (gx > cdiv(units, 8)) gx = cdiv(units, 8);            // a slab is 393 KB: at least eight passes per workgroup
  p.x = x;
  p.dz = static_cast<const __bf16*>(dz);
  p.partial = workspace;
  p.bias_part = workspace + (long long)kWgradSlotsMax * (3 * CIN * WLD);
  constexpr int lds = 2 * kWgradTiles * (4 * 34 * 4 + NPH * 2 * 32 * 4) * 16;
  rc = set_lds(s1_wgrad1d_kernel<kWgradTiles>, lds, "stage1_1d_wgrad");
  if (rc) return rc;
  s1_wgrad1d_kernel<kWgradTiles><<<(int)gx, 512, lds, (hipStream_t)stream>>>(p);
  s1_wgrad_sum_kernel<<<cdiv(3 * CIN * WLD, 256), 256, 0, (hipStream_t)stream>>>(workspace, p.bias_part, (int)gx, dwbig, dbias);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
