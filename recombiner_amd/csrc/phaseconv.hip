// N1 for grids of any dimension (1-D audio / protein, 3-D video; also valid for 2-D): the `nearest-upsample(2) -> conv(3, pad 1)`
// stages of the upsampling net (prior_model.py:52-54 with Conv1d / Conv3d: up2/conv2/act2, up3/conv3) as DIRECT sub-pixel
// convolutions on bf16 MFMA -- no window matrix.  Per axis, output pixel 2 i + a (phase a in {0, 1}) reads the two source
// pixels i + a + t - 1 (tap t in {0, 1}) with pre-summed weights
//     Weff[a][t] = sum of the kernel taps k with floor((a + k - 1) / 2) = a + t - 1:   a = 0: {0} | {1, 2};  a = 1: {0, 1} | {2}
// (products over the axes), so a stage costs 2^d instead of 3^d taps and the up-sampled tensor never exists.
//
//   forward : y[b, 2 i + a, co] = bias[co] + sum_t sum_ci x[b, i + a + t - 1, ci] Weff[a][t][ci][co]        (+ LeakyReLU)
//   dgrad   : dx[b, j, ci]      = LeakyReLU'(x[b, j, ci]) sum_{u in {-1..2}^d} sum_co dy[b, 2 j + u, co] V[u][co][ci]
//             with V[u] = Weff[a][t] for 2 j + u = 2 i + a, i.e. per axis u = -1: (1,1), 0: (0,1), 1: (1,0), 2: (0,0)
//
// Both are implicit GEMMs on v_mfma_f32_32x32x16_bf16 with the POSITION on the lane: one 32-position segment along the
// last active axis is the N dimension, channels are M, and every k-step of 16 is 16 contiguous channels of ONE shifted
// pixel -- a 16-byte load per lane straight from global memory (the activations of a whole stage are a few tens of MB and
// stay in L2 / MALL; every line is consumed completely by the k-steps of the pixel).  Forward: a wave owns one (phase,
// 32-channel output block) and keeps its 2^d x 4 weight fragments in registers across all its tiles.  Data gradient: the
// 4^d x COUT/16 fragments of an input-channel block do not fit and are streamed (one 1-KB fragment per MFMA, shared by all
// waves: L1 hits).  Activations are stored POST-LeakyReLU, so LeakyReLU' is the sign of the stored value.
#include "rcb_common.h"

using namespace rcb;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CIN = 64;
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

struct PcArgs {
  const __bf16* x;      // fwd: input activations [B][g...][64];  dgrad: dy [B][2g...][COUT]
  const uint4* frags;   // packed weight fragments
  const float* bias;    // fwd
  const __bf16* xact;   // dgrad: stored activations of the stage input (sign = LeakyReLU'), nullable
  __bf16* y;            // fwd: [B][2g...][COUT];  dgrad: dx [B][g...][64]
  int B, g0, g1, g2;    // SOURCE grid; axes k >= nd have size 1
  int tiles_per_row, n_tiles;
};

// ---- kernel-tap sums -------------------------------------------------------------------------------------------------
// R(a, t, k) = 1 iff kernel tap k of output phase a reads source tap t
__device__ __forceinline__ bool tap_hits(int a, int t, int k) { return a == 0 ? (t == 0 ? k == 0 : k >= 1) : (t == 0 ? k <= 1 : k == 2); }

template <int ND>
__device__ __forceinline__ float weff_elem(const float* __restrict__ W, int cout_total, int co, int ci, int a, int t) {
  // W [co][ci][3]^ND row-major;  a, t: bit (ND - 1 - k) belongs to axis k
  float s = 0.f;
  constexpr int KK = ND == 1 ? 3 : (ND == 2 ? 9 : 27);
  const float* w = W + ((long long)co * CIN + ci) * KK;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    bool on = true;
    int rem = kk;
#pragma unroll
    for (int ax = ND - 1; ax >= 0; --ax) {
      const int k = rem % 3;
      rem /= 3;
      on = on && tap_hits((a >> (ND - 1 - ax)) & 1, (t >> (ND - 1 - ax)) & 1, k);
    }
    if (on) s += w[kk];
  }
  return s;
}

// forward fragments: [phase][mb][tap][c16][64 lanes] -- lane (m = co, h) holds k = ci = 16 c16 + 8 h + j
template <int ND>
__global__ void pc_pack_fwd_kernel(const float* __restrict__ W, int cout, uint4* __restrict__ out) {
  constexpr int NP = 1 << ND;
  const int mbs = (cout + 31) / 32;
  const int total = NP * mbs * NP * 4 * 64;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int lane = e & 63, c16 = (e >> 6) & 3;
    int rest = e >> 8;
    const int t = rest % NP;
    rest /= NP;
    const int mb = rest % mbs, a = rest / mbs;
    const int co = 32 * mb + (lane & 31), h = lane >> 5;
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)(co < cout ? weff_elem<ND>(W, cout, co, 16 * c16 + 8 * h + j, a, t) : 0.f);
    out[e] = f.u;
  }
}

// data-gradient fragments: [mb][u][cb][64 lanes] -- lane (m = ci, h) holds k = co = 16 cb + 8 h + j of tap u in {-1..2}^ND
template <int ND>
__global__ void pc_pack_dgrad_kernel(const float* __restrict__ W, int cout, uint4* __restrict__ out) {
  constexpr int NU = 1 << (2 * ND);
  const int cbs = cout / 16;
  const int total = 2 * NU * cbs * 64;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int lane = e & 63;
    int rest = e >> 6;
    const int cb = rest % cbs;
    rest /= cbs;
    const int u = rest % NU, mb = rest / NU;
    // per axis the 2-bit digit of u (axis 0 most significant) is u_k + 1 in 0..3: -1 -> (a, t) = (1, 1); 0 -> (0, 1); 1 -> (1, 0); 2 -> (0, 0)
    int a = 0, t = 0;
#pragma unroll
    for (int ax = 0; ax < ND; ++ax) {
      const int d = (u >> (2 * (ND - 1 - ax))) & 3;
      const int ak = (d == 0 || d == 2) ? 1 : 0, tk = (d == 0 || d == 1) ? 1 : 0;
      a |= ak << (ND - 1 - ax);
      t |= tk << (ND - 1 - ax);
    }
    const int ci = 32 * mb + (lane & 31), h = lane >> 5;
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)weff_elem<ND>(W, cout, 16 * cb + 8 * h + j, ci, a, t);
    out[e] = f.u;
  }
}

// ---- forward -----------------------------------------------------------------------------------------------------------
template <int ND, int COUT, int ACT>
__global__ void __launch_bounds__(256) pc_fwd_kernel(PcArgs p) {
  constexpr int NP = 1 << ND, MB = (COUT + 31) / 32, NROLE = NP * MB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane & 31, h = lane >> 5;
  const int role = blockIdx.y * 4 + wave;
  if (role >= NROLE) return;
  const int a = role / MB, mb = role % MB;
  bf16x8 wf[NP][4];
#pragma unroll
  for (int t = 0; t < NP; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      Frag f;
      f.u = p.frags[((role * NP + t) * 4 + c) * 64 + lane];
      wf[t][c] = f.v;
    }
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = 32 * mb + rho(r, h);
    bv[r] = co < COUT ? p.bias[co] : 0.f;
  }
  const int g[3] = {p.g0, p.g1, p.g2};
  const int gl = g[ND - 1];
  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    int rest = tile / p.tiles_per_row;
    const int ts = tile - rest * p.tiles_per_row;
    int idx[3] = {0, 0, 0};
    idx[ND - 1] = 32 * ts + q;
#pragma unroll
    for (int ax = ND - 2; ax >= 0; --ax) {
      idx[ax] = rest % g[ax];
      rest /= g[ax];
    }
    const int b = rest;
    const bool valid = idx[ND - 1] < gl;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bv[r];
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      bool inb = true;
      long long off = b;
#pragma unroll
      for (int ax = 0; ax < ND; ++ax) {
        const int s = idx[ax] + ((a >> (ND - 1 - ax)) & 1) + ((t >> (ND - 1 - ax)) & 1) - 1;
        inb = inb && s >= 0 && s < g[ax];
        off = off * g[ax] + s;
      }
      const uint4* src = reinterpret_cast<const uint4*>(p.x + off * CIN + 8 * h);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        Frag f;
        f.u = inb ? src[2 * c] : make_uint4(0, 0, 0, 0);
        acc = mfma16(wf[t][c], f.v, acc);
      }
    }
    if (!valid) continue;
    long long oo = b;
#pragma unroll
    for (int ax = 0; ax < ND; ++ax) oo = oo * (2 * g[ax]) + 2 * idx[ax] + ((a >> (ND - 1 - ax)) & 1);
    __bf16* dst = p.y + oo * COUT + 32 * mb + 4 * h;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      if (32 * mb + 8 * g4 < COUT) {
        bf16x4 ob;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v = acc[4 * g4 + k];
          if (ACT) v = v > 0.f ? v : v * SLOPE;
          ob[k] = (__bf16)v;
        }
        *reinterpret_cast<bf16x4*>(dst + 8 * g4) = ob;
      }
    }
  }
}

// ---- data gradient -------------------------------------------------------------------------------------------------------
template <int ND, int COUT>
__global__ void __launch_bounds__(256) pc_dgrad_kernel(PcArgs p) {
  constexpr int NU = 1 << (2 * ND), CB = COUT / 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane & 31, h = lane >> 5;
  const int mb = wave & 1;
  const int g[3] = {p.g0, p.g1, p.g2};
  const int gl = g[ND - 1];
  const uint4* __restrict__ fr = p.frags + (long long)mb * NU * CB * 64 + lane;
  for (int tile = 2 * blockIdx.x + (wave >> 1); tile < p.n_tiles; tile += 2 * gridDim.x) {
    int rest = tile / p.tiles_per_row;
    const int ts = tile - rest * p.tiles_per_row;
    int idx[3] = {0, 0, 0};
    idx[ND - 1] = 32 * ts + q;
#pragma unroll
    for (int ax = ND - 2; ax >= 0; --ax) {
      idx[ax] = rest % g[ax];
      rest /= g[ax];
    }
    const int b = rest;
    const bool valid = idx[ND - 1] < gl;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
    for (int u = 0; u < NU; ++u) {
      bool inb = true;
      long long off = b;
#pragma unroll
      for (int ax = 0; ax < ND; ++ax) {
        const int s = 2 * idx[ax] + ((u >> (2 * (ND - 1 - ax))) & 3) - 1;
        inb = inb && s >= 0 && s < 2 * g[ax];
        off = off * (2 * g[ax]) + s;
      }
      const uint4* src = reinterpret_cast<const uint4*>(p.x + off * COUT + 8 * h);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        Frag f, w;
        f.u = inb ? src[2 * cb] : make_uint4(0, 0, 0, 0);
        w.u = fr[(u * CB + cb) * 64];
        acc = mfma16(w.v, f.v, acc);
      }
    }
    if (!valid) continue;
    long long oo = b;
#pragma unroll
    for (int ax = 0; ax < ND; ++ax) oo = oo * g[ax] + idx[ax];
    const long long e0 = oo * CIN + 32 * mb + 4 * h;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 ob;
      bf16x4 xa;
      if (p.xact) xa = *reinterpret_cast<const bf16x4*>(p.xact + e0 + 8 * g4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = acc[4 * g4 + k];
        if (p.xact) v *= ((float)xa[k] > 0.f ? 1.0f : SLOPE);
        ob[k] = (__bf16)v;
      }
      *reinterpret_cast<bf16x4*>(p.y + e0 + 8 * g4) = ob;
    }
  }
}

int check_pc(const char* who, int nd, int B, int g0, int g1, int g2, int cout, PcArgs& p) {
  RCB_REQUIRE(nd >= 1 && nd <= 3 && B > 0 && g0 > 0 && g1 > 0 && g2 > 0 && (nd > 1 || g1 == 1) && (nd > 2 || g2 == 1), RCB_ERR_SHAPE,
              "%s: B=%d grid=%dx%dx%d nd=%d (axes >= nd must have size 1)", who, B, g0, g1, g2, nd);
  RCB_REQUIRE(cout == 16 || cout == 64, RCB_ERR_UNSUPPORTED, "%s: cout=%d (64 and 16 are instantiated; cin is 64)", who, cout);
  const int g[3] = {g0, g1, g2};
  const long long rows = (long long)B * (nd > 1 ? g0 : 1) * (nd > 2 ? g1 : 1);
  p.tiles_per_row = (g[nd - 1] + 31) / 32;
  const long long nt = rows * p.tiles_per_row;
  RCB_REQUIRE(nt < (1ll << 30) && (long long)B * g0 * g1 * g2 * 8 < (1ll << 31), RCB_ERR_SHAPE, "%s: grid too large", who);
  p.n_tiles = (int)nt;
  p.B = B;
  p.g0 = g0;
  p.g1 = g1;
  p.g2 = g2;
  return RCB_OK;
}

}  // namespace

extern "C" int64_t rcb_phaseconv_pack_uint4(int32_t nd, int32_t cout, int32_t which) {
  if (nd < 1 || nd > 3 || (cout != 16 && cout != 64)) return -1;
  const int np = 1 << nd;
  if (which == 0) return (int64_t)np * ((cout + 31) / 32) * np * 4 * 64;
  return (int64_t)2 * (1 << (2 * nd)) * (cout / 16) * 64;
}

extern "C" int rcb_phaseconv_pack(const float* conv_weight, int32_t nd, int32_t cout, void* fwd_frags, void* dgrad_frags,
                                  rcb_stream_t stream) {
  RCB_REQUIRE(conv_weight && (fwd_frags || dgrad_frags), RCB_ERR_ARG, "phaseconv_pack: null pointer");
  RCB_REQUIRE(nd >= 1 && nd <= 3 && (cout == 16 || cout == 64), RCB_ERR_UNSUPPORTED, "phaseconv_pack: nd=%d cout=%d", nd, cout);
  hipStream_t s = (hipStream_t)stream;
#define RCB_PACK(NDv)                                                                                            \
  if (nd == NDv) {                                                                                               \
    if (fwd_frags) pc_pack_fwd_kernel<NDv><<<64, 256, 0, s>>>(conv_weight, cout, static_cast<uint4*>(fwd_frags));       \
    if (dgrad_frags) pc_pack_dgrad_kernel<NDv><<<64, 256, 0, s>>>(conv_weight, cout, static_cast<uint4*>(dgrad_frags)); \
  }
  RCB_PACK(1)
  RCB_PACK(2)
  RCB_PACK(3)
#undef RCB_PACK
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_phaseconv_fwd(const void* x, const void* fwd_frags, const float* bias, void* y, int32_t B, int32_t g0,
                                 int32_t g1, int32_t g2, int32_t nd, int32_t cout, int32_t leaky_out, rcb_stream_t stream) {
  RCB_REQUIRE(x && fwd_frags && bias && y, RCB_ERR_ARG, "phaseconv_fwd: null pointer");
  PcArgs p;
  memset(&p, 0, sizeof(p));
  int rc = check_pc("phaseconv_fwd", nd, B, g0, g1, g2, cout, p);
  if (rc) return rc;
  p.x = static_cast<const __bf16*>(x);
  p.frags = static_cast<const uint4*>(fwd_frags);
  p.bias = bias;
  p.y = static_cast<__bf16*>(y);
  const int nrole = (1 << nd) * ((cout + 31) / 32);
  int gx = p.n_tiles < 2048 ? p.n_tiles : 2048;      // waves keep their fragments across the tiles they walk
  dim3 grid(gx, (nrole + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
#define RCB_FWD(NDv, Cv)                                                                  \
  if (nd == NDv && cout == Cv) {                                                          \
    if (leaky_out) pc_fwd_kernel<NDv, Cv, 1><<<grid, 256, 0, s>>>(p);                     \
    else pc_fwd_kernel<NDv, Cv, 0><<<grid, 256, 0, s>>>(p);                               \
  }
  RCB_FWD(1, 64) RCB_FWD(1, 16) RCB_FWD(2, 64) RCB_FWD(2, 16) RCB_FWD(3, 64) RCB_FWD(3, 16)
#undef RCB_FWD
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_phaseconv_dgrad(const void* dy, const void* dgrad_frags, const void* x_act, void* dx, int32_t B, int32_t g0,
                                   int32_t g1, int32_t g2, int32_t nd, int32_t cout, rcb_stream_t stream) {
  RCB_REQUIRE(dy && dgrad_frags && dx, RCB_ERR_ARG, "phaseconv_dgrad: null pointer");
  PcArgs p;
  memset(&p, 0, sizeof(p));
  int rc = check_pc("phaseconv_dgrad", nd, B, g0, g1, g2, cout, p);
  if (rc) return rc;
  p.x = static_cast<const __bf16*>(dy);
  p.frags = static_cast<const uint4*>(dgrad_frags);
  p.xact = static_cast<const __bf16*>(x_act);
  p.y = static_cast<__bf16*>(dx);
  int gx = (p.n_tiles + 1) / 2;
  if (gx > 4096) gx = 4096;
  hipStream_t s = (hipStream_t)stream;
#define RCB_DG(NDv, Cv) \
  if (nd == NDv && cout == Cv) pc_dgrad_kernel<NDv, Cv><<<gx, 256, 0, s>>>(p);
  RCB_DG(1, 64) RCB_DG(1, 16) RCB_DG(2, 64) RCB_DG(2, 16) RCB_DG(3, 64) RCB_DG(3, 16)
#undef RCB_DG
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
