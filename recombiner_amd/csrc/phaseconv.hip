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
  // Per axis the taps of (phase bit, source-tap bit) are {0}, {1, 2}, {0, 1} or {2}: at most two.  The 2^ND candidate index
  // tuples are loaded without a branch (a missing second candidate repeats the first and is dropped by a select): 2 / 4 / 8
  // independent loads per element.  Under a branch per kernel tap the loads went out one at a time, each behind its own wait
  // (tools/serial_loads.py: 420 serialised pairs in the 3-D pack kernel); loading all 3^ND taps unconditionally was slower in 3-D.
  int k0[ND], k1[ND];
  bool two[ND];
#pragma unroll
  for (int ax = 0; ax < ND; ++ax) {
    const int ab = (a >> (ND - 1 - ax)) & 1, tb = (t >> (ND - 1 - ax)) & 1;
    k0[ax] = ab == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2);
    two[ax] = (ab == 0 && tb == 1) || (ab == 1 && tb == 0);
    k1[ax] = two[ax] ? k0[ax] + 1 : k0[ax];
  }
  float wv[1 << ND];
  bool on[1 << ND];
#pragma unroll
  for (int c = 0; c < (1 << ND); ++c) {
    int idx = 0;
    bool ok = true;
#pragma unroll
    for (int ax = 0; ax < ND; ++ax) {
      const int pick = (c >> (ND - 1 - ax)) & 1;
      idx = idx * 3 + (pick ? k1[ax] : k0[ax]);
      ok = ok && (pick == 0 || two[ax]);
    }
    wv[c] = w[idx];
    on[c] = ok;
  }
  // (candidates in ascending tap order: the order in which the taps were summed before)
#pragma unroll
  for (int c = 0; c < (1 << ND); ++c) s += on[c] ? wv[c] : 0.f;
  return s;
}

// forward fragments: [phase][mb][tap][c16][64 lanes] -- lane (m = co, h) holds k = ci = 16 c16 + 8 h + j
template <int ND>
__global__ void pc_pack_fwd_kernel(const float* __restrict__ W, int cout, uint4* __restrict__ out) {
  constexpr int NP = 1 << ND;
  const int mbs = (cout + 31) / 32;
  const int total = NP * mbs * NP * 4 * 64;
  if (cout == 16) {
    // 16 output channels are ONE row block of v_mfma_f32_16x16x32_bf16: [phase][tap][kb2][64 lanes], lane = (co = lane & 15,
    // k-group lane >> 4) holds ci = 32 kb2 + 8 (lane >> 4) + j (the first half of the buffer; the rest stays unused)
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < NP * NP * 2 * 64; e += gridDim.x * blockDim.x) {
      const int lane = e & 63, kb2 = (e >> 6) & 1, t = (e >> 7) % NP, a = (e >> 7) / NP;
      Frag f;
#pragma unroll
      for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)weff_elem<ND>(W, cout, lane & 15, 32 * kb2 + 8 * (lane >> 4) + j, a, t);
      out[e] = f.u;
    }
    return;
  }
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
// One workgroup = 8 waves = all 2^ND phases of TPB = 8 / 2^ND consecutive tiles of one row (blockIdx.y = 32-channel output
// block).  The 3^(ND-1) source rows the phases share -- (32 TPB + 2) pixels each, one pixel of halo -- are staged ONCE in
// LDS with coalesced 16-byte loads (8 lanes per pixel); the B fragments (16 bytes of one shifted pixel per lane) are then
// ds_read_b128 from an image whose 16-byte chunk index is XORed with (pixel >> 1) & 7: conflict-free for the 16-lane groups
// of the instruction.  (Read straight from global memory the same fragments cost one cache-line lookup per lane: the first
// version of this kernel was bound by exactly that, at 5x the time.)
template <int ND, int COUT, int ACT>
__global__ void __launch_bounds__(512) pc_fwd_kernel(PcArgs p) {
  constexpr int NP = 1 << ND, MB = (COUT + 31) / 32, TPB = 8 / NP;
  constexpr int NROW = ND == 1 ? 1 : (ND == 2 ? 3 : 9), PW = 32 * TPB + 2, NCH = NROW * PW * 8, NIT = (NCH + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char pc_smem[];
  uint4* img = reinterpret_cast<uint4*>(pc_smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane & 31, h = lane >> 5;
  // the two 32-channel output blocks of a tile group run as workgroups 8 apart in the 1-D grid: the same XCD (workgroup ids go
  // round-robin over the 8 XCDs), dispatched together -- the second one finds the source rows in that XCD's L2 and the two
  // 64-byte halves of every output line are written close in time.  (As grid.y the blocks ran a whole grid apart.)
  const int n_walkers = gridDim.x / MB;
  const int pw = MB == 1 ? (int)blockIdx.x : (int)((blockIdx.x & 7) | ((blockIdx.x >> 4) << 3));      // (pw & 7 = this workgroup's XCD)
  // 1-D and 3-D grids: XCD k walks the k-th contiguous eighth of every sweep over the tile groups (round-robin ids put every
  // neighbour on another XCD).  Same-box, 64 channels: audio stage 2 1.48 -> 1.33 ms, video stage 2 0.185 -> 0.172 ms; the 2-D
  // photo grid lost what the pairing had gained (0.162 -> 0.180 ms) and keeps the round-robin order; no effect at 16 channels
  const int walker = (ND != 2 && (n_walkers & 7) == 0) ? (pw & 7) * (n_walkers >> 3) + (pw >> 3) : pw;
  const int a = wave % NP, tsub = wave / NP, mb = MB == 1 ? 0 : (blockIdx.x >> 3) & 1;
  const int role = a * MB + mb;
  constexpr bool C16 = (COUT == 16);            // one row block of the 16 x 16 x 32 MFMA (see pc_pack_fwd_kernel)
  bf16x8 wf[NP][C16 ? 2 : 4];
#pragma unroll
  for (int t = 0; t < NP; ++t)
#pragma unroll
    for (int c = 0; c < (C16 ? 2 : 4); ++c) {
      Frag f;
      f.u = C16 ? p.frags[((a * NP + t) * 2 + c) * 64 + lane] : p.frags[((role * NP + t) * 4 + c) * 64 + lane];
      wf[t][c] = f.v;
    }
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = C16 ? 4 * (lane >> 4) + (r & 3) : 32 * mb + rho(r, h);
    bv[r] = co < COUT ? p.bias[co] : 0.f;
  }
  const int g[3] = {p.g0, p.g1, p.g2};
  const int gl = g[ND - 1];
  const int groups_per_row = (p.tiles_per_row + TPB - 1) / TPB;
  const int n_groups = (p.n_tiles / p.tiles_per_row) * groups_per_row;
  // the source rows of group n + 1 travel from memory while group n is multiplied (registers stg: free once stored to LDS)
  uint4 stg[NIT];
  auto fetch = [&](int grp_) {
    int rest = grp_ / groups_per_row;
    const int l0 = 32 * TPB * (grp_ - rest * groups_per_row);      // first pixel of the group along the last active axis
    int lead[3] = {0, 0, 0};
#pragma unroll
    for (int ax = ND - 2; ax >= 0; --ax) {
      lead[ax] = rest % g[ax];
      rest /= g[ax];
    }
    const int b = rest;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = threadIdx.x + 512 * it;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < NCH) {
        const int c = e & 7, pp = (e >> 3) % PW, r = (e >> 3) / PW;
        bool inb = true;
        long long off = b;
        if (ND == 3) {
          const int s0 = lead[0] + r / 3 - 1, s1 = lead[1] + r % 3 - 1;
          inb = s0 >= 0 && s0 < g[0] && s1 >= 0 && s1 < g[1];
          off = (off * g[0] + s0) * g[1] + s1;
        } else if (ND == 2) {
          const int s0 = lead[0] + r - 1;
          inb = s0 >= 0 && s0 < g[0];
          off = off * g[0] + s0;
        }
        const int sl = l0 - 1 + pp;
        inb = inb && sl >= 0 && sl < gl;
        if (inb) v = reinterpret_cast<const uint4*>(p.x + (off * gl + sl) * CIN)[c];
      }
      stg[it] = v;
    }
  };
  if (walker < n_groups) fetch(walker);
  for (int grp = walker; grp < n_groups; grp += n_walkers) {
    int rest = grp / groups_per_row;
    const int tg = grp - rest * groups_per_row;
    int lead[3] = {0, 0, 0};
#pragma unroll
    for (int ax = ND - 2; ax >= 0; --ax) {
      lead[ax] = rest % g[ax];
      rest /= g[ax];
    }
    const int b = rest;
    __syncthreads();                               // every wave is done reading the previous group's image
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = threadIdx.x + 512 * it;
      if (e < NCH) {
        const int c = e & 7, pix = e >> 3, pp = pix % PW;
        img[pix * 8 + (c ^ ((pp >> 1) & 7))] = stg[it];
      }
    }
    __syncthreads();
    if (grp + n_walkers < n_groups) fetch(grp + n_walkers);
    // ---- this wave's phase of its tile -----------------------------------------------------------------------------------
    const int ts = tg * TPB + tsub;
    if (ts >= p.tiles_per_row) continue;           // (no barrier below this point inside the iteration)
    if constexpr (C16) {
      // lane = (position j = lane & 15 of a 16-position half, k-group kg = lane >> 4); D = [16 co x 16 positions]: lane holds
      // channels 4 kg + r.  Two halves per tile; per tap two MFMAs of K = 32 input channels each.
      typedef float f32x4v __attribute__((ext_vector_type(4)));
      const int j16 = lane & 15, kg = lane >> 4;
      f32x4v acc4[2];
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[ph][r] = bv[r];
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        int r = 0;
        if (ND == 3) r = (((a >> 2) & 1) + ((t >> 2) & 1)) * 3 + (((a >> 1) & 1) + ((t >> 1) & 1));
        if (ND == 2) r = ((a >> 1) & 1) + ((t >> 1) & 1);
        Frag f[2][2];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          const int pp = 32 * tsub + 16 * ph + j16 + (a & 1) + (t & 1);
          const uint4* src = img + (r * PW + pp) * 8;
          const int sw = (pp >> 1) & 7;
#pragma unroll
          for (int kb2 = 0; kb2 < 2; ++kb2) f[ph][kb2].u = src[(4 * kb2 + kg) ^ sw];
        }
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int kb2 = 0; kb2 < 2; ++kb2)
            acc4[ph] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][kb2], f[ph][kb2].v, acc4[ph], 0, 0, 0);
      }
      long long ob = b;
#pragma unroll
      for (int ax = 0; ax < ND - 1; ++ax) ob = ob * (2 * g[ax]) + 2 * lead[ax] + ((a >> (ND - 1 - ax)) & 1);
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        // neighbouring k-groups exchange: even kg ends with 8 consecutive channels (its own four + its odd neighbour's)
        float lo[4], hi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc4[ph][r];
          if (ACT) v = v > 0.f ? v : v * SLOPE;
          auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), 0u, false, false);
          lo[r] = __uint_as_float(sw[0]);
          hi[r] = __uint_as_float(sw[1]);
        }
        const int il16 = 32 * ts + 16 * ph + j16;
        if (il16 < gl && (kg & 1) == 0) {
          Frag o;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            o.v[k] = (__bf16)lo[k];
            o.v[4 + k] = (__bf16)hi[k];
          }
          *reinterpret_cast<uint4*>(p.y + (ob * (2 * gl) + 2 * il16 + (a & 1)) * COUT + 4 * kg) = o.u;
        }
      }
      continue;
    }
    const int il = 32 * ts + q;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bv[r];
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      int r = 0;
      if (ND == 3) r = (((a >> 2) & 1) + ((t >> 2) & 1)) * 3 + (((a >> 1) & 1) + ((t >> 1) & 1));
      if (ND == 2) r = ((a >> 1) & 1) + ((t >> 1) & 1);
      const int pp = 32 * tsub + q + (a & 1) + (t & 1);
      const uint4* src = img + (r * PW + pp) * 8;
      const int sw = (pp >> 1) & 7;
      Frag f[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) f[c].u = src[(2 * c + h) ^ sw];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc = mfma16(wf[t][c], f[c].v, acc);
    }
    long long oo = b;
#pragma unroll
    for (int ax = 0; ax < ND - 1; ++ax) oo = oo * (2 * g[ax]) + 2 * lead[ax] + ((a >> (ND - 1 - ax)) & 1);
    oo = oo * (2 * gl) + 2 * il + (a & 1);
    // the lane halves swap (v_permlane32_swap: every lane takes part, the bounds check comes after) so that lane (q, h) owns
    // the 16 consecutive channels 32 mb + 16 h .. + 15 of its pixel: two 16-byte stores instead of four 8-byte pieces
    uint4* dst = reinterpret_cast<uint4*>(p.y + oo * COUT + 32 * mb + 16 * h);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      Frag ob;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v0 = acc[4 * hf + k], v1 = acc[8 + 4 * hf + k];
        if (ACT) {
          v0 = v0 > 0.f ? v0 : v0 * SLOPE;
          v1 = v1 > 0.f ? v1 : v1 * SLOPE;
        }
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
        ob.v[k] = (__bf16)__uint_as_float(sw[0]);
        ob.v[4 + k] = (__bf16)__uint_as_float(sw[1]);
      }
      if (il < gl) dst[hf] = ob.u;
    }
  }
}

// ---- data gradient -------------------------------------------------------------------------------------------------------
// One workgroup = 4 waves = four consecutive tiles (in row-major tile order: rows may be as short as one tile) x both
// 32-channel input blocks; wave (mb, pair) owns two tiles, so every streamed weight fragment (the 4^ND x COUT/16 fragments
// of a block fit neither in registers nor, in 3-D at COUT = 64, in LDS) feeds two MFMAs; the fragments of the next pass
// are requested before the MFMAs of the current one.  The upstream gradient is staged through LDS per tile as one dy row
// segment (66 pixels: 2 x 32 + 2) and one 16-channel block at a time, double-buffered: coalesced 16-byte loads in, and the
// B fragments (pixel 2 j + u of the segment, stride-2 over the lanes) out by ds_read_b128 from an image whose chunk index
// is XORed with (pixel >> 3) & 3 -- conflict-free for the stride-2 pattern (checked exhaustively off line).
template <int ND, int COUT>
__global__ void __launch_bounds__(256) pc_dgrad_kernel(PcArgs p) {
  constexpr int NU = 1 << (2 * ND), CB = COUT / 16, NLEAD = 1 << (2 * (ND - 1)), PWD = 66, TCH = 136, NCHK = 4 * TCH;
  constexpr int NIT = (4 * PWD * 2 + 255) / 256;
  __shared__ uint4 img[2][NCHK];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane & 31, h = lane >> 5;
  const int mb = wave & 1, pair = wave >> 1;
  const int g[3] = {p.g0, p.g1, p.g2};
  const int gl = g[ND - 1];
  const uint4* __restrict__ fr = p.frags + (long long)mb * NU * CB * 64 + lane;
  const int n_groups = (p.n_tiles + 3) / 4;
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    // staging role of this thread: chunk e -> (tile slot, pixel of its segment, 16-byte half)
    int s_slot[NIT], s_pp[NIT], s_c[NIT];
    long long s_base[NIT];
    int s_lead0[NIT], s_lead1[NIT], s_j0[NIT];
    bool s_on[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = threadIdx.x + 256 * it;
      s_on[it] = e < 4 * PWD * 2;
      const int ee = s_on[it] ? e : 0;
      s_slot[it] = ee / (PWD * 2);
      const int rem = ee - s_slot[it] * (PWD * 2);
      s_pp[it] = rem >> 1;
      s_c[it] = rem & 1;
      const int tile = grp * 4 + s_slot[it];
      s_on[it] = s_on[it] && tile < p.n_tiles;
      int rest = min(tile, p.n_tiles - 1) / p.tiles_per_row;
      s_j0[it] = 32 * (min(tile, p.n_tiles - 1) - rest * p.tiles_per_row);
      int l0 = 0, l1 = 0;
      if (ND == 3) {
        l1 = rest % g[1];
        rest /= g[1];
        l0 = rest % g[0];
        rest /= g[0];
      } else if (ND == 2) {
        l0 = rest % g[0];
        rest /= g[0];
      }
      s_lead0[it] = l0;
      s_lead1[it] = l1;
      s_base[it] = rest;                            // batch index
    }
    f32x16 acc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    uint4 stg[NIT];
    auto fetch = [&](int pass) {
      const int cb = pass / NLEAD, ld = pass % NLEAD;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        bool inb = s_on[it];
        long long off = s_base[it];
        if (ND == 3) {
          const int s0 = 2 * s_lead0[it] + ((ld >> 2) & 3) - 1, s1 = 2 * s_lead1[it] + (ld & 3) - 1;
          inb = inb && s0 >= 0 && s0 < 2 * g[0] && s1 >= 0 && s1 < 2 * g[1];
          off = (off * (2 * g[0]) + s0) * (2 * g[1]) + s1;
        } else if (ND == 2) {
          const int s0 = 2 * s_lead0[it] + (ld & 3) - 1;
          inb = inb && s0 >= 0 && s0 < 2 * g[0];
          off = off * (2 * g[0]) + s0;
        }
        const int sl = 2 * s_j0[it] - 1 + s_pp[it];
        uint4 v = make_uint4(0, 0, 0, 0);
        if (inb && sl >= 0 && sl < 2 * gl) v = reinterpret_cast<const uint4*>(p.x + (off * (2 * gl) + sl) * COUT + 16 * cb)[s_c[it]];
        stg[it] = v;
      }
    };
    auto stash = [&](int buf) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e = threadIdx.x + 256 * it;
        if (e < 4 * PWD * 2) {
          const int L = s_pp[it] * 2 + s_c[it];
          img[buf][s_slot[it] * TCH + (L ^ ((s_pp[it] >> 3) & 3))] = stg[it];
        }
      }
    };
    constexpr int NPASS = CB * NLEAD;
    Frag wnext[4];
    auto fetch_w = [&](int pass) {
      const int cb = pass / NLEAD, ld = pass % NLEAD;
#pragma unroll
      for (int dl = 0; dl < 4; ++dl) wnext[dl].u = fr[((ld * 4 + dl) * CB + cb) * 64];
    };
    fetch(0);
    fetch_w(0);
    __syncthreads();                               // the previous group's last pass is done with both buffers
    stash(0);
    __syncthreads();
    for (int pass = 0; pass < NPASS; ++pass) {
      const int buf = pass & 1;
      Frag w[4];
#pragma unroll
      for (int dl = 0; dl < 4; ++dl) w[dl] = wnext[dl];
      if (pass + 1 < NPASS) {
        fetch(pass + 1);
        fetch_w(pass + 1);
      }
#pragma unroll
      for (int dl = 0; dl < 4; ++dl) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int pp = 2 * q + dl;
          Frag f;
          f.u = img[buf][(2 * pair + k) * TCH + ((pp * 2 + h) ^ ((pp >> 3) & 3))];
          acc[k] = mfma16(w[dl].v, f.v, acc[k]);
        }
      }
      if (pass + 1 < NPASS) {
        stash(buf ^ 1);                            // (buffer buf ^ 1 was last read in pass - 1: every wave passed the barrier since)
        __syncthreads();
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int tile = grp * 4 + 2 * pair + k;
      if (tile >= p.n_tiles) continue;               // (uniform over the wave)
      int rest = tile / p.tiles_per_row;
      const int il = 32 * (tile - rest * p.tiles_per_row) + q;
      // the lane halves swap (v_permlane32_swap) so that lane (q, h) owns the 16 consecutive channels 32 mb + 16 h .. + 15 of its
      // pixel: two 16-byte loads of the stored activation and two 16-byte stores instead of four 8-byte pieces each
      // rest = flattened (batch, leading indices): exactly the row index of dx
      const long long e0 = ((long long)rest * gl + il) * CIN + 32 * mb + 16 * h;
      const bool on = il < gl;
      Frag xa[2];
      if (p.xact && on) {
        xa[0].u = reinterpret_cast<const uint4*>(p.xact + e0)[0];
        xa[1].u = reinterpret_cast<const uint4*>(p.xact + e0)[1];
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float o[8];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[k][4 * hf + kk]), __float_as_uint(acc[k][8 + 4 * hf + kk]),
                                                     false, false);
          o[kk] = __uint_as_float(sw[0]);
          o[4 + kk] = __uint_as_float(sw[1]);
        }
        Frag ob;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          float v = o[kk];
          if (p.xact) v *= ((float)xa[hf].v[kk] > 0.f ? 1.0f : SLOPE);
          ob.v[kk] = (__bf16)v;
        }
        if (on) reinterpret_cast<uint4*>(p.y + e0)[hf] = ob.u;
      }
    }
  }
}

// The 64-channel stage (stage 2: dy has 64 channels): the kernel above stages ONE 16-channel block of the dy segment per pass,
// i.e. 4 x 4^(ND-1) passes of eight MFMAs per wave with a barrier each (16 passes per tile group on the photo grids, 64 on the
// video grids).  Here a pass stages whole 128-byte pixels of a segment: 4^(ND-1) passes of 32 MFMAs, the weight fragments of
// the next 16-channel block requested while the current block is multiplied.  One image set (34 KB: four workgroups per CU), the
// next pass's loads in flight during the MFMAs; adjacent pixels swapped where bit 1 of the pixel index is set and the chunk index
// XORed with (pixel >> 2) & 7: the lanes' stride-2 gather covers all 64 banks once per 16-lane group (tools/lds_banks.py model).
__device__ __forceinline__ int dg64_slot(int P, int chunk) { return (P ^ ((P >> 1) & 1)) * 8 + (chunk ^ ((P >> 2) & 7)); }

template <int ND>
__global__ void __launch_bounds__(256) pc_dgrad64_kernel(PcArgs p) {
  constexpr int COUT = 64, CB = 4, NU = 1 << (2 * ND), NLEAD = 1 << (2 * (ND - 1)), PWD = 66, TCH = PWD * 8, NCH = 4 * TCH;
  constexpr int NIT = (NCH + 255) / 256;
  __shared__ uint4 img[NCH];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane & 31, h = lane >> 5;
  const int mb = wave & 1, pair = wave >> 1;
  const int g[3] = {p.g0, p.g1, p.g2};
  const int gl = g[ND - 1];
  const uint4* __restrict__ fr = p.frags + (long long)mb * NU * CB * 64 + lane;
  const int n_groups = (p.n_tiles + 3) / 4;
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    // the four tiles of the group (uniform): batch index, leading source indices, first pixel
    long long t_base[4];
    int t_l0[4], t_l1[4], t_j0[4];
    bool t_on[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int tile = grp * 4 + sl;
      t_on[sl] = tile < p.n_tiles;
      const int tc = t_on[sl] ? tile : p.n_tiles - 1;
      int rest = tc / p.tiles_per_row;
      t_j0[sl] = 32 * (tc - rest * p.tiles_per_row);
      int l0 = 0, l1 = 0;
      if (ND == 3) {
        l1 = rest % g[1];
        rest /= g[1];
        l0 = rest % g[0];
        rest /= g[0];
      } else if (ND == 2) {
        l0 = rest % g[0];
        rest /= g[0];
      }
      t_l0[sl] = l0;
      t_l1[sl] = l1;
      t_base[sl] = rest;
    }
    uint4 stg[NIT];
    auto fetch = [&](int ld) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e = threadIdx.x + 256 * it;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (e < NCH) {
          const int sl = e / TCH, rem = e - sl * TCH, pp = rem >> 3, c = rem & 7;
          // (the tile of the chunk is one of four uniform records: selected, not indexed)
          const long long base = sl == 0 ? t_base[0] : (sl == 1 ? t_base[1] : (sl == 2 ? t_base[2] : t_base[3]));
          const int l0 = sl == 0 ? t_l0[0] : (sl == 1 ? t_l0[1] : (sl == 2 ? t_l0[2] : t_l0[3]));
          const int l1 = sl == 0 ? t_l1[0] : (sl == 1 ? t_l1[1] : (sl == 2 ? t_l1[2] : t_l1[3]));
          const int j0 = sl == 0 ? t_j0[0] : (sl == 1 ? t_j0[1] : (sl == 2 ? t_j0[2] : t_j0[3]));
          bool inb = sl == 0 ? t_on[0] : (sl == 1 ? t_on[1] : (sl == 2 ? t_on[2] : t_on[3]));
          long long off = base;
          if (ND == 3) {
            const int s0 = 2 * l0 + ((ld >> 2) & 3) - 1, s1 = 2 * l1 + (ld & 3) - 1;
            inb = inb && s0 >= 0 && s0 < 2 * g[0] && s1 >= 0 && s1 < 2 * g[1];
            off = (off * (2 * g[0]) + s0) * (2 * g[1]) + s1;
          } else if (ND == 2) {
            const int s0 = 2 * l0 + (ld & 3) - 1;
            inb = inb && s0 >= 0 && s0 < 2 * g[0];
            off = off * (2 * g[0]) + s0;
          }
          const int px = 2 * j0 - 1 + pp;
          if (inb && px >= 0 && px < 2 * gl) v = reinterpret_cast<const uint4*>(p.x + (off * (2 * gl) + px) * COUT)[c];
        }
        stg[it] = v;
      }
    };
    f32x16 acc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    Frag wnext[4];
    auto fetch_w = [&](int ld, int cb) {
#pragma unroll
      for (int dl = 0; dl < 4; ++dl) wnext[dl].u = fr[((ld * 4 + dl) * CB + cb) * 64];
    };
    fetch(0);
    fetch_w(0, 0);
#pragma unroll 1
    for (int ld = 0; ld < NLEAD; ++ld) {
      __syncthreads();                             // every wave is done reading the previous pass's image
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e = threadIdx.x + 256 * it;
        if (e < NCH) {
          const int sl = e / TCH, rem = e - sl * TCH;
          img[sl * TCH + dg64_slot(rem >> 3, rem & 7)] = stg[it];
        }
      }
      __syncthreads();
      if (ld + 1 < NLEAD) fetch(ld + 1);
#pragma unroll 1
      for (int cb = 0; cb < CB; ++cb) {             // (rolled: unrolled, the compiler hoists all 32 image reads of the pass -- 128 registers)
        Frag w[4];
#pragma unroll
        for (int dl = 0; dl < 4; ++dl) w[dl] = wnext[dl];
        if (cb + 1 < CB) fetch_w(ld, cb + 1);
        else if (ld + 1 < NLEAD) fetch_w(ld + 1, 0);
#pragma unroll
        for (int dl = 0; dl < 4; ++dl) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            Frag f;
            f.u = img[(2 * pair + k) * TCH + dg64_slot(2 * q + dl, 2 * cb + h)];
            acc[k] = mfma16(w[dl].v, f.v, acc[k]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int tile = grp * 4 + 2 * pair + k;
      if (tile >= p.n_tiles) continue;               // (uniform over the wave)
      int rest = tile / p.tiles_per_row;
      const int il = 32 * (tile - rest * p.tiles_per_row) + q;
      const long long e0 = ((long long)rest * gl + il) * CIN + 32 * mb + 16 * h;      // (lane halves swap: see pc_dgrad_kernel)
      const bool on = il < gl;
      Frag xa[2];
      if (p.xact && on) {
        xa[0].u = reinterpret_cast<const uint4*>(p.xact + e0)[0];
        xa[1].u = reinterpret_cast<const uint4*>(p.xact + e0)[1];
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float o[8];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[k][4 * hf + kk]), __float_as_uint(acc[k][8 + 4 * hf + kk]),
                                                     false, false);
          o[kk] = __uint_as_float(sw[0]);
          o[4 + kk] = __uint_as_float(sw[1]);
        }
        Frag ob;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          float v = o[kk];
          if (p.xact) v *= ((float)xa[hf].v[kk] > 0.f ? 1.0f : SLOPE);
          ob.v[kk] = (__bf16)v;
        }
        if (on) reinterpret_cast<uint4*>(p.y + e0)[hf] = ob.u;
      }
    }
  }
}

// ---- weight gradient -------------------------------------------------------------------------------------------------------
//   dWeff[a][t][ci][co] = sum_{b, i} x[b, i + a + t - 1, ci] dy[b, 2 i + a, co]        dbias[co] = sum dy
// The contraction runs over positions, i.e. over the lanes' axis of both operands: the [pixel][channel] images of a tile
// are staged in LDS (64-byte rows: 32 channels of this workgroup's block) and both MFMA operands are read TRANSPOSED with
// ds_read_b64_tr_b16 (four consecutive pixels x 64 bytes per 32-lane group: every bank once, no swizzle needed).
// grid.y = (32-channel input block mb) x (32-column output block nb); a workgroup = 8 waves walks its share of the tiles:
//   3-D: wave = phase a, its 8 taps;   2-D: wave = (phase, tap pair);   1-D: waves 0..3 = (phase, tap)
// so a wave reads its dy operand once per k-step and one shifted x operand per tap.  Per tile: the 3^(ND-1) x 34-pixel
// source rows and the 2^ND phase images of dy.  Sums stay in registers across the walk and leave as one fp32 slab per
// workgroup; pc_wgrad_finish_kernel adds the slabs in a fixed order (no atomics: bitwise reproducible) and folds the phase
// weights back onto the conv taps (transpose of the tap sums of the forward pass).
typedef short s16x4 __attribute__((ext_vector_type(4)));

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

struct WgArgs {
  const __bf16* x;    // [B][g...][64] stage input activations
  const __bf16* dy;   // [B][2g...][COUT]
  float* partial;     // [gridDim.x][gridDim.y][combos][32][32] (+ [gridDim.x][COUT] bias partials behind)
  float* bias_part;
  int B, g0, g1, g2, tiles_per_row, n_tiles;
};

template <int ND, int COUT>
__global__ void __launch_bounds__(512) pc_wgrad_kernel(WgArgs p) {
  constexpr int NP = 1 << ND, NBLK = (COUT + 31) / 32, NROW = ND == 1 ? 1 : (ND == 2 ? 3 : 9), PWX = 34;
  constexpr int TPW = ND == 3 ? 8 : (ND == 2 ? 2 : 1), TG = NP / TPW;       // taps per wave, tap groups per phase
  constexpr int NWAVE = NP * TG;                                           // active waves (8, 8, 4)
  constexpr int NXC = NROW * PWX * 4;                                      // 16-byte chunks of the x image
  constexpr int DYROWS = 1 << (ND - 1), NCD = COUT == 16 ? 2 : 4;           // dy rows per tile, chunks per dy pixel (this block)
  constexpr int NDC = DYROWS * 64 * NCD;
  constexpr int NITX = (NXC + 511) / 512, NITD = (NDC + 511) / 512;
  __shared__ uint4 ximg[2][NROW * PWX * 4];
  __shared__ uint4 dyimg[2][NP * 32 * 4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5;
  const int mb = blockIdx.y / NBLK, nb = blockIdx.y % NBLK;
  const int a = wave / TG, tgp = wave % TG;
  const int g[3] = {p.g0, p.g1, p.g2};
  const int gl = g[ND - 1];
  for (int e = threadIdx.x; e < 2 * NP * 32 * 4; e += 512) (&dyimg[0][0])[e] = make_uint4(0, 0, 0, 0);   // (padding columns at COUT = 16 stay zero)
  f32x16 acc[TPW];
#pragma unroll
  for (int k = 0; k < TPW; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  float dbsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbsum[j] = 0.f;
  uint4 sx[NITX], sd[NITD];
  auto fetch = [&](int tile) {
    int rest = tile / p.tiles_per_row;
    const int l0 = 32 * (tile - rest * p.tiles_per_row);
    int lead[3] = {0, 0, 0};
#pragma unroll
    for (int ax = ND - 2; ax >= 0; --ax) {
      lead[ax] = rest % g[ax];
      rest /= g[ax];
    }
    const int b = rest;
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int e = threadIdx.x + 512 * it;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < NXC) {
        const int c = e & 3, pp = (e >> 2) % PWX, r = (e >> 2) / PWX;
        bool inb = true;
        long long off = b;
        if (ND == 3) {
          const int s0 = lead[0] + r / 3 - 1, s1 = lead[1] + r % 3 - 1;
          inb = s0 >= 0 && s0 < g[0] && s1 >= 0 && s1 < g[1];
          off = (off * g[0] + s0) * g[1] + s1;
        } else if (ND == 2) {
          const int s0 = lead[0] + r - 1;
          inb = s0 >= 0 && s0 < g[0];
          off = off * g[0] + s0;
        }
        const int sl = l0 - 1 + pp;
        if (inb && sl >= 0 && sl < gl) v = reinterpret_cast<const uint4*>(p.x + (off * gl + sl) * CIN + 32 * mb)[c];
      }
      sx[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int e = threadIdx.x + 512 * it;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < NDC) {
        const int c = e % NCD, d = (e / NCD) & 63, rr = e / (NCD * 64);
        long long off = b;
        if (ND == 3) off = (off * (2 * g[0]) + 2 * lead[0] + (rr >> 1)) * (2 * g[1]) + 2 * lead[1] + (rr & 1);
        else if (ND == 2) off = off * (2 * g[0]) + 2 * lead[0] + rr;
        const int sl = 2 * l0 + d;
        if (sl < 2 * gl) v = reinterpret_cast<const uint4*>(p.dy + (off * (2 * gl) + sl) * COUT + 32 * nb)[c];
      }
      sd[it] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int e = threadIdx.x + 512 * it;
      if (e < NXC) ximg[buf][e] = sx[it];
    }
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int e = threadIdx.x + 512 * it;
      if (e < NDC) {
        const int c = e % NCD, d = (e / NCD) & 63, rr = e / (NCD * 64);
        dyimg[buf][((rr * 2 + (d & 1)) * 32 + (d >> 1)) * 4 + c] = sd[it];
        if (mb == 0) {
          Frag f;
          f.u = sd[it];
#pragma unroll
          for (int j = 0; j < 8; ++j) dbsum[j] += (float)f.v[j];
        }
      }
    }
  };
  // tile n + 1 travels from global memory while tile n is multiplied; one barrier per tile (two image buffers)
  int tile = blockIdx.x, buf = 0;
  if (tile < p.n_tiles) fetch(tile);
  __syncthreads();                                 // zero-fill of the dy images is complete
  if (tile < p.n_tiles) stash(0);
  __syncthreads();
  for (; tile < p.n_tiles; tile += gridDim.x, buf ^= 1) {
    const int next = tile + gridDim.x;
    if (next < p.n_tiles) fetch(next);
    if (wave < NWAVE) {
      const __bf16* dyi = reinterpret_cast<const __bf16*>(&dyimg[buf][0]) + a * 32 * 32;
      const __bf16* xi = reinterpret_cast<const __bf16*>(&ximg[buf][0]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 bop = tr_read_plain(dyi, ks, lane);
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          const int t = tgp * TPW + k;
          int r = 0;
          if (ND == 3) r = (((a >> 2) & 1) + ((t >> 2) & 1)) * 3 + (((a >> 1) & 1) + ((t >> 1) & 1));
          if (ND == 2) r = ((a >> 1) & 1) + ((t >> 1) & 1);
          const int sl = (a & 1) + (t & 1);
          const bf16x8 aop = tr_read_plain(xi + (r * PWX + sl) * 32, ks, lane);
          acc[k] = mfma16(aop, bop, acc[k]);
        }
      }
    }
    if (next < p.n_tiles) stash(buf ^ 1);          // (buffer buf ^ 1 was last read before the previous barrier)
    __syncthreads();
  }
  // ---- slabs -----------------------------------------------------------------------------------------------------------
  if (wave < NWAVE) {
    float* slab = p.partial + ((long long)blockIdx.x * gridDim.y + blockIdx.y) * (NP * NP) * 1024;
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int combo = a * NP + tgp * TPW + k;
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[(combo * 32 + rho(r, h)) * 32 + (lane & 31)] = acc[k][r];
    }
  }
  if (mb == 0) {       // bias partials: thread e owns the channels 8 (e % NCD) .. + 7 of this block in every chunk it staged
    __shared__ float red_sm[512 * 8];
#pragma unroll
    for (int j = 0; j < 8; ++j) red_sm[threadIdx.x * 8 + j] = dbsum[j];
    __syncthreads();
    if (threadIdx.x < 8 * NCD) {
      const int c = threadIdx.x / 8, j = threadIdx.x % 8;
      float sacc = 0.f;
      for (int t = c; t < 512; t += NCD) sacc += red_sm[t * 8 + j];          // fixed order: deterministic
      if (8 * c + j < (COUT < 32 ? COUT : 32)) p.bias_part[(long long)blockIdx.x * COUT + 32 * nb + 8 * c + j] = sacc;
    }
  }
}

// 1-D grids (audio, protein): a tile of the kernel above is two MFMAs per wave between two barriers, four of the eight waves
// have no (phase, tap) to work on, and every (mb, nb) block re-reads its operands -- at a rank's shard of the audio preset
// (61 440 INRs) that ran at 1 TB/s.  Here a workgroup owns ALL of dWeff (4 combos x 64 x COUT: wave = (combo, input block),
// both output blocks in its accumulators), so x and dy are read from memory exactly once, in whole 128- / 32-byte pixels;
// TB consecutive tiles are staged per barrier, and the small footprint (two image sets) lets two or three workgroups share a
// CU so that one's staging overlaps another's loads.  Same slab layout as above: the sum and the fold are shared.
template <int COUT, int TB>
__global__ void __launch_bounds__(512) pc_wgrad1d_kernel(WgArgs p) {
  constexpr int NBLK = (COUT + 31) / 32, PWX = 34, NC = COUT / 8;           // NC: 16-byte chunks of a dy pixel
  constexpr int XT = 2 * PWX * 4, DT = 2 * NBLK * 32 * 4;                    // uint4 per tile: x [mb][34][4], dy [phase][nb][32][4]
  constexpr int XPT = PWX * 8, DPT = 64 * NC;                                // chunks fetched per tile
  constexpr int NXC = TB * XPT, NDC = TB * DPT;
  constexpr int NITX = (NXC + 511) / 512, NITD = (NDC + 511) / 512;
  static_assert(512 % NC == 0, "a thread stages the same channels of dy in every pass (bias sums)");
  extern __shared__ uint4 wg1d_smem[];
  uint4* ximg = wg1d_smem;                     // [2][TB][XT]
  uint4* dyimg = wg1d_smem + 2 * TB * XT;      // [2][TB][DT]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5;
  const int combo = wave >> 1, a = combo >> 1, t = combo & 1, mb = wave & 1;
  const int gl = p.g0;
  for (int e = threadIdx.x; e < 2 * TB * DT; e += 512) dyimg[e] = make_uint4(0, 0, 0, 0);     // (padding columns at COUT = 16 stay zero)
  f32x16 acc[NBLK];
#pragma unroll
  for (int k = 0; k < NBLK; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  float dbsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbsum[j] = 0.f;
  uint4 sx[NITX], sd[NITD];
  auto fetch = [&](int st) {
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int e = threadIdx.x + 512 * it;
      uint4 v = make_uint4(0, 0, 0, 0);
      const int tt = e / XPT, r = e - tt * XPT, tile = st * TB + tt;
      if (e < NXC && tile < p.n_tiles) {
        const int b = tile / p.tiles_per_row, sl = 32 * (tile - b * p.tiles_per_row) - 1 + (r >> 3);
        if (sl >= 0 && sl < gl) v = reinterpret_cast<const uint4*>(p.x + ((long long)b * gl + sl) * CIN)[r & 7];
      }
      sx[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int e = threadIdx.x + 512 * it;
      uint4 v = make_uint4(0, 0, 0, 0);
      const int tt = e / DPT, r = e - tt * DPT, tile = st * TB + tt;
      if (e < NDC && tile < p.n_tiles) {
        const int b = tile / p.tiles_per_row, sl = 64 * (tile - b * p.tiles_per_row) + r / NC;
        if (sl < 2 * gl) v = reinterpret_cast<const uint4*>(p.dy + ((long long)b * 2 * gl + sl) * COUT)[r % NC];
      }
      sd[it] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int e = threadIdx.x + 512 * it;
      const int tt = e / XPT, r = e - tt * XPT, pp = r >> 3, c8 = r & 7;
      if (e < NXC) ximg[(buf * TB + tt) * XT + ((c8 >> 2) * PWX + pp) * 4 + (c8 & 3)] = sx[it];
    }
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int e = threadIdx.x + 512 * it;
      const int tt = e / DPT, r = e - tt * DPT, d = r / NC, c = r % NC;
      if (e < NDC) {
        dyimg[(buf * TB + tt) * DT + (((d & 1) * NBLK + (c >> 2)) * 32 + (d >> 1)) * 4 + (c & 3)] = sd[it];
        Frag f;
        f.u = sd[it];
#pragma unroll
        for (int j = 0; j < 8; ++j) dbsum[j] += (float)f.v[j];
      }
    }
  };
  const int n_st = (p.n_tiles + TB - 1) / TB;
  int st = blockIdx.x, buf = 0;
  if (st < n_st) fetch(st);
  __syncthreads();                                 // zero-fill of the dy images is complete
  if (st < n_st) stash(0);
  __syncthreads();
  for (; st < n_st; st += gridDim.x, buf ^= 1) {
    const int next = st + gridDim.x;
    if (next < n_st) fetch(next);
#pragma unroll
    for (int tt = 0; tt < TB; ++tt) {
      const __bf16* xi = reinterpret_cast<const __bf16*>(ximg + (buf * TB + tt) * XT + mb * PWX * 4) + (a + t) * 32;
      const __bf16* dyi = reinterpret_cast<const __bf16*>(dyimg + (buf * TB + tt) * DT + a * NBLK * 128);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 aop = tr_read_plain(xi, ks, lane);
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) acc[nb] = mfma16(aop, tr_read_plain(dyi + nb * 1024, ks, lane), acc[nb]);
      }
    }
    if (next < n_st) stash(buf ^ 1);               // (buffer buf ^ 1 was last read before the previous barrier)
    __syncthreads();
  }
  float* slab = p.partial + (long long)blockIdx.x * (2 * NBLK * 4 * 1024);
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[(((mb * NBLK + nb) * 4 + combo) * 32 + rho(r, h)) * 32 + (lane & 31)] = acc[nb][r];
  // bias partials: thread e stages the channels 8 (e % NC) .. + 7 in every chunk of dy it handles
  float* red_sm = reinterpret_cast<float*>(wg1d_smem);
#pragma unroll
  for (int j = 0; j < 8; ++j) red_sm[threadIdx.x * 8 + j] = dbsum[j];
  __syncthreads();
  if (threadIdx.x < 8 * NC) {
    const int c = threadIdx.x / 8, j = threadIdx.x % 8;
    float sacc = 0.f;
    for (int th = c; th < 512; th += NC) sacc += red_sm[th * 8 + j];          // fixed order: deterministic
    p.bias_part[(long long)blockIdx.x * COUT + 8 * c + j] = sacc;
  }
}
template <int COUT, int TB>
constexpr int wg1d_lds_bytes() {
  return 2 * TB * (2 * 34 * 4 + 2 * ((COUT + 31) / 32) * 32 * 4) * 16;
}

// 2-D grids (stitched photos): the same idea.  A workgroup owns all of dWeff (16 combos x 64 x COUT: wave = (phase, tap pair),
// both input blocks and all output blocks in its accumulators), so dy is read from memory once and x in whole pixels; a pass
// covers TB VERTICALLY adjacent tiles, which share their source rows (TB + 2 instead of 3 TB rows staged), and costs one barrier.
template <int COUT, int TB>
__global__ void __launch_bounds__(512) pc_wgrad2d_kernel(WgArgs p) {
  constexpr int NBLK = (COUT + 31) / 32, PWX = 34, NC = COUT / 8, XR = TB + 2;
  constexpr int XT = 2 * XR * PWX * 4;                         // uint4 of the x images of a pass: [mb][row][34][4]
  constexpr int DT = 4 * NBLK * 32 * 4;                        // dy images of one tile: [phase][nb][32][4]
  constexpr int NXC = XR * PWX * 8, DPT = 2 * 64 * NC, NDC = TB * DPT;
  constexpr int NITX = (NXC + 511) / 512, NITD = (NDC + 511) / 512;
  static_assert(512 % NC == 0, "a thread stages the same channels of dy in every pass (bias sums)");
  extern __shared__ uint4 wg2d_smem[];
  uint4* ximg = wg2d_smem;                     // [2][XT]
  uint4* dyimg = wg2d_smem + 2 * XT;           // [2][TB][DT]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5;
  const int a = wave >> 1, tgp = wave & 1, ay = a >> 1, ax = a & 1;
  const int g0 = p.g0, gl = p.g1;
  const int n_rb = (g0 + TB - 1) / TB, per_b = n_rb * p.tiles_per_row;
  const int n_st = p.B * per_b;
  for (int e = threadIdx.x; e < 2 * TB * DT; e += 512) dyimg[e] = make_uint4(0, 0, 0, 0);     // (padding columns at COUT = 16 stay zero)
  f32x16 acc[2][2][NBLK];                      // [tap of the pair][input block][output block]
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NBLK; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][m][n][r] = 0.f;
  float dbsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbsum[j] = 0.f;
  uint4 sx[NITX], sd[NITD];
  auto fetch = [&](int st) {
    const int b = st / per_b, rem = st - b * per_b, rb = rem / p.tiles_per_row, l0 = 32 * (rem - rb * p.tiles_per_row);
    const int r0 = rb * TB;
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int e = threadIdx.x + 512 * it;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < NXC) {
        const int c8 = e & 7, pp = (e >> 3) % PWX, j = (e >> 3) / PWX;
        const int s0 = r0 - 1 + j, sl = l0 - 1 + pp;
        if (s0 >= 0 && s0 < g0 && sl >= 0 && sl < gl) v = reinterpret_cast<const uint4*>(p.x + (((long long)b * g0 + s0) * gl + sl) * CIN)[c8];
      }
      sx[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int e = threadIdx.x + 512 * it;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < NDC) {
        const int tt = e / DPT, r = e - tt * DPT, c = r % NC, d = (r / NC) & 63, rr = r / (NC * 64);
        const int row = r0 + tt, sl = 2 * l0 + d;
        if (row < g0 && sl < 2 * gl)
          v = reinterpret_cast<const uint4*>(p.dy + (((long long)b * 2 * g0 + 2 * row + rr) * (2 * gl) + sl) * COUT)[c];
      }
      sd[it] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int e = threadIdx.x + 512 * it;
      if (e < NXC) {
        const int c8 = e & 7, pix = e >> 3;                   // pix = row * 34 + pixel
        ximg[buf * XT + ((c8 >> 2) * XR * PWX + pix) * 4 + (c8 & 3)] = sx[it];
      }
    }
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int e = threadIdx.x + 512 * it;
      if (e < NDC) {
        const int tt = e / DPT, r = e - tt * DPT, c = r % NC, d = (r / NC) & 63, rr = r / (NC * 64);
        dyimg[(buf * TB + tt) * DT + (((rr * 2 + (d & 1)) * NBLK + (c >> 2)) * 32 + (d >> 1)) * 4 + (c & 3)] = sd[it];
        Frag f;
        f.u = sd[it];
#pragma unroll
        for (int j = 0; j < 8; ++j) dbsum[j] += (float)f.v[j];
      }
    }
  };
  int st = blockIdx.x, buf = 0;
  if (st < n_st) fetch(st);
  __syncthreads();                                 // zero-fill of the dy images is complete
  if (st < n_st) stash(0);
  __syncthreads();
  for (; st < n_st; st += gridDim.x, buf ^= 1) {
    const int next = st + gridDim.x;
    if (next < n_st) fetch(next);
#pragma unroll
    for (int tt = 0; tt < TB; ++tt) {
      const __bf16* dyi = reinterpret_cast<const __bf16*>(dyimg + (buf * TB + tt) * DT + a * NBLK * 128);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 bop[NBLK];
#pragma unroll
        for (int n = 0; n < NBLK; ++n) bop[n] = tr_read_plain(dyi + n * 1024, ks, lane);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int t = tgp * 2 + k, row = tt + ay + (t >> 1), sl = ax + (t & 1);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const __bf16* xi = reinterpret_cast<const __bf16*>(ximg + buf * XT + ((m * XR + row) * PWX + sl) * 4);
            const bf16x8 aop = tr_read_plain(xi, ks, lane);
#pragma unroll
            for (int n = 0; n < NBLK; ++n) acc[k][m][n] = mfma16(aop, bop[n], acc[k][m][n]);
          }
        }
      }
    }
    if (next < n_st) stash(buf ^ 1);               // (buffer buf ^ 1 was last read before the previous barrier)
    __syncthreads();
  }
  float* slab = p.partial + (long long)blockIdx.x * (2 * NBLK * 16 * 1024);
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NBLK; ++n) {
        const int combo = a * 4 + tgp * 2 + k;
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(((m * NBLK + n) * 16 + combo) * 32 + rho(r, h)) * 32 + (lane & 31)] = acc[k][m][n][r];
      }
  float* red_sm = reinterpret_cast<float*>(wg2d_smem);
#pragma unroll
  for (int j = 0; j < 8; ++j) red_sm[threadIdx.x * 8 + j] = dbsum[j];
  __syncthreads();
  if (threadIdx.x < 8 * NC) {
    const int c = threadIdx.x / 8, j = threadIdx.x % 8;
    float sacc = 0.f;
    for (int th = c; th < 512; th += NC) sacc += red_sm[th * 8 + j];          // fixed order: deterministic
    p.bias_part[(long long)blockIdx.x * COUT + 8 * c + j] = sacc;
  }
}
template <int COUT, int TB>
constexpr int wg2d_lds_bytes() {
  return 2 * (2 * (TB + 2) * 34 * 4 + TB * 4 * ((COUT + 31) / 32) * 32 * 4) * 16;
}

// sum of the slabs in a fixed order (one thread per element of dWeff, coalesced over the slabs) ...
__global__ void __launch_bounds__(256) pc_wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias_part,
                                                              int n_slabs, long long slab_floats, int cout,
                                                              float* __restrict__ weff_sum, float* __restrict__ db) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e < slab_floats) {
    // sixteen slabs in flight per thread (the sum over 128 slabs is otherwise a chain of dependent L2 round trips); the
    // association is fixed: slab g goes to accumulator g % 16, the accumulators are added pairwise in index order
    float sa[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) sa[k] = 0.f;
    int gI = 0;
    for (; gI + 15 < n_slabs; gI += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) sa[k] += partial[(long long)(gI + k) * slab_floats + e];
    }
    for (int k = 0; gI < n_slabs; ++gI, ++k) sa[k] += partial[(long long)gI * slab_floats + e];
#pragma unroll
    for (int st = 8; st >= 1; st >>= 1)
#pragma unroll
      for (int k = 0; k < st; ++k) sa[k] += sa[k + st];
    weff_sum[e] = sa[0];
  }
  // bias: 256 / cout threads per channel, each with eight loads in flight (a single chain over 512 partials cost ~100 us);
  // fixed association: partial g -> (thread g % P, accumulator (g / P) % 8), accumulators pairwise, threads in index order
  __shared__ float bias_sm[256];
  if (blockIdx.x == 0) {
    const int P = 256 / cout, part = threadIdx.x / cout, ch = threadIdx.x % cout;       // cout = 16 or 64
    float sb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sb[k] = 0.f;
    for (int g0 = part; g0 < n_slabs; g0 += 8 * P) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int gI = g0 + k * P;
        if (gI < n_slabs) sb[k] += bias_part[(long long)gI * cout + ch];
      }
    }
#pragma unroll
    for (int st = 4; st >= 1; st >>= 1)
#pragma unroll
      for (int k = 0; k < st; ++k) sb[k] += sb[k + st];
    bias_sm[threadIdx.x] = sb[0];
    __syncthreads();
    if (threadIdx.x < cout) {
      float s2 = 0.f;
      for (int q = 0; q < P; ++q) s2 += bias_sm[q * cout + threadIdx.x];
      db[threadIdx.x] = s2;
    }
  }
}

// ... and the fold of the phase weights back onto the conv taps (transpose of the tap sums of the forward pass):
//   dW[co][ci][k...] = sum over phases a of dWeff[a][t(a, k)]      (tap k of phase a reads source tap t(a, k))
template <int ND>
__global__ void __launch_bounds__(256) pc_wgrad_fold_kernel(const float* __restrict__ weff_sum, int cout, float* __restrict__ dW) {
  constexpr int NP = 1 << ND, KK = ND == 1 ? 3 : (ND == 2 ? 9 : 27);
  const int nblk = (cout + 31) / 32;
  const int total = cout * CIN * KK;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int kk = e % KK, ci = (e / KK) % CIN, co = e / (KK * CIN);
  const int y = (ci >> 5) * nblk + (co >> 5);
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < NP; ++a) {
    int t = 0, rem = kk;
#pragma unroll
    for (int ax = ND - 1; ax >= 0; --ax) {
      const int k = rem % 3;
      rem /= 3;
      const int ak = (a >> (ND - 1 - ax)) & 1;
      const int tk = ak == 0 ? (k == 0 ? 0 : 1) : (k == 2 ? 1 : 0);
      t |= tk << (ND - 1 - ax);
    }
    s += weff_sum[(((long long)y * (NP * NP) + a * NP + t) * 32 + (ci & 31)) * 32 + (co & 31)];
  }
  dW[e] = s;
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
    if (fwd_frags) pc_pack_fwd_kernel<NDv><<<cdiv(rcb_phaseconv_pack_uint4(NDv, cout, 0), 256), 256, 0, s>>>(conv_weight, cout, static_cast<uint4*>(fwd_frags));       \
    if (dgrad_frags) pc_pack_dgrad_kernel<NDv><<<cdiv(rcb_phaseconv_pack_uint4(NDv, cout, 1), 256), 256, 0, s>>>(conv_weight, cout, static_cast<uint4*>(dgrad_frags)); \
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
  const int tpb = 8 >> nd;                            // tiles per workgroup (8 waves = all phases of tpb tiles of one row)
  const long long groups = (long long)(p.n_tiles / p.tiles_per_row) * ((p.tiles_per_row + tpb - 1) / tpb);
  int gx = (int)(groups < 1024 ? groups : 1024);      // waves keep their fragments across the groups they walk
  gx = (gx + 7) / 8 * 8;                              // (the kernel pairs workgroups 8 apart: see its comment)
  dim3 grid(gx * ((cout + 31) / 32));
  const size_t lds = (size_t)(nd == 1 ? 1 : (nd == 2 ? 3 : 9)) * (32 * tpb + 2) * 128;
  hipStream_t s = (hipStream_t)stream;
#define RCB_FWD(NDv, Cv)                                                                  \
  if (nd == NDv && cout == Cv) {                                                          \
    if (leaky_out) pc_fwd_kernel<NDv, Cv, 1><<<grid, 512, lds, s>>>(p);                   \
    else pc_fwd_kernel<NDv, Cv, 0><<<grid, 512, lds, s>>>(p);                             \
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
  const int groups = (p.n_tiles + 3) / 4;             // four consecutive tiles per workgroup
  int gx = groups < 4096 ? groups : 4096;
  hipStream_t s = (hipStream_t)stream;
  // RCB_PC_DGRAD64=0: the 16-channel-block kernel also at 64 channels (same-box A/B)
  static const bool whole_pixels = [] { const char* e = getenv("RCB_PC_DGRAD64"); return !(e && e[0] == '0'); }();
#define RCB_DG(NDv, Cv) \
  if (nd == NDv && cout == Cv) pc_dgrad_kernel<NDv, Cv><<<gx, 256, 0, s>>>(p);
  if (cout == 64 && whole_pixels) {
    if (nd == 1) pc_dgrad64_kernel<1><<<gx, 256, 0, s>>>(p);
    if (nd == 2) pc_dgrad64_kernel<2><<<gx, 256, 0, s>>>(p);
    if (nd == 3) pc_dgrad64_kernel<3><<<gx, 256, 0, s>>>(p);
  } else {
    RCB_DG(1, 64) RCB_DG(1, 16) RCB_DG(2, 64) RCB_DG(2, 16) RCB_DG(3, 64) RCB_DG(3, 16)
  }
#undef RCB_DG
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

// workgroups per (mb, nb) block of the weight gradient = slabs to add.  The 16-channel stage needs the 128 for its parallelism
// (64: 95 -> 157 us on the video grid); the 64-channel stage has four times the (mb, nb) blocks and a 1 MB slab, whose sum
// over the slabs is what costs (128 slabs = 134 MB: 30 us): 64 there (wgrad 47 -> 39 us, sum 30 -> 15 us)
constexpr int kWgSlotsMax = 128;
__host__ constexpr int wg_slots(int cout) { return cout == 64 ? 64 : 128; }
// 1-D grids (pc_wgrad1d_kernel): one 64 KB / 32 KB slab per workgroup, two workgroups per CU
constexpr int kWg1dSlots = 512;
constexpr int kWg1dTiles64 = 2, kWg1dTiles16 = 4;      // tiles staged per barrier (COUT = 64 / 16)
// 2-D grids (pc_wgrad2d_kernel): one workgroup per CU (100 / 118 KB of images), slabs of 256 / 128 KB
constexpr int kWg2dSlots = 256;
constexpr int kWg2dTiles64 = 2, kWg2dTiles16 = 4;      // vertically adjacent tiles per pass

extern "C" int64_t rcb_phaseconv_wgrad_workspace(int32_t nd, int32_t cout) {
  if (nd < 1 || nd > 3 || (cout != 16 && cout != 64)) return -1;
  const int np = 1 << nd, ny = 2 * ((cout + 31) / 32);
  const int64_t slab = (int64_t)ny * np * np * 1024;
  return (int64_t)(nd == 1 ? kWg1dSlots : (nd == 2 ? kWg2dSlots : kWgSlotsMax)) * (slab + cout) + slab;
}

extern "C" int rcb_phaseconv_wgrad(const void* x_act, const void* dy, float* dW, float* dbias, float* workspace,
                                   int64_t workspace_floats, int32_t B, int32_t g0, int32_t g1, int32_t g2, int32_t nd,
                                   int32_t cout, rcb_stream_t stream) {
  RCB_REQUIRE(x_act && dy && dW && dbias && workspace, RCB_ERR_ARG, "phaseconv_wgrad: null pointer");
  PcArgs pc;
  memset(&pc, 0, sizeof(pc));
  int rc = check_pc("phaseconv_wgrad", nd, B, g0, g1, g2, cout, pc);
  if (rc) return rc;
  RCB_REQUIRE(workspace_floats >= rcb_phaseconv_wgrad_workspace(nd, cout), RCB_ERR_SHAPE, "phaseconv_wgrad: workspace too small");
  const int np = 1 << nd, ny = 2 * ((cout + 31) / 32);
  const long long slab = (long long)ny * np * np * 1024;
  const int tb1 = cout == 64 ? kWg1dTiles64 : kWg1dTiles16, tb2 = cout == 64 ? kWg2dTiles64 : kWg2dTiles16;
  const int kWgSlots = nd == 1 ? kWg1dSlots : (nd == 2 ? kWg2dSlots : wg_slots(cout));
  const int units = nd == 1 ? cdiv(pc.n_tiles, tb1) : (nd == 2 ? B * cdiv(g0, tb2) * pc.tiles_per_row : pc.n_tiles);
  int gx = units < kWgSlots ? units : kWgSlots;
  if (nd <= 2 && gx > cdiv(units, 4)) gx = cdiv(units, 4);      // at least four passes per workgroup: its slab is 64 ... 256 KB
  WgArgs w;
  w.x = static_cast<const __bf16*>(x_act);
  w.dy = static_cast<const __bf16*>(dy);
  w.partial = workspace;
  w.bias_part = workspace + (long long)kWgSlots * slab;
  float* weff_sum = w.bias_part + (long long)kWgSlots * cout;
  w.B = B;
  w.g0 = g0;
  w.g1 = g1;
  w.g2 = g2;
  w.tiles_per_row = pc.tiles_per_row;
  w.n_tiles = pc.n_tiles;
  dim3 grid(gx, ny);
  hipStream_t s = (hipStream_t)stream;
  const int kk = nd == 1 ? 3 : (nd == 2 ? 9 : 27);
  const int fin = cdiv((long long)cout * CIN * kk, 256);
#define RCB_WG(NDv, Cv)                                                                                              \
  if (nd == NDv && cout == Cv) {                                                                                     \
    pc_wgrad_kernel<NDv, Cv><<<grid, 512, 0, s>>>(w);                                                                \
    pc_wgrad_reduce_kernel<<<cdiv(slab, 256), 256, 0, s>>>(w.partial, w.bias_part, gx, slab, cout, weff_sum, dbias); \
    pc_wgrad_fold_kernel<NDv><<<fin, 256, 0, s>>>(weff_sum, cout, dW);                                               \
  }
#define RCB_WG1(Cv, TBv)                                                                                             \
  if (nd == 1 && cout == Cv) {                                                                                       \
    constexpr int lds = wg1d_lds_bytes<Cv, TBv>();                                                                   \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pc_wgrad1d_kernel<Cv, TBv>),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);                             \
    RCB_REQUIRE(e == hipSuccess, (int)e, "phaseconv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));          \
    pc_wgrad1d_kernel<Cv, TBv><<<gx, 512, lds, s>>>(w);                                                              \
    pc_wgrad_reduce_kernel<<<cdiv(slab, 256), 256, 0, s>>>(w.partial, w.bias_part, gx, slab, cout, weff_sum, dbias); \
    pc_wgrad_fold_kernel<1><<<fin, 256, 0, s>>>(weff_sum, cout, dW);                                                 \
  }
  RCB_WG1(64, kWg1dTiles64) RCB_WG1(16, kWg1dTiles16)
#define RCB_WG2(Cv, TBv)                                                                                             \
  if (nd == 2 && cout == Cv) {                                                                                       \
    constexpr int lds = wg2d_lds_bytes<Cv, TBv>();                                                                   \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pc_wgrad2d_kernel<Cv, TBv>),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);                             \
    RCB_REQUIRE(e == hipSuccess, (int)e, "phaseconv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));          \
    pc_wgrad2d_kernel<Cv, TBv><<<gx, 512, lds, s>>>(w);                                                              \
    pc_wgrad_reduce_kernel<<<cdiv(slab, 256), 256, 0, s>>>(w.partial, w.bias_part, gx, slab, cout, weff_sum, dbias); \
    pc_wgrad_fold_kernel<2><<<fin, 256, 0, s>>>(weff_sum, cout, dW);                                                 \
  }
  RCB_WG2(64, kWg2dTiles64) RCB_WG2(16, kWg2dTiles16)
  RCB_WG(3, 64) RCB_WG(3, 16)
#undef RCB_WG2
#undef RCB_WG1
#undef RCB_WG
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
