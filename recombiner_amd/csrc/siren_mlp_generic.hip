// fp32 SIREN for the hidden widths the MFMA kernels do not cover (fp32 parity mode at widths other than 32, up to 64):
// the reference builds its INR from any `hidden_dims` (prior_model.py:84-85, 129-179); this kernel keeps the fp32 mode
// available for them, so that the width-48 / width-64 presets can be held to the reference's goldens at fp32 tolerance
// rather than within 16-bit operand rounding.
//
// Deliberately simple: plain fp32 FMAs, no matrix cores (a parity path, not a fast one).  One 256-thread workgroup per
// (INR, sample) walks tiles of 32 pixels; thread (p, og) = (tid & 31, tid >> 5) computes features og, og + 8, ... of pixel p.
// The tile's activations (sin, cos of every hidden layer) and the running dZ live in LDS.  Every thread OWNS entries
// tid, tid + 256, ... of each layer vector and accumulates their gradient over all pixels in registers, pixels in ascending
// order: complete sums, no reduction across threads, deterministic.
#include "rcb_common.h"

using namespace rcb;

#include "siren_common.h"

namespace {

// as in siren_mlp.hip (exact variant): sin / cos of w0 z with a two-float range reduction in revolutions
__device__ __forceinline__ void sincos_w0_exact(float z, float k_hi, float k_lo, float& s, float& c) {
  float th = z * k_hi;
  float tl = __builtin_fmaf(z, k_hi, -th) + z * k_lo;
  float fr = (th - rintf(th)) + tl;
  sincospif(2.0f * fr, &s, &c);
}

constexpr int GEN_MAXW = 64;                                      // hidden width limit
constexpr int GEN_J = (GEN_MAXW * (GEN_MAXW + 1) + 255) / 256;    // gradient entries per thread and layer

struct GenGeo {
  int W, NH, in0, dnet;        // W = the widest hidden layer
  int wid[MAXL];               // output width of every layer (hidden widths, then C)
  int off[MAXL + 1];
  // LDS map (floats)
  int x0, x0s;                 // inputs [32][x0s]
  int s_base, c_base, ws;      // sin / cos [NH][32][ws]
  int dza, dzb;                // [32][ws]
  int red;                     // [256]
  int total;
};

template <int MODE>
__global__ void __launch_bounds__(256) siren_generic_kernel(SirenArgs a, GenGeo geo) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, p = tid & 31, og = tid >> 5;
  const int g = blockIdx.x, n = g / a.S;
  const int W = geo.W, NH = geo.NH, NL = NH + 1, in0 = geo.in0, F = a.F, E = a.E, C = a.C, P = a.P;
  float* wl = smem;
  float* X0 = smem + geo.x0;
  float* Sb = smem + geo.s_base;
  float* Cb = smem + geo.c_base;
  float* dz_cur = smem + geo.dza;
  float* dz_nxt = smem + geo.dzb;
  float* red = smem + geo.red;
  const int ws = geo.ws, x0s = geo.x0s;
  auto lin = [&](int l) { return l == 0 ? in0 : geo.wid[l - 1]; };
  auto lout = [&](int l) { return geo.wid[l]; };

  {
    const float* src = a.wvec + (long long)g * a.w_stride;
    for (int i = tid; i < geo.dnet; i += 256) wl[i] = src[i];
  }
  float gacc[MAXL][GEN_J];
#pragma unroll
  for (int l = 0; l < MAXL; ++l)
#pragma unroll
    for (int j = 0; j < GEN_J; ++j) gacc[l][j] = 0.f;
  float sse_t = 0.f;
  __syncthreads();

  const int ntiles = (P + 31) >> 5;
  for (int t = 0; t < ntiles; ++t) {
    const int pix = t * 32 + p;
    const bool valid = pix < P;
    const int pc = valid ? pix : P - 1;
    // ---- inputs -> LDS (thread (p, og) copies features og, og + 8, ...) -------------------------------------------------------
    for (int i = og; i < in0; i += 8)
      X0[p * x0s + i] = i < F ? a.xf[(long long)n * a.xf_stride + (long long)pc * F + i] : a.pe[((long long)g * P + pc) * E + (i - F)];
    __syncthreads();
    // ---- forward (layer loops are unrolled over MAXL: the gradient registers are indexed statically) ------------------------
#pragma unroll
    for (int l = 0; l < MAXL - 1; ++l) {
      if (l >= NH) continue;
      const float* prev = l == 0 ? X0 + p * x0s : Sb + ((l - 1) * 32 + p) * ws;
      const float* Bl = wl + geo.off[l];
      const int ni = lin(l), no = lout(l);
      const float* Wl = Bl + no;
      float acc[GEN_MAXW / 8];
#pragma unroll
      for (int k = 0; k < GEN_MAXW / 8; ++k) acc[k] = (og + 8 * k < no) ? Bl[og + 8 * k] : 0.f;
      for (int i = 0; i < ni; ++i) {
        const float s = prev[i];
#pragma unroll
        for (int k = 0; k < GEN_MAXW / 8; ++k)
          if (og + 8 * k < no) acc[k] = __builtin_fmaf(Wl[i * no + og + 8 * k], s, acc[k]);
      }
#pragma unroll
      for (int k = 0; k < GEN_MAXW / 8; ++k) {
        const int o = og + 8 * k;
        if (o < no) {
          float s, c;
          sincos_w0_exact(acc[k], a.k_hi, a.k_lo, s, c);
          Sb[(l * 32 + p) * ws + o] = s;
          Cb[(l * 32 + p) * ws + o] = c;
        }
      }
      __syncthreads();
    }
    {
      // output layer: features og, og + 8, ... < C
      const float* prev = Sb + ((NH - 1) * 32 + p) * ws;
      const float* Bl = wl + geo.off[NH];
      const float* Wl = Bl + C;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int o = og + 8 * k;
        if (o < C) {
          float y = Bl[o];
          for (int i = 0; i < lin(NH); ++i) y = __builtin_fmaf(Wl[i * C + o], prev[i], y);
          if (MODE == MODE_FWD) {
            if (valid) a.yout[((long long)g * P + pix) * C + o] = y;
          } else if (MODE == MODE_LOSS) {
            const float diff = valid ? y - a.yin[((long long)n * P + pc) * C + o] : 0.f;
            sse_t += diff * diff;
            dz_cur[p * ws + o] = 2.0f * a.dy_scale * diff;
          } else {
            dz_cur[p * ws + o] = valid ? a.yin[((long long)g * P + pc) * C + o] : 0.f;
          }
        }
      }
    }
    if (MODE == MODE_FWD) {
      __syncthreads();
      continue;
    }
    __syncthreads();
    // ---- backward ---------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int l = MAXL - 1; l >= 0; --l) {
      if (l >= NL) continue;
      const int ni = lin(l), no = lout(l);
      const float* prev = l == 0 ? X0 : Sb + (l - 1) * 32 * ws;        // [32][stride]
      const int ps = l == 0 ? x0s : ws;
      // weight gradient: this thread's entries of the layer vector [bias(no) | W(in, out)]
      const int size = no * (ni + 1);
#pragma unroll
      for (int j = 0; j < GEN_J; ++j) {
        const int e = tid + 256 * j;
        if (e < size) {
          float v = gacc[l][j];
          if (e < no) {
            for (int pp = 0; pp < 32; ++pp) v += dz_cur[pp * ws + e];
          } else {
            const int idx = e - no, i = idx / no, o = idx - i * no;
            for (int pp = 0; pp < 32; ++pp) v = __builtin_fmaf(prev[pp * ps + i], dz_cur[pp * ws + o], v);
          }
          gacc[l][j] = v;
        }
      }
      // data gradient
      const float* Wl = wl + geo.off[l] + no;
      if (l > 0) {
#pragma unroll
        for (int k = 0; k < GEN_MAXW / 8; ++k) {
          const int i = og + 8 * k;
          if (i < ni) {
            float dh = 0.f;
            for (int o = 0; o < no; ++o) dh = __builtin_fmaf(Wl[i * no + o], dz_cur[p * ws + o], dh);
            dz_nxt[p * ws + i] = dh * Cb[((l - 1) * 32 + p) * ws + i] * a.w0;
          }
        }
      } else if (a.dpe != nullptr) {
        for (int e = og; e < E; e += 8) {
          float dx = 0.f;
          for (int o = 0; o < no; ++o) dx = __builtin_fmaf(Wl[(F + e) * no + o], dz_cur[p * ws + o], dx);
          if (valid) a.dpe[((long long)g * P + pix) * E + e] = dx;
        }
      }
      __syncthreads();
      float* tmp = dz_cur;
      dz_cur = dz_nxt;
      dz_nxt = tmp;
    }
  }
  if (MODE == MODE_FWD) return;
  float* dst = a.dwvec + (long long)g * a.w_stride;
#pragma unroll
  for (int l = 0; l < MAXL; ++l) {
    if (l >= NL) continue;
    const int size = lout(l) * (lin(l) + 1);
#pragma unroll
    for (int j = 0; j < GEN_J; ++j) {
      const int e = tid + 256 * j;
      if (e < size) dst[geo.off[l] + e] = gacc[l][j];
    }
  }
  if (MODE == MODE_LOSS) {
    red[tid] = sse_t;
    __syncthreads();
    if (tid == 0) {
      float v = 0.f;
      for (int i = 0; i < 256; ++i) v += red[i];
      a.sse[g] = v;
    }
  }
}

template <int MODE>
int launch_generic(const SirenArgs& a, const GenGeo& geo, hipStream_t st) {
  auto kfn = siren_generic_kernel<MODE>;
  // per launch, not once per process: the attribute belongs to the (function, device) pair and a process may drive
  // several devices or call from several threads; the call is a table write
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return fail((int)e, "siren(generic): hipFuncSetAttribute: %s", hipGetErrorString(e));
  kfn<<<a.G, 256, (size_t)geo.total * sizeof(float), st>>>(a, geo);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

}  // namespace

namespace rcb {
int siren_generic_dispatch(int mode, const rcb_siren_desc* d, SirenArgs& a, hipStream_t st) {
  RCB_REQUIRE(d->precision == 0 && d->hidden >= 1 && d->hidden <= GEN_MAXW && d->n_hidden + 1 <= MAXL && d->out_dim <= 32 &&
                  d->pixel_chunks <= 1,
              RCB_ERR_UNSUPPORTED, "siren(generic fp32): hidden=%d n_hidden=%d out_dim=%d pixel_chunks=%d", d->hidden, d->n_hidden,
              d->out_dim, d->pixel_chunks);
  GenGeo geo;
  geo.NH = d->n_hidden;
  geo.in0 = d->fourier_dim + d->pe_dim;
  const bool listed = d->hidden_dims[0] != 0;
  geo.W = 0;
  for (int l = 0; l < geo.NH; ++l) {
    geo.wid[l] = listed ? d->hidden_dims[l] : d->hidden;
    RCB_REQUIRE(geo.wid[l] >= 1 && geo.wid[l] <= GEN_MAXW, RCB_ERR_UNSUPPORTED, "siren(generic fp32): hidden layer %d is %d wide (1..%d)", l,
                geo.wid[l], GEN_MAXW);
    geo.W = geo.wid[l] > geo.W ? geo.wid[l] : geo.W;
  }
  geo.wid[geo.NH] = d->out_dim;
  int o = 0;
  for (int l = 0; l <= geo.NH; ++l) {
    geo.off[l] = o;
    const int li = l == 0 ? geo.in0 : geo.wid[l - 1], lo = geo.wid[l];
    o += lo * (li + 1);
  }
  geo.dnet = o;
  a.dnet = o;
  geo.x0s = geo.in0 | 1;
  geo.ws = (geo.W > d->out_dim ? geo.W : d->out_dim) | 1;
  int f = (geo.dnet + 3) & ~3;
  geo.x0 = f;
  f += 32 * geo.x0s;
  geo.s_base = f;
  f += geo.NH * 32 * geo.ws;
  geo.c_base = f;
  f += geo.NH * 32 * geo.ws;
  geo.dza = f;
  f += 32 * geo.ws;
  geo.dzb = f;
  f += 32 * geo.ws;
  geo.red = f;
  f += 256;
  geo.total = f;
  RCB_REQUIRE((size_t)f * sizeof(float) <= 160 * 1024, RCB_ERR_UNSUPPORTED, "siren(generic fp32): %zu B of LDS needed",
              (size_t)f * sizeof(float));
  if (mode == MODE_FWD) return launch_generic<MODE_FWD>(a, geo, st);
  if (mode == MODE_BWD) return launch_generic<MODE_BWD>(a, geo, st);
  return launch_generic<MODE_LOSS>(a, geo, st);
}
}  // namespace rcb
