// K3 + K4: batched SIREN coordinate-MLP with per-INR weights, forward / backward / fused-loss.
//
// One 256-thread workgroup (4 waves) per (INR, sample).  The INR's weights (<= ~13 KB) are staged
// once in LDS (natural layout for forward, transposed copy for the data-gradient); each wave then
// walks 32-pixel tiles.  All contractions run on the matrix cores in the *transposed* orientation
//
//      Z^T[out, pix] = W^T[out, in] * H^T[in, pix]          (mfma 32x32: col = pixel, rows = features)
//
// so the 32x32 fp32 accumulator of one layer (pixel on the lane, features in the 16 registers) is
// directly the B operand of the next layer's MFMA: activations never leave registers in the
// forward chain or in the data-gradient chain.  Only the weight gradient
//      dW^T[out, in] = dZ^T[out, pix] * H[pix, in]
// contracts over pixels, i.e. over the lane index; for it the two 32x32 tiles take one trip through
// a per-wave LDS buffer (padded stride, conflict-free b128 reads) to get pixels into registers.
// The backward pass recomputes the forward in registers: nothing but inputs and weights is read.
//
// fp32 path: v_mfma_f32_32x32x2_f32 (exact fp32 products, k-ordered fma chain).
#include "rcb_common.h"

using namespace rcb;

#include "siren_common.h"

namespace {

// sin / cos of w0*z with a two-float range reduction in revolutions
template <bool FAST>
__device__ __forceinline__ void sincos_w0(float z, float k_hi, float k_lo, float& s, float& c) {
  float th = z * k_hi;
  float tl = __builtin_fmaf(z, k_hi, -th) + z * k_lo;
  float fr = (th - rintf(th)) + tl;  // revolutions in [-0.5, 0.5]
  if (FAST) {
    s = __builtin_amdgcn_sinf(fr);
    c = __builtin_amdgcn_cosf(fr);
  } else {
    sincospif(2.0f * fr, &s, &c);
  }
}

template <int NH, int KS0, int NB0, int MODE>
__global__ void __launch_bounds__(256) siren_kernel(SirenArgs a) {
  constexpr int NL = NH + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = lane & 31, h = lane >> 5;   // q: pixel (B/D role) or feature (A role)
  const int g = blockIdx.x;
  const int n = g / a.S;
  const int F = a.F, E = a.E, C = a.C, P = a.P;
  const int in0 = F + E;

  // layer geometry (runtime values, static loop)
  int lin[NL], lout[NL], off[NL], toff[NL];
  {
    int o = 0, t = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      lin[l] = (l == 0) ? in0 : HID;
      lout[l] = (l == NL - 1) ? C : HID;
      off[l] = o;
      toff[l] = t;
      o += lout[l] * (lin[l] + 1);
      t += lout[l] * lin[l];
    }
  }
  float* wl = smem;                 // [dnet]  natural layer vectors  [bias | W(in,out)]
  float* wt = smem + a.dnet;        // transposed weights WT_l[out][in]
  float* tiles = smem + a.tile_base; // per wave: bufA [32][TS], bufB [32*NB0][TS]   (16-B aligned)
  float* bufA = tiles + wave * (32 + 32 * NB0) * TS;
  float* bufB = bufA + 32 * TS;

  // ---- stage weights -------------------------------------------------------------------
  {
    const float* src = a.wvec + (long long)g * a.w_stride;
    // loads in batches of eight before their LDS stores (a plain `wl[i] = src[i]` loop serialises one HBM round trip per element)
    for (int i0 = tid; i0 < a.dnet; i0 += 8 * 256) {
      float stage[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = i0 + 256 * k;
        stage[k] = src[i < a.dnet ? i : a.dnet - 1];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = i0 + 256 * k;
        if (i < a.dnet) wl[i] = stage[k];
      }
    }
    __syncthreads();
    if (MODE != MODE_FWD) {
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const int ni = lin[l], no = lout[l];
        const float* W = wl + off[l] + no;
        float* T = wt + toff[l];
        for (int idx = tid; idx < ni * no; idx += 256) {
          int i = idx / no, o = idx - i * no;
          T[o * ni + i] = W[idx];
        }
      }
      __syncthreads();
    }
  }

  f32x16 gW[NL + NB0 - 1];   // weight-gradient accumulators (layer 0 may span NB0 input blocks)
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

  const int ntiles = (P + 31) >> 5;
  for (int t = wave; t < ntiles; t += 4) {
    const int p = t * 32 + q;
    const bool valid = p < P;
    // ---- layer-0 input: half-wave 0 holds the Fourier row, half-wave 1 the pe row -------
    float xin[KS0];
    const int kh = (h == 0) ? F : E;
    {
      const float* src = (h == 0) ? (a.xf + (long long)n * a.xf_stride + (long long)p * F)
                                  : (a.pe + ((long long)g * P + p) * E);
#pragma unroll
      for (int s = 0; s < KS0; s += 2) {
        float2 v = make_float2(0.f, 0.f);
        if (valid && s + 1 < kh) v = *reinterpret_cast<const float2*>(src + s);
        else if (valid && s < kh) v.x = src[s];
        xin[s] = v.x;
        xin[s + 1] = v.y;
      }
    }
    // ---- forward ---------------------------------------------------------------------------
    f32x16 S[NH], Cs[NH], acc;
    {
      const float* B0 = wl + off[0];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = B0[rho(r, h)];
      const float* W0 = B0 + HID;
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        int irow = (h == 0) ? s : F + s;
        float aw = (s < kh) ? W0[irow * HID + q] : 0.f;
        acc = mfma2(aw, xin[s], acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv, cv;
        sincos_w0<false>(acc[r], a.k_hi, a.k_lo, sv, cv);
        S[0][r] = sv;
        Cs[0][r] = cv;
      }
    }
#pragma unroll
    for (int l = 1; l < NH; ++l) {
      const float* Bl = wl + off[l];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = Bl[rho(r, h)];
      const float* Wl = Bl + HID;
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = mfma2(Wl[rho(s, h) * HID + q], S[l - 1][s], acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv, cv;
        sincos_w0<false>(acc[r], a.k_hi, a.k_lo, sv, cv);
        S[l][r] = sv;
        Cs[l][r] = cv;
      }
    }
    {
      const float* Bl = wl + off[NL - 1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = (rho(r, h) < C) ? Bl[rho(r, h)] : 0.f;
      const float* Wl = Bl + C;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        float aw = (q < C) ? Wl[rho(s, h) * C + q] : 0.f;
        acc = mfma2(aw, S[NH - 1][s], acc);
      }
    }
    if (MODE == MODE_FWD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = rho(r, h);
        if (valid && row < C) a.yout[((long long)g * P + p) * C + row] = acc[r];
      }
      continue;
    }
    // ---- output gradient --------------------------------------------------------------------
    f32x16 dz;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = rho(r, h);
      float v = 0.f;
      if (valid && row < C) {
        if (MODE == MODE_LOSS) {
          float diff = acc[r] - a.yin[((long long)n * P + p) * C + row];
          sse_local += diff * diff;
          v = 2.0f * a.dy_scale * diff;
        } else {
          v = a.yin[((long long)g * P + p) * C + row];
        }
      }
      dz[r] = v;
    }
    // ---- backward, last layer down to layer 0 ----------------------------------------------
#pragma unroll
    for (int l = NL - 1; l >= 0; --l) {
      // (1) weight gradient: dW_l^T[o][i] += sum_pix dZ_l[o][pix] * H_{l-1}[i][pix]
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int r = 0; r < 16; ++r) bufA[rho(r, h) * TS + q] = dz[r];
      if (l > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bufB[rho(r, h) * TS + q] = S[l - 1][r];
      } else {
#pragma unroll
        for (int s = 0; s < KS0; ++s) {
          int irow = (h == 0) ? s : F + s;
          if (s < kh) bufB[irow * TS + q] = xin[s];
        }
        // zero padding rows [in0, 32*NB0)
        for (int rr = in0 + h; rr < 32 * NB0; rr += 2) bufB[rr * TS + q] = 0.f;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      {
        float av[16];
        const float4* pa = reinterpret_cast<const float4*>(bufA + q * TS + 16 * h);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          float4 v = pa[c4];
          av[4 * c4] = v.x; av[4 * c4 + 1] = v.y; av[4 * c4 + 2] = v.z; av[4 * c4 + 3] = v.w;
        }
        float bsum = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) bsum += av[s];
        gb[l] += bsum;
        constexpr int NBL = 1;
#pragma unroll
        for (int blk = 0; blk < ((l == 0) ? NB0 : NBL); ++blk) {
          float bv[16];
          const float4* pb = reinterpret_cast<const float4*>(bufB + (32 * blk + q) * TS + 16 * h);
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            float4 v = pb[c4];
            bv[4 * c4] = v.x; bv[4 * c4 + 1] = v.y; bv[4 * c4 + 2] = v.z; bv[4 * c4 + 3] = v.w;
          }
          const int gi = (l == 0) ? blk : (l + NB0 - 1);
#pragma unroll
          for (int s = 0; s < 16; ++s) gW[gi] = mfma2(av[s], bv[s], gW[gi]);
        }
      }
      // (2) data gradient
      if (l > 0) {
        // dH_{l-1}^T[i][pix] = sum_o W_l[i][o] dZ_l^T[o][pix] ; A = WT_l[o][i]
        const float* T = wt + toff[l];
        const int no = lout[l];
        f32x16 dh;
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          if (l == NL - 1 && rho(s, 0) >= C) continue;  // rows beyond out_dim are zero (uniform branch)
          int o = rho(s, h);
          float aw = (o < no) ? T[o * HID + q] : 0.f;
          dh = mfma2(aw, dz[s], dh);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[r] = dh[r] * (a.w0 * Cs[l - 1][r]);
      } else if (a.dpe != nullptr) {
        // dpe^T[e][pix] = sum_o W_0[F+e][o] dZ_0^T[o][pix]
        const float* T = wt + toff[0];
        f32x16 dx;
#pragma unroll
        for (int r = 0; r < 16; ++r) dx[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          float aw = (q < E) ? T[rho(s, h) * in0 + F + q] : 0.f;
          dx = mfma2(aw, dz[s], dx);
        }
        if (valid) {
          float* dst = a.dpe + ((long long)g * P + p) * E;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int e = rho(r, h);
            if (e < E) dst[e] = dx[r];
          }
        }
      }
    }
  }
  if (MODE == MODE_FWD) return;

  // ---- deterministic cross-wave reduction of the weight gradients --------------------------
  __syncthreads();
  float* part = smem + wave * a.dnet;
  for (int i = lane; i < a.dnet; i += 64) part[i] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int ni = lin[l], no = lout[l];
    float bt = gb[l] + __shfl_xor(gb[l], 32, 64);
    if (h == 0 && q < no) part[off[l] + q] = bt;
#pragma unroll
    for (int blk = 0; blk < ((l == 0) ? NB0 : 1); ++blk) {
      const int gi = (l == 0) ? blk : (l + NB0 - 1);
      const int i = 32 * blk + q;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int o = rho(r, h);
        if (i < ni && o < no) part[off[l] + no + i * no + o] = gW[gi][r];
      }
    }
  }
  __syncthreads();
  {
    float* dst = a.dwvec + (long long)g * a.w_stride;
    for (int i = tid; i < a.dnet; i += 256)
      dst[i] = ((smem[i] + smem[a.dnet + i]) + smem[2 * a.dnet + i]) + smem[3 * a.dnet + i];
  }
  if (MODE == MODE_LOSS) {
    float v = wave_sum(sse_local);
    __syncthreads();
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    if (tid == 0) a.sse[g] = ((smem[0] + smem[1]) + smem[2]) + smem[3];
  }
}


template <int NH, int KS0, int NB0, int MODE>
int launch(const SirenArgs& a, size_t smem_bytes, hipStream_t st) {
    auto kfn = siren_kernel<NH, KS0, NB0, MODE>;
  // (per launch: the attribute belongs to the (function, device) pair; a process may drive several devices)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024);
  if (e != hipSuccess) return fail((int)e, "siren: hipFuncSetAttribute: %s", hipGetErrorString(e));
  kfn<<<a.G, 256, smem_bytes, st>>>(a);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

template <int MODE>
int dispatch(const rcb_siren_desc* d, SirenArgs& a, hipStream_t st) {
  if (d->hidden != HID || d->hidden_dims[0] != 0) return siren_generic_dispatch(MODE, d, a, st);   // fp32 at other widths / per-layer widths: siren_mlp_generic.hip
  const int in0 = d->fourier_dim + d->pe_dim;
  const int ks = d->fourier_dim > d->pe_dim ? d->fourier_dim : d->pe_dim;
  const int nb0 = (in0 + 31) / 32;
  const int NL = d->n_hidden + 1;
  int dnet = 0, wt = 0;
  for (int l = 0; l < NL; ++l) {
    int li = l == 0 ? in0 : HID, lo = l == NL - 1 ? d->out_dim : HID;
    dnet += lo * (li + 1);
    wt += lo * li;
  }
  a.dnet = dnet;
  a.wt_total = wt;
  a.tile_base = (dnet + wt + 3) & ~3;
  int need = a.tile_base + 4 * (32 + 32 * nb0) * TS;
  if (need < 4 * dnet) need = 4 * dnet;
  need = (need + 3) & ~3;
  a.smem_floats = need;
  size_t smem = (size_t)need * sizeof(float);
  RCB_REQUIRE(smem <= 160 * 1024, RCB_ERR_UNSUPPORTED, "siren: %zu B of LDS needed", smem);
#define RCB_SIREN_CASE(NHv, KSv, NBv)                                       \
  if (d->n_hidden == NHv && ks <= KSv && nb0 == NBv) return launch<NHv, KSv, NBv, MODE>(a, smem, st);
  RCB_SIREN_CASE(3, 16, 1)
  RCB_SIREN_CASE(2, 16, 1)
  RCB_SIREN_CASE(3, 18, 2)
  RCB_SIREN_CASE(1, 16, 1)
  RCB_SIREN_CASE(4, 16, 1)
  RCB_SIREN_CASE(3, 32, 2)
#undef RCB_SIREN_CASE
  return fail(RCB_ERR_UNSUPPORTED, "siren: no kernel for n_hidden=%d, F=%d, E=%d", d->n_hidden, d->fourier_dim, d->pe_dim);
}

int fill_args(const rcb_siren_desc* d, SirenArgs& a) {
  RCB_REQUIRE(d, RCB_ERR_ARG, "siren: null descriptor");
  RCB_REQUIRE(d->hidden == HID || (d->precision >= 1 && (d->hidden == 48 || d->hidden == 64)) ||
                  (d->precision == 0 && d->hidden >= 1 && d->hidden <= 64),
              RCB_ERR_UNSUPPORTED, "siren: hidden width %d (fp32 mode: up to 64; 16-bit modes: 32, 48, 64)", d->hidden);
  RCB_REQUIRE(d->n_hidden >= 1 && d->n_hidden <= 4, RCB_ERR_UNSUPPORTED, "siren: n_hidden=%d", d->n_hidden);
  RCB_REQUIRE(d->hidden_dims[0] == 0 || d->precision == 0, RCB_ERR_UNSUPPORTED,
              "siren: per-layer hidden widths exist in the fp32 mode only (the 16-bit kernels take one width: 32, 48 or 64)");
  RCB_REQUIRE(d->out_dim >= 1 && d->out_dim <= 32, RCB_ERR_UNSUPPORTED, "siren: out_dim=%d", d->out_dim);
  RCB_REQUIRE(d->fourier_dim >= 1 && d->pe_dim >= 0 && d->fourier_dim + d->pe_dim <= 64, RCB_ERR_UNSUPPORTED,
              "siren: F=%d E=%d", d->fourier_dim, d->pe_dim);
  RCB_REQUIRE(d->fourier_dim % 2 == 0 && d->pe_dim % 2 == 0, RCB_ERR_UNSUPPORTED, "siren: odd feature dims");
  RCB_REQUIRE(d->n_rows > 0 && d->samples > 0 && d->n_pix > 0 && d->n_rows % d->samples == 0, RCB_ERR_SHAPE,
              "siren: rows=%d samples=%d pix=%d", d->n_rows, d->samples, d->n_pix);
  RCB_REQUIRE(d->precision >= 0 && d->precision <= 2, RCB_ERR_UNSUPPORTED, "siren: precision %d not built", d->precision);
  RCB_REQUIRE((long long)d->n_pix * d->out_dim < (1ll << 31), RCB_ERR_SHAPE, "siren: n_pix x out_dim = %lld elements per row of targets (32-bit offsets inside a row)",
              (long long)d->n_pix * d->out_dim);
  RCB_REQUIRE(d->pe_bf16 == 0 || (d->pe_bf16 == 1 && d->precision >= 1 && d->pe_dim % 8 == 0), RCB_ERR_UNSUPPORTED,
              "siren: bf16 pe storage needs a 16-bit precision mode and pe_dim %% 8 == 0 (precision=%d, E=%d)", d->precision,
              d->pe_dim);
  RCB_REQUIRE(d->dw_bf16 == nullptr || d->precision >= 1, RCB_ERR_UNSUPPORTED,
              "siren: the bf16 copy of the gradient exists in the 16-bit kernels only");
  RCB_REQUIRE(d->dw_lo == nullptr || d->dw_bf16 != nullptr, RCB_ERR_ARG, "siren: dw_lo (low plane) comes with dw_bf16 (high plane)");
  {
    const int nl = d->n_hidden + 1;
    long long dn = 0;
    for (int l = 0; l < nl; ++l) {
      const int hw_in = l == 0 ? 0 : (d->hidden_dims[0] ? d->hidden_dims[l - 1] : d->hidden);
      const int hw_out = d->hidden_dims[0] && l < d->n_hidden ? d->hidden_dims[l] : d->hidden;
      const int li = l == 0 ? d->fourier_dim + d->pe_dim : hw_in, lo = l == nl - 1 ? d->out_dim : hw_out;
      dn += (long long)lo * (li + 1);
    }
    RCB_REQUIRE(d->w_row_stride >= dn, RCB_ERR_SHAPE, "siren: w_row_stride %lld below the %lld elements of a row of layer vectors",
                (long long)d->w_row_stride, dn);
    // (a C caller that sets dw_bf16 but leaves the stride field zero would make every row of the copy overlap)
    RCB_REQUIRE(d->dw_bf16 == nullptr || d->dw_bf16_stride >= dn, RCB_ERR_SHAPE,
                "siren: dw_bf16_stride %lld below the %lld elements of a row of layer vectors", (long long)d->dw_bf16_stride, dn);
  }
  const int chunks = d->pixel_chunks > 1 ? d->pixel_chunks : 1;
  RCB_REQUIRE(chunks == 1 || (d->precision >= 1 && chunks <= (d->n_pix + 31) / 32 && d->dw_bf16 == nullptr &&
                              (long long)d->n_rows * chunks < (1ll << 30)),
              RCB_ERR_UNSUPPORTED,
              "siren: pixel_chunks=%d needs a 16-bit precision mode, at most one chunk per 32-pixel tile and dw_bf16 == NULL "
              "(rcb_siren_reduce_chunks emits it)", d->pixel_chunks);
  memset(&a, 0, sizeof(a));
  if (d->pe_grid_dims != 0) {
    const int nd = d->pe_grid_dims;
    RCB_REQUIRE(nd >= 1 && nd <= 3 && d->precision >= 1 && d->pe_dim > 0, RCB_ERR_UNSUPPORTED,
                "siren: pe_grid_dims=%d (1..3, 16-bit kernels, pe_dim > 0)", nd);
    long long np = 1, pp = 1;
    for (int i = 0; i < 3; ++i) a.pe_pn[i] = a.pe_ps[i] = 1;
    for (int i = 0; i < nd; ++i) {                                  // right-aligned: unused leading axes are 1
      RCB_REQUIRE(d->pe_patch_nums[i] >= 1 && d->pe_patch_size[i] >= 1, RCB_ERR_SHAPE, "siren: stitched pe axis %d: %d x %d", i,
                  d->pe_patch_nums[i], d->pe_patch_size[i]);
      a.pe_pn[3 - nd + i] = d->pe_patch_nums[i];
      a.pe_ps[3 - nd + i] = d->pe_patch_size[i];
      np *= d->pe_patch_nums[i];
      pp *= d->pe_patch_size[i];
    }
    RCB_REQUIRE(pp == d->n_pix && (d->n_rows / d->samples) % np == 0 && pp * 4096 < (1ll << 32) && a.pe_ps[1] <= 4096 &&
                    a.pe_ps[2] <= 4096,
                RCB_ERR_SHAPE, "siren: stitched pe layout: patch of %lld pixels for n_pix=%d, %lld patches per datapoint for %d INRs",
                pp, d->n_pix, np, d->n_rows / d->samples);
    a.pe_nd = nd;
    a.pe_ndc = (int)((d->n_rows / d->samples) / np);
    a.pe_m1 = (unsigned)(((1ull << 32) + a.pe_ps[1] - 1) / a.pe_ps[1]);
    a.pe_m2 = (unsigned)(((1ull << 32) + a.pe_ps[2] - 1) / a.pe_ps[2]);
  }
  a.chunks = chunks;
  a.pe_bf16 = d->pe_bf16;
  a.dw16 = reinterpret_cast<__bf16*>(d->dw_bf16);
  a.dwlo = reinterpret_cast<__bf16*>(d->dw_lo);
  a.dw16_stride = d->dw_bf16_stride;
  a.xf16 = d->xf_bf16;
  a.clock_probe = reinterpret_cast<unsigned long long*>(d->clock_probe);
  a.G = d->n_rows;
  a.S = d->samples;
  a.P = d->n_pix;
  a.F = d->fourier_dim;
  a.E = d->pe_dim;
  a.C = d->out_dim;
  a.xf_stride = d->xf_inr_stride;
  a.w_stride = d->w_row_stride;
  a.w0 = d->w0;
  double k = (double)d->w0 / (2.0 * 3.14159265358979323846);
  a.k_hi = (float)k;
  a.k_lo = (float)(k - (double)a.k_hi);
  return RCB_OK;
}

// ---- sum of the per-chunk partials of a pixel_chunks launch (fixed chunk order: deterministic) ------------------------------
struct ReduceArgs {
  const float* part;
  const float* sse_part;
  float* dw;
  float* sse;
  __bf16* dw16;
  __bf16* dwlo;
  long long stride, stride16;
  int G, chunks, dnet;
};

__global__ void __launch_bounds__(256) siren_reduce_chunks_kernel(ReduceArgs r) {
  const long long total = (long long)r.G * r.dnet;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int g = (int)(e / r.dnet), j = (int)(e - (long long)g * r.dnet);
    float v = 0.f;
    for (int k = 0; k < r.chunks; ++k) v += r.part[((long long)k * r.G + g) * r.stride + j];
    if (r.dw) r.dw[(long long)g * r.stride + j] = v;
    if (r.dw16) r.dw16[(long long)g * r.stride16 + j] = (__bf16)v;
    if (r.dwlo) r.dwlo[(long long)g * r.stride16 + j] = (__bf16)(v - (float)(__bf16)v);
  }
  if (r.sse && blockIdx.x == 0) {
    for (int g = threadIdx.x; g < r.G; g += 256) {
      float v = 0.f;
      for (int k = 0; k < r.chunks; ++k) v += r.sse_part[(long long)k * r.G + g];
      r.sse[g] = v;
    }
  }
}

}  // namespace

extern "C" int rcb_siren_fwd(const rcb_siren_desc* d, const float* xf, const void* pe, const float* wvec,
                             float* y_out, rcb_stream_t stream) {
  SirenArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  RCB_REQUIRE(xf && wvec && y_out && (pe || d->pe_dim == 0), RCB_ERR_ARG, "siren_fwd: null pointer");
  a.xf = xf;
  a.pe = static_cast<const float*>(pe);
  a.wvec = wvec;
  a.yout = y_out;
  if (d->precision >= 1) return siren_16bit_dispatch(MODE_FWD, d, a, (hipStream_t)stream);
  return dispatch<MODE_FWD>(d, a, (hipStream_t)stream);
}

extern "C" int rcb_siren_bwd(const rcb_siren_desc* d, const float* xf, const void* pe, const float* wvec,
                             const float* dy, float* dwvec, void* dpe, rcb_stream_t stream) {
  SirenArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  RCB_REQUIRE(xf && wvec && dy && (pe || d->pe_dim == 0), RCB_ERR_ARG, "siren_bwd: null pointer");
  RCB_REQUIRE(dwvec || (d->dw_bf16 && d->dw_lo && a.chunks == 1), RCB_ERR_ARG,
              "siren_bwd: dwvec may be NULL only when both planes (dw_bf16, dw_lo) receive the gradient of an unchunked launch");
  a.xf = xf;
  a.pe = static_cast<const float*>(pe);
  a.wvec = wvec;
  a.yin = dy;
  a.dwvec = dwvec;
  a.dpe = static_cast<float*>(dpe);
  if (d->precision >= 1) return siren_16bit_dispatch(MODE_BWD, d, a, (hipStream_t)stream);
  return dispatch<MODE_BWD>(d, a, (hipStream_t)stream);
}

extern "C" int rcb_siren_loss_bwd(const rcb_siren_desc* d, const float* xf, const void* pe, const float* wvec,
                                  const float* target, float dy_scale, float* sse, float* dwvec, void* dpe,
                                  rcb_stream_t stream) {
  SirenArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  RCB_REQUIRE(xf && wvec && target && sse && (pe || d->pe_dim == 0), RCB_ERR_ARG, "siren_loss_bwd: null pointer");
  RCB_REQUIRE(dwvec || (d->dw_bf16 && d->dw_lo && a.chunks == 1), RCB_ERR_ARG,
              "siren_loss_bwd: dwvec may be NULL only when both planes (dw_bf16, dw_lo) receive the gradient of an unchunked launch");
  a.xf = xf;
  a.pe = static_cast<const float*>(pe);
  a.wvec = wvec;
  a.yin = target;
  a.sse = sse;
  a.dwvec = dwvec;
  a.dpe = static_cast<float*>(dpe);
  a.dy_scale = dy_scale;
  if (d->precision >= 1) return siren_16bit_dispatch(MODE_LOSS, d, a, (hipStream_t)stream);
  return dispatch<MODE_LOSS>(d, a, (hipStream_t)stream);
}

extern "C" int rcb_siren_reduce_chunks(const rcb_siren_desc* d, const float* dw_partial, const float* sse_partial, float* dwvec,
                                       float* sse, rcb_stream_t stream) {
  RCB_REQUIRE(d && dw_partial && ((sse_partial == nullptr) == (sse == nullptr)), RCB_ERR_ARG, "siren_reduce_chunks: null pointer");
  RCB_REQUIRE(dwvec || (d->dw_bf16 && d->dw_lo), RCB_ERR_ARG, "siren_reduce_chunks: dwvec may be NULL only when both planes are given");
  RCB_REQUIRE(d->dw_lo == nullptr || d->dw_bf16 != nullptr, RCB_ERR_ARG, "siren_reduce_chunks: dw_lo comes with dw_bf16");
  RCB_REQUIRE(d->pixel_chunks >= 1 && d->n_rows > 0 && d->n_hidden >= 1 && d->n_hidden <= 4, RCB_ERR_SHAPE,
              "siren_reduce_chunks: chunks=%d rows=%d", d->pixel_chunks, d->n_rows);
  ReduceArgs r;
  memset(&r, 0, sizeof(r));
  r.part = dw_partial;
  r.sse_part = sse_partial;
  r.dw = dwvec;
  r.sse = sse;
  r.dw16 = reinterpret_cast<__bf16*>(d->dw_bf16);
  r.dwlo = reinterpret_cast<__bf16*>(d->dw_lo);
  r.stride16 = d->dw_bf16_stride;
  r.stride = d->w_row_stride;
  r.G = d->n_rows;
  r.chunks = d->pixel_chunks;
  const int nl = d->n_hidden + 1;
  int o = 0;
  for (int l = 0; l < nl; ++l) {
    const int li = l == 0 ? d->fourier_dim + d->pe_dim : d->hidden, lo = l == nl - 1 ? d->out_dim : d->hidden;
    o += lo * (li + 1);
  }
  r.dnet = o;
  RCB_REQUIRE(r.dw16 == nullptr || r.stride16 >= r.dnet, RCB_ERR_SHAPE, "siren_reduce_chunks: bf16 row stride %lld < %d", r.stride16, r.dnet);
  RCB_REQUIRE(r.stride >= r.dnet, RCB_ERR_SHAPE, "siren_reduce_chunks: row stride %lld < %d", r.stride, r.dnet);
  const long long total = (long long)r.G * r.dnet;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  siren_reduce_chunks_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(r);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
