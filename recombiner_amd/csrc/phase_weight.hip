// The window-GEMM form of a  nearest-upsample(f) -> conv(k, pad)  stage (stage 1 of the upsampling net on the 1-D / 3-D and
// stitched 2-D grids, reference prior_model.py:23-59) contracts every 3^d-pixel window of the source grid with ONE matrix
//
//   Wbig[(n_0 .. n_d-1, ci), (a_0 .. a_d-1, co)] = sum over the kernel taps kk with win_i(a_i, kk_i) == n_i for every axis of
//                                                  W[co][ci][kk],      win(a, kk) = floor((a + kk - pad) / f) + 1
//
// (phase a of an axis reads source pixel i + floor((a + kk - pad) / f) for tap kk; all reference stages stay inside the
// 3-pixel window).  The mapping is linear with 0 / 1 coefficients; built with einsums it cost the video step ~210 us per step
// in fp32 batched GEMMs and permute copies.  Here: one gather-sum kernel each way, coalesced along co.
//   rcb_phase_bigweight      : Wt [k^d][ci][co] fp32 (the conv weight moved to channel-last) -> Wbig bf16 or fp32
//   rcb_phase_bigweight_grad : dWbig (bf16 or fp32) -> dWt [k^d][ci][co] fp32  (every tap gathers its prod(f) entries)
#include "rcb_common.h"

using namespace rcb;

namespace {

struct PhaseWGeo {
  int nd, k, cin, cout;
  int f[3];                      // right-aligned: unused leading axes have f = 1, and their single tap / window cell is 0
  int kd[3];                     // taps per axis (k, or 1 on unused axes)
  int wd[3];                     // window cells per axis (3, or 1)
  // the tap <-> window-cell relation win(a, kk), packed so that the kernels never index an argument array dynamically
  // (that would go through scratch memory):
  unsigned long long tapmask[3][3];   // [axis][cell n]: byte a = bit mask of the taps kk with win(a, kk) == n
  unsigned long long cellcode[3][2];  // [axis][kk >> 2]: 16-bit field kk & 3 = the cells of phases 0..7, two bits each
};

__device__ __forceinline__ unsigned long long pick3(const unsigned long long (&v)[3], int n) {
  return n == 0 ? v[0] : (n == 1 ? v[1] : v[2]);
}

// one WORKGROUP per (window cell, ci): the k^d x cout weights of the input channel are staged in LDS (every output is a sum of
// a few of them: from L2 each dependent load cost its full latency), then one wave per phase, lanes along co
template <typename TO>
__global__ void __launch_bounds__(256) phase_bigweight_kernel(const float* __restrict__ wt, TO* __restrict__ big, PhaseWGeo g) {
  extern __shared__ float wl[];                                         // [k^d][cout]
  const int ktot = g.kd[0] * g.kd[1] * g.kd[2], nph = g.f[0] * g.f[1] * g.f[2];
  int r = blockIdx.x;                                                   // (n0, n1, n2, ci), ci fastest
  const int ci = r % g.cin; r /= g.cin;
  const int n2 = r % g.wd[2]; r /= g.wd[2];
  const int n1 = r % g.wd[1];
  const int n0 = r / g.wd[1];
  // loads in batches of eight before their LDS stores (a load-store loop pays one L2 round trip per iteration)
  for (int e0 = threadIdx.x; e0 < ktot * g.cout; e0 += 8 * 256) {
    float stage[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u < ktot * g.cout ? e0 + 256 * u : ktot * g.cout - 1;
      const int t = e / g.cout, co = e - t * g.cout;
      stage[u] = wt[((long long)t * g.cin + ci) * g.cout + co];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + 256 * u < ktot * g.cout) wl[e0 + 256 * u] = stage[u];
  }
  __syncthreads();
  const unsigned long long t0 = pick3(g.tapmask[0], n0), t1 = pick3(g.tapmask[1], n1), t2 = pick3(g.tapmask[2], n2);
  const int lane = threadIdx.x & 63;
  TO* out = big + (long long)blockIdx.x * nph * g.cout;                 // row (n, ci), columns (a0, a1, a2, co)
  for (int a = threadIdx.x >> 6; a < nph; a += 4) {
    int q = __builtin_amdgcn_readfirstlane(a);
    const int a2 = q % g.f[2]; q /= g.f[2];
    const int a1 = q % g.f[1];
    const int a0 = q / g.f[1];
    const unsigned m0 = (unsigned)(t0 >> (8 * a0)) & 0xffu, m1 = (unsigned)(t1 >> (8 * a1)) & 0xffu, m2 = (unsigned)(t2 >> (8 * a2)) & 0xffu;
    for (int co = lane; co < g.cout; co += 64) {
      float s = 0.f;
      for (unsigned b0 = m0; b0; b0 &= b0 - 1) {
        const int k0 = __builtin_ctz(b0);
        for (unsigned b1 = m1; b1; b1 &= b1 - 1) {
          const int k1 = __builtin_ctz(b1);
          for (unsigned b2 = m2; b2; b2 &= b2 - 1) {
            const int k2 = __builtin_ctz(b2);
            s += wl[((k0 * g.kd[1] + k1) * g.kd[2] + k2) * g.cout + co];
          }
        }
      }
      out[a * g.cout + co] = (TO)s;
    }
  }
}

// The same forward map evaluated SEPARABLY for the stage geometries of the reference nets (trailing factors (1, 4) and (4, 4)):
// the direct sum above takes prod_i |taps_i(a_i, n_i)| terms per output -- 43 on average in the centre window cell of the
// video stage ((6,4,4) x 5^3: 4100 dependent LDS reads per lane and workgroup, 128 such workgroups set the kernel's time) --
// whereas q[a2] = sum_{k2} w[k0][k1][k2], acc[a1][a2] += q[a2] over k1, over k0 needs ~30 operations per (k0, k1) pair and
// reads every staged weight once.  The waves split the phases of the leading axis.
template <typename TO, int F1, int F2>
__global__ void __launch_bounds__(256) phase_bigweight_sep_kernel(const float* __restrict__ wt, TO* __restrict__ big, PhaseWGeo g) {
  extern __shared__ float wl[];                                         // [k^d][cout]
  const int K1 = g.kd[1], K2 = g.kd[2], ktot = g.kd[0] * K1 * K2, nph = g.f[0] * F1 * F2;
  int r = blockIdx.x;                                                   // (n0, n1, n2, ci), ci fastest
  const int ci = r % g.cin; r /= g.cin;
  const int n2 = r % g.wd[2]; r /= g.wd[2];
  const int n1 = r % g.wd[1];
  const int n0 = r / g.wd[1];
  for (int e0 = threadIdx.x; e0 < ktot * g.cout; e0 += 8 * 256) {
    float stage[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u < ktot * g.cout ? e0 + 256 * u : ktot * g.cout - 1;
      const int t = e / g.cout, co = e - t * g.cout;
      stage[u] = wt[((long long)t * g.cin + ci) * g.cout + co];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + 256 * u < ktot * g.cout) wl[e0 + 256 * u] = stage[u];
  }
  __syncthreads();
  const unsigned long long t0 = pick3(g.tapmask[0], n0), t1 = pick3(g.tapmask[1], n1), t2 = pick3(g.tapmask[2], n2);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  TO* out = big + (long long)blockIdx.x * nph * g.cout;                 // row (n, ci), columns (a0, a1, a2, co)
  for (int a0 = wave; a0 < g.f[0]; a0 += 4) {
    const unsigned m0 = (unsigned)(t0 >> (8 * a0)) & 0xffu;
    for (int co = lane; co < g.cout; co += 64) {
      float acc[F1][F2];
#pragma unroll
      for (int a1 = 0; a1 < F1; ++a1)
#pragma unroll
        for (int a2 = 0; a2 < F2; ++a2) acc[a1][a2] = 0.f;
      for (unsigned b0 = m0; b0; b0 &= b0 - 1) {
        const int k0 = __builtin_ctz(b0);
        for (int k1 = 0; k1 < K1; ++k1) {
          unsigned s1 = 0;                                               // the phases a1 whose window cell n1 takes tap k1
#pragma unroll
          for (int a1 = 0; a1 < F1; ++a1) s1 |= ((unsigned)(t1 >> (8 * a1 + k1)) & 1u) << a1;
          if (!s1) continue;
          float w2[8];
#pragma unroll
          for (int k2 = 0; k2 < 8; ++k2) w2[k2] = k2 < K2 ? wl[((k0 * K1 + k1) * K2 + k2) * g.cout + co] : 0.f;
          float q[F2];
#pragma unroll
          for (int a2 = 0; a2 < F2; ++a2) {
            const unsigned m2 = (unsigned)(t2 >> (8 * a2)) & 0xffu;
            float v = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2)
              if ((m2 >> k2) & 1u) v += w2[k2];
            q[a2] = v;
          }
#pragma unroll
          for (int a1 = 0; a1 < F1; ++a1)
            if ((s1 >> a1) & 1u) {
#pragma unroll
              for (int a2 = 0; a2 < F2; ++a2) acc[a1][a2] += q[a2];
            }
        }
      }
#pragma unroll
      for (int a1 = 0; a1 < F1; ++a1)
#pragma unroll
        for (int a2 = 0; a2 < F2; ++a2) out[((a0 * F1 + a1) * F2 + a2) * g.cout + co] = (TO)acc[a1][a2];
    }
  }
}

// one wave per (tap, ci): gathers the prod(f) entries of dWbig the tap contributes to, lanes along co.  (Dealing the entries to
// the four waves of a workgroup and joining partial sums through LDS measured slower: 156 vs 84 us on the video stage.)
template <typename TI>
__global__ void __launch_bounds__(256) phase_bigweight_grad_kernel(const TI* __restrict__ dbig, float* __restrict__ dwt, PhaseWGeo g) {
  const int nph = g.f[0] * g.f[1] * g.f[2];
  const long long cols = (long long)nph * g.cout;
  const long long n_items = (long long)g.kd[0] * g.kd[1] * g.kd[2] * g.cin;
  const int lane = threadIdx.x & 63;
  for (long long it = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += (long long)gridDim.x * 4) {
    int r = __builtin_amdgcn_readfirstlane((int)it);
    const int ci = r % g.cin; r /= g.cin;
    const int k2 = r % g.kd[2]; r /= g.kd[2];
    const int k1 = r % g.kd[1];
    const int k0 = r / g.kd[1];
    const unsigned c0 = (unsigned)(((k0 >> 2) ? g.cellcode[0][1] : g.cellcode[0][0]) >> (16 * (k0 & 3))) & 0xffffu;
    const unsigned c1 = (unsigned)(((k1 >> 2) ? g.cellcode[1][1] : g.cellcode[1][0]) >> (16 * (k1 & 3))) & 0xffffu;
    const unsigned c2 = (unsigned)(((k2 >> 2) ? g.cellcode[2][1] : g.cellcode[2][0]) >> (16 * (k2 & 3))) & 0xffffu;
    for (int co = lane; co < g.cout; co += 64) {
      float s = 0.f;
      for (int a0 = 0; a0 < g.f[0]; ++a0) {
        const int n0 = (c0 >> (2 * a0)) & 3;
        for (int a1 = 0; a1 < g.f[1]; ++a1) {
          const int n1 = (c1 >> (2 * a1)) & 3;
#pragma unroll 4
          for (int a2 = 0; a2 < g.f[2]; ++a2) {
            const int n2 = (c2 >> (2 * a2)) & 3;
            const long long row = (((long long)n0 * g.wd[1] + n1) * g.wd[2] + n2) * g.cin + ci;
            const long long col = (((long long)a0 * g.f[1] + a1) * g.f[2] + a2) * g.cout + co;
            s += (float)dbig[row * cols + col];
          }
        }
      }
      dwt[it * g.cout + co] = s;
    }
  }
}

// the adjoint with the trailing factors known at compile time: the F1 * F2 loads of a leading-axis phase are issued together
template <typename TI, int F1, int F2>
__global__ void __launch_bounds__(256) phase_bigweight_grad_fixed_kernel(const TI* __restrict__ dbig, float* __restrict__ dwt, PhaseWGeo g) {
  const int nph = g.f[0] * F1 * F2;
  const long long cols = (long long)nph * g.cout;
  const long long n_items = (long long)g.kd[0] * g.kd[1] * g.kd[2] * g.cin;
  const int lane = threadIdx.x & 63;
  for (long long it = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += (long long)gridDim.x * 4) {
    int r = __builtin_amdgcn_readfirstlane((int)it);
    const int ci = r % g.cin; r /= g.cin;
    const int k2 = r % g.kd[2]; r /= g.kd[2];
    const int k1 = r % g.kd[1];
    const int k0 = r / g.kd[1];
    const unsigned c0 = (unsigned)(((k0 >> 2) ? g.cellcode[0][1] : g.cellcode[0][0]) >> (16 * (k0 & 3))) & 0xffffu;
    const unsigned c1 = (unsigned)(((k1 >> 2) ? g.cellcode[1][1] : g.cellcode[1][0]) >> (16 * (k1 & 3))) & 0xffffu;
    const unsigned c2 = (unsigned)(((k2 >> 2) ? g.cellcode[2][1] : g.cellcode[2][0]) >> (16 * (k2 & 3))) & 0xffffu;
    for (int co = lane; co < g.cout; co += 64) {
      float s = 0.f;
      for (int a0 = 0; a0 < g.f[0]; ++a0) {
        const int n0 = (c0 >> (2 * a0)) & 3;
        TI v[F1][F2];
#pragma unroll
        for (int a1 = 0; a1 < F1; ++a1)
#pragma unroll
          for (int a2 = 0; a2 < F2; ++a2) {
            const int n1 = (c1 >> (2 * a1)) & 3, n2 = (c2 >> (2 * a2)) & 3;
            const long long row = (((long long)n0 * g.wd[1] + n1) * g.wd[2] + n2) * g.cin + ci;
            v[a1][a2] = dbig[row * cols + (long long)((a0 * F1 + a1) * F2 + a2) * g.cout + co];
          }
#pragma unroll
        for (int a1 = 0; a1 < F1; ++a1)
#pragma unroll
          for (int a2 = 0; a2 < F2; ++a2) s += (float)v[a1][a2];
      }
      dwt[it * g.cout + co] = s;
    }
  }
}

int fill_geo(PhaseWGeo& g, int nd, const int32_t* f, int k, int pad, int cin, int cout) {
  RCB_REQUIRE(nd >= 1 && nd <= 3 && f && k >= 1 && k <= 8 && pad >= 0 && cin >= 1 && cout >= 1, RCB_ERR_ARG,
              "phase_bigweight: nd=%d k=%d pad=%d cin=%d cout=%d", nd, k, pad, cin, cout);
  memset(&g, 0, sizeof(g));
  g.nd = nd;
  g.k = k;
  g.cin = cin;
  g.cout = cout;
  for (int i = 0; i < 3; ++i) {
    g.f[i] = 1;
    g.kd[i] = 1;
    g.wd[i] = 1;
    g.tapmask[i][0] = 1;          // unused axis: phase 0, tap 0, cell 0
  }
  for (int i = 0; i < nd; ++i) {
    const int ax = 3 - nd + i, fi = f[i];
    RCB_REQUIRE(fi >= 1 && fi <= 8, RCB_ERR_UNSUPPORTED, "phase_bigweight: upsampling factor %d", fi);
    g.f[ax] = fi;
    g.kd[ax] = k;
    g.wd[ax] = 3;
    g.tapmask[ax][0] = 0;
    for (int a = 0; a < fi; ++a)
      for (int kk = 0; kk < k; ++kk) {
        const int num = a + kk - pad;
        const int fl = num >= 0 ? num / fi : -((-num + fi - 1) / fi);       // floor division
        RCB_REQUIRE(fl >= -1 && fl <= 1, RCB_ERR_UNSUPPORTED,
                    "phase_bigweight: (f, k, pad) = (%d, %d, %d) reaches outside the 3-pixel window", fi, k, pad);
        g.tapmask[ax][fl + 1] |= 1ull << (8 * a + kk);
        g.cellcode[ax][kk >> 2] |= (unsigned long long)(fl + 1) << (16 * (kk & 3) + 2 * a);
      }
  }
  return RCB_OK;
}

}  // namespace

extern "C" int rcb_phase_bigweight(const float* wt, void* big, int32_t out_bf16, int32_t nd, const int32_t* f, int32_t k, int32_t pad,
                                   int32_t cin, int32_t cout, rcb_stream_t stream) {
  RCB_REQUIRE(wt && big, RCB_ERR_ARG, "phase_bigweight: null pointer");
  PhaseWGeo g;
  int rc = fill_geo(g, nd, f, k, pad, cin, cout);
  if (rc) return rc;
  const long long blocks = (long long)g.wd[0] * g.wd[1] * g.wd[2] * cin;
  const size_t lds = (size_t)g.kd[0] * g.kd[1] * g.kd[2] * cout * sizeof(float);
  RCB_REQUIRE(blocks < (1ll << 31) && lds <= 64 * 1024, RCB_ERR_UNSUPPORTED, "phase_bigweight: %lld rows, %zu B of LDS per input channel",
              blocks, lds);
  hipStream_t st = (hipStream_t)stream;
#define RCB_BW_LAUNCH(KERNEL)                                                                  \
  do {                                                                                         \
    if (out_bf16)                                                                              \
      KERNEL<__bf16><<<(int)blocks, 256, lds, st>>>(wt, static_cast<__bf16*>(big), g);         \
    else                                                                                       \
      KERNEL<float><<<(int)blocks, 256, lds, st>>>(wt, static_cast<float*>(big), g);           \
  } while (0)
  if (g.f[1] == 4 && g.f[2] == 4) {
    if (out_bf16)
      phase_bigweight_sep_kernel<__bf16, 4, 4><<<(int)blocks, 256, lds, st>>>(wt, static_cast<__bf16*>(big), g);
    else
      phase_bigweight_sep_kernel<float, 4, 4><<<(int)blocks, 256, lds, st>>>(wt, static_cast<float*>(big), g);
  } else if (g.f[1] == 1 && g.f[2] == 4) {
    if (out_bf16)
      phase_bigweight_sep_kernel<__bf16, 1, 4><<<(int)blocks, 256, lds, st>>>(wt, static_cast<__bf16*>(big), g);
    else
      phase_bigweight_sep_kernel<float, 1, 4><<<(int)blocks, 256, lds, st>>>(wt, static_cast<float*>(big), g);
  } else {
    RCB_BW_LAUNCH(phase_bigweight_kernel);
  }
#undef RCB_BW_LAUNCH
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_phase_bigweight_grad(const void* dbig, int32_t in_bf16, float* dwt, int32_t nd, const int32_t* f, int32_t k,
                                        int32_t pad, int32_t cin, int32_t cout, rcb_stream_t stream) {
  RCB_REQUIRE(dbig && dwt, RCB_ERR_ARG, "phase_bigweight_grad: null pointer");
  PhaseWGeo g;
  int rc = fill_geo(g, nd, f, k, pad, cin, cout);
  if (rc) return rc;
  const long long items = (long long)g.kd[0] * g.kd[1] * g.kd[2] * cin;                                   // one wave each
  RCB_REQUIRE(items < (1ll << 31), RCB_ERR_UNSUPPORTED, "phase_bigweight_grad: %lld taps x channels", items);
  const int grid = (int)((items + 3) / 4 > 65536 ? 65536 : (items + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (g.f[1] == 4 && g.f[2] == 4) {
    if (in_bf16)
      phase_bigweight_grad_fixed_kernel<__bf16, 4, 4><<<grid, 256, 0, st>>>(static_cast<const __bf16*>(dbig), dwt, g);
    else
      phase_bigweight_grad_fixed_kernel<float, 4, 4><<<grid, 256, 0, st>>>(static_cast<const float*>(dbig), dwt, g);
  } else if (g.f[1] == 1 && g.f[2] == 4) {
    if (in_bf16)
      phase_bigweight_grad_fixed_kernel<__bf16, 1, 4><<<grid, 256, 0, st>>>(static_cast<const __bf16*>(dbig), dwt, g);
    else
      phase_bigweight_grad_fixed_kernel<float, 1, 4><<<grid, 256, 0, st>>>(static_cast<const float*>(dbig), dwt, g);
  } else if (in_bf16) {
    phase_bigweight_grad_kernel<__bf16><<<grid, 256, 0, st>>>(static_cast<const __bf16*>(dbig), dwt, g);
  } else {
    phase_bigweight_grad_kernel<float><<<grid, 256, 0, st>>>(static_cast<const float*>(dbig), dwt, g);
  }
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
