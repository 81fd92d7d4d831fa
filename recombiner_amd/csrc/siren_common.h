// Declarations shared by the fp32 and bf16 SIREN kernels.
#pragma once
#include <cstdlib>

#include "rcb_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace rcb {

constexpr int HID = 32;      // hidden width
constexpr int TS = 36;       // LDS tile row stride in floats (32 + 4 pad: conflict-free b128 reads)
constexpr int MAXL = 5;      // max linear layers

enum { MODE_FWD = 0, MODE_BWD = 1, MODE_LOSS = 2 };

struct SirenArgs {
  const float* xf;
  const float* pe;
  const float* wvec;
  const float* yin;   // MODE_LOSS: target [N,P,C];  MODE_BWD: dy [G,P,C]
  float* yout;        // MODE_FWD: [G,P,C]
  float* sse;         // MODE_LOSS: [G]
  float* dwvec;       // [G, w_stride]
  float* dpe;         // [G,P,E] or null
  long long xf_stride, w_stride;
  int G, S, P, F, E, C;
  int dnet, wt_total, tile_base, smem_floats;
  float k_hi, k_lo;   // w0 / (2 pi) split in two floats
  float w0, dy_scale;
  int pe_bf16;        // pe / dpe hold bf16 elements (16-bit kernels only)
  __bf16* dw16;       // nullable: bf16 copy of dwvec, rows dw16_stride elements apart (rcb_siren_desc.dw_bf16)
  __bf16* dwlo;       // nullable (with dw16): the low plane bf16(dwvec - dw16), same layout; dwvec may then be NULL
  long long dw16_stride;
  int chunks;         // >= 1: workgroups per row of wvec (pixel tiles split; dwvec / sse hold per-chunk partials)
  const void* xf16;   // nullable: 16-bit copy of xf in the operand format, rows padded to 8 features (rcb_siren_desc.xf_bf16)
  unsigned long long* clock_probe;   // nullable: rcb_siren_desc.clock_probe
  // pe / dpe layout (rcb_siren_desc.pe_grid_dims): 0 = [G][P][E]; else the rows are the patches of stitched grids
  // [images][G0][G1][G2][E], G_i = pe_pn[i] * pe_ps[i] (unused leading axes are 1), image = sample * pe_ndc + datapoint
  int pe_nd, pe_ndc;
  int pe_pn[3], pe_ps[3];
  unsigned pe_m1, pe_m2;   // ceil(2^32 / pe_ps[1]), ceil(2^32 / pe_ps[2]): exact quotients for p * ps < 2^32
};

// pixel index (to be multiplied by E) of pixel 0 of row g in the pe / dpe array, and of pixel p relative to it
__device__ __forceinline__ long long pe_row_base(const SirenArgs& a, int g) {
  if (a.pe_nd == 0) return (long long)g * a.P;
  const int n = g / a.S, s = g - n * a.S;
  const int np = a.pe_pn[0] * a.pe_pn[1] * a.pe_pn[2];
  const int d = n / np;
  int pl = n - d * np;
  const int c2 = pl % a.pe_pn[2];
  pl /= a.pe_pn[2];
  const int c1 = pl % a.pe_pn[1], c0 = pl / a.pe_pn[1];
  const long long img = (long long)s * a.pe_ndc + d;
  const int G0 = a.pe_pn[0] * a.pe_ps[0], G1 = a.pe_pn[1] * a.pe_ps[1], G2 = a.pe_pn[2] * a.pe_ps[2];
  return ((img * G0 + c0 * a.pe_ps[0]) * G1 + c1 * a.pe_ps[1]) * G2 + c2 * a.pe_ps[2];
}
__device__ __forceinline__ int pe_pix_off(const SirenArgs& a, int p) {
  if (a.pe_nd == 0) return p;
  // (an axis of one pixel has no 32-bit reciprocal: its quotient is the dividend)
  const unsigned t = a.pe_ps[2] == 1 ? (unsigned)p : __umulhi((unsigned)p, a.pe_m2), y2 = (unsigned)p - t * (unsigned)a.pe_ps[2];
  const unsigned y0 = a.pe_ps[1] == 1 ? t : __umulhi(t, a.pe_m1), y1 = t - y0 * (unsigned)a.pe_ps[1];
  const int G1 = a.pe_pn[1] * a.pe_ps[1], G2 = a.pe_pn[2] * a.pe_ps[2];
  return (int)((y0 * G1 + y1) * G2 + y2);
}

// row of accumulator register r for lane half h (32x32 MFMA C/D layout)
__device__ __forceinline__ constexpr int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}


int siren_bf16_dispatch(int mode, const rcb_siren_desc* d, SirenArgs& a, hipStream_t st);
int siren_wide_dispatch(int mode, const rcb_siren_desc* d, SirenArgs& a, hipStream_t st);   // hidden width 48 / 64
int siren_generic_dispatch(int mode, const rcb_siren_desc* d, SirenArgs& a, hipStream_t st);   // fp32, widths other than 32
// width 32, loss / backward, one wave per row (siren_mlp_wave.hip); *taken = false when that family has no instance for the request
int& siren_wave_tiles();      // 1: one wave per row where instantiated, 0: the workgroup kernel everywhere (rcb_debug_siren_wave_tiles)
int siren_wave_dispatch(int mode, const rcb_siren_desc* d, SirenArgs& a, hipStream_t st, int variant, bool* taken);

// the 16-bit kernel family of a descriptor: width 32 (siren_mlp_bf16.hip) or the kernel with dealt gradient tiles for widths
// 48 / 64 (siren_mlp_wide.hip; in a -DRCB_SIREN_DEALT32 build RCB_SIREN_W32_DEALT=1 sends the width-32 loss / backward
// launches there too, for A/B runs)
inline int siren_16bit_dispatch(int mode, const rcb_siren_desc* d, SirenArgs& a, hipStream_t st) {
#ifdef RCB_SIREN_DEALT32
  static const bool dealt32 = getenv("RCB_SIREN_W32_DEALT") != nullptr;
  if (dealt32 && mode != MODE_FWD) return siren_wide_dispatch(mode, d, a, st);
#endif
  if (d->hidden > HID) return siren_wide_dispatch(mode, d, a, st);
  const int wave_variant = siren_wave_tiles();
  if (wave_variant > 0 && mode != MODE_FWD) {
    bool taken = false;
    const int rc = siren_wave_dispatch(mode, d, a, st, wave_variant, &taken);
    if (taken) return rc;
  }
  return siren_bf16_dispatch(mode, d, a, st);
}

}  // namespace rcb
