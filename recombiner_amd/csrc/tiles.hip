// Overlapping-tile views of a large channel-last bf16 image, for running the small-image phase-conv kernels
// (upconv.hip: one zero-halo G x G image per workgroup pass) on the STITCHED grids of the patched presets
// (reference: utils.py:71-116 stitches the patches of a datapoint into one grid before the upsampling net).
//
//   tile (ty, tx) holds source pixels  ty * step - off .. + T - 1  (rows) x  tx * step - off .. + T - 1  (columns)
//
// rcb_tile_gather : image -> tiles, zero outside the image and (ring = 1) on the outermost row / column of every tile
//                   (source tiles: T = G, step = G - 1, off = 1, ring = 0; upstream-gradient tiles: T = 2G,
//                   step = 2G - 2, off = 2, ring = 1)
// rcb_tile_crop   : tiles -> image from the inner T - 2 ring rows / columns of every tile (the valid outputs)
// rcb_tile_fold   : tiles -> image, SUM of every tile element that maps to the pixel (the adjoint of the ring-0 gather:
//                   pixels on a tile border belong to two tiles per axis), fp32 sum rounded once to bf16
// rcb_window_gather / _fold: the 3^d-pixel window matrix of a channel-last grid and its adjoint (see below).
// Pure data movement: one 16-byte chunk (8 channels) per thread and iteration, coalesced along the channel / column axis;
// HBM-bound (bytes = image + tiles).
#include "rcb_common.h"

using namespace rcb;

namespace {

struct TileGeo {
  int n, H, W, C8, Ty, Tx, T, step, off, ring;
};

__global__ void __launch_bounds__(256) tile_gather_kernel(const uint4* __restrict__ img, uint4* __restrict__ tiles, TileGeo g) {
  const long long total = (long long)g.n * g.Ty * g.Tx * g.T * g.T * g.C8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    long long r = e;
    const int c = (int)(r % g.C8); r /= g.C8;
    const int j = (int)(r % g.T); r /= g.T;
    const int i = (int)(r % g.T); r /= g.T;
    const int tx = (int)(r % g.Tx); r /= g.Tx;
    const int ty = (int)(r % g.Ty);
    const int b = (int)(r / g.Ty);
    const int y = ty * g.step + i - g.off, x = tx * g.step + j - g.off;
    const bool ok = y >= 0 && y < g.H && x >= 0 && x < g.W && i >= g.ring && i < g.T - g.ring && j >= g.ring && j < g.T - g.ring;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ok) v = img[(((long long)b * g.H + y) * g.W + x) * g.C8 + c];
    tiles[e] = v;
  }
}

__global__ void __launch_bounds__(256) tile_crop_kernel(const uint4* __restrict__ tiles, uint4* __restrict__ img, TileGeo g) {
  const long long total = (long long)g.n * g.H * g.W * g.C8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    long long r = e;
    const int c = (int)(r % g.C8); r /= g.C8;
    const int x = (int)(r % g.W); r /= g.W;
    const int y = (int)(r % g.H);
    const int b = (int)(r / g.H);
    const int uy = y + g.off, ux = x + g.off;
    const int ty = uy / g.step, i = uy % g.step + g.ring, tx = ux / g.step, j = ux % g.step + g.ring;
    img[e] = tiles[(((((long long)b * g.Ty + ty) * g.Tx + tx) * g.T + i) * g.T + j) * g.C8 + c];
  }
}

__device__ __forceinline__ void add8(float (&acc)[8], const uint4& u) {
  acc[0] += __uint_as_float(u.x << 16); acc[1] += __uint_as_float(u.x & 0xffff0000u);
  acc[2] += __uint_as_float(u.y << 16); acc[3] += __uint_as_float(u.y & 0xffff0000u);
  acc[4] += __uint_as_float(u.z << 16); acc[5] += __uint_as_float(u.z & 0xffff0000u);
  acc[6] += __uint_as_float(u.w << 16); acc[7] += __uint_as_float(u.w & 0xffff0000u);
}

__global__ void __launch_bounds__(256) tile_fold_kernel(const uint4* __restrict__ tiles, uint4* __restrict__ img, TileGeo g) {
  const long long total = (long long)g.n * g.H * g.W * g.C8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    long long r = e;
    const int c = (int)(r % g.C8); r /= g.C8;
    const int x = (int)(r % g.W); r /= g.W;
    const int y = (int)(r % g.H);
    const int b = (int)(r / g.H);
    const int py = y + g.off, px = x + g.off;
    // candidates per axis: (t, i) = (p / step, p % step) and, on a tile border, (t - 1, i + step)
    int tys[2] = {py / g.step, py / g.step - 1}, is[2] = {py % g.step, py % g.step + g.step};
    int txs[2] = {px / g.step, px / g.step - 1}, js[2] = {px % g.step, px % g.step + g.step};
    const bool oky[2] = {tys[0] < g.Ty, is[0] == 0 && tys[1] >= 0}, okx[2] = {txs[0] < g.Tx, js[0] == 0 && txs[1] >= 0};
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int d = 0; d < 2; ++d)
        if (oky[a] && okx[d])      // fixed order (0,0), (0,1), (1,0), (1,1): bitwise reproducible
          add8(acc, tiles[(((((long long)b * g.Ty + tys[a]) * g.Tx + txs[d]) * g.T + is[a]) * g.T + js[d]) * g.C8 + c]);
    union { __bf16 h[8]; uint4 u; } o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.h[k] = (__bf16)acc[k];
    img[e] = o.u;
  }
}

// ---- 3^d-pixel windows (torch-level phase form of the 1-D / 3-D upsampling nets: one GEMM per stage over these windows) ----
// cols[b, p, k, :] = x[b, p + o(k) - 1, :] (zero outside the grid), k = window tap, o(k) its base-3 digits (axis 0 most
// significant), nd = 1..3 windowed axes; fold is the adjoint: dx[b, p, :] = sum_k dcols[b, p - o(k) + 1, k, :].
struct WinGeo {
  int B, g[3], nd, C8, taps;
};

__global__ void __launch_bounds__(256) window_gather_kernel(const uint4* __restrict__ x, uint4* __restrict__ cols, WinGeo w) {
  const long long npos = (long long)w.B * w.g[0] * w.g[1] * w.g[2];
  const long long total = npos * w.taps * w.C8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    long long r = e;
    const int c = (int)(r % w.C8); r /= w.C8;
    int k = (int)(r % w.taps); r /= w.taps;
    int p[3];
    p[2] = (int)(r % w.g[2]); r /= w.g[2];
    p[1] = (int)(r % w.g[1]); r /= w.g[1];
    p[0] = (int)(r % w.g[0]);
    const long long b = r / w.g[0];
    bool ok = true;
#pragma unroll
    for (int d = 2; d >= 0; --d) {
      if (d < w.nd) {
        const int o = k % 3;
        k /= 3;
        p[d] += o - 1;
        ok = ok && p[d] >= 0 && p[d] < w.g[d];
      }
    }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ok) v = x[(((b * w.g[0] + p[0]) * w.g[1] + p[1]) * w.g[2] + p[2]) * w.C8 + c];
    cols[e] = v;
  }
}

__global__ void __launch_bounds__(256) window_fold_kernel(const uint4* __restrict__ dcols, uint4* __restrict__ dx, WinGeo w) {
  const long long total = (long long)w.B * w.g[0] * w.g[1] * w.g[2] * w.C8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    long long r = e;
    const int c = (int)(r % w.C8); r /= w.C8;
    int p[3];
    p[2] = (int)(r % w.g[2]); r /= w.g[2];
    p[1] = (int)(r % w.g[1]); r /= w.g[1];
    p[0] = (int)(r % w.g[0]);
    const long long b = r / w.g[0];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < w.taps; ++k) {        // fixed order: bitwise reproducible
      int kk = k, q[3] = {p[0], p[1], p[2]};
      bool ok = true;
#pragma unroll
      for (int d = 2; d >= 0; --d) {
        if (d < w.nd) {
          const int o = kk % 3;
          kk /= 3;
          q[d] += 1 - o;
          ok = ok && q[d] >= 0 && q[d] < w.g[d];
        }
      }
      if (ok) add8(acc, dcols[((((b * w.g[0] + q[0]) * w.g[1] + q[1]) * w.g[2] + q[2]) * w.taps + k) * w.C8 + c]);
    }
    union { __bf16 h[8]; uint4 u; } o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.h[j] = (__bf16)acc[j];
    dx[e] = o.u;
  }
}

int check_win(const char* who, const void* a, const void* b, int B, int g0, int g1, int g2, int C, int nd, WinGeo& w) {
  RCB_REQUIRE(a && b, RCB_ERR_ARG, "%s: null pointer", who);
  RCB_REQUIRE(B > 0 && g0 > 0 && g1 > 0 && g2 > 0 && C > 0 && C % 8 == 0 && nd >= 1 && nd <= 3 && (nd > 1 || g1 == 1) &&
                  (nd > 2 || g2 == 1),
              RCB_ERR_SHAPE, "%s: B=%d grid=%dx%dx%d C=%d nd=%d (unused axes must have size 1)", who, B, g0, g1, g2, C, nd);
  w.B = B; w.g[0] = g0; w.g[1] = g1; w.g[2] = g2; w.nd = nd; w.C8 = C / 8;
  w.taps = nd == 1 ? 3 : (nd == 2 ? 9 : 27);
  return RCB_OK;
}

int check_geo(const char* who, const void* a, const void* b, int n, int H, int W, int C, int Ty, int Tx, int T, int step,
              int off, int ring, TileGeo& g) {
  RCB_REQUIRE(a && b, RCB_ERR_ARG, "%s: null pointer", who);
  RCB_REQUIRE(n > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && Ty > 0 && Tx > 0 && T > 1 && step > 0 && step <= T && off >= 0 &&
                  (ring == 0 || ring == 1) && T > 2 * ring,
              RCB_ERR_SHAPE, "%s: n=%d H=%d W=%d C=%d tiles=%dx%d T=%d step=%d off=%d ring=%d", who, n, H, W, C, Ty, Tx, T, step,
              off, ring);
  g = TileGeo{n, H, W, C / 8, Ty, Tx, T, step, off, ring};
  return RCB_OK;
}

inline int blocks_for(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int rcb_tile_gather(const void* img, void* tiles, int32_t n, int32_t H, int32_t W, int32_t C, int32_t Ty, int32_t Tx,
                               int32_t T, int32_t step, int32_t off, int32_t ring, rcb_stream_t stream) {
  TileGeo g;
  int rc = check_geo("tile_gather", img, tiles, n, H, W, C, Ty, Tx, T, step, off, ring, g);
  if (rc) return rc;
  const long long total = (long long)n * Ty * Tx * T * T * g.C8;
  tile_gather_kernel<<<blocks_for(total), 256, 0, (hipStream_t)stream>>>(static_cast<const uint4*>(img), static_cast<uint4*>(tiles), g);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_tile_crop(const void* tiles, void* img, int32_t n, int32_t H, int32_t W, int32_t C, int32_t Ty, int32_t Tx,
                             int32_t T, int32_t off, rcb_stream_t stream) {
  TileGeo g;
  int rc = check_geo("tile_crop", tiles, img, n, H, W, C, Ty, Tx, T, T - 2, off, 1, g);
  if (rc) return rc;
  // every image pixel must fall on a valid row / column of an existing tile
  RCB_REQUIRE((H - 1 + off) / (T - 2) < Ty && (W - 1 + off) / (T - 2) < Tx, RCB_ERR_SHAPE,
              "tile_crop: %dx%d tiles of %d valid rows do not cover a %dx%d image at offset %d", Ty, Tx, T - 2, H, W, off);
  const long long total = (long long)n * H * W * g.C8;
  tile_crop_kernel<<<blocks_for(total), 256, 0, (hipStream_t)stream>>>(static_cast<const uint4*>(tiles), static_cast<uint4*>(img), g);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_tile_fold(const void* tiles, void* img, int32_t n, int32_t H, int32_t W, int32_t C, int32_t Ty, int32_t Tx,
                             int32_t T, int32_t off, rcb_stream_t stream) {
  TileGeo g;
  int rc = check_geo("tile_fold", tiles, img, n, H, W, C, Ty, Tx, T, T - 1, off, 0, g);
  if (rc) return rc;
  RCB_REQUIRE((long long)Ty * (T - 1) >= H - 1 + off && (long long)Tx * (T - 1) >= W - 1 + off, RCB_ERR_SHAPE,
              "tile_fold: %dx%d tiles of step %d do not cover a %dx%d image at offset %d", Ty, Tx, T - 1, H, W, off);
  const long long total = (long long)n * H * W * g.C8;
  tile_fold_kernel<<<blocks_for(total), 256, 0, (hipStream_t)stream>>>(static_cast<const uint4*>(tiles), static_cast<uint4*>(img), g);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_window_gather(const void* x, void* cols, int32_t B, int32_t g0, int32_t g1, int32_t g2, int32_t C, int32_t nd,
                                 rcb_stream_t stream) {
  WinGeo w;
  int rc = check_win("window_gather", x, cols, B, g0, g1, g2, C, nd, w);
  if (rc) return rc;
  const long long total = (long long)B * g0 * g1 * g2 * w.taps * w.C8;
  window_gather_kernel<<<blocks_for(total), 256, 0, (hipStream_t)stream>>>(static_cast<const uint4*>(x), static_cast<uint4*>(cols), w);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}

extern "C" int rcb_window_fold(const void* dcols, void* dx, int32_t B, int32_t g0, int32_t g1, int32_t g2, int32_t C, int32_t nd,
                               rcb_stream_t stream) {
  WinGeo w;
  int rc = check_win("window_fold", dcols, dx, B, g0, g1, g2, C, nd, w);
  if (rc) return rc;
  const long long total = (long long)B * g0 * g1 * g2 * w.C8;
  window_fold_kernel<<<blocks_for(total), 256, 0, (hipStream_t)stream>>>(static_cast<const uint4*>(dcols), static_cast<uint4*>(dx), w);
  RCB_LAUNCH_CHECK();
  return RCB_OK;
}
