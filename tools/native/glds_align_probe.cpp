// Probe: does global_load_lds_dwordx4 (LDS-DMA, 16 B per lane) accept source addresses that are only 4-byte aligned?
// build: hipcc --offload-arch=gfx950 -O3 glds_align_probe.cpp -o /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* __restrict__ src, float* __restrict__ dst, int shift) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4];
  const int lane = threadIdx.x;
  const float* g = src + shift + lane * 4;           // 16 B per lane, base misaligned by 4 * shift bytes
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) dst[lane * 4 + i] = lds[lane * 4 + i];
}

int main() {
  const int n = 1024;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    hipMemset(o, 0, 256 * 4);
    probe<<<1, 64>>>(d, o, shift);
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[i] != (float)(i + shift);
    printf("shift %d floats: %s, %d mismatches (first values %g %g %g %g)\n", shift, hipGetErrorString(e), bad, r[0], r[1], r[2], r[3]);
  }
  return 0;
}
