// Known-bytes probe for the HBM read counter: how does rocprofv3's FETCH_SIZE count LDS-DMA loads
// (global_load_lds_dwordx4, what atrans.hip streams its operands with) compared with ordinary 16-byte global loads?
// Each kernel reads the SAME buffer exactly once (N bytes, N > the 256 MB Infinity Cache), so FETCH_SIZE x (its unit) should
// read N for both if the counter treats them alike; the ratio of the two readings settles whether the x2 correction of the
// streaming-load case (MI355X_MICROARCH.md, HBM / rocprofv3 section) also applies to LDS-DMA traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/native/glds_fetch_probe.cpp -o /tmp/glds_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/glds_probe -- /tmp/glds_fetch
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int NT = 256;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void __launch_bounds__(NT) read_plain(const float4* __restrict__ src, float* __restrict__ sink, long long n16) {
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n16; i += (long long)gridDim.x * NT) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) sink[0] = acc;          // never true: keeps the loads alive
}

__global__ void __launch_bounds__(NT) read_dma(const float4* __restrict__ src, float* __restrict__ sink, long long n16) {
  __shared__ __attribute__((aligned(16))) float4 stage[2][NT];
  float acc = 0.f;
  const int wave = threadIdx.x >> 6;
  int it = 0;
  for (long long i0 = (long long)blockIdx.x * NT; i0 < n16; i0 += (long long)gridDim.x * NT, ++it) {
    // every wave moves its 64 x 16 bytes straight into LDS (wave-uniform destination + 16 * lane)
    const long long i = i0 + threadIdx.x < n16 ? i0 + threadIdx.x : n16 - 1;
    __builtin_amdgcn_global_load_lds(src + i, (lds_ptr_t)(&stage[it & 1][wave * 64]), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float4 v = stage[it & 1][threadIdx.x];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const long long bytes = 1ll << 30;              // 1 GiB: four times the Infinity Cache
  const long long n16 = bytes / 16;
  float4* buf;
  float* sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    read_plain<<<2048, NT>>>(buf, sink, n16);
    hipDeviceSynchronize();
    read_dma<<<2048, NT>>>(buf, sink, n16);
    hipDeviceSynchronize();
  }
  printf("read %lld bytes per launch, 3 launches of read_plain and of read_dma: %s\n", bytes, hipGetErrorString(hipGetLastError()));
  return 0;
}
