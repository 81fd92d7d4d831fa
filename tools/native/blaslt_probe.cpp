// hipBLASLt algorithm sweep for the GEMM shapes of the CIFAR training step (diagnostic: which of the library's own
// solutions is fastest for each shape, against its default heuristic pick).
//   hipcc --offload-arch=gfx950 -O2 tools/native/blaslt_probe.cpp -lhipblaslt -o gpurun_out/blaslt_probe && gpurun_out/blaslt_probe
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

struct Shape {
  const char* name;
  int m, n, k, batch;          // row-major C[m,n] = A[m,k] B[k,n]
  bool transB;                 // B given as [n,k]
  bool transA;                 // A given as [k,m]
  bool out_f32;
  long long ldc;               // row stride of C (row-major), 0 = n
  long long strideC;           // batch stride of C, 0 = m * ldc
};

int main() {
  hipblasLtHandle_t h;
  CK(hipblasLtCreate(&h));
  const int N = 4096, W = 1056;
  std::vector<Shape> shapes = {
      {"A fwd   [3] 4096x2112 @ 2112x1056 -> f32 (strided into [N,3267])", N, W, 2 * W, 3, false, false, true, 3267, W},
      {"A fwd   [3] same, contiguous output", N, W, 2 * W, 3, false, false, true, 0, 0},
      {"A dgrad [3] 4096x2112 @ (1056x2112)^T -> f32 strided", N, W, 2 * W, 3, true, false, true, 3267, W},
      {"A wgrad [3] (4096x1056)^T @ 4096x1056 -> f32", W, W, N, 3, false, true, true, 0, 0},
      {"stage1 fwd 4096x512 @ 512x4096 -> bf16", N, 4096, 512, 1, false, false, false, 0, 0},
      {"stage1 dgrad 4096x4096 @ (512x4096)^T -> bf16", N, 512, 4096, 1, true, false, false, 0, 0},
      {"stage1 wgrad (4096x512)^T @ 4096x4096 -> bf16", 512, 4096, N, 1, false, true, false, 0, 0},
  };
  size_t ws_size = 128ull << 20;
  void* ws;
  CK(hipMalloc(&ws, ws_size));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (auto& s : shapes) {
    // hipBLASLt is column-major: compute C^T[n,m] = B^T[n,k] A^T[k,m]; a row-major X[r,c] is a column-major [c,r] with ld = c
    const long long ldc = s.ldc ? s.ldc : s.n;
    const long long sC = s.strideC ? s.strideC : (long long)s.m * ldc;
    const size_t eA = 2, eB = 2, eC = s.out_f32 ? 4 : 2;
    void *dA, *dB, *dC;
    CK(hipMalloc(&dA, (size_t)s.batch * s.m * s.k * eA));
    CK(hipMalloc(&dB, (size_t)s.batch * s.k * s.n * eB));
    CK(hipMalloc(&dC, (size_t)(s.strideC ? (long long)s.m * ldc : (long long)s.batch * s.m * ldc) * eC + 4096));
    CK(hipMemset(dA, 0, (size_t)s.batch * s.m * s.k * eA));
    CK(hipMemset(dB, 0, (size_t)s.batch * s.k * s.n * eB));
    hipblasLtMatmulDesc_t md;
    CK(hipblasLtMatmulDescCreate(&md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    // column-major problem: (first operand) = row-major B, (second) = row-major A
    hipblasOperation_t op1 = s.transB ? HIPBLAS_OP_T : HIPBLAS_OP_N;   // B row-major [k,n] == col-major [n,k] (no transpose needed)
    hipblasOperation_t op2 = s.transA ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSA, &op1, sizeof(op1)));
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSB, &op2, sizeof(op2)));
    hipblasLtMatrixLayout_t l1, l2, lc;
    // first operand: col-major view of row-major B: if !transB: [n,k] ld n ; if transB (B stored [n,k] row-major = col-major [k,n] ld k)
    if (!s.transB) CK(hipblasLtMatrixLayoutCreate(&l1, HIP_R_16BF, s.n, s.k, s.n)); else CK(hipblasLtMatrixLayoutCreate(&l1, HIP_R_16BF, s.k, s.n, s.k));
    if (!s.transA) CK(hipblasLtMatrixLayoutCreate(&l2, HIP_R_16BF, s.k, s.m, s.k)); else CK(hipblasLtMatrixLayoutCreate(&l2, HIP_R_16BF, s.m, s.k, s.m));
    CK(hipblasLtMatrixLayoutCreate(&lc, s.out_f32 ? HIP_R_32F : HIP_R_16BF, s.n, s.m, ldc));
    if (s.batch > 1) {
      int32_t bc = s.batch;
      long long s1 = (long long)s.k * s.n, s2 = (long long)s.m * s.k;
      for (auto l : {l1, l2, lc}) CK(hipblasLtMatrixLayoutSetAttribute(l, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)));
      CK(hipblasLtMatrixLayoutSetAttribute(l1, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &s1, sizeof(s1)));
      CK(hipblasLtMatrixLayoutSetAttribute(l2, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &s2, sizeof(s2)));
      CK(hipblasLtMatrixLayoutSetAttribute(lc, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sC, sizeof(sC)));
    }
    hipblasLtMatmulPreference_t pref;
    CK(hipblasLtMatmulPreferenceCreate(&pref));
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_size, sizeof(ws_size)));
    const int want = 96;
    std::vector<hipblasLtMatmulHeuristicResult_t> res(want);
    int got = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(h, md, l1, l2, lc, lc, pref, want, res.data(), &got));
    float alpha = 1.f, beta = 0.f;
    double flops = 2.0 * s.m * s.n * s.k * s.batch;
    printf("%s: %d algos\n", s.name, got);
    double best = 1e30, first = 0;
    int besti = -1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < got; ++i) {
      auto run = [&]() { return hipblasLtMatmul(h, md, &alpha, dB, l1, dA, l2, &beta, dC, lc, dC, lc, &res[i].algo, ws, ws_size, st); };
      if (run() != HIPBLAS_STATUS_SUCCESS) continue;
      for (int r = 0; r < 3; ++r) run();
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      const int reps = 20;
      for (int r = 0; r < reps; ++r) run();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      double us = ms * 1e3 / reps;
      if (i == 0) first = us;
      if (us < best) { best = us; besti = i; }
    }
    printf("   default pick %.1f us (%.0f TF/s)   best #%d %.1f us (%.0f TF/s)\n", first, flops / first / 1e6, besti, best, flops / best / 1e6);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}
