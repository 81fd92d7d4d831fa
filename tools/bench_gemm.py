"""Times the A-transform GEMM shapes under different BLAS back-ends: python tools/bench_gemm.py"""
import os, sys, time
import torch
dev = "cuda"
N, L = 4096, 1056
h = torch.randn(N, 3267, device=dev)
A = torch.randn(L, L, device=dev)
dw = torch.randn(N, 3267, device=dev)

def bench(tag):
    out = torch.empty(N, 3267, device=dev)
    def run():
        torch.mm(h[:, :L], A, out=out[:, :L])                 # fwd
        torch.mm(dw[:, :L], A.t(), out=out[:, L:2 * L])       # dgrad
        return torch.mm(h[:, :L].t(), dw[:, :L])              # wgrad
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print("%-28s %.1f us per (fwd+dgrad+wgrad) = %.1f TFLOP/s" % (tag, dt * 1e6, 3 * 2 * N * L * L / dt / 1e12), flush=True)

bench("default (hipBLASLt)")
torch.backends.cuda.preferred_blas_library("cublas")
bench("rocBLAS")
torch.backends.cuda.preferred_blas_library("cublaslt")
import torch.cuda.tunable as tun
tun.enable(True)
tun.set_max_tuning_duration(200)
tun.set_max_tuning_iterations(30)
bench("TunableOp (tuned)")
