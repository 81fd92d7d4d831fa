"""A-transform GEMM shapes (4096 x 1056 x 1056): fp32 library GEMM vs bf16 and split-bf16 (hi/lo, 3 products in one
K-concatenated GEMM with an fp32 result: torch.mm(..., out_dtype=float32)).  Reports time and max-norm error against fp64.   python tools/bench_split_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import tuning

tuning.enable_tuned_gemms()
dev = "cuda"
torch.manual_seed(0)
N, D = 4096, 1056
h = torch.randn(N, D, device=dev) * 0.03            # latent sample
A = torch.randn(D, D, device=dev) / D ** 0.5        # mapping
g = torch.randn(N, D, device=dev) * 1e-3            # upstream gradient


def timed(fn, reps=30):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, out


def split(x):
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    return hi, lo


def cat3_left(x):      # [hi | lo | hi] along K (last dim)
    hi, lo = split(x)
    return torch.cat([hi, lo, hi], -1)


def cat3_right(w):     # [hi ; hi ; lo] along K (first dim)
    hi, lo = split(w)
    return torch.cat([hi, hi, lo], 0)


def rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


cases = {
    "fwd   h @ A      ": (lambda: h @ A, lambda: h.bfloat16() @ A.bfloat16(),
                          lambda: cat3_left(h), lambda: cat3_right(A), lambda l, r: torch.mm(l, r, out_dtype=torch.float32), h.double() @ A.double()),
    "dgrad g @ A^T    ": (lambda: g @ A.t(), lambda: g.bfloat16() @ A.bfloat16().t(),
                          lambda: cat3_left(g), lambda: cat3_right(A.t().contiguous()), lambda l, r: torch.mm(l, r, out_dtype=torch.float32),
                          g.double() @ A.double().t()),
    "wgrad h^T @ g    ": (lambda: h.t() @ g, lambda: h.bfloat16().t() @ g.bfloat16(),
                          lambda: cat3_left(h.t().contiguous()), lambda: cat3_right(g), lambda l, r: torch.mm(l, r, out_dtype=torch.float32),
                          h.double().t() @ g.double()),
}
for name, (f32, b16, mkl, mkr, mm, ref) in cases.items():
    t32, o32 = timed(f32)
    t16, o16 = timed(b16)
    tl, L = timed(mkl)
    tr, R = timed(mkr)
    t3, o3 = timed(lambda: mm(L, R))
    print("%s fp32 %6.1f us err %.1e | bf16 (incl. casts) %6.1f us err %.1e | split-bf16 gemm %6.1f us (+ split L %5.1f, R %5.1f us) err %.1e"
          % (name, t32, rel(o32, ref), t16, rel(o16, ref), t3, tl, tr, rel(o3.float(), ref)))
