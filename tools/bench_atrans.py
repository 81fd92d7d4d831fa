"""Times the hand-written A-transform kernels (atrans.hip).
usage: python tools/bench_atrans.py [rows] [terms] [pad floats per row]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recombiner_amd import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
terms = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sizes = [1056, 1056, 1056, 99]
cum = [0]
for n in sizes:
    cum.append(cum[-1] + n)
slices = list(zip(cum[:-1], cum[1:]))
D = cum[-1]
dev = "cuda"
torch.manual_seed(0)
pad = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # extra floats per row (1: rows of 3268 floats = 16-byte aligned)
x = (torch.randn(rows, D + pad, device=dev) * 0.03)[:, :D]
dw = (torch.randn(rows, D + pad, device=dev) * 1e-3)[:, :D]
A = [torch.randn(n, n, device=dev) / n ** 0.5 for n in sizes]
out = torch.empty(rows, D, device=dev)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tr = ops.ATransform(slices, dev, terms=terms)
t_pack = timeit(lambda: tr.prepare(A))
t_fwd = timeit(lambda: tr.forward(x, out))
t_dg = timeit(lambda: tr.dgrad(dw, out))
flops = 2.0 * rows * sum(n * n for n in sizes) * terms
print(f"atrans rows={rows} terms={terms} pad={pad}: pack {t_pack:.1f} us, forward {t_fwd:.1f} us ({flops / t_fwd / 1e6:.0f} TFLOP/s), "
      f"dgrad {t_dg:.1f} us ({flops / t_dg / 1e6:.0f} TFLOP/s)")
xp, dp = ops.Planes.from_float(x.contiguous()), ops.Planes.from_float(dw.contiguous())
t_fwd_p = timeit(lambda: tr.forward(xp, out))
t_dg_p = timeit(lambda: tr.dgrad(dp, out))
print(f"  operands as (hi, lo) planes: forward {t_fwd_p:.1f} us ({flops / t_fwd_p / 1e6:.0f} TFLOP/s), dgrad {t_dg_p:.1f} us "
      f"({flops / t_dg_p / 1e6:.0f} TFLOP/s)")
t_wg_p = timeit(lambda: tr.wgrad(xp, dp))
print(f"  wgrad on the planes {t_wg_p:.1f} us")
h16 = x.bfloat16()
d16 = dw.bfloat16()
t_wg = timeit(lambda: tr.wgrad(x, dw, h16, d16, True))
print(f"wgrad (batched bf16 library GEMM on the producers' copies + narrow-layer kernel) {t_wg:.1f} us")
