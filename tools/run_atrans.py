"""Launches the A-transform kernels a few times (profiling target).  usage: python tools/run_atrans.py [rows] [terms] [reps] [planes]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recombiner_amd import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
terms = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
sizes = [1056, 1056, 1056, 99]
cum = [0]
for n in sizes:
    cum.append(cum[-1] + n)
slices = list(zip(cum[:-1], cum[1:]))
D = cum[-1]
torch.manual_seed(0)
x = torch.randn(rows, D, device="cuda") * 0.03
dw = torch.randn(rows, D, device="cuda") * 1e-3
A = [torch.randn(n, n, device="cuda") / n ** 0.5 for n in sizes]
out = torch.empty(rows, D, device="cuda")
tr = ops.ATransform(slices, "cuda", terms=terms)
planes = len(sys.argv) > 4 and sys.argv[4] == "planes"          # operands as the producers' (hi, lo) bf16 planes
if planes:
    x, dw = ops.Planes.from_float(x), ops.Planes.from_float(dw)
for _ in range(reps):
    tr.prepare(A)
    tr.forward(x, out)
    tr.dgrad(dw, out)
    if hasattr(tr, "wgrad"):
        tr.wgrad(x, dw)
torch.cuda.synchronize()
print("done")
