"""Fused SIREN training kernel time vs pixels per INR: the intercept is the per-INR prologue/epilogue (weight
fragments into LDS, cross-wave gradient reduction), the slope the per-tile cost.  python tools/siren_scaling.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import ops
from recombiner_amd.ops import SirenMeta

dev, n = "cuda", 4096
res = []
for P in (128, 256, 512, 1024, 2048):
    meta = SirenMeta(1, P, 16, 16, 3, 32, 3, precision=1)
    xf = torch.randn(P, 16, device=dev)
    pe = (torch.randn(n, P, 16, device=dev) * 0.1).bfloat16()
    y = torch.rand(n, P, 3, device=dev)
    wv = (torch.rand(n, meta.d_net, device=dev) * 2 - 1) * 0.02
    for _ in range(3):
        ops.siren_loss_bwd(xf, pe, wv, y, 1.0 / (3 * P), meta)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.siren_loss_bwd(xf, pe, wv, y, 1.0 / (3 * P), meta)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    res.append((P, t))
    print("P=%5d  %.1f us  (%.2f ns per INR-pixel)" % (P, t, t * 1e3 / (n * P)))
(p0, t0), (p1, t1) = res[-3], res[-1]
slope = (t1 - t0) / (p1 - p0)
print("slope %.4f us/pixel, intercept %.1f us (of %.1f us at P=1024)" % (slope, t0 - slope * p0, res[-2][1]))
