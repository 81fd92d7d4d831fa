"""Phase-conv kernel time vs batch: the intercept is the per-workgroup prologue (weight fragments into registers),
the slope the streaming cost per INR.   python tools/upconv_scaling.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import ops

dev = "cuda"
bf = torch.bfloat16


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


W2 = torch.randn(2, 2, 64, 2, 2, 64, device=dev) * 0.05
W3 = torch.randn(2, 2, 64, 2, 2, 16, device=dev) * 0.05
b2, b3 = torch.zeros(64, device=dev), torch.zeros(16, device=dev)
use_pack = not (len(sys.argv) > 1 and sys.argv[1] == "nopack")
pack = None
if use_pack:      # fragments pre-packed by the effective-weight kernel (what the training step does)
    cw = [torch.randn(64, 128, 5, 5, device=dev) * 0.05, torch.zeros(64, device=dev), torch.randn(64, 64, 3, 3, device=dev) * 0.05,
          torch.randn(16, 64, 3, 3, device=dev) * 0.05]
    _, _, W2, W3, pack = ops.upconv_weff_build(*cw, True)
print("fragment pack:", use_pack)
rows = {}
for B in (256, 1024, 4096, 8192):
    z1 = torch.randn(B, 8, 8, 64, device=dev).to(bf)
    h2 = torch.randn(B, 16, 16, 64, device=dev).to(bf)
    dz2 = torch.randn(B, 16, 16, 64, device=dev).to(bf)
    dpe = torch.randn(B, 32, 32, 16, device=dev).to(bf)
    t = {
        "fwd2": timed(lambda: ops.upconv_fwd(z1, W2, b2, 8, 64, out_f32=False, preact=True, pack=pack)),
        "fwd3": timed(lambda: ops.upconv_fwd(h2, W3, b3, 16, 16, out_f32=False, linear_bf16=True, pack=pack)),
        "dgrad2": timed(lambda: ops.upconv_dgrad(dz2, W2, z1, 8, 64, preact=True, pack=pack)),
        "dgrad3": timed(lambda: ops.upconv_dgrad(dpe, W3, h2, 16, 16, pack=pack)),
        "wgrad2": timed(lambda: ops.upconv_wgrad(z1, dz2, 8, 64, preact=True)),
        "wgrad3": timed(lambda: ops.upconv_wgrad(h2, dpe, 16, 16)),
    }
    rows[B] = t
    print("B=%5d  " % B + "  ".join("%s %6.1f" % kv for kv in t.items()), flush=True)
for k in rows[4096]:
    slope = (rows[8192][k] - rows[4096][k]) / 4096
    print("%-7s slope %.4f us/INR   intercept %.1f us (of %.1f us at B=4096)" % (k, slope, rows[4096][k] - slope * 4096, rows[4096][k]))
