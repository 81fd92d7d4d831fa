"""Wall-clock (s_memrealtime, 100 MHz) stamps of workgroup 0 in the stage-2 weight-gradient kernel and the stage-3 forward:
kernel entry -> loop start (prologue) -> loop end -> kernel end (epilogue), against the launch time by events.
   python -m recombiner_amd.build --variant wc --only upconv.hip -DRCB_B3_STAMPS=1 -DRCB_WC_STAMPS=1 ;  RCB_LIB=.../librcb_wc.so python tools/wc_stamps.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recombiner_amd import ops, _lib

B = 4096
W1 = torch.randn(64, 128, 5, 5, device="cuda") * 0.02
W2c = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
W3c = torch.randn(16, 64, 3, 3, device="cuda") * 0.05
_, _, weff2, weff3, pack = ops.upconv_weff_build(W1, torch.zeros(64, device="cuda"), W2c, W3c, True)
z1 = torch.randn(B, 8, 8, 64, device="cuda").bfloat16()
dy2 = (torch.randn(B, 16, 16, 64, device="cuda") * 1e-3).bfloat16()
h2 = torch.randn(B, 16, 16, 64, device="cuda").bfloat16()
b3 = torch.zeros(16, device="cuda")
lib = _lib.load()


def t(fn, n=10):
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


us_w = t(lambda: ops.upconv_wgrad(z1, dy2, 8, 64, preact=True))
us_f = t(lambda: ops.upconv_fwd(h2, weff3, b3, 16, 16, False, linear_bf16=True, pack=pack))
buf = (C.c_uint64 * 8)()
assert lib.rcb_debug_wc_stamps(buf, 8) == 0
st = np.array(buf, dtype=np.int64).reshape(2, 4)
for name, us, r in (("stage-2 weight gradient (+ slab reduce)", us_w, st[0]), ("stage-3 forward", us_f, st[1])):
    print("%-40s launch %.1f us | workgroup 0: prologue %.1f, loop %.1f, epilogue %.1f us" % (
        name, us, (r[1] - r[0]) / 100.0, (r[2] - r[1]) / 100.0, (r[3] - r[2]) / 100.0))
