"""Where the stage-2 data-gradient kernel spends its cycles (diagnostic build):
   python -m recombiner_amd.build --variant d2 --only upconv.hip -DRCB_B3_STAMPS=1 -DRCB_D2_STAMPS=1
   RCB_LIB=.../librcb_d2.so python tools/d2_stamps.py
Per wave of workgroup 0, s_memtime ticks summed over its 16 INRs: barrier A, staging, barrier B, MFMA loop (incl. prefetch
issue), exchange write, barrier C, epilogue."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recombiner_amd import ops, _lib

B = 4096
dy = (torch.randn(B, 16, 16, 64, device="cuda") * 1e-3).bfloat16()
x = torch.randn(B, 8, 8, 64, device="cuda").bfloat16()
# the production path: effective weights and their pre-ordered MFMA fragments from the conv weights (rcb_upconv_weff_build)
W1 = torch.randn(64, 128, 5, 5, device="cuda") * 0.02
W2c = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
W3c = torch.randn(16, 64, 3, 3, device="cuda") * 0.05
_, _, weff2_, weff3_, pack = ops.upconv_weff_build(W1, torch.zeros(64, device="cuda"), W2c, W3c, True)
weff = weff2_
for _ in range(5):
    out = ops.upconv_dgrad(dy, weff, x, 8, 64, preact=True, pack=pack)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.upconv_dgrad(dy, weff, x, 8, 64, preact=True, pack=pack)
e1.record()
torch.cuda.synchronize()
print("avg us per launch:", e0.elapsed_time(e1) * 100)
lib = _lib.load()
try:
    f = lib.rcb_debug_d2_stamps
except AttributeError:
    sys.exit("not a stamps build")
buf = (C.c_uint64 * 64)()
assert f(buf, 64) == 0
raw = np.array(buf, dtype=np.int64).reshape(8, 8)
st = raw[:, :7].copy()
prologue_us, loop_us = raw[:, 2].mean() / 100.0, raw[:, 7].mean() / 100.0       # s_memrealtime: 100 MHz (slot 2: no barrier B any more)
st[:, 2] = 0
print("workgroup 0: prologue %.1f us, INR loop %.1f us by the 100 MHz wall clock -> s_memtime ticks at %.2f GHz" % (
    prologue_us, loop_us, float(st.sum(1).mean() / (loop_us * 1e3))))
np.set_printoptions(linewidth=200)
print("ticks per wave [barrier A, staging, barrier B, MFMA loop, exchange write, barrier C, epilogue]:")
print(st)
print("share:", (st / st.sum(1, keepdims=True)).round(3).mean(0), " total per wave:", st.sum(1))
