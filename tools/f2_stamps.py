"""Where the stage-2 forward kernel spends its time (diagnostic build):
   python -m recombiner_amd.build --variant f2 --only upconv.hip -DRCB_B3_STAMPS=1 -DRCB_F2_STAMPS=1
   RCB_LIB=.../librcb_f2.so python tools/f2_stamps.py
Per wave of workgroup 0, s_memtime ticks summed over its passes (4 INRs each): first barrier, staging, second barrier,
gathers + MFMAs of its four (INR, tile) pairs, their epilogues (bias, LeakyReLU, lane swap, stores)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recombiner_amd import ops, _lib

B = 4096
x = torch.randn(B, 8, 8, 64, device="cuda").bfloat16()
# the production path: effective weights and their pre-ordered MFMA fragments from the conv weights (rcb_upconv_weff_build)
W1 = torch.randn(64, 128, 5, 5, device="cuda") * 0.02
W2c = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
W3c = torch.randn(16, 64, 3, 3, device="cuda") * 0.05
_, _, weff2_, weff3_, pack = ops.upconv_weff_build(W1, torch.zeros(64, device="cuda"), W2c, W3c, True)
weff = weff2_
bias = torch.randn(64, device="cuda") * 0.1
for _ in range(5):
    ops.upconv_fwd(x, weff, bias, 8, 64, False, preact=True, pack=pack)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.upconv_fwd(x, weff, bias, 8, 64, False, preact=True, pack=pack)
e1.record()
torch.cuda.synchronize()
print("avg us per launch:", e0.elapsed_time(e1) * 100)
lib = _lib.load()
try:
    f = lib.rcb_debug_f2_stamps
except AttributeError:
    sys.exit("not a stamps build")
buf = (C.c_uint64 * 64)()
assert f(buf, 64) == 0
raw = np.array(buf, dtype=np.int64).reshape(8, 8)
st = raw[:, :5]
wall_us = (raw[:, 6] - raw[:, 5]) / 100.0        # s_memrealtime: 100 MHz
print("workgroup 0: prologue (weights, LDS clear, first fetch issued) %.1f us;" % ((raw[:, 5] - raw[:, 7]).mean() / 100.0), end=" ")
print("pass loop %.1f us by the 100 MHz wall clock -> s_memtime ticks at %.2f GHz" % (
    wall_us.mean(), float(st.sum(1).mean() / (wall_us.mean() * 1e3))))
np.set_printoptions(linewidth=200)
print("ticks per wave [barrier1, staging, barrier2, gathers + MFMAs, epilogues], summed over the workgroup's passes:")
print(st)
print("share of the wave's time:", (st / st.sum(1, keepdims=True)).round(3).mean(0))
print("ticks total per wave:", st.sum(1))
