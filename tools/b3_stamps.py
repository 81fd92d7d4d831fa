"""Where the stage-3 backward kernel spends its cycles (diagnostic build):
   python -m recombiner_amd.build --variant b3 --only upconv.hip -DRCB_B3_STAMPS=1 ;  RCB_LIB=.../librcb_b3.so python tools/b3_stamps.py
Per wave of workgroup 0, cycles (s_memtime, 100 MHz ticks) summed over its 16 INRs: first barrier, staging, second barrier,
data gradient, weight gradient."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recombiner_amd import ops, _lib

B = 4096
dy = (torch.randn(B, 32, 32, 16, device="cuda") * 1e-3).bfloat16()
x = torch.randn(B, 16, 16, 64, device="cuda").bfloat16()
# the production path: effective weights and their pre-ordered MFMA fragments from the conv weights (rcb_upconv_weff_build)
W1 = torch.randn(64, 128, 5, 5, device="cuda") * 0.02
W2c = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
W3c = torch.randn(16, 64, 3, 3, device="cuda") * 0.05
_, _, weff2_, weff3_, pack = ops.upconv_weff_build(W1, torch.zeros(64, device="cuda"), W2c, W3c, True)
weff = weff3_
for _ in range(5):
    out = ops.upconv_bwd_fused(dy, weff, x, 16, 16, pack=pack)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.upconv_bwd_fused(dy, weff, x, 16, 16, pack=pack)
e1.record()
torch.cuda.synchronize()
print("avg us per launch (incl. slab reduction):", e0.elapsed_time(e1) * 100)
lib = _lib.load()
if hasattr(lib, "rcb_debug_b3_stamps") or True:
    try:
        f = lib.rcb_debug_b3_stamps
    except AttributeError:
        sys.exit("not a stamps build")
    buf = (C.c_uint64 * 64)()
    assert f(buf, 64) == 0
    raw = np.array(buf, dtype=np.int64).reshape(8, 8)
    st = raw[:, :5]
    wall_us = (raw[:, 6] - raw[:, 5]) / 100.0        # s_memrealtime: 100 MHz
    end_wg = raw[7, 7]
    print("workgroup 0: prologue %.1f us, epilogue (bias sums, slab stores) %.1f us;" % (
        (raw[:7, 5] - raw[:7, 7]).mean() / 100.0, (end_wg - raw[0, 6]) / 100.0), end=" ")
    print("INR loop %.1f us by the 100 MHz wall clock -> s_memtime ticks at %.2f GHz" % (
        wall_us.mean(), float(st.sum(1).mean() / (wall_us.mean() * 1e3))))
    np.set_printoptions(linewidth=200)
    print("ticks per wave [barrier1, staging, barrier2, dgrad, wgrad], summed over the workgroup's INRs:")
    print(st)
    print("share of the wave's time:", (st / st.sum(1, keepdims=True)).round(3).mean(0))
    print("ticks total per wave:", st.sum(1))
