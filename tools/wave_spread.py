"""Spread of the wave-per-row SIREN kernel's workgroups in time (rcb_siren_desc.clock_probe: wave 0 of the first 256 workgroups
stamps s_memtime / s_memrealtime at its start and at its end):   python tools/wave_spread.py [N=4096]
Prints, in us relative to the earliest start: start and end percentiles, per-workgroup duration and clock, and the kernel time by events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import _lib, ops, utils
from recombiner_amd.ops import SirenMeta

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev, P = "cuda", 1024
lib = _lib.load()
X, Y = utils.synthetic_inputs([32, 32], 16, n, 3, seed=0)
meta = SirenMeta(1, P, 16, 16, 3, 32, 3, precision=1)
Xd, Yd = X.to(dev), Y.to(dev)
pe = (torch.randn(n, P, 16, device=dev) * 0.1).bfloat16()
wv = torch.empty(n, (meta.d_net + 31) // 32 * 32, device=dev)[:, :meta.d_net]
wv.copy_((torch.rand(n, meta.d_net, device=dev) * 2 - 1) * 0.02)
xf16 = ops.xf_bf16(Xd)
probe = torch.zeros(256, 4, device=dev, dtype=torch.int64)
for it in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / (3 * P), meta, want_bf16=True, xf16=xf16)
    e0.record()
    ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / (3 * P), meta, want_bf16=True, xf16=xf16, clock_probe=probe)
    e1.record()
    torch.cuda.synchronize()
    if it < 3:
        continue
    p = probe.cpu().double().numpy()
    t0 = p[:, 1].min()
    st, en = (p[:, 1] - t0) / 100.0, (p[:, 3] - t0) / 100.0
    dur = en - st
    clk = (p[:, 2] - p[:, 0]) / (dur * 1e-6) / 1e9
    q = lambda v: " ".join("%7.1f" % x for x in np.percentile(v, [0, 10, 50, 90, 100]))
    print("kernel %.1f us by events | start %s | end %s | duration %s | clock GHz %s" % (e0.elapsed_time(e1) * 1e3, q(st), q(en), q(dur), q(clk)))
    xcd = np.arange(256) % 8
    print("   end by XCD (median):", " ".join("%6.1f" % np.median(en[xcd == k]) for k in range(8)), "| clock by XCD:", " ".join("%5.2f" % np.median(clk[xcd == k]) for k in range(8)))
