"""Launches the fused SIREN kernel alone (for rocprofv3): python tools/run_siren.py [bf16|fp32] [N] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import ops, utils
from recombiner_amd.ops import SirenMeta

prec = 1 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda"
X, Y = utils.synthetic_inputs([32, 32], 16, n, 3, seed=0)
meta = SirenMeta(1, 1024, 16, 16, 3, 32, 3, precision=prec)
Xd, Yd = X.to(dev), Y.to(dev)
pe = torch.randn(n, 1024, 16, device=dev) * 0.1
wv = (torch.rand(n, meta.d_net, device=dev) * 2 - 1) * 0.02
for _ in range(reps):
    sse, dw, dpe = ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / 3072, meta)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / 3072, meta)
e1.record()
torch.cuda.synchronize()
print("avg ms", e0.elapsed_time(e1) / reps, "sse", float(sse.sum()))
