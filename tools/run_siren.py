"""Launches the fused SIREN kernel alone (for rocprofv3): python tools/run_siren.py [bf16|f16|fp32] [N] [reps] [pe16] [width] [step]
`step`: as the training step launches it (rows on 128-byte lines, bf16 copy of the gradient written by the epilogue)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import ops, utils
from recombiner_amd.ops import SirenMeta

prec = {"bf16": 1, "f16": 2}.get(sys.argv[1] if len(sys.argv) > 1 else "fp32", 0)
width = int(sys.argv[5]) if len(sys.argv) > 5 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda"
X, Y = utils.synthetic_inputs([32, 32], 16, n, 3, seed=0)
meta = SirenMeta(1, 1024, 16, 16, 3, width, 3, precision=prec)
Xd, Yd = X.to(dev), Y.to(dev)
pe = torch.randn(n, 1024, 16, device=dev) * 0.1
wv = (torch.rand(n, meta.d_net, device=dev) * 2 - 1) * 0.02
as_step = len(sys.argv) > 6 and sys.argv[6] == "step" and prec != 0
kw = {}
if as_step:
    wv_p = torch.empty(n, (meta.d_net + 31) // 32 * 32, device=dev)[:, :meta.d_net]
    wv_p.copy_(wv)
    wv = wv_p
    kw = dict(want_bf16=True, xf16=ops.xf_bf16(Xd, prec) if prec in (1, 2) else None)
if len(sys.argv) > 4 and sys.argv[4] == "pe16":
    pe = pe.bfloat16()


def timed(fn):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


t, out = timed(lambda: ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / 3072, meta, **kw))
print("loss_bwd (pe %s): avg ms %.4f  sse %.6f" % (pe.dtype, t, float(out[0].sum())))
t, _ = timed(lambda: ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / 3072, meta, want_dpe=False))
print("loss_bwd without dpe: avg ms %.4f" % t)
t, _ = timed(lambda: ops.siren_fwd(Xd, pe, wv, meta))
print("fwd: avg ms %.4f" % t)
