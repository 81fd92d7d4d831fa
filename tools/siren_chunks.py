"""SIREN fused loss + backward at few rows: one workgroup per row vs pixel tiles split over several (pixel_chunks).
python tools/siren_chunks.py [rows] [pixels] [width]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import ops
from recombiner_amd.ops import SirenMeta

G = int(sys.argv[1]) if len(sys.argv) > 1 else 192
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
W = int(sys.argv[3]) if len(sys.argv) > 3 else 32
meta = SirenMeta(1, P, 16, 16, 3, W, 3, precision=1)
dev = "cuda"
xf = torch.rand(P, 16, device=dev) * 2 - 1
pe = (torch.randn(G, P, 16, device=dev) * 0.1).bfloat16()
wv = (torch.rand(G, meta.d_net, device=dev) * 2 - 1) * 0.02
y = torch.rand(G, P, 3, device=dev)
for c in (1, 2, 3, 4, 6, 8, 16):
    if c > (P + 31) // 32:
        continue
    f = lambda: ops.siren_loss_bwd(xf, pe, wv, y, 1.0 / (3 * P), meta, want_bf16=True, pixel_chunks=c)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    print("rows %d px %d width %d chunks %2d: %.1f us (kernel + reduction)" % (G, P, W, c, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
