"""Weight gradient of the 1-D upsampling stages at a rank's shard of the audio preset (configs[3]: 1024 clips = 61 440 INRs):
stage 2 (x [B, 200, 64], dy [B, 400, 64]) and stage 3 (x [B, 400, 64], dy [B, 800, 16]); time per call and bytes / s over the
operands read once.   python tools/bench_wgrad1d.py [n_inrs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import ops

CASES = [(int(sys.argv[1]), 200, 64), (int(sys.argv[1]), 400, 16)] if len(sys.argv) > 1 else \
    [(61440, 200, 64), (61440, 400, 16), (480, 200, 64), (480, 400, 16), (4096, 24, 64), (4096, 48, 16)]   # audio shard, 8 clips, protein
for B, g, cout in CASES:
    torch.manual_seed(0)
    x = torch.randn(B, g, 64, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(B, 2 * g, cout, device="cuda", dtype=torch.bfloat16) * 0.1
    for _ in range(3):
        dW, db = ops.phaseconv_wgrad(x, dy)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        dW, db = ops.phaseconv_wgrad(x, dy)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    nbytes = x.numel() * 2 + dy.numel() * 2
    print("wgrad 1-D g %d cout %d, %d INRs: %.3f ms per call (kernel + slab sum + fold), %.2f TB/s over %.2f GB of operands"
          % (g, cout, B, ms, nbytes / ms / 1e9, nbytes / 1e9), flush=True)
    del x, dy
