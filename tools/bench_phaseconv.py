"""The direct sub-pixel kernels (rcb_phaseconv_fwd / _dgrad / _wgrad) at the shapes of a rank's shard of the patch presets;
time per call and bytes / s over the tensors each call must touch once.  RCB_LIB selects an alternative build (same-box A/B).
    python tools/bench_phaseconv.py [audio|kodak|video ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import ops

# stage-2 input grids [B, *g, 64] (stage 3 works on twice the grid, 64 -> 16 channels)
SHAPES = {"audio": (61440, 200), "kodak": (24, 192, 128), "video": (32, 6, 32, 32), "audio8": (480, 200), "kodak2": (2, 192, 128),
          "video4": (4, 6, 32, 32)}
which = sys.argv[1:] or ["audio", "kodak", "video"]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


for name in which:
    B, g = SHAPES[name][0], list(SHAPES[name][1:])
    nd = len(g)
    for stage, cout in ((2, 64), (3, 16)):
        gs = g if stage == 2 else [2 * v for v in g]
        torch.manual_seed(0)
        x = torch.nn.functional.leaky_relu(torch.randn(B, *gs, 64, device="cuda"), 0.01).bfloat16()
        W = torch.randn(cout, 64, *([3] * nd), device="cuda") * (0.5 / (64 * 3 ** nd) ** 0.5)
        b = torch.randn(cout, device="cuda") * 0.1
        ff, fd = ops.phaseconv_pack(W)
        y = ops.phaseconv_fwd(x, ff, b, cout, stage == 2)
        dy = (torch.randn_like(y.float()) * 0.1).bfloat16()
        nx, ny = x.numel() * 2, y.numel() * 2
        t_f = timed(lambda: ops.phaseconv_fwd(x, ff, b, cout, stage == 2))
        t_d = timed(lambda: ops.phaseconv_dgrad(dy, fd, x))
        t_w = timed(lambda: ops.phaseconv_wgrad(x, dy))
        print("%-7s stage %d (%s x 64 -> %d): fwd %.3f ms (%.2f TB/s)  dgrad %.3f ms (%.2f TB/s)  wgrad %.3f ms (%.2f TB/s)"
              % (name, stage, "x".join(map(str, gs)), cout, t_f, (nx + ny) / t_f / 1e9, t_d, (2 * nx + ny) / t_d / 1e9,
                 t_w, (nx + ny) / t_w / 1e9), flush=True)
        del x, y, dy
