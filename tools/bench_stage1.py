"""Stage 1 of the 1-D upsampling net at a rank's shard of the audio preset (1024 clips: latent grid [1024, 3000, 128]): the direct
kernels (rcb_stage1_1d_*) against the window-GEMM form they replace (cast + rcb_window_gather + library GEMM + LeakyReLU pass;
backward: two GEMMs, column sum, rcb_window_fold, cast).   python tools/bench_stage1.py [B] [g]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import ops, tuning, upsample_fast as UF

tuning.enable_tuned_gemms()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = int(sys.argv[2]) if len(sys.argv) > 2 else 3000


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


torch.manual_seed(0)
W = (torch.randn(64, 128, 5, device="cuda") * 0.04).requires_grad_(True)
b = (torch.randn(64, device="cuda") * 0.1).requires_grad_(True)
x = (torch.randn(B, g, 128, device="cuda") * 0.5).requires_grad_(True)
st = UF.PhaseStage(4, 5, 2, 1)
wbig = ops.phase_bigweight(W.detach(), st.f, st.k, st.pad, torch.bfloat16)
dz = (torch.randn(B, 4 * g, 64, device="cuda") * 0.1).bfloat16()
t_f = timed(lambda: ops.stage1_1d_fwd(x.detach(), wbig, b.detach()))
t_d = timed(lambda: ops.stage1_1d_dgrad(dz, wbig))
t_w = timed(lambda: ops.stage1_1d_wgrad(x.detach(), dz))
nx, ny = x.numel() * 4, dz.numel() * 2
print("direct, %d x %d: fwd %.3f ms (%.2f TB/s)  dgrad %.3f ms (%.2f TB/s)  wgrad %.3f ms (%.2f TB/s)"
      % (B, g, t_f, (nx + ny) / t_f / 1e9, t_d, (nx + ny) / t_d / 1e9, t_w, (nx + ny) / t_w / 1e9), flush=True)


def window_fwd():
    z1 = st.forward_gemm(x, W, b, torch.bfloat16)
    return F.leaky_relu(z1, 0.01)


def window_fwd_bwd():
    y = window_fwd()
    torch.autograd.grad(y, [x, W, b], dz)


def direct_fwd_bwd():
    y = UF._Stage1Direct1dFn.apply(x, W, b, st)
    torch.autograd.grad(y, [x, W, b], dz)


with torch.no_grad():
    t_wf = timed(window_fwd)
t_wfb = timed(window_fwd_bwd)
t_dfb = timed(direct_fwd_bwd)
print("forward + backward of the stage: window-GEMM form %.3f ms (forward alone %.3f), direct %.3f ms" % (t_wfb, t_wf, t_dfb))
