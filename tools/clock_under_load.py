"""Shader clock and package power while ONE kernel of the training step runs back to back (rocm-smi sampled from a thread):
    python tools/clock_under_load.py
The LDS / MFMA-dense kernels of the step do not run at the 2.4 GHz the peak figures assume."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import ops, utils
from recombiner_amd.ops import SirenMeta

dev = "cuda"
B = 4096
torch.manual_seed(0)
# operands of each kernel
dy3 = (torch.randn(B, 32, 32, 16, device=dev) * 1e-3).bfloat16()
h2 = torch.randn(B, 16, 16, 64, device=dev).bfloat16()
w3 = torch.randn(2, 2, 64, 2, 2, 16, device=dev) * 0.05
z1 = torch.randn(B, 8, 8, 64, device=dev).bfloat16()
w2 = torch.randn(2, 2, 64, 2, 2, 64, device=dev) * 0.05
b2 = torch.randn(64, device=dev) * 0.1
dy2 = (torch.randn(B, 16, 16, 64, device=dev) * 1e-3).bfloat16()
X, Y = utils.synthetic_inputs([32, 32], 16, B, 3, seed=0)
Xd, Yd = X.to(dev), Y.to(dev)
meta = SirenMeta(1, 1024, 16, 16, 3, 32, 3, precision=1)
pe = (torch.randn(B, 1024, 16, device=dev) * 0.1).bfloat16()
wv = torch.empty(B, (meta.d_net + 31) // 32 * 32, device=dev)[:, :meta.d_net]
wv.copy_((torch.rand(B, meta.d_net, device=dev) * 2 - 1) * 0.02)
xf16 = ops.xf_bf16(Xd)
big = torch.randn(64 * 1024 * 1024, device=dev)
ga, gb = torch.randn(4096, 4096, device=dev).bfloat16(), torch.randn(4096, 4096, device=dev).bfloat16()
cases = [
    ("idle", None),
    ("siren_wave_kernel (rcb_siren_loss_bwd)", lambda: ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / 3072, meta, want_bf16=True, xf16=xf16)),
    ("upconv_bwd3_fused_kernel", lambda: ops.upconv_bwd_fused(dy3, w3, h2, 16, 16)),
    ("upconv_fwd2_reg_kernel", lambda: ops.upconv_fwd(z1, w2, b2, 8, 64, False, preact=True)),
    ("upconv_dgrad2_reg_kernel", lambda: ops.upconv_dgrad(dy2, w2, z1, 8, 64, preact=True)),
    ("upconv_fwd3_lds_kernel", lambda: ops.upconv_fwd(h2, w3, torch.zeros(16, device=dev), 16, 16, False, linear_bf16=True)),
    ("streaming copy (torch, 256 MB)", lambda: big.add_(1.0)),
    ("bf16 GEMM 4096^3 (hipBLASLt)", lambda: torch.mm(ga, gb)),
]


def sample(out, stop):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        except Exception as e:      # noqa: BLE001
            out.append(("error", str(e)))
            return
        sclk = re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
        pw = re.findall(r"Power \(W\): ([0-9.]+)", txt)
        out.append((int(sclk[0]) if sclk else None, float(pw[0]) if pw else None))
        time.sleep(0.2)


for name, fn in cases:
    rec, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(rec, stop))
    t0 = time.time()
    n = 0
    if fn is not None:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    th.start()
    while time.time() - t0 < 4.0:
        if fn is None:
            time.sleep(0.1)
        else:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
    stop.set()
    th.join()
    rec = [r for r in rec[1:] if r[0] != "error"]
    sc = [r[0] for r in rec if r[0]]
    pw = [r[1] for r in rec if r[1]]
    print("%-42s sclk MHz %s  power W %s  (%d samples, %d launches)" % (
        name, ("%d..%d" % (min(sc), max(sc))) if sc else "n/a", ("%.0f..%.0f" % (min(pw), max(pw))) if pw else "n/a", len(rec), n))
