"""Phase breakdown of the fused bf16 SIREN kernel from in-kernel s_memtime stamps (DIAGNOSTIC build, -DRCB_SIREN_STAMPS:
built here into gpurun_out/librcb_stamps.so, loaded through RCB_LIB; the shipped library carries no stamps).
    python tools/siren_stamps.py            (on the GPU box)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
lib = os.path.join(out_dir, "librcb_stamps.so")
from recombiner_amd import build as B  # noqa: E402
cmd = [B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DRCB_SIREN_STAMPS", "-Wno-unused-result"] + \
      [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", lib]
subprocess.check_call(cmd)
os.environ["RCB_LIB"] = lib
import torch  # noqa: E402
from recombiner_amd import _lib, ops, utils  # noqa: E402
from recombiner_amd.ops import SirenMeta  # noqa: E402

WIDE = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] == "wide" else 0      # python tools/siren_stamps.py wide 64 [n]
if WIDE:
    del sys.argv[1:3]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X, Y = utils.synthetic_inputs([32, 32], 16, n, 3, seed=0)
meta = SirenMeta(1, 1024, 16, 16, 3, WIDE or 32, 3, precision=1)
Xd, Yd = X.cuda(), Y.cuda()
pe = (torch.randn(n, 1024, 16, device="cuda") * 0.1).bfloat16()
wv = (torch.rand(n, meta.d_net, device="cuda") * 2 - 1) * 0.02
for _ in range(5):
    ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / 3072, meta)
torch.cuda.synchronize()
L = _lib.load()
nb = min(n, 8192)
buf = (C.c_uint64 * (nb * 16))()
rc = (L.rcb_debug_read_stamps_wide if WIDE else L.rcb_debug_read_stamps)(buf, nb * 16)
assert rc == 0, rc
st = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 16).astype(np.int64)
# persistent kernel: the stamps of a workgroup's LAST row survive.  1 = row start (its weights were requested during the row
# before), 2 = fragments stored (+sync), 3 = wave 0 done with its tiles, 4 = all waves done, 5 = partials in LDS and the next
# row's weights requested (+sync), 6 = gradients summed and stored, 7 = row end (+sync)
names = ["fragments: wait for the prefetched weights, convert, store (+sync)", "tile loop (wave 0: 8 tiles)", "wait for the other waves",
         "partials -> LDS, request next weights (+sync)", "reduce + dw / split stores", "sse, sync"]
nb = min(nb, 512)
st = st[:nb]
tot = (st[:, 7] - st[:, 1]).astype(np.float64)
print("workgroups %d; shader-clock ticks per row (last row of each workgroup): median %.0f" % (nb, np.median(tot)))
for k, nm in enumerate(names):
    d = (st[:, k + 2] - st[:, k + 1]).astype(np.float64)
    print("  %-72s median %8.0f  (%5.1f %%)   p10 %8.0f  p90 %8.0f" % (nm, np.median(d), 100 * np.median(d) / np.median(tot),
                                                                      np.percentile(d, 10), np.percentile(d, 90)))
# first tile of wave 0: 14 = tile start, 8 = after forward, 9..12 = before backward layer 3,2,1,0, 15 = start of the wave's second tile
t = st
if WIDE:
    # wide kernel: 0 = launch, 2 = fragments built, 3 = passes done, 6 = gradient tiles stored, 7 = end
    for nm, a_, b_ in [("weights staged, fragments built", 0, 2), ("passes over the pixel tiles", 2, 3), ("gradient tiles -> HBM", 3, 6), ("sse", 6, 7)]:
        d = (st[:, b_] - st[:, a_]).astype(np.float64)
        print("  wide: %-50s median %8.0f  (%5.1f %% of %0.f)" % (nm, np.median(d), 100 * np.median(d) / np.median(st[:, 7] - st[:, 0]), np.median(st[:, 7] - st[:, 0])))
    print("  wide: wave 0 inside the passes' barriers, summed       median %8.0f  (%5.1f %% of the passes)" % (
        np.median(st[:, 5]), 100 * np.median(st[:, 5]) / np.median(st[:, 3] - st[:, 2])))
seg = [("forward (3 sine layers + output)", 14, 8), ("loss / dz", 8, 9), ("backward layer 3", 9, 10), ("backward layer 2", 10, 11),
       ("backward layer 1", 11, 12), ("backward layer 0 (+dpe)", 12, 15)]
tile = (t[:, 15] - t[:, 14]).astype(np.float64)
print("first tile of wave 0: median %.0f ticks" % np.median(tile))
for nm, a, b in seg:
    d = (t[:, b] - t[:, a]).astype(np.float64)
    print("  %-34s median %7.0f  (%5.1f %%)" % (nm, np.median(d), 100 * np.median(d) / np.median(tile)))
# whole-kernel view: every workgroup starts at launch; the kernel ends with the slowest one
life = (st[:, 7] - st[:, 0]).astype(np.float64)
print("workgroup lifetime (launch -> last row done), ticks: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f  (max / median = %.3f)" % (
    life.min(), np.percentile(life, 10), np.median(life), np.percentile(life, 90), life.max(), life.max() / np.median(life)))
