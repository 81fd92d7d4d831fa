"""In-kernel timeline of the A-transform chunk loop (diagnostic build -DATRANS_STAMPS=1, selected with RCB_LIB).
Stamps of workgroup 0: 0 top of iteration, 1 after the vmcnt wait, 2 after the barrier, 3 at the middle block (before the
wait for the next x pieces), 4 after the last block's MFMAs were issued, 5 end of iteration (after the conversion)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recombiner_amd import ops, _lib

rows = 4096
sizes = [1056, 1056, 1056, 99]
cum = [0]
for n in sizes:
    cum.append(cum[-1] + n)
slices = list(zip(cum[:-1], cum[1:]))
pad = int(sys.argv[1]) if len(sys.argv) > 1 else 0
x = (torch.randn(rows, cum[-1] + pad, device="cuda") * 0.03)[:, :cum[-1]]
A = [torch.randn(n, n, device="cuda") / n ** 0.5 for n in sizes]
out = torch.empty(rows, cum[-1], device="cuda")
tr = ops.ATransform(slices, "cuda", terms=2)
tr.prepare(A)
for _ in range(20):
    tr.forward(x, out)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_uint64 * (8 * 40 * 6))()
assert lib.rcb_debug_atrans_stamps(buf, 8 * 40 * 6) == 0
st = np.array(buf, dtype=np.int64).reshape(8, 40, 6)
t0 = st[:, 0, 0].min()
np.set_printoptions(linewidth=200)
print("per-wave iteration length (stamp 0 to next stamp 0), iterations 5..25, mean per wave:", np.diff(st[:, 5:26, 0], axis=1).mean(1).round())
for w in (0, 4, 3):
    d = st[w, 5:25]
    print(f"wave {w}: mean cycles  wait {np.mean(d[:, 1] - d[:, 0]):.0f}  barrier {np.mean(d[:, 2] - d[:, 1]):.0f}  "
          f"first half {np.mean(d[:, 3] - d[:, 2]):.0f}  second half {np.mean(d[:, 4] - d[:, 3]):.0f}  convert {np.mean(d[:, 5] - d[:, 4]):.0f}")
print("iteration 10, all waves, stamps relative to wave 0 stamp 0:")
print(st[:, 10, :] - st[0, 10, 0])
