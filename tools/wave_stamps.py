"""In-kernel phase stamps of the wave-per-row SIREN kernel (diagnostic build, -DRCB_WAVE_STAMPS):
    python -m recombiner_amd.build --variant wstamps --only siren_mlp_wave.hip -DRCB_WAVE_STAMPS=1
    RCB_LIB=recombiner_amd/lib/librcb_wstamps.so python tools/wave_stamps.py [N=4096]
Wave 0 of workgroup 0, its LAST row: prologue, the phases of its first tiles (forward layers, backward layers), epilogue; s_memtime ticks."""
import ctypes as C
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import _lib, ops, utils
from recombiner_amd.ops import SirenMeta
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
dev, P = "cuda", 1024
X, Y = utils.synthetic_inputs([32, 32], 16, n, 3, seed=0)
meta = SirenMeta(1, P, 16, 16, 3, 32, 3, precision=1)
Xd, Yd = X.to(dev), Y.to(dev)
pe = (torch.randn(n, P, 16, device=dev) * 0.1).bfloat16()
wv = torch.empty(n, (meta.d_net + 31) // 32 * 32, device=dev)[:, :meta.d_net]
wv.copy_((torch.rand(n, meta.d_net, device=dev) * 2 - 1) * 0.02)
xf16 = ops.xf_bf16(Xd)
lib.rcb_debug_siren_wave_tiles(1)
for _ in range(5):
    ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / (3 * P), meta, want_bf16=True, xf16=xf16)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert raw.rcb_debug_wave_stamps(buf, 256) == 0
st = np.array(buf[:], dtype=np.int64)
print("shader clock GHz", (st[3] - st[0]) / ((st[5] - st[4]) * 10.0), " row us", (st[5] - st[4]) / 100.0)
print("prologue", st[1] - st[0], " tile loop", st[2] - st[1], " epilogue", st[3] - st[2], " row", st[3] - st[0])
print("   prologue phases: DMA of the first half issued + landed", st[6] - st[0], " its fragments", st[7] - st[6], " second half landed", st[8] - st[7], " its fragments", st[9] - st[8], " rest (accumulators, first fetch)", st[1] - st[9])
for k in range(6):
    b = st[10 + 12 * k: 10 + 12 * k + 9]
    print(f"tile {k}: forward layers", np.diff(b[:5]), " backward layers", np.diff(b[4:9]), " tile", b[8] - b[0], " to next", st[10 + 12 * (k + 1)] - b[0])
