"""In-kernel phase stamps of the wave-per-row SIREN kernel (diagnostic build, -DRCB_WAVE_STAMPS):
    python -m recombiner_amd.build --variant wstamps --only siren_mlp_wave.hip -DRCB_WAVE_STAMPS=1
    RCB_LIB=recombiner_amd/lib/librcb_wstamps.so python tools/wave_stamps.py [tiles=2] [N=4096]
Wave 0 of workgroup 0, its first row: prologue, every step of rounds 0 and 1 (forward j, backward j), epilogue; s_memtime ticks."""
import ctypes as C
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import _lib, ops, utils
from recombiner_amd.ops import SirenMeta
kt = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
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
lib.rcb_debug_siren_wave_tiles(kt)
for _ in range(5):
    ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / (3 * P), meta, want_bf16=True, xf16=xf16)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert raw.rcb_debug_wave_stamps(buf, 256) == 0
st = np.array(buf[:], dtype=np.int64)
NL = 4
nj = NL * kt + 1
print("prologue", st[1] - st[0], " row total", st[101] - st[0], " epilogue", st[101] - st[100])
for rnd in range(2):
    f = st[2 + rnd * 40: 2 + rnd * 40 + nj]
    b = st[2 + rnd * 40 + 20: 2 + rnd * 40 + 20 + nj]
    print(f"round {rnd}: forward steps", np.diff(f), " fwd->bwd", b[0] - f[-1], " backward steps", np.diff(b), " round total", b[-1] - f[0])
print("round 0 start -> round 1 start", st[2 + 40] - st[2])
