"""Debug aid: where does the wave-per-row SIREN kernel differ from the workgroup kernel?  python tools/dbg_wave.py [N] [tiles]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import _lib, ops, utils
from recombiner_amd.ops import SirenMeta
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kt = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # 2: the wave family whatever the number of rows
dev = "cuda"
lib = _lib.load()
P = 1024
X, Y = utils.synthetic_inputs([32, 32], 16, n, 3, seed=0)
meta = SirenMeta(1, P, 16, 16, 3, 32, 3, precision=1)
Xd, Yd = X.to(dev), Y.to(dev)
torch.manual_seed(1)
pe = (torch.randn(n, P, 16, device=dev) * 0.1).bfloat16()
wv = torch.empty(n, (meta.d_net + 31) // 32 * 32, device=dev)[:, :meta.d_net]
wv.copy_((torch.rand(n, meta.d_net, device=dev) * 2 - 1) * 0.02)
xf16 = ops.xf_bf16(Xd)
def run(v):
    lib.rcb_debug_siren_wave_tiles(v)
    o = ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / (3 * P), meta, want_bf16=True, xf16=xf16)
    torch.cuda.synchronize()
    return o
a, b = run(0), run(kt)
print("sse rel diff per row", ((a[0] - b[0]).abs() / a[0]).cpu().numpy())
d = (a[2].float() - b[2].float()).abs().amax(dim=2)        # [n, P]
scale = a[2].float().abs().max()
print("dpe: max diff per 32-pixel tile / max, row 0:", (d[0].view(32, 32).amax(dim=1) / scale).cpu().numpy().round(4))
print("dpe: per-pixel-in-tile max over tiles, row 0:", (d[0].view(32, 32).amax(dim=0) / scale).cpu().numpy().round(4))
print("dpe: max diff per feature, row 0:", ((a[2].float() - b[2].float()).abs()[0].amax(dim=0) / scale).cpu().numpy().round(4))
dw = (a[1] - b[1]).abs()
sizes = [32 * 33, 32 * 33, 32 * 33, 3 * 33]
o = 0
for l, s in enumerate(sizes):
    seg = dw[:, o:o + s]
    ref = a[1][:, o:o + s].abs().max()
    no = 32 if l < 3 else 3
    print(f"layer {l}: bias grad max diff/max {float(seg[:, :no].max() / ref):.3e}  weight grad {float(seg[:, no:].max() / ref):.3e}  per row {(seg.amax(dim=1) / ref).cpu().numpy().round(3)}")
    if l < 3:
        wd = seg[0, no:].view(32, 32)      # [in i][out o]
        print("   row 0: max over out per in-feature:", (wd.amax(dim=1) / ref).cpu().numpy().round(3))
        print("   row 0: max over in per out-feature:", (wd.amax(dim=0) / ref).cpu().numpy().round(3))
    o += s
