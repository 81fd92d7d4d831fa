"""Same-process A/B of the two width-32 bf16 SIREN loss / backward families (rcb_debug_siren_wave_tiles):
    python tools/ab_siren_wave.py [N=4096] [rounds=5] [reps=20]
0 = one workgroup per row (siren_mlp_bf16.hip), 1 = one wave per row (siren_mlp_wave.hip).
Checks each against the fp32 kernel (exact fp32 products) on the same inputs, then times them interleaved, launched as the
training step launches them (rows on 128-byte lines, bf16 copy of the gradient, bf16 pe / dpe, bf16 coordinate grid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import _lib, ops, utils
from recombiner_amd.ops import SirenMeta

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
variants = [int(v) for v in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["0", "1"])]
grid = [int(v) for v in (sys.argv[5].split("x") if len(sys.argv) > 5 else ["32", "32"])]
P = grid[0] * grid[1]
dev = "cuda"
lib = _lib.load()
X, Y = utils.synthetic_inputs(grid, 16, n, 3, seed=0)
meta = SirenMeta(1, P, 16, 16, 3, 32, 3, precision=1)
meta32 = SirenMeta(1, P, 16, 16, 3, 32, 3, precision=0)
Xd, Yd = X.to(dev), Y.to(dev)
torch.manual_seed(1)
pe = (torch.randn(n, P, 16, device=dev) * 0.1).bfloat16()
wv_p = torch.empty(n, (meta.d_net + 31) // 32 * 32, device=dev)[:, :meta.d_net]
wv_p.copy_((torch.rand(n, meta.d_net, device=dev) * 2 - 1) * 0.02)
xf16 = ops.xf_bf16(Xd)


def run(v, **kw):
    lib.rcb_debug_siren_wave_tiles(v)
    return ops.siren_loss_bwd(Xd, pe, wv_p, Yd, 1.0 / (3 * P), meta, want_bf16=True, xf16=xf16, **kw)


# reference: fp32 kernel on the bf16-rounded inputs
m = min(n, 64)
ref = ops.siren_loss_bwd(Xd, pe[:m].float(), wv_p[:m].contiguous(), Yd[:m], 1.0 / (3 * P), meta32)
for v in variants:
    out = run(v)
    torch.cuda.synchronize()
    sse, dw, dpe, dw16 = out
    e_sse = float(((sse[:m] - ref[0]).abs() / ref[0].abs().clamp_min(1e-6)).max())
    e_dw = float((dw[:m] - ref[1]).abs().max() / ref[1].abs().max())
    e_dpe = float((dpe[:m].float() - ref[2]).abs().max() / ref[2].abs().max())
    e16 = float((dw16[:m].float() - dw[:m]).abs().max() / dw[:m].abs().max())
    print(f"variant {v}: rel err vs fp32 kernel: sse {e_sse:.2e} dw {e_dw:.2e} dpe {e_dpe:.2e}; bf16 copy {e16:.2e}; finite {bool(torch.isfinite(dw).all())}")
    if v == variants[0]:
        base = out
    else:
        print(f"   vs variant {variants[0]}: dw max diff / max {float((dw - base[1]).abs().max() / base[1].abs().max()):.2e}, "
              f"dpe {float((dpe.float() - base[2].float()).abs().max() / base[2].float().abs().max()):.2e}, sse {float(((sse - base[0]).abs() / base[0]).max()):.2e}")

times = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        for _ in range(3):
            run(v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run(v)
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / reps * 1000)
for v in variants:
    t = sorted(times[v])
    print(f"variant {v}: us per launch median {t[len(t) // 2]:.1f} min {t[0]:.1f} max {t[-1]:.1f}  ({n} rows, {P} pixels)")
lib.rcb_debug_siren_wave_tiles(0)
