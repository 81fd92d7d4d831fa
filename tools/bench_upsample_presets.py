"""fwd+bwd of the upsampling net on the stitched latent grids of the other presets: torch.nn (MIOpen) vs the torch-level
phase form (UpsampleFast), fp32 and bf16 autocast.   python tools/bench_upsample_presets.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import config
from recombiner_amd import prior_model as PM
from recombiner_amd.upsample_fast import UpsampleFast

dev = "cuda"
BATCH = {"kodak": 4, "audio": 16, "video": 4, "protein": 4096}
for name, nb in BATCH.items():
    c = config.configs[name]
    dd = c["data_dim"]
    lat = [c["pixel_sizes"][i] // c["upsample_factors"][i] for i in range(dd)]
    grid = [lat[i] * (c["patch_nums"][i] if c["patch"] else 1) for i in range(dd)]
    torch.manual_seed(0)
    net = PM.Upsample(dd, c["paddings"], c["layerwise_scale_factors"]).to(dev)
    fast = UpsampleFast(net)
    x = torch.randn(nb, 128, *grid, device=dev, requires_grad=True)
    with torch.no_grad():
        out_shape = net(x[:1]).shape[1:]
    g = torch.randn(nb, *out_shape, device=dev)

    def run(fn, ac, reps=3):
        def once():
            if ac:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = fn(x)
            else:
                y = fn(x)
            return torch.autograd.grad(y, [x] + list(net.parameters()), g.to(y.dtype))
        for _ in range(2):
            r = once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            once()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, r

    res = {}
    only = sys.argv[1:]
    if only and name not in only:
        continue
    xs = {"": x}
    if dd == 2:
        xs["cl "] = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    variants = [("nn fp32", net, False), ("nn bf16", net, True), ("phase fp32", fast, False), ("phase bf16", fast, True)]
    if dd == 2:
        variants += [("cl nn fp32", net, False), ("cl nn bf16", net, True), ("cl phase bf16", fast, True)]
    if dd == 2 and c["patch"]:
        from recombiner_amd.upsample_fast import stitched2d_module
        variants += [("cl hip tiles", stitched2d_module(net), False)]
    x_nchw = x
    for label, fn, ac in variants:
        x = xs["cl "] if label.startswith("cl ") else x_nchw
        try:
            res[label] = run(fn, ac)
        except Exception as e:      # noqa: BLE001
            res[label] = (float("nan"), None)
            print(name, label, "failed:", repr(e)[:120])
    ref = res["nn fp32"][1]
    line = "%-8s grid %-14s batch %-5d out %-18s" % (name, grid, nb, tuple(out_shape))
    for label, (ms, r) in res.items():
        err = max(((a.float() - b).abs().max() / (b.abs().max() + 1e-20)).item() for a, b in zip(r, ref)) if r is not None else float("nan")
        line += "  %s %.2f ms (err %.0e)" % (label, ms, err)
    print(line, flush=True)
