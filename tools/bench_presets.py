"""Prior-training step time of every reference preset in the bf16 throughput mode (production path: no injected noise,
graph replay).   python tools/bench_presets.py"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import config, tuning, utils
from recombiner_amd import prior_model as PM

tuning.enable_tuned_gemms()
if os.environ.get("RCB_PHASE_ALL"):        # A/B: torch-level phase form also where MIOpen is the default (audio)
    PM.phase_form_preferred = lambda *a: True
# (label, preset, datapoints (images / clips / ...), hidden width, precision mode): the five reference presets, then
# BASELINE.json's width variants (configs[2]: Kodak patches at width 48; configs[4]: video at width 64 in f16)
RUNS = [("cifar", "cifar", 4096, 32, 1), ("protein", "protein", 4096, 32, 1), ("kodak", "kodak", 2, 32, 1),
        ("audio", "audio", 8, 32, 1), ("video", "video", 4, 32, 1),
        ("kodak-w48", "kodak", 2, 48, 1), ("video-w64-f16", "video", 4, 64, 2), ("cifar-w64", "cifar", 4096, 64, 1)]
only = {a.split(":")[0]: (int(a.split(":")[1]) if ":" in a else None) for a in sys.argv[1:]}     # label[:datapoints]
for label, name, n_data, width, prec in RUNS:
    if only and label not in only:
        continue
    if only and only[label]:
        n_data = only[label]
    cfg = dict(config.configs[name])
    cfg["hidden_dims"] = [width] * len(cfg["hidden_dims"])
    per = int(np.prod(cfg["patch_nums"])) if cfg["patch"] else 1
    n = n_data * per
    X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, cfg["output_dim"], seed=0)
    m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                         cfg["upsample_factors"], cfg["latent_dim"], cfg["patch"], cfg["patch_nums"],
                         cfg["hierarchical_patch_nums"], random_seed=42, device="cuda")
    m.precision = prec
    torch.manual_seed(1)
    lt = PM.LinearTransform(m.dims).cuda()
    up = PM.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"]).cuda()
    s0 = 0.0211547
    D = m._d_net
    lat = list(m.lpe_loc.shape[1:])
    pri = [torch.zeros(D).cuda(), torch.full((D,), s0).cuda(), torch.zeros(lat).cuda(), torch.full(lat, s0).cuda()]
    pri += ([torch.zeros(D).cuda(), torch.full((D,), s0).cuda()] * 2) if cfg["patch"] else [None] * 4
    Xd, Yd = X.cuda()[None].expand(n, -1, -1), Y.cuda()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.train(8, 2e-4, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 30
        mse, kl, elbo = m.train(steps, 2e-4, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    graph = m._ws is not None and m._ws["graphs"] is not None
    px = int(np.prod(cfg["pixel_sizes"]))
    print("%-14s %5d INRs (%d datapoints) x %5d px: %7.2f ms/step = %9.0f INR-steps/s, %6.1f Mpx-steps/s; graph replay: %s; finite: %s%s"
          % (label, n, n_data, px, dt * 1e3, n / dt, n * px / dt / 1e6, graph, bool(np.isfinite(elbo).all()),
             ("; warnings: " + "; ".join(str(x.message)[:80] for x in w)) if w else "")
          + "; peak HBM %.1f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
    del m, lt, up, Xd, Yd, X, Y, pri
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
