"""Stability of the 16-bit training paths over a longer run: every preset, 10 x 200 Adam steps (graph replay), mean ELBO per
block must be finite and improve.   python tools/long_run.py [preset ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import config, tuning, utils
from recombiner_amd import prior_model as PM

tuning.enable_tuned_gemms()
RUNS = [("cifar-4096", "cifar", 4096, 32, 1),      # (the batch size at which the one-wave-per-INR SIREN family runs)
        ("cifar", "cifar", 512, 32, 1), ("protein", "protein", 512, 32, 1), ("kodak", "kodak", 1, 32, 1), ("audio", "audio", 4, 32, 1),
        ("video", "video", 2, 32, 1), ("kodak-w48", "kodak", 1, 48, 1), ("video-w64-f16", "video", 2, 64, 2)]
only = sys.argv[1:]
bad = 0
for label, name, n_data, width, prec in RUNS:
    if only and label not in only:
        continue
    cfg = dict(config.configs[name])
    cfg["hidden_dims"] = [width] * 3
    n = n_data * (int(np.prod(cfg["patch_nums"])) if cfg["patch"] else 1)
    X, _ = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, cfg["output_dim"], seed=0)
    # smooth targets (a few sinusoids of the first Fourier features): something an INR can actually fit
    g = torch.Generator().manual_seed(3)
    Wt = torch.randn(X.shape[1], cfg["output_dim"], generator=g) * 0.3
    Y = (0.5 + 0.25 * torch.tanh(X @ Wt))[None].repeat(n, 1, 1) + 0.02 * torch.randn(n, X.shape[0], cfg["output_dim"], generator=g)
    m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                         cfg["upsample_factors"], cfg["latent_dim"], cfg["patch"], cfg["patch_nums"],
                         cfg["hierarchical_patch_nums"], random_seed=42, device="cuda")
    m.precision = prec
    torch.manual_seed(1)
    lt = PM.LinearTransform(m.dims).cuda()
    up = PM.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"]).cuda()
    s0, D, lat = 0.0211547, m._d_net, list(m.lpe_loc.shape[1:])
    pri = [torch.zeros(D).cuda(), torch.full((D,), s0).cuda(), torch.zeros(lat).cuda(), torch.full(lat, s0).cuda()]
    pri += ([torch.zeros(D).cuda(), torch.full((D,), s0).cuda()] * 2) if cfg["patch"] else [None] * 4
    Xd, Yd = X.cuda()[None].expand(n, -1, -1), Y.cuda()
    means, mses = [], []
    for blk in range(10):
        mse, kl, elbo = m.train(200, 2e-4, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=True)
        means.append(float(np.mean(elbo)))
        mses.append(mse)
    ok = bool(np.isfinite(means).all() and means[-1] > means[0] and mses[-1] < mses[0])
    bad += not ok
    print("%-14s %5d INRs, 2000 steps: mean ELBO per 200-step block %s; MSE/INR first %.4g last %.4g (PSNR %.1f dB) -> %s"
          % (label, n, " ".join("%.4g" % v for v in means), mses[0], mses[-1], 10 * np.log10(1 / max(mses[-1], 1e-12)),
             "ok" if ok else "NOT IMPROVING"), flush=True)
sys.exit(1 if bad else 0)
