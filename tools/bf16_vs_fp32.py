"""Trains the same INRs in fp32 and bf16-operand mode with identical noise; prints loss / PSNR gaps."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import config, utils
from recombiner_amd import prior_model as PM

def smooth_targets(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 32), torch.linspace(-1, 1, 32), indexing="ij")
    out = []
    for _ in range(n):
        img = torch.zeros(3, 32, 32)
        for _ in range(6):
            fx, fy, ph = (torch.rand(3, generator=g) * 6).tolist()
            amp = torch.rand(3, 1, 1, generator=g) * 0.25
            img += amp * torch.sin(fx * xx + fy * yy + ph)
        out.append((img * 0.5 + 0.5).clamp(0, 1).reshape(3, -1).T)
    return torch.stack(out)

def run(precision, n, steps, lr, lowp=False, stage1=False, split=False, wgrad16=False, terms=3, dterms=None, seed=5):
    cfg = config.configs["cifar"]
    X, _ = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, 3)
    Y = smooth_targets(n)
    m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                         cfg["upsample_factors"], cfg["latent_dim"], False, None, None, random_seed=42, device="cuda")
    m.precision = precision
    m.lowp_gemm = lowp
    m.stage1_bf16 = stage1
    m.split_gemm = split
    m.wgrad_bf16 = wgrad16
    m.split_terms = terms
    m.split_dgrad_terms = dterms
    torch.manual_seed(123); lt = PM.LinearTransform(m.dims).cuda()
    torch.manual_seed(124); up = PM.Upsample(2, cfg["paddings"], cfg["layerwise_scale_factors"]).cuda()
    gen = torch.Generator(device="cuda").manual_seed(seed)
    m.noise_source = lambda shape: torch.randn(shape, device="cuda", generator=gen)
    D = m._d_net; s0 = 0.0211547
    pri = [torch.zeros(D).cuda(), torch.full((D,), s0).cuda(), torch.zeros(2, 2, 128).cuda(), torch.full((2, 2, 128), s0).cuda()] + [None] * 4
    mse, kl, elbo = m.train(steps, lr, X.cuda()[None].expand(n, -1, -1), Y.cuda(), *pri, lt, up, 1e-8, training_mappings=True)
    return mse, kl, np.array(elbo)

if __name__ == "__main__":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    variants = (("fp32", dict(precision=0)), ("bf16", dict(precision=1)),
                ("bf16 + bf16 stage-1 GEMMs", dict(precision=1, stage1=True)),
                ("bf16 + stage-1 + split-bf16 A transform", dict(precision=1, stage1=True, split=True)),
                ("... + bf16 A weight gradient", dict(precision=1, stage1=True, split=True, wgrad16=True)),
                ("bf16 + f16/bf16 A-transform GEMMs", dict(precision=1, lowp=True)),
                ("... + bf16 A weight gradient, 2-term split", dict(precision=1, stage1=True, split=True, wgrad16=True, terms=2)),
                ("... 2-term split, 1-term data gradient", dict(precision=1, stage1=True, split=True, wgrad16=True, terms=2, dterms=1)))
    if len(sys.argv) > 4:
        keep = sys.argv[4].split(",")
        variants = tuple(v for i, v in enumerate(variants) if str(i) in keep)
    res = {}
    for name, kw in variants:      # repetition r of every variant sees the same noise stream (seed 5 + r): paired gaps
        ps = [10 * np.log10(1 / run(n=n, steps=steps, lr=1e-3, seed=5 + r, **kw)[0]) for r in range(reps)]
        res[name] = np.array(ps)
        print("%-40s PSNR mean %.3f  std %.3f  (%s)" % (name, np.mean(ps), np.std(ps), " ".join("%.2f" % p for p in ps)), flush=True)
    base = variants[0][0]
    for name in [v[0] for v in variants[1:]]:
        d = res[name] - res[base]
        print("paired gap %s - %s: %.3f +- %.3f dB" % (name, base, d.mean(), d.std(ddof=1) / np.sqrt(reps)))
