"""Throughput of the test-time path on synthetic priors: S=5 posterior optimisation steps and A* encode rounds
for a CIFAR test batch (N=500 images, main_compression.py defaults).  python tools/bench_compress.py [bf16|fp32]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import config, utils
from recombiner_amd import prior_model as PM, test_model as TM, tuning

tuning.enable_tuned_gemms()

prec = 1 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else 0
cfg = config.configs["cifar"]
N, dev = 500, "cuda"
X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], N, 3, seed=0)
dims = [cfg["input_dim"]] + cfg["hidden_dims"] + [cfg["output_dim"]]
torch.manual_seed(123); lt = PM.LinearTransform(dims).to(dev)
torch.manual_seed(124); up = PM.Upsample(2, cfg["paddings"], cfg["layerwise_scale_factors"]).to(dev)
D = 3267 + 512
rng = np.random.RandomState(0)
bits = rng.gamma(0.7, 6.0, size=D).astype(np.float32)          # ~4 bits/param -> ~4 params per 16-bit group
gi, gs, ge, g2p, p2g, G, gk, w = PM.get_grouping_by_kl(bits)
p_loc = torch.zeros(D); p_ls = torch.full((D,), -2.0)
m = TM.TestBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], N, cfg["upsample_factors"], cfg["latent_dim"],
                    2, cfg["pixel_sizes"], False, None, None, "cifar", linear_transform=lt, upsample_net=up,
                    p_loc=p_loc[p2g], p_log_scale=p_ls[p2g], init_log_scale=torch.full((D,), -4.0), param_to_group=p2g,
                    group_to_param=g2p, n_groups=G, group_start_index=gs, group_end_index=ge, group_idx=gi, device=dev,
                    initial_beta=1e-8)
m.precision = prec
if len(sys.argv) > 2 and sys.argv[2] == "nosplit":
    m.split_gemm = False
with torch.no_grad():
    m.loc.add_(0.02 * torch.randn_like(m.loc))
Xd, Yd = X.to(dev)[None].expand(N, -1, -1), Y.to(dev)
opt = torch.optim.Adam(m.parameters(), lr=2e-4)
m.train(Xd, Yd, 24, opt, False, sample_size=5)       # warm-up: also captures the two step graphs
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 200
m.train(Xd, Yd, steps, torch.optim.Adam(m.parameters(), lr=2e-4), False, sample_size=5)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("groups %d; test-time training: %.2f ms/step (N=500, S=5) = %.0f INR-steps/s (%.0f INR-sample-steps/s)" % (
    G, dt / steps * 1e3, N * steps / dt, 5 * N * steps / dt))
for glen in np.unique(ge - gs):           # one-off: Sobol/ppf candidate tables for every group length, Gumbel table
    m._table(m._l1, int(glen), 65536)
m._encode_round(m._l1, True, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
rounds = 10
for r in range(rounds):
    m._encode_round(m._l1, True, r + 1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("A* encode round (500 rows, K=65536, batched): %.2f ms/round = %.0f group-encodes/s" % (dt / rounds * 1e3, N * rounds / dt))
