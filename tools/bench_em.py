"""Wall time of the EM prior-training loop (drivers.train_prior: train -> beta rule -> prior refit [-> grouping +
checkpoint every 10 iterations]) on synthetic CIFAR-shaped data, bf16 mode.   python tools/bench_em.py [N] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import config, drivers, utils

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = config.configs["cifar"]
X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, 3, seed=0)
Xd, Yd = X.cuda()[None].expand(n, -1, -1), Y.cuda()
stamps = []


def log(msg):
    torch.cuda.synchronize()
    stamps.append(time.perf_counter())


t0 = time.perf_counter()
out = drivers.train_prior(cfg, "cifar", Xd, Yd, max_bitrate=0.5, device="cuda", n_em_iter=iters, first_epochs=200, epochs=100,
                          lr=2e-4, precision=1, checkpoint_path=None, checkpoint_every=1, log=log)
torch.cuda.synchronize()
t1 = time.perf_counter()
d = [b - a for a, b in zip(stamps[1:-1], stamps[2:])]          # steady-state iterations (100 steps each)
per_iter = sum(d) / len(d)
print("EM loop, N=%d: %d iterations in %.2f s; steady-state iteration (100 Adam steps + beta rule + prior refit + log) %.1f ms"
      % (n, iters, t1 - t0, per_iter * 1e3))
sched = (2 + 549) * per_iter                                   # reference schedule: 200 + 549 x 100 steps
print("reference schedule (200 + 549 x 100 steps) would take %.1f s -> %.1f INRs trained / s on one GPU" % (sched, n / sched))
