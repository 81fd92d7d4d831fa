"""Test-time optimisation step (TestBNNmodel.train, S = 5) of a patch preset for rocprofv3: a prior from a few training steps on
two datapoints, then `datapoints` test datapoints optimised together.  Eager (no graph): kernels appear individually.
    python tools/prof_testtime.py kodak [datapoints] [width] [steps]"""
import contextlib
import io
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import config, drivers, tuning, utils
from recombiner_amd import prior_model as PM

tuning.enable_tuned_gemms()
name = sys.argv[1] if len(sys.argv) > 1 else "kodak"
n_data = int(sys.argv[2]) if len(sys.argv) > 2 else 8
width = int(sys.argv[3]) if len(sys.argv) > 3 else 48
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
graph = len(sys.argv) > 5 and sys.argv[5] == "graph"
dev = "cuda"
cfg = dict(config.configs[name])
cfg["hidden_dims"] = [width] * len(cfg["hidden_dims"])
per = int(np.prod(cfg["patch_nums"])) if cfg["patch"] else 1
n = 2 * per
X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, cfg["output_dim"], seed=0)
m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                     cfg["upsample_factors"], cfg["latent_dim"], cfg["patch"], cfg["patch_nums"], cfg["hierarchical_patch_nums"],
                     random_seed=42, device=dev)
m.precision = 1
torch.manual_seed(1)
lt = PM.LinearTransform(m.dims).to(dev)
up = PM.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"]).to(dev)
s0, D, lat = 0.0211547, m._d_net, list(m.lpe_loc.shape[1:])
pri = [torch.zeros(D, device=dev), torch.full((D,), s0, device=dev), torch.zeros(lat, device=dev), torch.full(lat, s0, device=dev)]
pri += ([torch.zeros(D, device=dev), torch.full((D,), s0, device=dev)] * 2) if cfg["patch"] else [None] * 4
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m.train(12, 1e-3, X.to(dev)[None].expand(n, -1, -1), Y.to(dev), *pri, lt, up, 1e-6, training_mappings=True)
ck = drivers.build_checkpoint(m, lt, up, *pri, 1e-6)
del m
Xn, Yn = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n_data * per, cfg["output_dim"], seed=3)
with contextlib.redirect_stdout(io.StringIO()):
    tm = drivers.build_test_model(cfg, name, ck, n_data * per, dev, 42)
tm.precision = 1
if not graph:
    tm.use_graph = False
Xd, Yd = Xn.to(dev)[None].expand(n_data * per, -1, -1), Yn.to(dev)
tm.train(Xd, Yd, 6, torch.optim.Adam(tm.parameters(), lr=2e-4), False, sample_size=5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
tm.train(Xd, Yd, steps, torch.optim.Adam(tm.parameters(), lr=2e-4), False, sample_size=5)
e1.record()
torch.cuda.synchronize()
print("test-time %s w%d, %d datapoints = %d INRs x 5 samples: %.3f ms/step (%s)" % (name, width, n_data, n_data * per,
                                                                                    e0.elapsed_time(e1) / steps, "graph" if graph else "eager"))
