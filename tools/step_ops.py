"""Which torch operators (with input shapes) the eager training step of a preset spends GPU time in -- the glue around the HIP
kernels: python tools/step_ops.py video [datapoints] [width] [prec]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import config, tuning, utils
from recombiner_amd import prior_model as PM
from torch.profiler import ProfilerActivity, profile

tuning.enable_tuned_gemms()
name = sys.argv[1] if len(sys.argv) > 1 else "video"
n_data = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = dict(config.configs[name])
if len(sys.argv) > 3:
    cfg["hidden_dims"] = [int(sys.argv[3])] * 3
prec = int(sys.argv[4]) if len(sys.argv) > 4 else 1
per = int(np.prod(cfg["patch_nums"])) if cfg["patch"] else 1
n = n_data * per
X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, cfg["output_dim"], seed=0)
m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                     cfg["upsample_factors"], cfg["latent_dim"], cfg["patch"], cfg["patch_nums"], cfg["hierarchical_patch_nums"],
                     random_seed=42, device="cuda")
m.precision = prec
m.use_graph = False
torch.manual_seed(1)
lt = PM.LinearTransform(m.dims).cuda()
up = PM.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"]).cuda()
s0, D, lat = 0.0211547, m._d_net, list(m.lpe_loc.shape[1:])
pri = [torch.zeros(D).cuda(), torch.full((D,), s0).cuda(), torch.zeros(lat).cuda(), torch.full(lat, s0).cuda()]
pri += ([torch.zeros(D).cuda(), torch.full((D,), s0).cuda()] * 2) if cfg["patch"] else [None] * 4
Xd, Yd = X.cuda()[None].expand(n, -1, -1), Y.cuda()
m.train(4, 2e-4, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=True)
torch.cuda.synchronize()
steps = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    m.train(steps, 2e-4, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=True)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    us = float(getattr(e, "self_device_time_total", 0.0)) / steps
    if us > 1.0:
        rows.append((us, e.count / steps, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
print("%s: eager step, operators by self GPU time" % name)
for us, c, k, sh in rows[:45]:
    print("%8.1f us  x%-4.1f %-34s %s" % (us, c, k[:34], sh))
