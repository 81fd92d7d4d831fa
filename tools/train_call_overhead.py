"""Per-call overhead of PriorBNNmodel.train (bf16 mode, graph replay): wall time of calls with K steps -> slope and intercept."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recombiner_amd import config, utils, prior_model as PM, tuning
tuning.enable_tuned_gemms()
DEV = "cuda"
cfg = config.configs["cifar"]; n = 4096
X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, 3, seed=0)
Xd, Yd = X.to(DEV)[None].expand(n, -1, -1), Y.to(DEV)
m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                     cfg["upsample_factors"], cfg["latent_dim"], False, None, None, random_seed=42, device=DEV)
m.precision = 1
torch.manual_seed(123); lt = PM.LinearTransform(m.dims).to(DEV)
torch.manual_seed(124); up = PM.Upsample(2, cfg["paddings"], cfg["layerwise_scale_factors"]).to(DEV)
D, s0 = m._d_net, 0.0211547
pri = [torch.zeros(D, device=DEV), torch.full((D,), s0, device=DEV), torch.zeros(2, 2, 128, device=DEV), torch.full((2, 2, 128), s0, device=DEV)] + [None] * 4
def call(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.train(k, 2e-4, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
call(10); call(10)
ks = [5, 10, 20, 40, 80]
ts = {k: min(call(k) for _ in range(4)) for k in ks}
A = np.polyfit(ks, [ts[k] for k in ks], 1)
print({k: round(v, 3) for k, v in ts.items()})
print("per step %.4f ms, per call %.3f ms" % (A[0], A[1]))
