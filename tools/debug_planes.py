"""debug: base vs operand_planes training, per-step differences"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recombiner_amd import config, utils
from recombiner_amd import prior_model as PM
DEV = "cuda"
cfg = config.configs["cifar"]
n = 8
X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, 3, seed=2)
Xd, Yd = X.to(DEV)[None].expand(n, -1, -1), Y.to(DEV)
res = {}
for tag, planes in (("base", False), ("planes", True)):
    torch.manual_seed(77)
    m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                         cfg["upsample_factors"], cfg["latent_dim"], False, None, None, random_seed=42, device=DEV)
    m.precision, m.operand_planes, m.use_graph = 1, planes, False
    torch.manual_seed(123)
    lt = PM.LinearTransform(m.dims).to(DEV)
    torch.manual_seed(124)
    up = PM.Upsample(2, cfg["paddings"], cfg["layerwise_scale_factors"]).to(DEV)
    D, s0 = m._d_net, 0.0211547
    pri = [torch.zeros(D, device=DEV), torch.full((D,), s0, device=DEV), torch.zeros(2, 2, 128, device=DEV),
           torch.full((2, 2, 128), s0, device=DEV)] + [None] * 4
    snaps = []
    for k in range(4):
        m.train(1, 2e-4, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=True)
        snaps.append([m.loc.detach().clone(), m.log_scale.detach().clone(), m.lpe_loc.detach().clone()] + [p.detach().clone() for p in lt.parameters()]
                     + [p.detach().clone() for p in up.parameters()])
    res[tag] = snaps
names = ["loc", "log_scale", "lpe_loc", "A0", "A1", "A2", "A3"] + ["up%d" % i for i in range(6)]
for k in range(4):
    print("after call", k + 1, " ".join("%s %.2e" % (nm, float((a - b).abs().max())) for nm, a, b in zip(names, res["base"][k], res["planes"][k])))
