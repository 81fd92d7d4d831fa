#!/usr/bin/env python3
"""Times the REFERENCE's own PriorBNNmodel.train on the host cores of the build container (TEST INFRASTRUCTURE; the
reference never travels) and, beside it, the oracle restatement on the same inputs -- the provenance link between the
`cpu_baseline` that bench.py measures on the GPU box (the oracle: kind "port") and the reference itself.
Writes profiles/r02_reference_cpu_timing.json.   python oracle/time_reference.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import config as ref_config          # noqa: E402
import prior_model as ref_prior      # noqa: E402
from oracle import ref_cpu as O      # noqa: E402
from recombiner_amd import utils     # noqa: E402


def run(n, steps, threads):
    torch.set_num_threads(threads)
    cfg = ref_config.configs["cifar"]
    X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, 3, seed=0)
    Xn = X[None].repeat(n, 1, 1)
    s0 = 0.0211547
    m = ref_prior.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                                cfg["upsample_factors"], cfg["latent_dim"], False, None, None, random_seed=42, device="cpu")
    torch.manual_seed(123)
    lt = ref_prior.LinearTransform(m.dims)
    torch.manual_seed(124)
    up = ref_prior.Upsample(2, cfg["paddings"], cfg["layerwise_scale_factors"])
    pri = [torch.zeros(3267), torch.full((3267,), s0), torch.zeros(2, 2, 128), torch.full((2, 2, 128), s0), None, None, None, None]
    m.train(1, 2e-4, Xn, Y, *pri, lt, up, 1e-8, training_mappings=True)
    t0 = time.perf_counter()
    m.train(steps, 2e-4, Xn, Y, *pri, lt, up, 1e-8, training_mappings=True)
    t_ref = (time.perf_counter() - t0) / steps
    geo = O.Geometry.from_config(cfg)
    p = O.init_prior_params(geo, n, seed=42)
    A = O.make_linear_transform(geo.dims, seed=123)
    upo = O.UpsampleNet(geo.data_dim, geo.paddings, geo.layerwise_scale_factors, seed=124)
    O.prior_train(geo, p, Xn, Y, pri, A, upo, 1, 2e-4, 1e-8, True, O.Noise())
    t0 = time.perf_counter()
    O.prior_train(geo, p, Xn, Y, pri, A, upo, steps, 2e-4, 1e-8, True, O.Noise())
    t_or = (time.perf_counter() - t0) / steps
    return {"n_inrs": n, "steps": steps, "reference_ms_per_step": round(t_ref * 1e3, 2), "reference_inr_steps_per_sec": round(n / t_ref, 1),
            "oracle_ms_per_step": round(t_or * 1e3, 2), "oracle_inr_steps_per_sec": round(n / t_or, 1)}


if __name__ == "__main__":
    threads = len(os.sched_getaffinity(0))
    out = {"what": "reference PriorBNNmodel.train (cambridge-mlg/RECOMBINER, CPU, training_mappings=True, CIFAR preset) and the oracle "
                   "restatement, same inputs, build container", "threads": threads, "torch": torch.__version__,
           "cases": [run(16, 30, threads), run(1024, 6, threads)]}
    path = os.path.join(ROOT, "profiles", "r02_reference_cpu_timing.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
