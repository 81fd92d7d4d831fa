"""Exploration for the R-D test: the product trains its own prior on the fixture's data / schedule and compresses the
fixture's test images, several seeds, both precision modes; prints PSNR / bpp next to the reference's."""
import contextlib
import io
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from golden_util import smooth_images, moment_stats
from recombiner_amd import bitstream, drivers, utils

d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "rd_trained_cifar.npz"), allow_pickle=False)
cfg = json.loads(str(d["cfg"]))
dev = "cuda"
Ytr = smooth_images(int(d["n_train"]), cfg["pixel_sizes"], int(d["train_seed"]))
Yte = smooth_images(int(d["n_test"]), cfg["pixel_sizes"], int(d["test_seed"]))
np.testing.assert_allclose(moment_stats(Ytr), d["Y_train_stats"], rtol=1e-9)
np.testing.assert_allclose(moment_stats(Yte), d["Y_test_stats"], rtol=1e-9)
X, _ = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], 1, 3, seed=0)
ntr, nte = Ytr.shape[0], Yte.shape[0]
for ri, rate in enumerate(d["max_bitrate"]):
    ref = d[f"r{ri}_psnr"]
    print(f"rate {rate}: reference {d[f'r{ri}_n_groups']} groups, bpp {float(d[f'r{ri}_bpp']):.3f}, PSNR {ref.mean():.2f} (train {d[f'r{ri}_psnr_train'].mean():.2f}, "
          f"after opt {d[f'r{ri}_psnr_after_opt'].mean():.2f}); final bits {d[f'r{ri}_traj'][-1, 0]:.0f} beta {d[f'r{ri}_traj'][-1, 1]:.2e}")
    for prec in (1, 0):
        for seed in ((42, 1, 2, 3, 4, 5) if prec == 1 else (42, 1)):
            t0 = time.time()
            out = drivers.train_prior(cfg, "cifar", X.to(dev)[None].expand(ntr, -1, -1), Ytr, float(rate), device=dev,
                                      n_em_iter=int(d["n_iter"]), first_epochs=int(d["first_epochs"]), epochs=int(d["epochs"]),
                                      lr=float(d["lr"]), precision=prec, log=lambda *_: None, seed=seed)
            pri = out["priors"]
            ck = drivers.build_checkpoint(out["model"], out["linear_transform"], out["upsample_net"], *pri, out["kl_beta"])
            t1 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                dist, model = drivers.compress(cfg, "cifar", ck, X.to(dev)[None].expand(nte, -1, -1), Yte.to(dev), device=dev,
                                               n_epochs=int(d["n_opt"]), lr=float(d["lr"]), precision=prec,
                                               finetune_epochs=int(d["n_ft"]), seed=seed)
            bpp = bitstream.payload_bits(bitstream.encode(model)) / (nte * 1024)
            tr = np.array(out["trajectory"])
            print(f"   product prec {prec} seed {seed}: groups {ck[0][5]}, bpp {bpp:.3f}, PSNR {np.mean(dist):.2f}; final bits {tr[-1, 0]:.0f} "
                  f"beta {tr[-1, 1]:.2e} mse {tr[-1, 2]:.2e}; train {t1 - t0:.1f} s, compress {time.time() - t1:.1f} s", flush=True)
