"""The reference's two drivers, restated call for call on the DROP-IN modules -- run with dropin/ first on the path:

    PYTHONPATH=dropin:. python tools/dropin_sequence.py <scratch dir>

Imports `config`, `prior_model`, `test_model`, `utils` by the reference's module names (they resolve to dropin/), then
follows main_prior_training.py:53-73 (construction, `.to(device)`), :114-172 (train -> beta rule -> prior refit), :186-338
(grouping, the eight pickle.dump calls incl. `linear_transform.cpu()` ... `.to(device)` on the LIVE modules, then further
training) and main_compression.py:37-167 (eight plain pickle.load calls, prior re-ordering, TestBNNmodel, optimise, compress).
Prints 'DROPIN OK ...'."""
import os
import pickle
import sys

import numpy as np
import torch

from config import configs                                            # noqa: E402  (dropin/config.py)
from prior_model import LinearTransform, PriorBNNmodel, Upsample, get_grouping   # noqa: E402  (dropin/prior_model.py)
from test_model import TestBNNmodel                                   # noqa: E402  (dropin/test_model.py)
from utils import metric                                               # noqa: E402  (dropin/utils.py)
import torch.nn.functional as F                                       # noqa: E402


def main(scratch):
    import prior_model as pm_mod
    assert os.path.basename(os.path.dirname(os.path.abspath(pm_mod.__file__))) == "dropin", pm_mod.__file__
    device = "cuda"
    config = configs["cifar"]
    in_dim, hidden_dims, out_dim = config["input_dim"], config["hidden_dims"], config["output_dim"]
    from recombiner_amd.utils import synthetic_inputs               # data loading is out of scope: synthetic tensors
    n = 6
    Xg, Y = synthetic_inputs(config["pixel_sizes"], config["fourier_dim"], n, out_dim, seed=1)
    X = Xg[None].repeat(n, 1, 1).to(device)
    Y = Y.to(device)
    train_size = X.shape[0]
    prior_model = PriorBNNmodel(in_dim=in_dim, hidden_dims=hidden_dims, out_dim=out_dim, train_size=train_size,
                                data_dim=config["data_dim"], pixel_sizes=config["pixel_sizes"],
                                upsample_factors=config["upsample_factors"], latent_dim=config["latent_dim"],
                                patch=config["patch"], patch_nums=config["patch_nums"],
                                hierarchical_patch_nums=config["hierarchical_patch_nums"], random_seed=42, device=device,
                                init_log_scale=-4, c=6., w0=30.).to(device)
    prior_model.precision = 1          # the one line a user adds to select the bf16 throughput mode (graph replay)
    linear_transform = LinearTransform(prior_model.dims).to(device)
    upsample_net = Upsample(kernel_dim=config["data_dim"], paddings=config["paddings"],
                            layerwise_scale_factors=config["layerwise_scale_factors"]).to(device)
    prior_loc = torch.zeros_like(prior_model.loc[0]).to(device)
    prior_scale = torch.ones_like(prior_model.loc[0]).to(device) * F.softplus(torch.tensor(-2.), beta=1, threshold=20) / 6
    prior_lpe_loc = torch.zeros_like(prior_model.lpe_loc[0]).to(device)
    prior_lpe_scale = torch.ones_like(prior_model.lpe_loc[0]).to(device) * F.softplus(torch.tensor(-2.), beta=1, threshold=20) / 6
    kl_beta, n_epoch, ELBOs = 1e-8, 8, []
    path = os.path.join(scratch, "PRIOR_train_size_%d_max_bitrate=%.3f.pkl" % (train_size, 0.5))
    a_track = []
    for it in range(3):
        _, _, _ELBOs = prior_model.train(n_epoch, 2e-4, X, Y, prior_loc, prior_scale, prior_lpe_loc, prior_lpe_scale, None, None,
                                         None, None, linear_transform, upsample_net, kl_beta, training_mappings=True, verbose=False)
        ELBOs = ELBOs + _ELBOs
        n_epoch = 6
        with torch.no_grad():
            kls = prior_model.calculate_kl(prior_loc, prior_scale, prior_lpe_loc, prior_lpe_scale, None, None, None, None)
            kls = kls.item() / np.log(2.) / train_size
            if kls > 0.5 * 1024:
                kl_beta = kl_beta * 1.5
            kl_beta = min(max(kl_beta, 1e-20), 1)
            prior_loc = prior_model.loc.clone().detach().mean(0)
            prior_scale = ((prior_model.st(prior_model.log_scale.clone().detach()) ** 2).mean(0) + prior_model.loc.clone().detach().var(0)) ** 0.5
            prior_lpe_loc = prior_model.lpe_loc.clone().detach().mean(0)
            prior_lpe_scale = ((prior_model.st(prior_model.lpe_log_scale.clone().detach()) ** 2).mean(0) + prior_model.lpe_loc.clone().detach().var(0)) ** 0.5
            y_hat = prior_model.forward(X, linear_transform, upsample_net, False)
            assert torch.isfinite(y_hat).all()
            average_training_log_scale = prior_model.log_scale.clone().detach().mean(0).cpu()
            average_training_lpe_log_scale = prior_model.lpe_log_scale.clone().detach().mean([0]).flatten().cpu()
            q_loc = torch.cat([prior_model.loc.flatten(start_dim=1), prior_model.lpe_loc.flatten(start_dim=1)], -1)
            q_scale = torch.cat([prior_model.st(prior_model.log_scale).flatten(start_dim=1),
                                 prior_model.st(prior_model.lpe_log_scale).flatten(start_dim=1)], -1)
            p_loc = torch.cat([prior_loc.flatten(), prior_lpe_loc.flatten()])
            p_scale = torch.cat([prior_scale.flatten(), prior_lpe_scale.flatten()])
            G = get_grouping(q_loc, q_scale, p_loc, p_scale)
        with open(path, "wb") as f:
            pickle.dump(tuple(G), f)
            pickle.dump((p_loc.cpu(), p_scale.cpu(), kl_beta, torch.cat([average_training_log_scale, average_training_lpe_log_scale])), f)
            pickle.dump((None,) * 8, f)
            pickle.dump((None, None, kl_beta, None), f)
            pickle.dump((None,) * 8, f)
            pickle.dump((None, None, kl_beta, None), f)
            pickle.dump(linear_transform.cpu(), f)            # the LIVE modules travel to the CPU and back, as upstream
            pickle.dump(upsample_net.cpu(), f)
        linear_transform.to(device)
        upsample_net.to(device)
        a_track.append(linear_transform.A[0].detach().clone())
    # the mappings kept training across the checkpoints (a stale captured graph would have frozen them)
    assert float((a_track[2] - a_track[1]).abs().max()) > 1e-6 and float((a_track[1] - a_track[0]).abs().max()) > 1e-6
    assert len(ELBOs) == 8 + 6 + 6 and np.isfinite(ELBOs).all()
    blob = open(path, "rb").read()
    assert b"prior_model" in blob and b"recombiner_amd" not in blob and b"_rcb_" not in blob

    # ---- main_compression.py:37-167 ----
    with open(path, "rb") as f:
        group_idx, group_start_index, group_end_index, group2param, param2group, n_groups, group_kls, weights = pickle.load(f)
        prior_loc, prior_scale, kl_beta, average_training_log_scale = pickle.load(f)
        for _ in range(4):
            pickle.load(f)
        linear_transform = pickle.load(f)
        upsample_net = pickle.load(f)
    _p_locs = prior_loc.clone()[param2group].to(device)
    _p_log_scales = torch.log(torch.exp(prior_scale * 6) - 1).clone()[param2group].to(device)
    _average_training_log_scale = average_training_log_scale[param2group].cpu().detach()
    x, y = X[:2], Y[:2]
    recombiner = TestBNNmodel(in_dim=in_dim, hidden_dims=hidden_dims, out_dim=out_dim, number_of_datapoints=x.shape[0],
                              upsample_factors=config["upsample_factors"], latent_dim=config["latent_dim"],
                              data_dim=config["data_dim"], pixel_sizes=config["pixel_sizes"], patch=config["patch"],
                              patch_nums=config["patch_nums"], hierarchical_patch_nums=config["hierarchical_patch_nums"],
                              dataset="cifar", linear_transform=linear_transform.to(device), upsample_net=upsample_net.to(device),
                              p_loc=_p_locs, p_log_scale=_p_log_scales, init_log_scale=_average_training_log_scale,
                              param_to_group=param2group, group_to_param=group2param, n_groups=n_groups,
                              group_start_index=group_start_index, group_end_index=group_end_index, group_idx=group_idx,
                              w0=30., c=6., random_seed=42, device=device, kl_upper_buffer=0., kl_lower_buffer=0.4,
                              kl_adjust_gap=10, initial_beta=kl_beta, beta_step_size=0.05).to(device)
    recombiner.optimize_posteriors(x, y, n_epochs=12, lr=2e-4, verbose=0)
    lv = recombiner._l1
    for r in range(3):                       # compress_posteriors' loop body for the first rounds (the full run: test suite)
        recombiner._encode_round(lv, True, r)
        recombiner.train(x, y, n_epochs=2, optimizer=torch.optim.Adam(recombiner.parameters(), lr=2e-4), verbose=False)
    assert recombiner.compressed_mask_groupwise.sum() == 3 * x.shape[0]
    with torch.no_grad():
        dist = metric(y.cpu().numpy(), recombiner.predict(x).cpu().numpy(), "cifar")
    assert np.isfinite(dist).all()
    print("DROPIN OK groups %d, first-round indices %s" % (n_groups, recombiner.compressed_idx_groupwise[recombiner.compressed_mask_groupwise][:4]))


if __name__ == "__main__":
    main(sys.argv[1])
