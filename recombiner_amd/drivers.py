"""Host drivers mirroring the reference's two entry points on top of the MI355X models:

  * `train_prior(...)`      <- main_prior_training.py:25-341  (EM loop: train q -> adjust beta -> refit prior ->
                               every 10 iterations group parameters and write the checkpoint)
  * `compress(...)`         <- main_compression.py:25-178     (load checkpoint -> reorder priors into group order
                               -> TestBNNmodel -> optimise -> A* encode -> distortion + index arrays)

Data loading (data/*.py) is out of scope: both take tensors X [N,P,F] / [P,F] and Y [N,P,C].  The checkpoint is
the reference's own layout -- eight sequential pickles in one file, in the same order with the same tuple
contents (main_prior_training.py:284-335) -- so priors interchange with the reference in both directions
(`dropin/prior_model.py` provides the `prior_model.LinearTransform` / `prior_model.Upsample` class paths).
Under torch.distributed the INRs of `train_prior` are the local shard; the prior refit, the KL that drives
beta and the grouping statistics are all-reduced (recombiner_amd.dist).
"""
import contextlib
import io
import copy
import gzip
import pickle
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

from . import dist, ops, tuning
from .prior_model import LinearTransform, PriorBNNmodel, Upsample, get_grouping_by_kl
from .test_model import TestBNNmodel

LN2 = np.log(2.)


def bit_budgets(config, dataset, max_bitrate):
    """(budget_max, budget_min) in bits per INR (main_prior_training.py:75-83)."""
    px = np.prod(config['pixel_sizes'])
    lo = max(config['lowest_bitrate'], (max_bitrate - config['bitrate_range']))
    if dataset != 'audio':
        return max_bitrate * px, lo * px
    k = px * (3 / 48000) * 1000          # kbps * samples per patch * seconds per sample
    return max_bitrate * k, lo * k


def adjust_beta(kl_beta, kl_bits, budget_max, budget_min):
    """main_prior_training.py:144-154."""
    if kl_bits > budget_max:
        kl_beta *= 1.5
    if kl_bits < budget_min:
        kl_beta /= 1.5
    return min(max(kl_beta, 1e-20), 1)


def _grouping(q_loc, q_log_scale, p_loc, p_scale, group=None):
    """get_grouping over all ranks' INRs: per-parameter KL summed on the device, all-reduced, packed on the host."""
    colsum = ops.gauss_kl_colsum_fx(q_loc, q_log_scale, p_loc, p_scale, q_is_log=True)
    return get_grouping_by_kl(dist.grouping_weights(colsum, q_loc.shape[0], group))


def train_prior(config, dataset, X, Y, max_bitrate, device="cuda", seed=42, n_em_iter=550, first_epochs=200,
                epochs=100, lr=2e-4, kl_beta=1e-8, training_mappings=True, checkpoint_path=None, checkpoint_every=10,
                precision=0, group=None, log=print, noise_source=None):
    """Coordinate-ascent prior learning (main_prior_training.py:112-172).  Returns a dict with the model, mappings, priors,
    beta, the ELBO curve and the trajectory of the loop: per iteration (KL bits per INR before the beta rule, beta after it,
    MSE per INR).  noise_source: PriorBNNmodel.noise_source (a given epsilon stream instead of the device generator)."""
    tuning.enable_tuned_gemms()
    train_size = Y.shape[0]
    patch = config['patch']
    m = PriorBNNmodel(config['input_dim'], config['hidden_dims'], config['output_dim'], train_size, config['data_dim'],
                      config['pixel_sizes'], config['upsample_factors'], config['latent_dim'], patch,
                      config['patch_nums'], config['hierarchical_patch_nums'], random_seed=seed, device=device)
    m.precision = precision
    m.noise_source = noise_source
    lt = LinearTransform(m.dims).to(device)
    up = Upsample(config['data_dim'], config['paddings'], config['layerwise_scale_factors']).to(device)
    if dist.is_dist():
        # sharded: this model's mapping gradients are summed over `group` (None = every rank), nothing else is; the
        # shared mappings start from rank 0's values rather than from an assumption about identical seeds
        m.dp_group = group if group is not None else torch.distributed.group.WORLD
        if training_mappings:
            for prm in list(lt.parameters()) + list(up.parameters()):
                torch.distributed.broadcast(prm.data, torch.distributed.get_global_rank(m.dp_group, 0), group=m.dp_group)
    budget_max, budget_min = bit_budgets(config, dataset, max_bitrate)
    assert budget_min <= budget_max
    s0 = F.softplus(torch.tensor(-2.), beta=1, threshold=20) / 6

    def init_prior(shape):
        return torch.zeros(shape, device=device), torch.ones(shape, device=device) * s0.to(device)
    p_loc, p_scale = init_prior(m.loc.shape[1])
    p_lpe_loc, p_lpe_scale = init_prior(m.lpe_loc.shape[1:])
    p_h_loc = p_h_scale = p_hh_loc = p_hh_scale = None
    if patch:
        p_h_loc, p_h_scale = init_prior(m.h_loc.shape[-1])
        p_hh_loc, p_hh_scale = init_prior(m.hh_loc.shape[-1])
    X, Y = X.to(device), Y.to(device)
    rank, ws = dist.world(group)
    elbos, n_epoch, traj = [], first_epochs, []
    for it in range(n_em_iter):
        mse, _, e = m.train(n_epoch, lr, X, Y, p_loc, p_scale, p_lpe_loc, p_lpe_scale, p_h_loc, p_h_scale, p_hh_loc,
                            p_hh_scale, lt, up, kl_beta, training_mappings=training_mappings)
        elbos += e
        n_epoch = epochs
        # average KL in bits per INR over all ranks -> beta rule
        pack = torch.stack([m._kl_value([p_loc, p_scale, p_lpe_loc, p_lpe_scale, p_h_loc, p_h_scale, p_hh_loc, p_hh_scale]),
                            torch.tensor(float(train_size), dtype=torch.float64, device=m.loc.device)])
        pack = dist.allreduce_scalar(pack, group)
        kls = float(pack[0] / LN2 / pack[1])
        kl_beta = adjust_beta(kl_beta, kls, budget_max, budget_min)
        traj.append((kls, kl_beta, mse))
        # closed-form prior refit (moment matching over every rank's INRs)
        p_loc, p_scale = dist.refit_prior(m.loc, m.log_scale, group)
        p_lpe_loc, p_lpe_scale = dist.refit_prior(m.lpe_loc, m.lpe_log_scale, group)
        if patch:
            p_h_loc, p_h_scale = dist.refit_prior(m.h_loc, m.h_log_scale, group)
            p_hh_loc, p_hh_scale = dist.refit_prior(m.hh_loc, m.hh_log_scale, group)
        if it % checkpoint_every == 0 or it == n_em_iter - 1:
            log("EM iter %d: KL %.4f bits/INR, beta %.3e" % (it, kls, kl_beta))
            if checkpoint_path is not None:
                ck = build_checkpoint(m, lt, up, p_loc, p_scale, p_lpe_loc, p_lpe_scale, p_h_loc, p_h_scale, p_hh_loc,
                                      p_hh_scale, kl_beta, group)
                if rank == 0:
                    save_checkpoint(checkpoint_path, ck)
    return dict(model=m, linear_transform=lt, upsample_net=up, kl_beta=kl_beta, elbo=elbos, trajectory=traj,
                priors=(p_loc, p_scale, p_lpe_loc, p_lpe_scale, p_h_loc, p_h_scale, p_hh_loc, p_hh_scale))


def build_checkpoint(m, lt, up, p_loc, p_scale, p_lpe_loc, p_lpe_scale, p_h_loc, p_h_scale, p_hh_loc, p_hh_scale,
                     kl_beta, group=None):
    """The eight objects of the reference checkpoint, in file order (main_prior_training.py:186-335)."""
    def mean_rows(t):
        s = t.detach().sum(0, dtype=torch.float64).flatten()
        pack = torch.cat([s, torch.tensor([float(t.shape[0])], dtype=torch.float64, device=t.device)])
        pack = dist.allreduce_scalar(pack, group)
        return (pack[:-1] / pack[-1]).float().cpu()
    n = m.loc.shape[0]
    q_loc = torch.cat([m.loc.detach().flatten(1), m.lpe_loc.detach().flatten(1)], -1).contiguous()
    q_ls = torch.cat([m.log_scale.detach().flatten(1), m.lpe_log_scale.detach().flatten(1)], -1).contiguous()
    pl = torch.cat([p_loc.flatten(), p_lpe_loc.flatten()])
    ps = torch.cat([p_scale.flatten(), p_lpe_scale.flatten()])
    g1 = _grouping(q_loc, q_ls, pl, ps, group)
    avg_ls = torch.cat([mean_rows(m.log_scale), mean_rows(m.lpe_log_scale)])
    none8 = (None,) * 8
    if m.patch:
        g2 = _grouping(m.h_loc.detach(), m.h_log_scale.detach(), p_h_loc, p_h_scale, group)
        g3 = _grouping(m.hh_loc.detach(), m.hh_log_scale.detach(), p_hh_loc, p_hh_scale, group)
        l2 = (p_h_loc.cpu(), p_h_scale.cpu(), kl_beta, mean_rows(m.h_log_scale))
        l3 = (p_hh_loc.cpu(), p_hh_scale.cpu(), kl_beta, mean_rows(m.hh_log_scale))
    else:
        g2 = g3 = none8
        l2 = l3 = (None, None, kl_beta, None)
    return [g1, (pl.cpu(), ps.cpu(), kl_beta, avg_ls), g2, l2, g3, l3, lt, up]


_REF_CLASSES = {"LinearTransform": LinearTransform, "Upsample": Upsample}


@contextlib.contextmanager
def _reference_class_paths():
    """While active, LinearTransform / Upsample pickle as `prior_model.LinearTransform` / `prior_model.Upsample` -- the
    paths the reference's own pickles carry (main_prior_training.py:334-335) and main_compression.py:44-45 resolves --
    whatever is on sys.path.  (pickle checks that the named module really holds the class: a stand-in module is
    registered for the duration and whatever was there is put back.)"""
    stand_in = types.ModuleType("prior_model")
    saved_mod = sys.modules.get("prior_model")
    saved_paths = {n: c.__module__ for n, c in _REF_CLASSES.items()}
    for n, c in _REF_CLASSES.items():
        setattr(stand_in, n, c)
        c.__module__ = "prior_model"
    sys.modules["prior_model"] = stand_in
    try:
        yield
    finally:
        for n, c in _REF_CLASSES.items():
            c.__module__ = saved_paths[n]
        if saved_mod is None:
            sys.modules.pop("prior_model", None)
        else:
            sys.modules["prior_model"] = saved_mod


class _CheckpointUnpickler(pickle.Unpickler):
    """`prior_model.LinearTransform` / `prior_model.Upsample` resolve to the MI355X classes (same attribute, sub-module and
    parameter names as the reference's, prior_model.py:16-59), so a checkpoint written by the reference loads without the
    reference -- or `dropin/` -- on sys.path."""

    def find_class(self, module, name):
        if module in ("prior_model", "recombiner_amd.prior_model") and name in _REF_CLASSES:
            return _REF_CLASSES[name]
        return super().find_class(module, name)


def save_checkpoint(path, ck):
    """Writes the eight pickles (main_prior_training.py:284-335), loadable by the reference's main_compression.py.
    The two modules are pickled as CPU *copies*: moving the live modules to the CPU and back (what the reference does,
    :334-338) re-allocates their parameter storages, i.e. the addresses the captured training graphs read and update
    (PriorBNNmodel.train re-captures if that ever happens, see its workspace key)."""
    lt, up = ck[6], ck[7]
    with open(path, "wb") as f, _reference_class_paths():
        for obj in ck[:6]:
            pickle.dump(obj, f)
        pickle.dump(copy.deepcopy(lt).cpu(), f)
        pickle.dump(copy.deepcopy(up).cpu(), f)


def load_checkpoint(path):
    """-> list of the eight objects (main_compression.py:37-45); `path` may be gzip-compressed (test fixtures)."""
    out = []
    with open(path, "rb") as f:
        magic = f.read(2)
    opener = gzip.open if magic == b"\x1f\x8b" else open
    with opener(path, "rb") as f:
        for _ in range(8):          # eight independent pickle streams: a fresh unpickler (memo) for each
            out.append(_CheckpointUnpickler(f).load())
    return out


def build_test_model(config, dataset, checkpoint, n_datapoints, device="cuda", seed=42):
    """TestBNNmodel initialised from a prior checkpoint exactly as main_compression.py:47-133 does (priors and average
    log-scales reordered into group order).  Shared by the encoder (`compress`) and the decoder (`bitstream.decode`)."""
    g1, l1, g2, l2, g3, l3, lt, up = checkpoint
    group_idx, start, end, group2param, param2group, n_groups, _, _ = g1
    prior_loc, prior_scale, kl_beta, avg_ls = l1

    def inv_st(s):
        return torch.log(torch.exp(s * 6) - 1)
    kw = dict(p_loc=prior_loc.clone()[param2group].to(device), p_log_scale=inv_st(prior_scale).clone()[param2group].to(device),
              init_log_scale=avg_ls[param2group].cpu().detach(), param_to_group=param2group, group_to_param=group2param,
              n_groups=n_groups, group_start_index=start, group_end_index=end, group_idx=group_idx)
    if config['patch']:
        for pre, g, l in (("h_", g2, l2), ("hh_", g3, l3)):
            gi, gs, ge, g2p, p2g, ng, _, _ = g
            ploc, pscale, _, als = l
            kw.update({pre + "p_loc": ploc.clone()[p2g].to(device), pre + "p_log_scale": inv_st(pscale).clone()[p2g].to(device),
                       pre + "init_log_scale": als[p2g].cpu().detach(), pre + "param_to_group": p2g,
                       pre + "group_to_param": g2p, pre + "n_groups": ng, pre + "group_start_index": gs,
                       pre + "group_end_index": ge, pre + "group_idx": gi})
    return TestBNNmodel(config['input_dim'], config['hidden_dims'], config['output_dim'], n_datapoints,
                        config['upsample_factors'], config['latent_dim'], config['data_dim'], config['pixel_sizes'],
                        config['patch'], config['patch_nums'], config['hierarchical_patch_nums'], dataset,
                        linear_transform=lt.to(device), upsample_net=up.to(device), w0=30., c=6., random_seed=seed,
                        device=device, kl_upper_buffer=0., kl_lower_buffer=0.4, kl_adjust_gap=10, initial_beta=kl_beta,
                        beta_step_size=0.05, **kw)


def compress(config, dataset, checkpoint, x, y, device="cuda", seed=42, n_epochs=30000, lr=2e-4, precision=0,
             verbose=0, finetune_epochs=None, bitstream_path=None):
    """main_compression.py:47-178 on an in-memory checkpoint (list from load_checkpoint / build_checkpoint).
    Returns (distortion, model); with `bitstream_path` the packed A* indices are written there (bitstream.py)."""
    tuning.enable_tuned_gemms()
    g1, _, g2, _, g3, _, _, _ = checkpoint
    n_groups = g1[5]
    h_n = hh_n = None
    if config['patch']:
        h_n, hh_n = g2[5], g3[5]
    x, y = x.to(device), y.to(device)
    model = build_test_model(config, dataset, checkpoint, y.shape[0], device, seed)
    model.precision = precision
    model.optimize_posteriors(x, y, n_epochs=n_epochs, lr=lr, verbose=verbose)
    ft = finetune_epochs
    distortion = model.compress_posteriors(
        x, y, n_epochs_finetune=ft if ft is not None else max(30000 // n_groups, 50),
        h_n_epochs_finetune=None if h_n is None else (ft if ft is not None else max(15000 // h_n, 20)),
        hh_n_epochs_finetune=None if hh_n is None else (ft if ft is not None else max(15000 // hh_n, 20)),
        verbose=verbose, lr=lr, fine_tune_gap=1, compress_from_group_with_largest_kl=True)
    if bitstream_path is not None:
        from . import bitstream
        with open(bitstream_path, "wb") as f:
            f.write(bitstream.encode(model))
    return distortion, model


def rd_point(config, dataset, X, Y_train, Y_test, max_bitrate, device="cuda", seed=42, n_em_iter=550, first_epochs=200, epochs=100,
             lr=2e-4, n_opt=30000, finetune_epochs=None, precision=0, noise_source=None):
    """One rate-distortion point from scratch with the product alone: learn the prior (and the mappings) on Y_train
    (train_prior), build the checkpoint in memory, compress Y_test from it (compress) -> dict(bpp, psnr (per datapoint),
    n_groups, trajectory).  The composition main_prior_training.py -> main_compression.py performs through a file."""
    from . import bitstream
    nt, ne = Y_train.shape[0], Y_test.shape[0]
    Xd = X.to(device)
    out = train_prior(config, dataset, Xd[None].expand(nt, -1, -1), Y_train, float(max_bitrate), device=device, seed=seed,
                      n_em_iter=n_em_iter, first_epochs=first_epochs, epochs=epochs, lr=lr, precision=precision,
                      log=lambda *_: None, noise_source=noise_source)
    ck = build_checkpoint(out["model"], out["linear_transform"], out["upsample_net"], *out["priors"], out["kl_beta"])
    with contextlib.redirect_stdout(io.StringIO()):           # (the test-time model prints its expected bpp, like the reference's)
        dist_, model = compress(config, dataset, ck, Xd[None].expand(ne, -1, -1), Y_test.to(device), device=device, seed=seed,
                                n_epochs=n_opt, lr=lr, precision=precision, finetune_epochs=finetune_epochs)
    bits = bitstream.payload_bits(bitstream.encode(model))
    return dict(bpp=bits / (ne * int(np.prod(config["pixel_sizes"]))), psnr=np.asarray(dist_, dtype=np.float64),
                n_groups=int(ck[0][5]), trajectory=np.array(out["trajectory"]))
