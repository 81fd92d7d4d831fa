#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE - runs only in the build container).

Imports the reference implementation from /root/reference (read-only, never copied), runs
small seeded cases of every hot-path function listed in SURVEY.md section 8(a) on CPU and
stores inputs + expected outputs as small .npz fixtures under tests/golden/.

The reference never travels to the GPU box: only these fixtures do.  Noise tensors drawn
through ``torch.randn_like`` are captured so that the oracle restatement and the HIP path
can be fed the identical epsilon ("eps injection").

Usage:  python oracle/make_golden.py [--out tests/golden]
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
import config as ref_config          # noqa: E402
import prior_model as ref_prior      # noqa: E402
import test_model as ref_test        # noqa: E402
import utils as ref_utils            # noqa: E402

torch.set_num_threads(4)

# ----------------------------------------------------------------------------------------
# mini presets: same code paths as the reference presets (config.py:28-137) at fixture size
# ----------------------------------------------------------------------------------------
def presets():
    c = ref_config.configs
    p = {}
    p["cifar"] = dict(c["cifar"])
    p["protein"] = dict(c["protein"])
    p["patch2d"] = dict(c["kodak"], pixel_sizes=[32, 32], patch_nums=[2, 2],
                        hierarchical_patch_nums={"level2": [1, 2], "level3": [2, 2]})
    p["patch1d"] = dict(c["audio"], pixel_sizes=[160], patch_nums=[4],
                        hierarchical_patch_nums={"level2": [2], "level3": [4]})
    p["patch3d"] = dict(c["video"], pixel_sizes=[24, 16, 16], patch_nums=[1, 2, 2],
                        hierarchical_patch_nums={"level2": [1, 1, 2], "level3": [1, 2, 2]})
    n_inr = {"cifar": 3, "protein": 3, "patch2d": 8, "patch1d": 8, "patch3d": 8}
    return p, n_inr


def jsonable(cfg):
    return json.dumps(cfg, sort_keys=True)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fourier_inputs(pixel_sizes, fourier_dim):
    """data/image.py:24-27 (same three lines in audio/video/protein loaders)."""
    datum = torch.zeros(1, *pixel_sizes)
    coords, _ = ref_utils.to_grid_coordinates_and_features(datum)
    dd = len(pixel_sizes)
    w = torch.exp(torch.linspace(0, np.log(1024), fourier_dim // (2 * dd)))
    inp = torch.matmul(coords.unsqueeze(-1), w.unsqueeze(0)).view(*coords.shape[:-1], -1)
    inp = torch.cat([torch.cos(np.pi * inp), torch.sin(np.pi * inp)], dim=-1)
    return coords, inp


class NoiseTap:
    """Records every tensor produced by torch.randn_like while active."""

    def __init__(self):
        self.log = []
        self._orig = torch.randn_like

    def __enter__(self):
        def tapped(t, *a, **k):
            out = self._orig(t, *a, **k)
            self.log.append(out.detach().clone())
            return out
        torch.randn_like = tapped
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig


def build_prior(cfg, n, seed=42):
    m = ref_prior.PriorBNNmodel(in_dim=cfg["input_dim"], hidden_dims=cfg["hidden_dims"],
                                out_dim=cfg["output_dim"], train_size=n,
                                data_dim=cfg["data_dim"], pixel_sizes=cfg["pixel_sizes"],
                                upsample_factors=cfg["upsample_factors"],
                                latent_dim=cfg["latent_dim"], patch=cfg["patch"],
                                patch_nums=cfg["patch_nums"],
                                hierarchical_patch_nums=cfg["hierarchical_patch_nums"],
                                random_seed=seed, device="cpu")
    return m


def build_maps(cfg, dims, seed_a=123, seed_u=124):
    torch.manual_seed(seed_a)
    lt = ref_prior.LinearTransform(dims)
    torch.manual_seed(seed_u)
    up = ref_prior.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"])
    return lt, up


def priors_for(m, patch, seed=7):
    g = torch.Generator().manual_seed(seed)
    s0 = float(torch.nn.functional.softplus(torch.tensor(-2.0)) / 6)

    def mk(shape):
        loc = 0.01 * torch.randn(shape, generator=g)
        sc = s0 * (1 + 0.2 * torch.rand(shape, generator=g))
        return loc, sc
    pl, ps = mk(m.loc.shape[1:])
    ll, ls = mk(m.lpe_loc.shape[1:])
    if patch:
        hl, hs = mk(m.h_loc.shape[1:])
        hhl, hhs = mk(m.hh_loc.shape[1:])
    else:
        hl = hs = hhl = hhs = None
    return [pl, ps, ll, ls, hl, hs, hhl, hhs]


def tnp(t):
    return None if t is None else t.detach().cpu().numpy().copy()   # copy: never alias live parameters


def stats(t):
    a = t.detach().double()
    return np.array([a.sum().item(), a.abs().sum().item(), (a * a).sum().item()])


def put(d, key, val):
    if val is not None:
        store(d, key, val)


FULL_LIMIT = 40000


class Bag(dict):
    """dict that routes big float arrays through store() (subsample + checksums)."""

    def __setitem__(self, k, v):
        if isinstance(v, np.ndarray) and v.dtype.kind == "f" and v.size > FULL_LIMIT \
                and not k.endswith(("__sub",)):
            store(self, k, v)
        else:
            if isinstance(v, np.ndarray) and v.dtype == np.int64 and v.size > 64:
                v = v.astype(np.int32)
            super().__setitem__(k, v)


def store(d, key, arr):
    """Store small arrays whole; large ones as a strided subsample + moment checksums."""
    if arr is None:
        return
    arr = np.asarray(arr)
    if arr.size <= FULL_LIMIT or arr.dtype.kind not in "f":
        dict.__setitem__(d, key, arr)
        return
    flat = arr.reshape(-1)
    stride = int(np.ceil(arr.size / 12000)) | 1
    dict.__setitem__(d, key + "__sub", flat[::stride].copy())
    d[key + "__stride"] = np.array(stride)
    d[key + "__shape"] = np.array(arr.shape)
    f64 = flat.astype(np.float64)
    d[key + "__stats"] = np.array([f64.sum(), np.abs(f64).sum(), (f64 * f64).sum()])


def store_noise(d, prefix, seed, log, keep_full):
    """Noise = torch.manual_seed(seed) followed by randn of these shapes, in this order."""
    d[prefix + "_seed"] = np.array(seed)
    d[prefix + "_shapes"] = np.array(json.dumps([list(e.shape) for e in log]))
    d[prefix + "_stats"] = np.stack([stats(e) for e in log])
    if keep_full:
        for i, e in enumerate(log):
            d[f"{prefix}{i}"] = tnp(e)


# ----------------------------------------------------------------------------------------
def gen_synthetic(out):
    d = {}
    for name, px, f in [("cifar", [32, 32], 16), ("audio", [800], 16), ("video", [24, 16, 16], 18),
                        ("protein", [96], 16), ("kodak", [64, 64], 16)]:
        coords, x = fourier_inputs(px, f)
        d[f"{name}_pixel_sizes"] = np.array(px)
        d[f"{name}_fourier_dim"] = np.array(f)
        xs = tnp(x)
        if xs.size <= 20000:
            d[f"{name}_X"] = xs
            d[f"{name}_coords"] = tnp(coords)
        else:
            d[f"{name}_X_rows"] = xs[::37]
            d[f"{name}_coords_rows"] = tnp(coords)[::37]
        d[f"{name}_X_stats"] = stats(x)
        d[f"{name}_X_sha"] = np.array(sha(xs))
    np.savez_compressed(os.path.join(out, "synthetic.npz"), **d)


def gen_prior_cases(out):
    P, NI = presets()
    for name, cfg in P.items():
        n = NI[name]
        d = Bag({"cfg": np.array(jsonable(cfg)), "n": np.array(n)})
        small = name in ("cifar", "protein")
        m = build_prior(cfg, n)
        # A1: init parity (seed 42)
        d["init_loc"] = tnp(m.loc)
        d["init_lpe_loc"] = tnp(m.lpe_loc)
        if cfg["patch"]:
            d["init_h_loc"] = tnp(m.h_loc)
            d["init_hh_loc"] = tnp(m.hh_loc)
        # make log-scales non-trivial
        g = torch.Generator().manual_seed(11)
        with torch.no_grad():
            m.log_scale.add_(0.7 * torch.randn(m.log_scale.shape, generator=g))
            m.lpe_log_scale.add_(0.7 * torch.randn(m.lpe_log_scale.shape, generator=g))
            if cfg["patch"]:
                m.h_log_scale.add_(0.7 * torch.randn(m.h_log_scale.shape, generator=g))
                m.hh_log_scale.add_(0.7 * torch.randn(m.hh_log_scale.shape, generator=g))
        for k in ["log_scale", "lpe_log_scale", "h_log_scale", "hh_log_scale"]:
            if hasattr(m, k):
                d["p_" + k] = tnp(getattr(m, k))
        lt, up = build_maps(cfg, m.dims)
        d["A_stats"] = np.stack([stats(a) for a in lt.A])
        d["up_stats"] = np.stack([stats(p) for p in up.parameters()])
        _, x = fourier_inputs(cfg["pixel_sizes"], cfg["fourier_dim"])
        torch.manual_seed(5)
        y = torch.rand(n, x.shape[0], cfg["output_dim"])
        d["X"] = tnp(x)
        d["Y"] = tnp(y)
        X = x[None].repeat(n, 1, 1)
        pri = priors_for(m, cfg["patch"])
        for k, v in zip(["pl", "ps", "ll", "ls", "hl", "hs", "hhl", "hhs"], pri):
            put(d, "prior_" + k, tnp(v))

        # A3-A6 forward with captured noise
        torch.manual_seed(1000)
        with NoiseTap() as tap, torch.no_grad():
            yhat = m.forward(X, lt, up)
        store_noise(d, "fwd_eps", 1000, tap.log, small)
        d["fwd_yhat"] = tnp(yhat)
        # intermediate: pe and h_w (re-run helper functions with the same noise)
        with torch.no_grad():
            lpe = m.lpe_loc + m.st(m.lpe_log_scale) * tap.log[0]
            pe = ref_utils.map_lpe_to_inr_inputs(up, lpe[None], m.latent_dim, m.pixel_sizes,
                                                 m.upsample_factors, m.patch, m.patch_nums,
                                                 m.data_dim)[:, 0]
        d["fwd_pe"] = tnp(pe)
        # A7 KL
        with torch.no_grad():
            d["kl"] = np.array(m.calculate_kl(*pri).item())

        # A8: 3 Adam steps, mappings trained / frozen
        for tm in (True, False):
            m2 = build_prior(cfg, n)
            with torch.no_grad():
                for k in ["log_scale", "lpe_log_scale", "h_log_scale", "hh_log_scale"]:
                    if hasattr(m, k):
                        getattr(m2, k).copy_(getattr(m, k))
            lt2, up2 = build_maps(cfg, m.dims)
            torch.manual_seed(2000)
            with NoiseTap() as tap2:
                mse, klv, elbo = m2.train(3, 2e-4, X, y, *pri, lt2, up2, 1e-4,
                                          training_mappings=tm)
            tag = "tm1" if tm else "tm0"
            store_noise(d, f"{tag}_eps", 2000, tap2.log, False)
            d[f"{tag}_ret"] = np.array([mse, klv])
            d[f"{tag}_elbo"] = np.array(elbo)
            for k in ["loc", "log_scale", "lpe_loc", "lpe_log_scale", "h_loc", "h_log_scale",
                      "hh_loc", "hh_log_scale"]:
                if hasattr(m2, k):
                    d[f"{tag}_{k}"] = tnp(getattr(m2, k))
            d[f"{tag}_A_stats"] = np.stack([stats(a) for a in lt2.A])
            d[f"{tag}_A0_rows"] = tnp(lt2.A[0][:4])
            d[f"{tag}_A3"] = tnp(lt2.A[-1])
            d[f"{tag}_up_stats"] = np.stack([stats(p) for p in up2.parameters()])
            d[f"{tag}_conv3_w"] = tnp(up2.conv3.weight)

        # A9: prior refit expressions (main_prior_training.py:157-172 evaluated on m2)
        with torch.no_grad():
            def refit(loc, ls):
                pl_ = loc.clone().detach().mean(0)
                ps_ = ((m2.st(ls.clone().detach()) ** 2).mean(0) + loc.clone().detach().var(0)) ** 0.5
                return pl_, ps_
            a, b = refit(m2.loc, m2.log_scale)
            d["refit_loc"], d["refit_scale"] = tnp(a), tnp(b)
            a, b = refit(m2.lpe_loc, m2.lpe_log_scale)
            d["refit_lpe_loc"], d["refit_lpe_scale"] = tnp(a), tnp(b)
            if cfg["patch"]:
                a, b = refit(m2.h_loc, m2.h_log_scale)
                d["refit_h_loc"], d["refit_h_scale"] = tnp(a), tnp(b)
                a, b = refit(m2.hh_loc, m2.hh_log_scale)
                d["refit_hh_loc"], d["refit_hh_scale"] = tnp(a), tnp(b)
        np.savez_compressed(os.path.join(out, f"prior_{name}.npz"), **d)
        print("prior", name, "ok", flush=True)


def gen_grouping(out):
    """A10: get_grouping_by_kl on synthetic fp32 bit-weights (exact integer outputs)."""
    d = {}
    rng = np.random.RandomState(3)
    for tag, D, scale in [("a", 3779, 0.06), ("b", 515, 3.0), ("c", 64, 9.0)]:
        w = (rng.gamma(0.7, scale, size=D)).astype(np.float32)
        if tag == "c":
            w[5] = 17.5  # single element above the 16-bit cap
        r = ref_prior.get_grouping_by_kl(w.copy())
        names = ["group_idx", "start", "end", "group2param", "param2group", "n_groups",
                 "group_kls", "weights"]
        d[f"{tag}_in"] = w
        for k, v in zip(names, r):
            d[f"{tag}_{k}"] = np.asarray(v)
    # and through get_grouping (KL in bits, mean over rows)
    g = torch.Generator().manual_seed(9)
    ql = 0.02 * torch.randn(6, 300, generator=g)
    qs = 0.003 + 0.01 * torch.rand(6, 300, generator=g)
    pl = 0.005 * torch.randn(300, generator=g)
    ps = 0.02 + 0.01 * torch.rand(300, generator=g)
    r = ref_prior.get_grouping(ql, qs, pl, ps)
    for k, v in zip(["ql", "qs", "pl", "ps"], [ql, qs, pl, ps]):
        d["g_" + k] = tnp(v)
    for k, v in zip(names, r):
        d[f"g_{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(out, "grouping.npz"), **d)
    print("grouping ok", flush=True)


def gen_tables(out):
    """A15/A16: Gumbel recurrence and Sobol-normal candidate tables (seed 42)."""
    d = {}

    class Stub:
        pass
    s = Stub()
    s.bit_per_group = 16
    s.random_seed = 42
    ref_test.TestBNNmodel.get_gumbel_sample(s)
    g = s.g_samples.numpy()
    d["gumbel_head"] = g[:512]
    d["gumbel_tail"] = g[-512:]
    d["gumbel_sha"] = np.array(sha(g))
    d["gumbel_stats"] = np.array([g.sum(), np.abs(g).sum()])
    os.makedirs(os.path.join(out, "tables"), exist_ok=True)
    np.save(os.path.join(out, "tables", "gumbel_seed42_f64.npy"), g)
    for gs in range(1, 13):
        t = ref_test.TestBNNmodel.get_sobol_normal_sample(s, gs, 65536)
        a = t.numpy()
        assert a.dtype == np.float64
        a32 = a.astype(np.float32)
        assert np.array_equal(a32.astype(np.float64), a), "table not fp32-representable"
        d[f"sobol_g{gs}_sha"] = np.array(sha(a))
        d[f"sobol_g{gs}_head"] = a[:64]
        if gs in (3, 5):
            np.save(os.path.join(out, "tables", f"sobol_normal_g{gs}_seed42_f32.npy"), a32)
    d["versions"] = np.array(json.dumps({"torch": torch.__version__, "numpy": np.__version__,
                                          "scipy": __import__("scipy").__version__}))
    np.savez_compressed(os.path.join(out, "tables.npz"), **d)
    print("tables ok", flush=True)


def build_test_model(cfg, name, n, pm, pri, lt, up, initial_beta=1e-5):
    """Mirror of main_compression.py:49-146 on in-memory priors."""
    q_loc = torch.cat([pm.loc.flatten(start_dim=1), pm.lpe_loc.flatten(start_dim=1)], -1)
    q_scale = torch.cat([pm.st(pm.log_scale).flatten(start_dim=1),
                         pm.st(pm.lpe_log_scale).flatten(start_dim=1)], -1)
    p_loc = torch.cat([pri[0].flatten(), pri[2].flatten()])
    p_scale = torch.cat([pri[1].flatten(), pri[3].flatten()])
    with torch.no_grad():
        G = ref_prior.get_grouping(q_loc, q_scale, p_loc, p_scale)
    avg_ls = torch.cat([pm.log_scale.detach().mean(0), pm.lpe_log_scale.detach().mean(0).flatten()])
    kw = {}
    groups = {"": G}
    if cfg["patch"]:
        with torch.no_grad():
            Gh = ref_prior.get_grouping(pm.h_loc, pm.st(pm.h_log_scale), pri[4], pri[5])
            Ghh = ref_prior.get_grouping(pm.hh_loc, pm.st(pm.hh_log_scale), pri[6], pri[7])
        groups["h_"] = Gh
        groups["hh_"] = Ghh

    def inv_st(s):
        return torch.log(torch.exp(s * 6) - 1)
    p2g = G[4]
    kw.update(p_loc=p_loc[p2g], p_log_scale=inv_st(p_scale)[p2g], init_log_scale=avg_ls[p2g],
              param_to_group=p2g, group_to_param=G[3], n_groups=G[5], group_start_index=G[1],
              group_end_index=G[2], group_idx=G[0])
    if cfg["patch"]:
        for pre, Gx, pl_, ps_, als in [("h_", Gh, pri[4], pri[5], pm.h_log_scale.detach().mean(0)),
                                       ("hh_", Ghh, pri[6], pri[7], pm.hh_log_scale.detach().mean(0))]:
            q = Gx[4]
            kw.update({pre + "p_loc": pl_[q], pre + "p_log_scale": inv_st(ps_)[q],
                       pre + "init_log_scale": als[q], pre + "param_to_group": q,
                       pre + "group_to_param": Gx[3], pre + "n_groups": Gx[5],
                       pre + "group_start_index": Gx[1], pre + "group_end_index": Gx[2],
                       pre + "group_idx": Gx[0]})
    tm = ref_test.TestBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n,
                               cfg["upsample_factors"], cfg["latent_dim"], cfg["data_dim"],
                               cfg["pixel_sizes"], cfg["patch"], cfg["patch_nums"],
                               cfg["hierarchical_patch_nums"], name if name in ("cifar", "protein") else
                               {"patch2d": "kodak", "patch1d": "audio", "patch3d": "video"}[name],
                               linear_transform=lt, upsample_net=up, device="cpu",
                               initial_beta=initial_beta, **kw)
    return tm, groups, kw


def gen_test_cases(out, names=("cifar", "patch2d", "patch1d")):
    """names may carry a width suffix ("patch3d_w64": the reference classes with hidden_dims = [64] * 3, BASELINE
    configs[4]); the dataset string and the geometry come from the part before it."""
    P, NI = presets()
    for full_name in names:
        name, _, wtag = full_name.partition("_w")
        cfg = dict(P[name])
        if wtag:
            cfg["hidden_dims"] = [int(wtag)] * 3
        n = {"cifar": 4, "patch2d": 8, "patch1d": 8, "patch3d": 8}[name]  # 2 datapoints for the patched presets
        d = Bag({"cfg": np.array(jsonable(cfg)), "n": np.array(n)})
        pm = build_prior(cfg, n)
        lt, up = build_maps(cfg, pm.dims)
        _, x = fourier_inputs(cfg["pixel_sizes"], cfg["fourier_dim"])
        torch.manual_seed(5)
        y = torch.rand(n, x.shape[0], cfg["output_dim"])
        X = x[None].repeat(n, 1, 1)
        d["X"], d["Y"] = tnp(x), tnp(y)
        # a short prior fit so that the KLs / groups are non-degenerate
        s0 = float(torch.nn.functional.softplus(torch.tensor(-2.0)) / 6)
        pri = [torch.zeros(pm.loc.shape[1]), torch.ones(pm.loc.shape[1]) * s0,
               torch.zeros(pm.lpe_loc.shape[1:]), torch.ones(pm.lpe_loc.shape[1:]) * s0]
        if cfg["patch"]:
            pri += [torch.zeros(pm.h_loc.shape[1]), torch.ones(pm.h_loc.shape[1]) * s0,
                    torch.zeros(pm.hh_loc.shape[1]), torch.ones(pm.hh_loc.shape[1]) * s0]
        else:
            pri += [None] * 4
        torch.manual_seed(77)
        pm.train(30, 2e-3, X, y, *pri, lt, up, 1e-6, training_mappings=False)
        with torch.no_grad():
            # tighten the posteriors so that groups hold a handful of parameters each
            pm.log_scale.sub_(1.0)
            pm.lpe_log_scale.sub_(1.0)
        for k in ["loc", "log_scale", "lpe_loc", "lpe_log_scale", "h_loc", "h_log_scale", "hh_loc",
                  "hh_log_scale"]:
            if hasattr(pm, k):
                d["pm_" + k] = tnp(getattr(pm, k))
        for k, v in zip(["pl", "ps", "ll", "ls", "hl", "hs", "hhl", "hhs"], pri):
            put(d, "prior_" + k, tnp(v))
        tm, groups, kw = build_test_model(cfg, name, n, pm, pri, lt, up)
        for pre, G in groups.items():
            for k, v in zip(["group_idx", "start", "end", "group2param", "param2group", "n_groups",
                             "group_kls", "weights"], G):
                d[f"{pre}G_{k}"] = np.asarray(v)
        for k, v in kw.items():
            if torch.is_tensor(v):
                d["kw_" + k] = tnp(v)
        d["bpp"] = np.array(tm.bpp)
        if cfg["patch"]:
            d["perm_x_g2p"] = tm.permute_patch_x_g2p.astype(np.int16)
            d["h_perm_x_g2p"] = tm.h_permute_patch_x_g2p.astype(np.int16)
        # perturb the test posteriors so rows differ
        g = torch.Generator().manual_seed(21)
        with torch.no_grad():
            tm.loc.add_(0.004 * torch.randn(tm.loc.shape, generator=g))
            tm.log_scale.add_(0.3 * torch.randn(tm.log_scale.shape, generator=g))
            if cfg["patch"]:
                tm.h_loc.add_(0.004 * torch.randn(tm.h_loc.shape, generator=g))
                tm.hh_loc.add_(0.004 * torch.randn(tm.hh_loc.shape, generator=g))
        # the posteriors the cases start from are inputs: always stored whole (dict.__setitem__ bypasses the subsampling)
        dict.__setitem__(d, "t_loc", tnp(tm.loc))
        dict.__setitem__(d, "t_log_scale", tnp(tm.log_scale))
        if cfg["patch"]:
            for k_ in ("h_loc", "h_log_scale", "hh_loc", "hh_log_scale"):
                dict.__setitem__(d, "t_" + k_, tnp(getattr(tm, k_)))
        # A12 predict, S=1 and S=5
        for S in (1, 5):
            with NoiseTap() as tap, torch.no_grad():
                yp = tm.predict(X, random_seed=3, sample_size=S)
            d[f"pred_S{S}"] = tnp(yp)
            store_noise(d, f"pred_S{S}_eps", 3, tap.log, False)
        # A13 / A14
        with torch.no_grad():
            d["kl_beta_weighted"] = np.array(tm.calculate_kl().item())
        r = tm.update_annealing_factors(update=False)
        if cfg["patch"]:
            d["kls"], d["h_kls"], d["hh_kls"] = r
        else:
            d["kls"] = r
        d["beta_before"] = tnp(tm.kl_beta)
        tm.update_annealing_factors(update=True)
        d["beta_after"] = tnp(tm.kl_beta)
        if cfg["patch"]:
            d["h_beta_after"] = tnp(tm.h_kl_beta)
            d["hh_beta_after"] = tnp(tm.hh_kl_beta)
        # A17/A18: encode a few groups (level-1) and record index, sample and top-2 margin
        enc = []
        for row, grp in [(0, 0), (1, 3), (n - 1, int(tm.n_groups) - 1), (2, 7)]:
            i, z, lw = tm.sample_group(row, grp, 65536)
            top2 = torch.topk(lw, 2).values
            enc.append((row, grp, i, float(top2[0] - top2[1])))
            d[f"enc_{row}_{grp}_z"] = tnp(z)
            d[f"enc_{row}_{grp}_lw_head"] = tnp(lw[:256])
            d[f"enc_{row}_{grp}_lw_max"] = np.array(lw.max().item())
        d["enc_table"] = np.array(enc, dtype=np.float64)
        if cfg["patch"]:
            i, z, lw = tm.h_sample_group(0, 1, 65536)
            d["h_enc_0_1"] = np.array([i])
            d["h_enc_0_1_z"] = tnp(z)
            i, z, lw = tm.hh_sample_group(0, 2, 65536)
            d["hh_enc_0_2"] = np.array([i])
            d["hh_enc_0_2_z"] = tnp(z)
        # A19: 3 training epochs (S=5), beta update at epoch 0
        opt = torch.optim.Adam(tm.parameters(), lr=2e-4)
        with NoiseTap() as tap:
            tm.train(X, y, 3, opt, False, sample_size=5)
        # the reference reseeds with the epoch number each step (test_model.py:285,623)
        store_noise(d, "train_eps", -1, tap.log, False)
        d["train_loc"], d["train_log_scale"] = tnp(tm.loc), tnp(tm.log_scale)
        d["train_beta"] = tnp(tm.kl_beta)
        if cfg["patch"]:
            d["train_h_loc"], d["train_hh_loc"] = tnp(tm.h_loc), tnp(tm.hh_loc)
            d["train_h_log_scale"], d["train_hh_log_scale"] = tnp(tm.h_log_scale), tnp(tm.hh_log_scale)
        np.savez_compressed(os.path.join(out, f"test_{full_name}.npz"), **d)
        print("test", full_name, "ok", flush=True)

        # A20/A21 mini end-to-end (only for cifar and patch1d: cheap)
        if full_name in ("cifar", "patch1d"):
            e = Bag({"cfg": d["cfg"], "n": d["n"]})
            tm2, _, _ = build_test_model(cfg, name, n, pm, pri, lt, up)
            import io
            import contextlib
            with contextlib.redirect_stderr(io.StringIO()):
                tm2.optimize_posteriors(X, y, n_epochs=12, lr=2e-4, verbose=False)
                e["opt_loc"], e["opt_log_scale"] = tnp(tm2.loc), tnp(tm2.log_scale)
                e["opt_beta"] = tnp(tm2.kl_beta)
                dist = tm2.compress_posteriors(X, y, n_epochs_finetune=2, h_n_epochs_finetune=2,
                                               hh_n_epochs_finetune=2, verbose=False, lr=2e-4,
                                               fine_tune_gap=1)
            e["distortion"] = np.asarray(dist)
            e["idx"] = tm2.compressed_idx_groupwise
            e["final_sample"] = tnp(tm2.compressed_sample)
            if cfg["patch"]:
                e["h_idx"] = tm2.h_compressed_idx_groupwise
                e["hh_idx"] = tm2.hh_compressed_idx_groupwise
            with torch.no_grad():
                e["final_pred"] = tnp(tm2.predict(X))
            np.savez_compressed(os.path.join(out, f"e2e_{name}.npz"), **e)
            print("e2e", name, "ok", flush=True)

        # 3-D patched geometry: the head of a compression run -- 12 optimisation epochs, then the first encode rounds of the
        # top level (level 3) exactly as compress_posteriors' loop body runs them (test_model.py:700-722): the row's
        # largest-KL group -> hh_compress_group -> fine-tune.  (The complete run is hours of CPU time at 6144 pixels.)
        if name == "patch3d":
            e = Bag({"cfg": d["cfg"], "n": d["n"]})
            tm2, _, _ = build_test_model(cfg, name, n, pm, pri, lt, up)
            import io
            import contextlib
            with contextlib.redirect_stderr(io.StringIO()):
                tm2.optimize_posteriors(X, y, n_epochs=12, lr=2e-4, verbose=False)
            e["opt_loc"], e["opt_log_scale"] = tnp(tm2.loc), tnp(tm2.log_scale)
            e["opt_hh_loc"], e["opt_h_loc"] = tnp(tm2.hh_loc), tnp(tm2.h_loc)
            e["opt_beta"] = tnp(tm2.kl_beta)
            rounds = []
            for _i in range(3):
                for row in range(tm2.hh_loc.shape[0]):
                    _, _, kls = tm2.update_annealing_factors(False)
                    kls = kls / np.log(2.)
                    kls[tm2.hh_compressed_mask_groupwise] = -1e10
                    grp = int(np.argmax(kls[row]))
                    tm2.hh_compress_group(row, grp)
                    rounds.append((row, grp, tm2.hh_compressed_idx_groupwise[row, grp]))
                opt = torch.optim.Adam(tm2.parameters(), lr=2e-4)
                tm2.train(X, y, n_epochs=2, optimizer=opt, verbose=False)
            e["hh_rounds"] = np.array(rounds, dtype=np.float64)
            e["hh_loc_after"] = tnp(tm2.hh_loc)
            e["loc_after"] = tnp(tm2.loc)
            with torch.no_grad():
                e["final_pred"] = tnp(tm2.predict(X, random_seed=11))
            np.savez_compressed(os.path.join(out, f"e2e_{full_name}.npz"), **e)
            print("e2e head", full_name, "ok", flush=True)


def gen_long_groups(out):
    """A17 on groups of hundreds of parameters: what low bit-rates produce (group_parameters packs until 16 bits of KL are
    reached, prior_model.py:301-316, no size cap).  The reference's own sample_group on a CIFAR test model whose grouping
    comes from ~0.05 bits per parameter."""
    P, _ = presets()
    cfg = P["cifar"]
    n = 3
    pm = build_prior(cfg, n)
    lt, up = build_maps(cfg, pm.dims)
    D = pm.loc.shape[1] + int(np.prod(pm.lpe_loc.shape[1:]))
    rng = np.random.RandomState(4)
    bits = rng.gamma(0.7, 0.05 / 0.7, size=D).astype(np.float32)
    G = ref_prior.get_grouping_by_kl(bits.copy())
    g = torch.Generator().manual_seed(31)
    p_loc = 0.01 * torch.randn(D, generator=g)
    p_ls = -2.0 + 0.3 * torch.randn(D, generator=g)
    p2g = G[4]
    tm = ref_test.TestBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["upsample_factors"],
                               cfg["latent_dim"], cfg["data_dim"], cfg["pixel_sizes"], False, None, None, "cifar",
                               linear_transform=lt, upsample_net=up, device="cpu", initial_beta=1e-5,
                               p_loc=p_loc[p2g], p_log_scale=p_ls[p2g], init_log_scale=torch.full((D,), -2.2)[p2g],
                               param_to_group=p2g, group_to_param=G[3], n_groups=G[5], group_start_index=G[1],
                               group_end_index=G[2], group_idx=G[0])
    with torch.no_grad():      # posteriors a fraction of a bit per parameter away from the prior
        tm.loc.add_(0.004 * torch.randn(tm.loc.shape, generator=g))
        tm.log_scale.add_(0.2 * torch.randn(tm.log_scale.shape, generator=g))
    d = {"bits": bits, "n_groups": np.array(G[5]), "start": np.asarray(G[1]), "end": np.asarray(G[2]),
         "p_loc": tnp(tm.p_loc), "p_log_scale": tnp(tm.p_log_scale)}
    lens = np.asarray(G[2]) - np.asarray(G[1])
    picks = [(0, int(np.argmax(lens))), (1, int(np.argmin(lens))), (2, int(G[5]) // 2)]
    enc = []
    for row, grp in picks:
        i, z, lw = tm.sample_group(row, grp, 65536)
        s_, e_ = int(G[1][grp]), int(G[2][grp])
        top2 = torch.topk(lw, 2).values
        enc.append((row, grp, i, float(top2[0] - top2[1]), e_ - s_))
        d[f"enc_{row}_{grp}_loc"] = tnp(tm.loc[row, s_:e_])
        d[f"enc_{row}_{grp}_log_scale"] = tnp(tm.log_scale[row, s_:e_])
        d[f"enc_{row}_{grp}_z"] = tnp(z)
        d[f"enc_{row}_{grp}_lw_head"] = tnp(lw[:256])
    d["enc_table"] = np.array(enc, dtype=np.float64)
    np.savez_compressed(os.path.join(out, "rec_long_groups.npz"), **d)
    print("long groups ok: lengths", [int(e[4]) for e in enc], flush=True)


def smooth_images(n, pixel_sizes, seed):
    """low-frequency synthetic images in [0.1, 0.9] (something an INR fits to > 25 dB): per channel a few 2-D sinusoids"""
    rng = np.random.RandomState(seed)
    h, w = pixel_sizes
    yy, xx = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing="ij")
    out = np.zeros([n, h * w, 3], dtype=np.float32)
    for i in range(n):
        for c in range(3):
            img = np.zeros([h, w])
            for _ in range(3):
                fx, fy = rng.uniform(-2.5, 2.5, size=2)
                img += rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * (fx * xx + fy * yy) + rng.uniform(0, 2 * np.pi))
            img = 0.5 + 0.4 * img / np.abs(img).max()
            out[i, :, c] = img.reshape(-1)
    return torch.from_numpy(out)


def structured_A(dims):
    """a well-conditioned, exactly reproducible stand-in for LEARNED mappings (the fixture cannot carry 13 MB of trained
    A): identity plus a small circulant pattern.  tests/golden_util.py::structured_A restates it."""
    mats = []
    for i in range(1, len(dims)):
        L = dims[i] * (dims[i - 1] + 1)
        r = torch.arange(L)
        base = torch.tensor([0.5, -0.25, 0.125, 0.375, -0.5, 0.25, -0.125, -0.375]) / 8
        mats.append(torch.eye(L) + base[(r[:, None] + 3 * r[None, :]) % 8] / L)
    return mats


def gen_checkpoint_and_psnr(out):
    """N2 + the PSNR@bpp half of the metric.  The reference's EM loop (main_prior_training.py:112-172: train -> beta rule ->
    prior refit) on 16 smooth synthetic images with its own classes, its checkpoint written in its own layout
    (main_prior_training.py:284-335: eight sequential pickles, the two modules pickled as `prior_model.*`), then the
    reference's compression of 4 other images from that file (main_compression.py:37-167): optimise, A*-encode every
    group, fine-tune in between.  Stored: the checkpoint (gzip), the test images' seed, per-image PSNR, bpp, all indices,
    and a predict() output on injected noise for the interchange test."""
    import gzip
    import io
    import pickle
    import contextlib
    import time
    P, _ = presets()
    cfg = P["cifar"]
    n_train, n_test = 16, 32      # 32 test images: per-image PSNR varies by ~0.3 dB between equally valid runs (every A* index is a
    #                               random draw), the mean over 32 by ~0.05 dB
    _, x = fourier_inputs(cfg["pixel_sizes"], cfg["fourier_dim"])
    Ytr = smooth_images(n_train, cfg["pixel_sizes"], 100)
    Yte = smooth_images(n_test, cfg["pixel_sizes"], 200)
    X = x[None].repeat(n_train, 1, 1)
    pm = build_prior(cfg, n_train)
    lt, up = build_maps(cfg, pm.dims)
    with torch.no_grad():
        for a, b in zip(lt.A, structured_A(pm.dims)):
            a.copy_(b)
    s0 = torch.nn.functional.softplus(torch.tensor(-2.0)) / 6
    pri = [torch.zeros(pm.loc.shape[1]), torch.ones(pm.loc.shape[1]) * s0,
           torch.zeros(pm.lpe_loc.shape[1:]), torch.ones(pm.lpe_loc.shape[1:]) * s0, None, None, None, None]
    max_bitrate = 3.0
    px = np.prod(cfg["pixel_sizes"])
    budget_max = max_bitrate * px
    budget_min = max(cfg["lowest_bitrate"], max_bitrate - cfg["bitrate_range"]) * px
    kl_beta, n_epoch = 1e-8, 200
    t0 = time.time()
    torch.manual_seed(0)
    for it in range(14):
        # mappings frozen: A is the structured stand-in, Upsample stays at its seeded initialisation
        _, kl, _ = pm.train(n_epoch, 2e-3, X, Ytr, *pri, lt, up, kl_beta, training_mappings=False)
        n_epoch = 100
        kls = kl / np.log(2.)
        if kls > budget_max:
            kl_beta *= 1.5
        if kls < budget_min:
            kl_beta /= 1.5
        kl_beta = min(max(kl_beta, 1e-20), 1)
        with torch.no_grad():                                 # main_prior_training.py:157-172
            pri[0] = pm.loc.clone().detach().mean(0)
            pri[1] = ((pm.st(pm.log_scale.clone().detach()) ** 2).mean(0) + pm.loc.clone().detach().var(0)) ** 0.5
            pri[2] = pm.lpe_loc.clone().detach().mean(0)
            pri[3] = ((pm.st(pm.lpe_log_scale.clone().detach()) ** 2).mean(0) + pm.lpe_loc.clone().detach().var(0)) ** 0.5
        print("  EM %d: %.1f bits/INR, beta %.2e (%.0f s)" % (it, kls, kl_beta, time.time() - t0), flush=True)
    # checkpoint, in the reference's order (main_prior_training.py:186-335)
    with torch.no_grad():
        avg_ls = torch.cat([pm.log_scale.clone().detach().mean(0), pm.lpe_log_scale.clone().detach().mean([0]).flatten()])
        q_loc = torch.cat([pm.loc.flatten(start_dim=1), pm.lpe_loc.flatten(start_dim=1)], -1)
        q_scale = torch.cat([pm.st(pm.log_scale).flatten(start_dim=1), pm.st(pm.lpe_log_scale).flatten(start_dim=1)], -1)
        p_loc = torch.cat([pri[0].flatten(), pri[2].flatten()])
        p_scale = torch.cat([pri[1].flatten(), pri[3].flatten()])
        G = ref_prior.get_grouping(q_loc, q_scale, p_loc, p_scale)
    none8 = (None,) * 8
    buf = io.BytesIO()
    pickle.dump(tuple(G), buf)
    pickle.dump((p_loc.cpu(), p_scale.cpu(), kl_beta, avg_ls), buf)
    pickle.dump(none8, buf)
    pickle.dump((None, None, kl_beta, None), buf)
    pickle.dump(none8, buf)
    pickle.dump((None, None, kl_beta, None), buf)
    pickle.dump(lt.cpu(), buf)
    pickle.dump(up.cpu(), buf)
    with gzip.open(os.path.join(out, "PRIOR_ref_smooth_cifar.pkl.gz"), "wb", compresslevel=9) as f:
        f.write(buf.getvalue())
    print("  checkpoint: %d groups, %.2f MB pickled" % (G[5], len(buf.getvalue()) / 1e6), flush=True)

    # compression from that file, the way main_compression.py:37-167 does it
    f = io.BytesIO(buf.getvalue())
    group_idx, group_start_index, group_end_index, group2param, param2group, n_groups, group_kls, weights = pickle.load(f)
    prior_loc, prior_scale, kl_beta_l, average_training_log_scale = pickle.load(f)
    for _ in range(4):
        pickle.load(f)
    linear_transform = pickle.load(f)
    upsample_net = pickle.load(f)
    _p_locs = prior_loc.clone()[param2group]
    _p_log_scales = torch.log(torch.exp(prior_scale * 6) - 1).clone()[param2group]
    _avg = average_training_log_scale[param2group].cpu().detach()
    tm = ref_test.TestBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n_test, cfg["upsample_factors"],
                               cfg["latent_dim"], cfg["data_dim"], cfg["pixel_sizes"], False, None, None, "cifar",
                               linear_transform=linear_transform, upsample_net=upsample_net, p_loc=_p_locs,
                               p_log_scale=_p_log_scales, init_log_scale=_avg, param_to_group=param2group,
                               group_to_param=group2param, n_groups=n_groups, group_start_index=group_start_index,
                               group_end_index=group_end_index, group_idx=group_idx, w0=30., c=6., random_seed=42,
                               device="cpu", kl_upper_buffer=0., kl_lower_buffer=0.4, kl_adjust_gap=10,
                               initial_beta=kl_beta_l, beta_step_size=0.05)
    Xte = x[None].repeat(n_test, 1, 1)
    d = Bag({"cfg": np.array(jsonable(cfg)), "n_test": np.array(n_test), "train_seed": np.array(100), "test_seed": np.array(200),
             "Y_test": tnp(Yte), "n_groups": np.array(n_groups), "bpp": np.array(tm.bpp), "kl_beta": np.array(kl_beta_l),
             "p_loc": tnp(prior_loc), "p_scale": tnp(prior_scale)})
    with NoiseTap() as tap, torch.no_grad():
        yp = tm.predict(Xte, random_seed=3, sample_size=1)
    d["pred0"] = tnp(yp)
    store_noise(d, "pred0_eps", 3, tap.log, False)
    n_opt, n_ft = 400, 6
    d["n_opt"], d["n_ft"], d["lr"] = np.array(n_opt), np.array(n_ft), np.array(2e-3)
    with contextlib.redirect_stderr(io.StringIO()):
        tm.optimize_posteriors(Xte, Yte, n_epochs=n_opt, lr=2e-3, verbose=False)
        d["opt_loc"], d["opt_log_scale"] = tnp(tm.loc), tnp(tm.log_scale)
        with torch.no_grad():
            d["psnr_after_opt"] = np.asarray(ref_utils.metric(Yte.numpy(), tm.predict(Xte).numpy(), "cifar"))
        print("  optimised (%.0f s): PSNR" % (time.time() - t0), d["psnr_after_opt"], flush=True)
        dist = tm.compress_posteriors(Xte, Yte, n_epochs_finetune=n_ft, h_n_epochs_finetune=None, hh_n_epochs_finetune=None,
                                      verbose=False, lr=2e-3, fine_tune_gap=1)
    d["psnr"] = np.asarray(dist)
    d["idx"] = tm.compressed_idx_groupwise
    d["final_sample"] = tnp(tm.compressed_sample)
    with torch.no_grad():
        d["final_pred"] = tnp(tm.predict(Xte))
    np.savez_compressed(os.path.join(out, "psnr_smooth_cifar.npz"), **d)
    print("checkpoint + PSNR ok: bpp %.3f, PSNR" % tm.bpp, d["psnr"], "(%.0f s)" % (time.time() - t0), flush=True)


def gen_rd_trained(out):
    """The PSNR@bpp half of the metric for a prior the reference TRAINED ITSELF, mappings included -- the path the
    throughput number times.  For two rate targets: the reference's EM loop (main_prior_training.py:112-172: train with
    training_mappings=True -> beta rule -> closed-form prior refit) on 64 smooth synthetic images with its own classes,
    then its grouping and its compression of 16 other images (main_compression.py:47-162).  Stored per rate: the
    trajectory of the loop (KL bits per INR before the beta rule, beta after it, MSE per INR), the number of groups, bpp
    and per-image PSNR.  The trained mappings themselves do not travel (13 MB per rate): the GPU test trains its own
    prior with the product on the same data and schedule and is held to the same rate-distortion points.
    Noise of the loop: torch.manual_seed(em_seed) before it, then per step randn_like(lpe) and randn_like(level 1)."""
    import contextlib
    import io
    import time
    P, _ = presets()
    cfg = P["cifar"]
    n_train, n_test = 64, int(os.environ.get("RD_N_TEST", "16"))
    # round 4: RD_N_TEST=64 RD_N_SEEDS=4 RD_RATE=<0|1> python oracle/make_golden.py --only rd   writes
    # rd_trained_cifar_n64_r<0|1>.npz: the same experiment with 64 held-out images (the per-run mean is then a mean over four
    # times as many images), one rate per process so that the two rates run side by side
    only_rate = os.environ.get("RD_RATE")
    seed0 = int(os.environ.get("RD_SEED0", "0"))     # first repetition index (more repetitions of an existing fixture: RD_SEED0=4 ...)
    fname = "rd_trained_cifar.npz" if (n_test == 16 and only_rate is None) else "rd_trained_cifar_n%d_r%s%s.npz" % (
        n_test, only_rate or "all", "" if seed0 == 0 else "_s%d" % seed0)
    n_iter, first_epochs, epochs, lr = 30, 200, 60, 1e-3
    n_opt, n_ft = 300, 4
    _, x = fourier_inputs(cfg["pixel_sizes"], cfg["fourier_dim"])
    Ytr = smooth_images(n_train, cfg["pixel_sizes"], 300)
    Yte = smooth_images(n_test, cfg["pixel_sizes"], 400)
    X = x[None].repeat(n_train, 1, 1)
    Xte = x[None].repeat(n_test, 1, 1)
    px = np.prod(cfg["pixel_sizes"])
    d = Bag({"cfg": np.array(jsonable(cfg)), "n_train": np.array(n_train), "n_test": np.array(n_test),
             "train_seed": np.array(300), "test_seed": np.array(400), "n_iter": np.array(n_iter),
             "first_epochs": np.array(first_epochs), "epochs": np.array(epochs), "lr": np.array(lr), "n_opt": np.array(n_opt),
             "n_ft": np.array(n_ft), "Y_train_stats": stats(Ytr), "Y_test_stats": stats(Yte)})
    px = np.prod(cfg["pixel_sizes"])
    rates = [3.0, 1.5]
    n_seeds = int(os.environ.get("RD_N_SEEDS", "4"))   # independent repetitions per rate: a single run's PSNR scatters by ~0.5 dB (every A* index is a
    #                               random draw and the prior differs with the noise), the comparison is between means
    d["max_bitrate"] = np.array(rates)
    d["n_seeds"] = np.array(n_seeds)
    t0 = time.time()
    for ri, max_bitrate in enumerate(rates):
        if only_rate is not None and ri != int(only_rate):
            continue
        acc = {k: [] for k in ("traj", "em_seed", "psnr_train", "n_groups", "bpp", "psnr_after_opt", "psnr")}
        for si in range(seed0, seed0 + n_seeds):
            pm = build_prior(cfg, n_train, seed=42 + si)
            lt, up = build_maps(cfg, pm.dims)
            s0 = torch.nn.functional.softplus(torch.tensor(-2.0)) / 6
            pri = [torch.zeros(pm.loc.shape[1]), torch.ones(pm.loc.shape[1]) * s0,
                   torch.zeros(pm.lpe_loc.shape[1:]), torch.ones(pm.lpe_loc.shape[1:]) * s0, None, None, None, None]
            budget_max = max_bitrate * px
            budget_min = max(cfg["lowest_bitrate"], max_bitrate - cfg["bitrate_range"]) * px
            kl_beta, n_epoch = 1e-8, first_epochs
            em_seed = 10 + ri + 10 * si
            torch.manual_seed(em_seed)
            traj = []
            with NoiseTap() as tap:
                for it in range(n_iter):
                    mse, kl, _ = pm.train(n_epoch, lr, X, Ytr, *pri, lt, up, kl_beta, training_mappings=True)
                    n_epoch = epochs
                    kls = kl / np.log(2.)
                    if kls > budget_max:
                        kl_beta *= 1.5
                    if kls < budget_min:
                        kl_beta /= 1.5
                    kl_beta = min(max(kl_beta, 1e-20), 1)
                    with torch.no_grad():                                 # main_prior_training.py:157-172
                        pri[0] = pm.loc.clone().detach().mean(0)
                        pri[1] = ((pm.st(pm.log_scale.clone().detach()) ** 2).mean(0) + pm.loc.clone().detach().var(0)) ** 0.5
                        pri[2] = pm.lpe_loc.clone().detach().mean(0)
                        pri[3] = ((pm.st(pm.lpe_log_scale.clone().detach()) ** 2).mean(0) + pm.lpe_loc.clone().detach().var(0)) ** 0.5
                    traj.append([kls, kl_beta, mse])
                if si == seed0:        # the noise stream of the first repetition is pinned (the fp32 test replays it)
                    d[f"r{ri}_noise_shapes"] = np.array(json.dumps([list(e.shape) for e in tap.log[:2]]))
                    d[f"r{ri}_noise_count"] = np.array(len(tap.log))
                    d[f"r{ri}_noise_first_stats"] = np.stack([stats(e) for e in tap.log[:2]])
                    d[f"r{ri}_noise_last_stats"] = np.stack([stats(e) for e in tap.log[-2:]])
            print("  rate %.1f seed %d: EM done, %.1f bits/INR, beta %.3e, mse %.3e (%.0f s)" % (max_bitrate, si, traj[-1][0], kl_beta,
                                                                                             traj[-1][2], time.time() - t0), flush=True)
            d[f"r{ri}_budget"] = np.array([budget_min, budget_max])
            with torch.no_grad():
                psnr_train = np.asarray(ref_utils.metric(Ytr.numpy(), pm.forward(X, lt, up).numpy(), "cifar"))
                avg_ls = torch.cat([pm.log_scale.clone().detach().mean(0), pm.lpe_log_scale.clone().detach().mean([0]).flatten()])
                q_loc = torch.cat([pm.loc.flatten(start_dim=1), pm.lpe_loc.flatten(start_dim=1)], -1)
                q_scale = torch.cat([pm.st(pm.log_scale).flatten(start_dim=1), pm.st(pm.lpe_log_scale).flatten(start_dim=1)], -1)
                p_loc = torch.cat([pri[0].flatten(), pri[2].flatten()])
                p_scale = torch.cat([pri[1].flatten(), pri[3].flatten()])
                group_idx, gs, ge, group2param, param2group, n_groups, group_kls, weights = ref_prior.get_grouping(q_loc, q_scale, p_loc, p_scale)
            _p_locs = p_loc.clone()[param2group]
            _p_log_scales = torch.log(torch.exp(p_scale * 6) - 1).clone()[param2group]
            _avg = avg_ls[param2group].cpu().detach()
            with contextlib.redirect_stdout(io.StringIO()):
                tm = ref_test.TestBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n_test, cfg["upsample_factors"],
                                           cfg["latent_dim"], cfg["data_dim"], cfg["pixel_sizes"], False, None, None, "cifar",
                                           linear_transform=lt, upsample_net=up, p_loc=_p_locs, p_log_scale=_p_log_scales,
                                           init_log_scale=_avg, param_to_group=param2group, group_to_param=group2param,
                                           n_groups=n_groups, group_start_index=gs, group_end_index=ge, group_idx=group_idx, w0=30.,
                                           c=6., random_seed=42 + si, device="cpu", kl_upper_buffer=0., kl_lower_buffer=0.4,
                                           kl_adjust_gap=10, initial_beta=kl_beta, beta_step_size=0.05)
            with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
                tm.optimize_posteriors(Xte, Yte, n_epochs=n_opt, lr=lr, verbose=False)
                with torch.no_grad():
                    after_opt = np.asarray(ref_utils.metric(Yte.numpy(), tm.predict(Xte).numpy(), "cifar"))
                dist = tm.compress_posteriors(Xte, Yte, n_epochs_finetune=n_ft, h_n_epochs_finetune=None, hh_n_epochs_finetune=None,
                                              verbose=False, lr=lr, fine_tune_gap=1)
            for k, v in (("traj", np.array(traj)), ("em_seed", em_seed), ("psnr_train", psnr_train), ("n_groups", n_groups),
                         ("bpp", tm.bpp), ("psnr_after_opt", after_opt), ("psnr", np.asarray(dist))):
                acc[k].append(v)
            print("rate %.1f seed %d: %d groups, bpp %.3f, train PSNR %.2f, after opt %.2f, compressed %.2f dB (%.0f s)" % (
                max_bitrate, si, n_groups, tm.bpp, psnr_train.mean(), after_opt.mean(), np.mean(dist), time.time() - t0), flush=True)
            for k, v in acc.items():
                d[f"r{ri}_{k}"] = np.array(v)
            np.savez_compressed(os.path.join(out, fname), **d)
    print("rd ok (%.0f s)" % (time.time() - t0), flush=True)


def gen_metrics(out):
    d = {}
    rng = np.random.RandomState(0)
    a = rng.rand(4, 1024, 3).astype(np.float32)
    b = (a + 0.05 * rng.randn(4, 1024, 3)).astype(np.float32)
    d["a"], d["b"] = a, b
    for ds in ["cifar", "kodak", "video", "audio", "protein"]:
        d["m_" + ds] = np.asarray(ref_utils.metric(a, b, ds))
    np.savez_compressed(os.path.join(out, "metrics.npz"), **d)


# ----------------------------------------------------------------------------------------
# BASELINE.json's width variants (configs[2]: Kodak patches at width 48; configs[4]: 3-D video at width 64): the
# reference's own PriorBNNmodel with `hidden_dims` overridden (prior_model.py:65,84-85), 2 Adam steps incl. the mappings
# ----------------------------------------------------------------------------------------
def gen_wide_cases(out):
    P, NI = presets()
    for name, width in (("patch2d", 48), ("patch3d", 64), ("cifar", 64)):
        cfg = dict(P[name], hidden_dims=[width] * 3)
        n = NI[name]
        d = Bag({"cfg": np.array(jsonable(cfg)), "n": np.array(n)})
        m = build_prior(cfg, n)
        d["init_loc"] = tnp(m.loc)
        lt, up = build_maps(cfg, m.dims)
        d["A_stats"] = np.stack([stats(a) for a in lt.A])
        _, x = fourier_inputs(cfg["pixel_sizes"], cfg["fourier_dim"])
        torch.manual_seed(5)
        y = torch.rand(n, x.shape[0], cfg["output_dim"])
        d["X"] = tnp(x)
        d["Y"] = tnp(y)
        pri = priors_for(m, cfg["patch"])
        for k, v in zip(["pl", "ps", "ll", "ls", "hl", "hs", "hhl", "hhs"], pri):
            put(d, "prior_" + k, tnp(v))
        torch.manual_seed(2000)
        with NoiseTap() as tap:
            mse, klv, elbo = m.train(2, 2e-4, x[None].repeat(n, 1, 1), y, *pri, lt, up, 1e-4, training_mappings=True)
        store_noise(d, "tm1_eps", 2000, tap.log, False)
        d["tm1_ret"] = np.array([mse, klv])
        d["tm1_elbo"] = np.array(elbo)
        for k in ["loc", "log_scale", "lpe_loc", "h_loc", "hh_loc"]:
            if hasattr(m, k):
                d[f"tm1_{k}"] = tnp(getattr(m, k))
        d["tm1_A1"] = tnp(lt.A[1])
        d["tm1_A3"] = tnp(lt.A[-1])
        d["tm1_conv3_w"] = tnp(up.conv3.weight)
        np.savez_compressed(os.path.join(out, f"wide_{name}_w{width}.npz"), **d)
        print("wide", name, width, "ok", flush=True)

def gen_hier_map(out):
    """round-4 addition: utils.map_hierarchical_model_to_int_weights itself (utils.py:122-198; A5), called directly on small
    random posteriors of every patched geometry and of the un-patched case, sample sizes 1 and 3.  The noise is the CPU
    generator's stream after torch.manual_seed(seed): three [N, S, D] draws in the order level 1, 2, 3 (one draw un-patched)."""
    P, n_inr = presets()
    d, D = {}, 37
    for name in ("patch1d", "patch2d", "patch3d", "cifar"):
        cfg, N = P[name], n_inr[name]
        pn, hp, dd = cfg["patch_nums"], cfg["hierarchical_patch_nums"], cfg["data_dim"]
        g = torch.Generator().manual_seed(1000 + len(name))
        r2 = N // int(np.prod(hp["level2"])) if cfg["patch"] else 1
        r3 = N // int(np.prod(hp["level3"])) if cfg["patch"] else 1
        loc, h_loc, hh_loc = (torch.randn(r, D, generator=g) for r in (N, r2, r3))
        sc, h_sc, hh_sc = (torch.rand(r, D, generator=g) * 0.3 + 0.01 for r in (N, r2, r3))
        for k, v in zip(("loc", "scale", "h_loc", "h_scale", "hh_loc", "hh_scale"), (loc, sc, h_loc, h_sc, hh_loc, hh_sc)):
            d[f"{name}_{k}"] = tnp(v)
        for S in (1, 3):
            torch.manual_seed(77 + S)
            o = ref_utils.map_hierarchical_model_to_int_weights(bool(cfg["patch"]), loc, sc, h_loc, h_sc, hh_loc, hh_sc, S,
                                                                hp, pn, dd)
            assert tuple(o.shape) == (N, S, D)
            d[f"{name}_S{S}_out"] = tnp(o)
            d[f"{name}_S{S}_seed"] = np.int64(77 + S)
        d[f"{name}_cfg"] = np.array(jsonable(cfg))
        d[f"{name}_n"] = np.int64(N)
    np.savez_compressed(os.path.join(out, "hier_map.npz"), **d)
    print("hier_map ok", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    todo = a.only.split(",") if a.only else ["synthetic", "prior", "grouping", "tables", "test", "metrics", "wide"]
    if "synthetic" in todo:
        gen_synthetic(a.out)
    if "metrics" in todo:
        gen_metrics(a.out)
    if "grouping" in todo:
        gen_grouping(a.out)
    if "tables" in todo:
        gen_tables(a.out)
    if "prior" in todo:
        gen_prior_cases(a.out)
    if "test" in todo:
        gen_test_cases(a.out)
    if "wide" in todo:
        gen_wide_cases(a.out)
    # round-2 additions (not in the default list: the round-1 fixtures above are not rewritten)
    if "test3d" in todo:
        gen_test_cases(a.out, names=("patch3d", "patch3d_w64"))
    if "long" in todo:
        gen_long_groups(a.out)
    if "ckpt" in todo:
        gen_checkpoint_and_psnr(a.out)
    # round-3 addition
    if "rd" in todo:
        gen_rd_trained(a.out)
    # round-4 addition
    if "hier" in todo:
        gen_hier_map(a.out)
    print("done")
