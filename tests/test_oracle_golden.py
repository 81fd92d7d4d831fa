"""Pins the CPU oracle (oracle/ref_cpu.py) against golden vectors generated from the reference
itself (oracle/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest
import torch

from golden_util import (DATASET_OF, xy_of, GOLDEN, O, cfg_of, check, level_kwargs, load, prior_inputs,
                         regen_noise, regen_noise_per_epoch, stats_of, t)

PRIOR_CASES = ["cifar", "protein", "patch2d", "patch1d", "patch3d"]


def test_synthetic_inputs():
    d = load("synthetic.npz")
    for name in ["cifar", "audio", "video", "protein", "kodak"]:
        X = O.fourier_features(list(d[f"{name}_pixel_sizes"]), int(d[f"{name}_fourier_dim"]))
        np.testing.assert_allclose(stats_of([X])[0], d[f"{name}_X_stats"], rtol=1e-6)
        if f"{name}_X" in d.files:
            np.testing.assert_allclose(X.numpy(), d[f"{name}_X"], rtol=0, atol=1e-6)
        else:
            np.testing.assert_allclose(X.numpy()[::37], d[f"{name}_X_rows"], rtol=0, atol=1e-6)


def test_metrics():
    d = load("metrics.npz")
    for ds in ["cifar", "kodak", "video", "audio", "protein"]:
        np.testing.assert_allclose(np.asarray(O.metric(d["a"], d["b"], ds)), d["m_" + ds], rtol=1e-6)


@pytest.mark.parametrize("name", PRIOR_CASES)
def test_prior_init_and_maps(name):
    d = load(f"prior_{name}.npz")
    cfg, geo, n, p, A, up, X, Y, pri = prior_inputs(d)
    p0 = O.init_prior_params(geo, n, seed=42)
    assert np.array_equal(p0["loc"].numpy(), d["init_loc"])
    assert np.array_equal(p0["lpe_loc"].numpy(), d["init_lpe_loc"])
    if geo.patch:
        assert np.array_equal(p0["h_loc"].numpy(), d["init_h_loc"])
        assert np.array_equal(p0["hh_loc"].numpy(), d["init_hh_loc"])
    np.testing.assert_allclose(stats_of(A), d["A_stats"], rtol=1e-12)
    np.testing.assert_allclose(stats_of(up.weights), d["up_stats"], rtol=1e-12)
    X2 = O.fourier_features(cfg["pixel_sizes"], cfg["fourier_dim"])
    np.testing.assert_allclose(X2.numpy(), X.numpy(), atol=1e-6)


@pytest.mark.parametrize("name", PRIOR_CASES)
def test_prior_forward_and_kl(name):
    d = load(f"prior_{name}.npz")
    cfg, geo, n, p, A, up, X, Y, pri = prior_inputs(d)
    eps = regen_noise(d, "fwd_eps")
    with torch.no_grad():
        y, pe, h_w = O.prior_forward(geo, p, X[None].repeat(n, 1, 1), A, up, O.Noise(eps), return_parts=True)
        kl = O.prior_kl(geo, p, pri).item()
    check(d, "fwd_pe", pe, rtol=1e-5, atol=1e-6)
    check(d, "fwd_yhat", y, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(kl, float(d["kl"]), rtol=1e-6)


@pytest.mark.parametrize("name", PRIOR_CASES)
@pytest.mark.parametrize("tm", [True, False])
def test_prior_train_3_steps(name, tm):
    d = load(f"prior_{name}.npz")
    cfg, geo, n, p, A, up, X, Y, pri = prior_inputs(d)
    tag = "tm1" if tm else "tm0"
    eps = regen_noise(d, f"{tag}_eps")
    mse, kl, elbo = O.prior_train(geo, p, X[None].repeat(n, 1, 1), Y, pri, A, up, 3, 2e-4, 1e-4, tm, O.Noise(eps))
    np.testing.assert_allclose([mse, kl], d[f"{tag}_ret"], rtol=2e-5)
    np.testing.assert_allclose(elbo, d[f"{tag}_elbo"], rtol=2e-5)
    for k in ["loc", "log_scale", "lpe_loc", "lpe_log_scale", "h_loc", "h_log_scale", "hh_loc", "hh_log_scale"]:
        if k in p:
            check(d, f"{tag}_{k}", p[k], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(stats_of(A), d[f"{tag}_A_stats"], rtol=1e-5)
    np.testing.assert_allclose(A[-1].numpy(), d[f"{tag}_A3"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(up.weights[4].detach().numpy(), d[f"{tag}_conv3_w"], rtol=1e-4, atol=1e-6)
    if not tm:  # refit expressions were evaluated on the tm0 model
        a, b = O.refit_prior(p["loc"], p["log_scale"])
        check(d, "refit_loc", a, atol=1e-6)
        check(d, "refit_scale", b, atol=1e-6)
        a, b = O.refit_prior(p["lpe_loc"], p["lpe_log_scale"])
        check(d, "refit_lpe_loc", a, atol=1e-6)
        check(d, "refit_lpe_scale", b, atol=1e-6)
        if geo.patch:
            a, b = O.refit_prior(p["h_loc"], p["h_log_scale"])
            check(d, "refit_h_scale", b, atol=1e-6)
            a, b = O.refit_prior(p["hh_loc"], p["hh_log_scale"])
            check(d, "refit_hh_loc", a, atol=1e-6)


@pytest.mark.parametrize("case", ["wide_patch2d_w48", "wide_patch3d_w64", "wide_cifar_w64"])
def test_prior_train_at_widths_48_and_64(case):
    """BASELINE.json's width variants: the REFERENCE's PriorBNNmodel run with hidden_dims = [48]*3 / [64]*3
    (oracle/make_golden.py --only wide), two Adam steps incl. the mappings: pins the oracle's width-generic path to the
    reference itself, not only to its width-32 presets."""
    d = load(case + ".npz")
    cfg, geo, n, p, A, up, X, Y, pri = prior_inputs(d)
    assert geo.hidden_dims[0] in (48, 64)
    check(d, "init_loc", p["loc"], rtol=0, atol=0)
    np.testing.assert_allclose(stats_of(A), d["A_stats"], rtol=1e-6)
    eps = regen_noise(d, "tm1_eps")
    mse, kl, elbo = O.prior_train(geo, p, X[None].repeat(n, 1, 1), Y, pri, A, up, 2, 2e-4, 1e-4, True, O.Noise(eps))
    np.testing.assert_allclose([mse, kl], d["tm1_ret"], rtol=2e-5)
    np.testing.assert_allclose(elbo, d["tm1_elbo"], rtol=2e-5)
    for k in ["loc", "log_scale", "lpe_loc", "h_loc", "hh_loc"]:
        if k in p:
            check(d, f"tm1_{k}", p[k], rtol=1e-5, atol=2e-6)
    check(d, "tm1_A1", A[1], rtol=1e-4, atol=2e-6)          # (Adam: elements with a ~0 gradient move by rounding-level amounts)
    np.testing.assert_allclose(A[-1].detach().numpy(), d["tm1_A3"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(up.weights[4].detach().numpy(), d["tm1_conv3_w"], rtol=1e-4, atol=1e-6)


def test_beta_rule():
    assert O.beta_rule(1e-8, 10.0, 5.0, 1.0) == pytest.approx(1.5e-8)
    assert O.beta_rule(1e-8, 0.5, 5.0, 1.0) == pytest.approx(1e-8 / 1.5)
    assert O.beta_rule(0.9, 10.0, 5.0, 1.0) == 1.0
    assert O.beta_rule(1.2e-20, 0.1, 5.0, 1.0) == 1e-20


def test_grouping_exact():
    d = load("grouping.npz")
    names = ["group_idx", "start", "end", "group2param", "param2group", "n_groups", "group_kls", "weights"]
    for tag in "abc":
        r = O.group_by_bits(d[f"{tag}_in"].copy())
        for k, v in zip(names, r):
            exp = d[f"{tag}_{k}"]
            if k in ("group_kls", "weights"):
                np.testing.assert_allclose(np.asarray(v), exp, rtol=1e-6)
            else:
                assert np.array_equal(np.asarray(v), exp), (tag, k)
    r = O.grouping(t(d, "g_ql"), t(d, "g_qs"), t(d, "g_pl"), t(d, "g_ps"))
    for k, v in zip(names, r):
        if k in ("group_kls", "weights"):
            np.testing.assert_allclose(np.asarray(v), d[f"g_{k}"], rtol=1e-5)
        else:
            assert np.array_equal(np.asarray(v), d[f"g_{k}"]), k


def test_gumbel_and_sobol_tables():
    d = load("tables.npz")
    g = O.gumbel_table(42, 65536)
    assert hashlib.sha256(g.tobytes()).hexdigest() == str(d["gumbel_sha"])
    assert np.array_equal(g[:512], d["gumbel_head"])
    shipped = np.load(os.path.join(GOLDEN, "tables", "gumbel_seed42_f64.npy"))
    assert np.array_equal(shipped, g)
    for gs in (1, 3, 5, 12):
        tb = O.sobol_normal_table(gs, 65536, 42).numpy()
        assert np.array_equal(tb[:64], d[f"sobol_g{gs}_head"])
        assert hashlib.sha256(np.ascontiguousarray(tb).tobytes()).hexdigest() == str(d[f"sobol_g{gs}_sha"])
    t5 = np.load(os.path.join(GOLDEN, "tables", "sobol_normal_g5_seed42_f32.npy"))
    assert np.array_equal(t5.astype(np.float64), O.sobol_normal_table(5).numpy())


def build_test_model(d, name):
    cfg = cfg_of(d)
    geo = O.Geometry.from_config(cfg)
    n = int(d["n"])
    A = O.make_linear_transform(geo.dims, seed=123)
    up = O.UpsampleNet(geo.data_dim, geo.paddings, geo.layerwise_scale_factors, seed=124)
    m = O.TestTimeModel(geo, n, DATASET_OF[name.partition("_w")[0]], A, up, level_kwargs(d, ""),
                        level_kwargs(d, "h_") if geo.patch else None,
                        level_kwargs(d, "hh_") if geo.patch else None, initial_beta=1e-5)
    return geo, n, m


def set_test_posteriors(d, geo, m):
    m.l1.loc, m.l1.log_scale = t(d, "t_loc").clone(), t(d, "t_log_scale").clone()
    if geo.patch:
        m.l2.loc, m.l2.log_scale = t(d, "t_h_loc").clone(), t(d, "t_h_log_scale").clone()
        m.l3.loc, m.l3.log_scale = t(d, "t_hh_loc").clone(), t(d, "t_hh_log_scale").clone()


@pytest.mark.parametrize("name", ["cifar", "patch2d", "patch1d", "patch3d", "patch3d_w64"])
def test_test_model(name):
    d = load(f"test_{name}.npz")
    geo, n, m = build_test_model(d, name)
    np.testing.assert_allclose(m.bpp, float(d["bpp"]), rtol=1e-12)
    if geo.patch:
        assert np.array_equal(m.l1.row_perm_g2p, d["perm_x_g2p"].astype(np.int64))
        assert np.array_equal(m.l2.row_perm_g2p, d["h_perm_x_g2p"].astype(np.int64))
    set_test_posteriors(d, geo, m)
    X, Y = xy_of(d)
    X = X[None].repeat(n, 1, 1)
    for S in (1, 5):
        eps = regen_noise(d, f"pred_S{S}_eps")
        with torch.no_grad():
            yp = m.predict(X, random_seed=None, S=S, noise=O.Noise(eps))
        check(d, f"pred_S{S}", yp, rtol=1e-5, atol=2e-6)
    with torch.no_grad():
        np.testing.assert_allclose(m.weighted_kl().item(), float(d["kl_beta_weighted"]), rtol=1e-5)
    r = m.update_annealing(False)
    if geo.patch:
        for a, k in zip(r, ["kls", "h_kls", "hh_kls"]):
            np.testing.assert_allclose(a, d[k], rtol=1e-6, atol=1e-9)
    else:
        np.testing.assert_allclose(r, d["kls"], rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal(m.l1.kl_beta.numpy(), d["beta_before"])
    m.update_annealing(True)
    np.testing.assert_array_equal(m.l1.kl_beta.numpy(), d["beta_after"])
    if geo.patch:
        np.testing.assert_array_equal(m.l2.kl_beta.numpy(), d["h_beta_after"])
        np.testing.assert_array_equal(m.l3.kl_beta.numpy(), d["hh_beta_after"])
    # A17: A* encode -- exact index, sample, log-weights
    for row, grp, idx, margin in d["enc_table"]:
        row, grp = int(row), int(grp)
        lv = m.l1
        s, e = int(lv.start[grp]), int(lv.end[grp])
        if e - s not in lv.tables:
            lv.tables[e - s] = O.sobol_normal_table(e - s)
        gum = torch.from_numpy(O.gumbel_table(42))
        i, z, lw = O.rec_score(lv.tables[e - s], lv.loc[row, s:e], O.st(lv.log_scale[row, s:e]),
                               lv.p_loc[s:e], O.st(lv.p_log_scale[s:e]), gum)
        assert i == int(idx)
        np.testing.assert_allclose(z.numpy(), d[f"enc_{row}_{grp}_z"], rtol=1e-12)
        np.testing.assert_allclose(lw[:256].numpy(), d[f"enc_{row}_{grp}_lw_head"], rtol=1e-10, atol=1e-10)
        top2 = torch.topk(lw, 2).values
        np.testing.assert_allclose(float(top2[0] - top2[1]), margin, rtol=1e-6, atol=1e-9)
    # A19: 3 epochs of S=5 training
    noises = [O.Noise(e) for e in regen_noise_per_epoch(d, "train_eps", 3)]
    m.train(X, Y, 3, 2e-4, S=5, noises=noises)
    check(d, "train_loc", m.l1.loc, rtol=1e-5, atol=2e-6)
    check(d, "train_log_scale", m.l1.log_scale, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(m.l1.kl_beta.numpy(), d["train_beta"], rtol=1e-6)
    if geo.patch:
        check(d, "train_h_loc", m.l2.loc, rtol=1e-5, atol=2e-6)
        check(d, "train_hh_log_scale", m.l3.log_scale, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("name", ["cifar"])
def test_end_to_end_compress(name):
    """Mini end-to-end: optimise 12 epochs, then encode every group with 2 fine-tune epochs per
    round.  Index selection must reproduce the reference exactly (same CPU arithmetic)."""
    d = load(f"test_{name}.npz")
    e = load(f"e2e_{name}.npz")
    geo, n, m = build_test_model(d, name)
    X = t(d, "X")[None].repeat(n, 1, 1)
    Y = t(d, "Y")
    m.train(X, Y, 12, 2e-4)
    check(e, "opt_loc", m.l1.loc, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(m.l1.kl_beta.numpy(), e["opt_beta"], rtol=1e-6)
    # encode only the first rounds here (the full 1500-round run is exercised by the generator)
    rounds = 6
    lv = m.l1
    for _ in range(rounds):
        for row in range(n):
            bits = lv.group_kls()[row] / np.log(2.)
            bits[lv.done[row]] = -1e10
            m.encode_group(lv, row, int(bits.argmax()))
        m.train(X, Y, 2, 2e-4)
    done = lv.done
    assert done.sum() == rounds * n
    agree = (lv.idx[done] == e["idx"][done]).mean()
    assert agree == 1.0, agree


def test_rec_long_groups():
    """the reference's sample_group on groups of 338 / 360 parameters (rec_long_groups.npz): exact index, sample, margin"""
    d = load("rec_long_groups.npz")
    gum = torch.from_numpy(O.gumbel_table(42))
    for row, grp, idx, margin, gl in d["enc_table"]:
        row, grp, gl = int(row), int(grp), int(gl)
        s0 = int(d["start"][grp])
        assert int(d["end"][grp]) - s0 == gl
        i, z, lw = O.rec_score(O.sobol_normal_table(gl), t(d, f"enc_{row}_{grp}_loc"), O.st(t(d, f"enc_{row}_{grp}_log_scale")),
                               t(d, "p_loc")[s0:s0 + gl], O.st(t(d, "p_log_scale")[s0:s0 + gl]), gum)
        assert i == int(idx)
        np.testing.assert_allclose(z.numpy(), d[f"enc_{row}_{grp}_z"], rtol=1e-12)
        np.testing.assert_allclose(lw[:256].numpy(), d[f"enc_{row}_{grp}_lw_head"], rtol=1e-9, atol=1e-9)
        top2 = torch.topk(lw, 2).values
        np.testing.assert_allclose(float(top2[0] - top2[1]), margin, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("name", ["patch1d", "patch2d", "patch3d", "cifar"])
def test_hierarchical_sampling_against_the_reference_function(name):
    """A5: the oracle's sample_latent_weights against utils.map_hierarchical_model_to_int_weights itself
    (tests/golden/hier_map.npz, written by oracle/make_golden.py --only hier): same noise stream, bit for bit."""
    import json
    d = load("hier_map.npz")
    cfg = json.loads(str(d[f"{name}_cfg"]))
    geo = O.Geometry.from_config(cfg)
    args = [torch.from_numpy(d[f"{name}_{k}"]) for k in ("loc", "scale", "h_loc", "h_scale", "hh_loc", "hh_scale")]
    N, D = args[0].shape
    for S in (1, 3):
        torch.manual_seed(int(d[f"{name}_S{S}_seed"]))
        eps = [torch.randn(N, S, D) for _ in range(3 if cfg["patch"] else 1)]
        got = O.sample_latent_weights(geo, *args, S, O.Noise(eps))
        assert np.array_equal(got.numpy(), d[f"{name}_S{S}_out"]), (name, S)
