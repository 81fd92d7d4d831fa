"""Size-independent properties at BASELINE.json's full size (CIFAR, 4096 INRs) where the oracle is too slow:
determinism, batch invariance (an INR's result does not depend on its neighbours), linearity of the
backward pass in dy, fused-loss == forward + explicit dy, A* job-order invariance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from recombiner_amd import ops, utils  # noqa: E402
from recombiner_amd.ops import SirenMeta  # noqa: E402

DEV = "cuda"
N = 4096


def _inputs(prec):
    g = torch.Generator().manual_seed(0)
    X, Y = utils.synthetic_inputs([32, 32], 16, N, 3, seed=0)
    meta = SirenMeta(1, 1024, 16, 16, 3, 32, 3, precision=prec)
    pe = torch.randn(N, 1024, 16, generator=g) * 0.2
    wv = (torch.rand(N, meta.d_net, generator=g) * 2 - 1) * 0.03
    return meta, X.to(DEV), Y.to(DEV), pe.to(DEV), wv.to(DEV)


@pytest.mark.parametrize("prec", [0, 1])
def test_determinism_and_batch_invariance(prec):
    meta, X, Y, pe, wv = _inputs(prec)
    sc = 1.0 / 3072
    s1, w1, p1 = ops.siren_loss_bwd(X, pe, wv, Y, sc, meta)
    s2, w2, p2 = ops.siren_loss_bwd(X, pe, wv, Y, sc, meta)
    assert torch.equal(s1, s2) and torch.equal(w1, w2) and torch.equal(p1, p2)
    sl = slice(1000, 1007)
    s3, w3, p3 = ops.siren_loss_bwd(X, pe[sl].contiguous(), wv[sl].contiguous(), Y[sl].contiguous(), sc, meta)
    assert torch.equal(s3, s1[sl]) and torch.equal(w3, w1[sl]) and torch.equal(p3, p1[sl])
    assert torch.isfinite(w1).all() and torch.isfinite(p1).all() and float(s1.min()) > 0


def test_backward_is_linear_in_dy_and_matches_fused_loss():
    meta, X, Y, pe, wv = _inputs(0)
    g = torch.Generator(device=DEV).manual_seed(1)
    d1 = torch.randn(N, 1024, 3, device=DEV, generator=g)
    d2 = torch.randn(N, 1024, 3, device=DEV, generator=g)
    wa, pa = ops.siren_bwd(X, pe, wv, d1, meta)
    wb, pb = ops.siren_bwd(X, pe, wv, d2, meta)
    wc, pc = ops.siren_bwd(X, pe, wv, 0.7 * d1 - 1.9 * d2, meta)
    np.testing.assert_allclose(wc.cpu().numpy(), (0.7 * wa - 1.9 * wb).cpu().numpy(), rtol=2e-4, atol=2e-4 * float(wa.abs().max()))
    np.testing.assert_allclose(pc.cpu().numpy(), (0.7 * pa - 1.9 * pb).cpu().numpy(), rtol=2e-4, atol=2e-4 * float(pa.abs().max()))
    # fused loss kernel == forward kernel + explicit dy through the plain backward kernel
    sc = 1.0 / 3072
    y = ops.siren_fwd(X, pe, wv, meta)
    sse, wf, pf = ops.siren_loss_bwd(X, pe, wv, Y, sc, meta)
    np.testing.assert_allclose(sse.cpu().numpy(), ((y - Y) ** 2).sum((1, 2)).cpu().numpy(), rtol=1e-5)
    we, pe2 = ops.siren_bwd(X, pe, wv, 2 * sc * (y - Y), meta)
    np.testing.assert_allclose(wf.cpu().numpy(), we.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(we.abs().max()))
    np.testing.assert_allclose(pf.cpu().numpy(), pe2.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(pe2.abs().max()))


def test_posterior_update_is_row_local():
    """fused grad + KL + Adam over [4096, 3267] equals the same update applied to a slice on its own."""
    g = torch.Generator().manual_seed(3)
    D, S = 3267, 1
    loc = (0.02 * torch.randn(N, D, generator=g)).to(DEV)
    ls = (-4 + 0.3 * torch.randn(N, D, generator=g)).to(DEV)
    eps = torch.randn(N, S, D, generator=g).to(DEV)
    dout = (1e-3 * torch.randn(N, S, D, generator=g)).to(DEV)
    pl, ps = torch.zeros(D, device=DEV), torch.full((D,), 0.02, device=DEV)

    def step(l, s, e, d):
        l, s = l.clone(), s.clone()
        lv = ops.LevelSpec(l, s, D, l.shape[0])
        st = {k: torch.zeros_like(l) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
        kl = torch.zeros(1024, device=DEV, dtype=torch.int64)             # fixed point, ops.KL_FX units per nat
        for t in (1, 2):
            ops.posterior_bwd(lv, pl, ps, False, 1e-4, d, e, S, adam=ops.adam_cfg(2e-4, t), state=st, kl_accum=kl)
        return l, s, kl.sum().double() / ops.KL_FX
    la, sa, kla = step(loc, ls, eps, dout)
    sl = slice(2048, 2056)
    lb, sb, _ = step(loc[sl].contiguous(), ls[sl].contiguous(), eps[sl].contiguous(), dout[sl].contiguous())
    assert torch.equal(la[sl], lb) and torch.equal(sa[sl], sb)
    rows, _ = ops.gauss_kl(loc, ls, pl, ps)
    # the kernel accumulated the pre-update KL of step 1 and of step 2: the first equals the standalone KL
    assert float(kla) > float(rows.sum())


def test_rec_job_order_invariance():
    from golden_util import GOLDEN, O
    import os
    g = torch.Generator().manual_seed(4)
    rows, D = 500, 3779
    loc = (0.02 * torch.randn(rows, D, generator=g)).to(DEV)
    scale = (0.002 + 0.004 * torch.rand(rows, D, generator=g)).to(DEV)
    pl = (0.01 * torch.randn(D, generator=g)).to(DEV)
    ps = (0.015 + 0.01 * torch.rand(D, generator=g)).to(DEV)
    gum = torch.from_numpy(np.load(os.path.join(GOLDEN, "tables", "gumbel_seed42_f64.npy"))).to(DEV)
    tabs = {k: torch.from_numpy(np.load(os.path.join(GOLDEN, "tables", f"sobol_normal_g{k}_seed42_f32.npy")).astype(np.float64)).to(DEV)
            for k in (3, 5)}
    rs = np.random.RandomState(1)
    jr = np.arange(rows)
    jg = np.array([3, 5])[rs.randint(0, 2, rows)]
    js = np.array([rs.randint(0, D - 5) for _ in range(rows)])
    i1, z1, b1, _ = ops.rec_score_argmax(loc, scale, pl, ps, tabs, gum, jr, js, jg)
    perm = rs.permutation(rows)
    i2, z2, b2, _ = ops.rec_score_argmax(loc, scale, pl, ps, tabs, gum, jr[perm], js[perm], jg[perm])
    assert torch.equal(i1[perm], i2) and torch.equal(z1[perm], z2) and torch.equal(b1[perm], b2)
    assert int(i1.min()) >= 0 and int(i1.max()) < 65536
    # spot-check three jobs against the fp64 oracle
    for b in (0, 123, 499):
        r, s, gl = jr[b], js[b], jg[b]
        i, zi, _ = O.rec_score(tabs[gl].cpu(), loc[r, s:s + gl].cpu(), scale[r, s:s + gl].cpu(), pl[s:s + gl].cpu(),
                               ps[s:s + gl].cpu(), gum.cpu())
        assert int(i1[b]) == i


# ---------------------------------------------------------------------------------------------------
# round 4: the WHOLE training step (sample -> upsampling net -> A transform -> SIREN -> backward -> posterior update, one
# replayed HIP graph) at BASELINE.json's sizes, through properties that need no oracle
# ---------------------------------------------------------------------------------------------------
def _preset_model(name, n, seed=42):
    from recombiner_amd import config
    from recombiner_amd import prior_model as PM
    cfg = config.configs[name]
    m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                         cfg["upsample_factors"], cfg["latent_dim"], cfg["patch"], cfg["patch_nums"],
                         cfg["hierarchical_patch_nums"], random_seed=seed, device=DEV)
    m.precision = 1
    torch.manual_seed(123)
    lt = PM.LinearTransform(m.dims).to(DEV)
    torch.manual_seed(124)
    up = PM.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"]).to(DEV)
    D, s0, lat = m._d_net, 0.0211547, list(m.lpe_loc.shape[1:])
    pri = [torch.zeros(D, device=DEV), torch.full((D,), s0, device=DEV), torch.zeros(lat, device=DEV), torch.full(lat, s0, device=DEV)]
    pri += ([torch.zeros(D, device=DEV), torch.full((D,), s0, device=DEV)] * 2) if cfg["patch"] else [None] * 4
    return cfg, m, lt, up, pri


def test_whole_step_at_configs1_size_is_reproducible_and_batch_invariant():
    """BASELINE configs[1] (CIFAR 32x32, 4096 INRs, bf16 mode, graph replay).  (1) Two runs from the same state are BITWISE
    identical: posteriors and the ELBO log (no atomics on floats, fixed-order reductions, counter-based noise).  (2) Rows
    1000..1007 trained inside the batch of 4096 equal the same eight INRs trained ALONE (frozen mappings; the small model
    draws the big batch's noise for those rows: rng_row_offset) -- an INR's result does not depend on its neighbours.  That
    comparison is bitwise for every hand-written kernel on its own (test_determinism_and_batch_invariance,
    test_posterior_update_is_row_local) but not for the step: stage 1 of the upsampling net is a library GEMM whose K-split
    follows the batch size, and the A transform cuts its contraction for launches of few rows; so it is held to fp32
    rounding (parameters 2e-6 absolute after 10 Adam steps of 2e-4, all but 0.1 % of the elements: Adam turns a gradient of
    rounding-noise size into a full step either way)."""
    from golden_util import assert_close_mostly
    from recombiner_amd import utils
    steps, lr = 10, 2e-4
    X, Y = utils.synthetic_inputs([32, 32], 16, N, 3, seed=0)
    Xd, Yd = X.to(DEV), Y.to(DEV)

    def run(rows=None):
        n = N if rows is None else rows.stop - rows.start
        cfg, m, lt, up, pri = _preset_model("cifar", N)
        if rows is not None:
            cfg, ms, lt, up, pri = _preset_model("cifar", n)
            with torch.no_grad():
                for k in ("loc", "log_scale", "lpe_loc", "lpe_log_scale"):
                    getattr(ms, k).copy_(getattr(m, k)[rows])
            ms.rng_row_offset = rows.start
            m = ms
        m.rng_seed_override = 0xC0FFEE
        y = Yd if rows is None else Yd[rows].contiguous()
        out = m.train(steps, lr, Xd[None].expand(n, -1, -1), y, *pri, lt, up, 1e-8, training_mappings=False)
        assert m._ws is not None and m._ws["graphs"] is not None, "the step must run as a replayed graph"
        return m, out

    m1, (mse1, kl1, e1) = run()
    m2, (mse2, kl2, e2) = run()
    for k in ("loc", "log_scale", "lpe_loc", "lpe_log_scale"):
        assert torch.equal(getattr(m1, k), getattr(m2, k)), k
    assert e1 == e2 and mse1 == mse2 and kl1 == kl2 and np.isfinite(e1).all()
    rows = slice(1000, 1008)
    m8, (mse8, kl8, e8) = run(rows)
    for k in ("loc", "log_scale", "lpe_loc", "lpe_log_scale"):
        assert_close_mostly(getattr(m8, k), getattr(m1, k)[rows].detach().cpu().numpy(), rtol=0, atol=2e-6, max_frac=1e-3,
                            hard_atol=2.5 * lr * steps, what=k)
    assert torch.isfinite(m1.loc).all() and abs(mse8 - mse1) < 0.2 * mse1        # (eight INRs against the mean over 4096)


def test_a_clip_of_the_audio_preset_trains_the_same_alone_and_inside_a_batch():
    """Three-level preset (audio: 60 patches per clip, levels of 4 and 60 patches): with every level's noise and the lpe's drawn
    in the kernels at the element index of the unsharded arrays (rng_row_offset), clip 5 of a batch of 8 trained alone (frozen
    mappings) follows the batch's rows for that clip -- a rank's shard sees the noise of the unsharded run.  The hand-written
    kernels are row-local; the A transform cuts its contraction differently for 60 and 480 rows, so the comparison is held to
    fp32 rounding like the CIFAR one."""
    from golden_util import assert_close_mostly
    from recombiner_amd import utils
    steps, lr, per, clips, pick = 8, 2e-4, 60, 8, 5
    n_all = per * clips
    cfg, m_all, lt, up, pri = _preset_model("audio", n_all)
    X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n_all, cfg["output_dim"], seed=0)
    Xd, Yd = X.to(DEV), Y.to(DEV)
    rows = slice(pick * per, (pick + 1) * per)
    _, m_one, _, _, _ = _preset_model("audio", per)
    r2, r3 = per // 4, 1                                   # rows of the coarser levels per clip
    with torch.no_grad():
        for k in ("loc", "log_scale", "lpe_loc", "lpe_log_scale"):
            getattr(m_one, k).copy_(getattr(m_all, k)[rows])
        for k in ("h_loc", "h_log_scale"):
            getattr(m_one, k).copy_(getattr(m_all, k)[pick * r2:(pick + 1) * r2])
        for k in ("hh_loc", "hh_log_scale"):
            getattr(m_one, k).copy_(getattr(m_all, k)[pick * r3:(pick + 1) * r3])
    m_one.rng_row_offset = rows.start
    for m_, n_, y_ in ((m_all, n_all, Yd), (m_one, per, Yd[rows].contiguous())):
        m_.rng_seed_override = 0xBEEF
        m_.train(steps, lr, Xd[None].expand(n_, -1, -1), y_, *pri, lt, up, 1e-8, training_mappings=False)
        assert m_._ws is not None and m_._ws["graphs"] is not None and "hier_eps" in m_._ws and "smp_lpe" in m_._ws
    for k, sl in (("loc", rows), ("log_scale", rows), ("lpe_loc", rows), ("lpe_log_scale", rows),
                  ("h_loc", slice(pick * r2, (pick + 1) * r2)), ("hh_loc", slice(pick * r3, (pick + 1) * r3))):
        assert_close_mostly(getattr(m_one, k), getattr(m_all, k)[sl].detach().cpu().numpy(), rtol=0, atol=2e-6, max_frac=2e-3,
                            hard_atol=2.5 * lr * steps, what=k)
        assert float((getattr(m_one, k) - getattr(m_all, k)[sl]).abs().max()) < 2.5 * lr * steps


def test_whole_step_at_configs3_shard_size_captures_replays_and_stays_finite():
    """BASELINE configs[3] per-GPU shard: LibriSpeech-shaped 1-D INRs, 1024 clips x 60 patches = 61 440 INRs with the
    three-level hierarchy, bf16 mode: the step captures as one HIP graph, replays on the second call, the ELBO improves and
    every posterior level stays finite; peak device memory stays far inside the 288 GB of one MI355X."""
    import warnings
    from recombiner_amd import utils
    clips = 1024
    cfg, m, lt, up, pri = _preset_model("audio", clips * 60)
    n = clips * 60
    X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, cfg["output_dim"], seed=0)
    Xd, Yd = X.to(DEV)[None].expand(n, -1, -1), Y.to(DEV)
    torch.cuda.reset_peak_memory_stats()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _, _, e1 = m.train(5, 1e-3, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=True)
        ws = m._ws
        assert ws is not None and ws["graphs"] is not None
        _, _, e2 = m.train(5, 1e-3, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=True)
    assert m._ws is ws and np.isfinite(e1).all() and np.isfinite(e2).all() and np.mean(e2) > np.mean(e1)
    for k in ("loc", "log_scale", "h_loc", "hh_loc", "lpe_loc"):
        assert torch.isfinite(getattr(m, k)).all(), k
    assert all(torch.isfinite(a).all() for a in lt.A) and all(torch.isfinite(p).all() for p in up.parameters())
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print("configs[3] shard: peak device memory %.1f GiB" % peak)
    assert peak < 200


def test_forked_step_equals_one_stream_step_at_full_size():
    """The captured step runs on three streams (PriorBNNmodel.stream_forks: the A transform's backward + the network level's
    posterior update beside the upsampling net's backward, whose weight-gradient side has a stream of its own).  Same kernels
    on the same operands, so the result must be BITWISE the one-stream step's -- at BASELINE configs[1] size with the
    mappings trained, where a missing dependency or a buffer freed under a stream that still reads it shows (round 4: the
    SIREN gradient, allocated on the main stream and read by the forked A transform, was released at the end of its scope
    and overwritten by the upsampling net's backward: non-finite ELBO at 4096 INRs, invisible at 8)."""
    from recombiner_amd import utils
    steps, lr = 12, 2e-4
    X, Y = utils.synthetic_inputs([32, 32], 16, N, 3, seed=0)
    Xd, Yd = X.to(DEV), Y.to(DEV)
    outs = []
    for forks in (0, 14, 14, 30, 46, 78):
        cfg, m, lt, up, pri = _preset_model("cifar", N)
        m.stream_forks, m.rng_seed_override = forks, 0xFACADE
        mse, kl, elbo = m.train(steps, lr, Xd[None].expand(N, -1, -1), Yd, *pri, lt, up, 1e-8, training_mappings=True)
        assert m._ws is not None and m._ws["graphs"] is not None
        assert np.isfinite(elbo).all() and np.isfinite(mse) and np.isfinite(kl), (forks, elbo)
        outs.append([torch.tensor(elbo), m.loc.detach().clone(), m.log_scale.detach().clone(), m.lpe_loc.detach().clone()]
                    + [a.detach().clone() for a in lt.A] + [p.detach().clone() for p in up.parameters()])
        del m, lt, up
    for other in outs[1:]:
        for a_, b_ in zip(outs[0], other):
            assert torch.equal(a_, b_)


def _variant_model(name, n, width, prec, seed=42):
    """a preset at another hidden width / operand type (BASELINE's "width-48" and "width-64 fp16" variants: the reference
    builds its INR from any hidden_dims, prior_model.py:84-85)"""
    from recombiner_amd import config
    from recombiner_amd import prior_model as PM
    cfg = dict(config.configs[name])
    cfg["hidden_dims"] = [width] * len(cfg["hidden_dims"])
    m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                         cfg["upsample_factors"], cfg["latent_dim"], cfg["patch"], cfg["patch_nums"],
                         cfg["hierarchical_patch_nums"], random_seed=seed, device=DEV)
    m.precision = prec
    torch.manual_seed(123)
    lt = PM.LinearTransform(m.dims).to(DEV)
    torch.manual_seed(124)
    up = PM.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"]).to(DEV)
    D, s0, lat = m._d_net, 0.0211547, list(m.lpe_loc.shape[1:])
    pri = [torch.zeros(D, device=DEV), torch.full((D,), s0, device=DEV), torch.zeros(lat, device=DEV), torch.full(lat, s0, device=DEV)]
    pri += [torch.zeros(D, device=DEV), torch.full((D,), s0, device=DEV)] * 2
    return cfg, m, lt, up, pri


def test_one_kodak_photo_at_full_geometry_width48():
    """BASELINE configs[2] at its FULL geometry (config.py:50-70 with hidden_dims = [48] * 3): one Kodak-sized photo = 8 x 12
    patches of 64 x 64 pixels = 96 INRs of 4096 pixels, stitched 2-D positional encodings, three-level hierarchy, bf16
    operands (the golden-vector tests run this preset on 2 x 2 patches of 32 x 32).  The prior-training step captures as one
    HIP graph, replays on the second call, stays finite and improves the ELBO; and photo 1 of a batch of two trained ALONE
    (frozen mappings, the batch's noise for its rows) follows the batch's rows for that photo to fp32 rounding -- the
    stitched upsampling couples the patches of a photo, never two photos."""
    import warnings
    from golden_util import assert_close_mostly
    from recombiner_amd import utils
    per, steps, lr = 96, 6, 2e-4
    cfg, m_all, lt, up, pri = _variant_model("kodak", 2 * per, 48, 1)
    assert cfg["pixel_sizes"] == [64, 64] and int(np.prod(cfg["patch_nums"])) == per and m_all.dims == [32, 48, 48, 48, 3]
    X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], 2 * per, cfg["output_dim"], seed=0)
    Xd, Yd = X.to(DEV), Y.to(DEV)
    # (1) one photo with the mappings trained: capture, replay, finite, improving
    _, m1, lt1, up1, _ = _variant_model("kodak", per, 48, 1)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _, _, e1 = m1.train(5, 1e-3, Xd[None].expand(per, -1, -1), Yd[:per].contiguous(), *pri, lt1, up1, 1e-8, training_mappings=True)
        ws = m1._ws
        assert ws is not None and ws["graphs"] is not None
        _, _, e2 = m1.train(5, 1e-3, Xd[None].expand(per, -1, -1), Yd[:per].contiguous(), *pri, lt1, up1, 1e-8, training_mappings=True)
    assert m1._ws is ws and np.isfinite(e1).all() and np.isfinite(e2).all() and np.mean(e2) > np.mean(e1)
    for k in ("loc", "log_scale", "h_loc", "hh_loc", "lpe_loc"):
        assert torch.isfinite(getattr(m1, k)).all(), k
    # (2) photo 1 alone == photo 1 inside the batch of two
    rows = slice(per, 2 * per)
    r2 = per // int(np.prod(cfg["hierarchical_patch_nums"]["level2"]))
    _, m_one, _, _, _ = _variant_model("kodak", per, 48, 1)
    with torch.no_grad():
        for k in ("loc", "log_scale", "lpe_loc", "lpe_log_scale"):
            getattr(m_one, k).copy_(getattr(m_all, k)[rows])
        for k in ("h_loc", "h_log_scale"):
            getattr(m_one, k).copy_(getattr(m_all, k)[r2:2 * r2])
        for k in ("hh_loc", "hh_log_scale"):
            getattr(m_one, k).copy_(getattr(m_all, k)[1:2])
    m_one.rng_row_offset = rows.start
    for m_, n_, y_ in ((m_all, 2 * per, Yd), (m_one, per, Yd[rows].contiguous())):
        m_.rng_seed_override = 0xC0DA
        m_.train(steps, lr, Xd[None].expand(n_, -1, -1), y_, *pri, lt, up, 1e-8, training_mappings=False)
        assert m_._ws is not None and m_._ws["graphs"] is not None
    for k, sl in (("loc", rows), ("log_scale", rows), ("lpe_loc", rows), ("h_loc", slice(r2, 2 * r2)), ("hh_loc", slice(1, 2))):
        assert_close_mostly(getattr(m_one, k), getattr(m_all, k)[sl].detach().cpu().numpy(), rtol=0, atol=2e-6, max_frac=2e-3,
                            hard_atol=2.5 * lr * steps, what=k)


def test_one_video_clip_at_full_geometry_width64_f16():
    """BASELINE configs[4] at its FULL geometry (config.py:94-114 with hidden_dims = [64] * 3, f16 operands): one clip = 1 x 8 x 8
    patches of 24 x 16 x 16 pixels = 64 INRs of 6144 pixels, stitched 3-D positional encodings.  Prior training: the step
    captures, replays, stays finite, improves.  Test time (main_compression.py:87-162): a prior from those steps, one clip
    optimised with five samples per step (finite, improving), then ONE A* encode round of level 1 -- every row's
    largest-KL group, 65 536 candidates, the certified fast scorer -- whose indices must be those of the exact scorer
    (the reference's arithmetic op for op, test_model.py:501-533) for every one of the 64 jobs."""
    import contextlib
    import io
    import warnings
    from recombiner_amd import drivers, utils
    per = 64
    cfg, m, lt, up, pri = _variant_model("video", 2 * per, 64, 2)
    assert cfg["pixel_sizes"] == [24, 16, 16] and int(np.prod(cfg["patch_nums"])) == per and m.dims == [34, 64, 64, 64, 3]
    X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], 2 * per, cfg["output_dim"], seed=0)
    Xd, Yd = X.to(DEV)[None].expand(2 * per, -1, -1), Y.to(DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _, _, e1 = m.train(6, 1e-3, Xd, Yd, *pri, lt, up, 1e-6, training_mappings=True)
        ws = m._ws
        assert ws is not None and ws["graphs"] is not None
        _, _, e2 = m.train(6, 1e-3, Xd, Yd, *pri, lt, up, 1e-6, training_mappings=True)
    assert m._ws is ws and np.isfinite(e1).all() and np.isfinite(e2).all() and np.mean(e2) > np.mean(e1)
    ck = drivers.build_checkpoint(m, lt, up, *pri, 1e-6)
    del m
    Xn, Yn = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], per, cfg["output_dim"], seed=3)
    with contextlib.redirect_stdout(io.StringIO()):
        tm = drivers.build_test_model(cfg, "video", ck, per, DEV, 42)
    tm.precision = 2
    Xt, Yt = Xn.to(DEV)[None].expand(per, -1, -1), Yn.to(DEV)

    def loss():
        with torch.no_grad():
            return float(((tm.predict(Xt, random_seed=7, sample_size=1) - Yt) ** 2).mean())
    l0 = loss()
    tm.train(Xt, Yt, 12, torch.optim.Adam(tm.parameters(), lr=2e-3), False, sample_size=5)
    l1 = loss()
    assert np.isfinite(l1) and l1 < l0
    for k in ("loc", "log_scale", "h_loc", "hh_loc"):
        assert torch.isfinite(getattr(tm, k)).all(), k
    # one encode round of level 1 through the batched driver path, against the exact scorer job by job
    lv = tm._l1
    LN2 = float(np.log(2.0))
    bits = tm._group_kls(lv) / LN2
    groups = torch.argmax(torch.where(lv.d_done.bool(), torch.full_like(bits, -1e10), bits), dim=1)
    rows = torch.arange(lv.rows, device=DEV)
    K = int(np.ceil(2 ** tm.bit_per_group))
    order = torch.sort(lv.d_glen[groups], stable=True)[1]
    exact = [tm._sample_group(lv, int(r), int(g_), K)[0] for r, g_ in zip(rows[order].tolist(), groups[order].tolist())]
    idx, _ = tm._encode_jobs(lv, rows, groups, K)
    assert lv.rows == per and K == 65536 and idx.cpu().tolist() == exact
    assert int(lv.d_done.sum()) == per                     # one group per row committed
