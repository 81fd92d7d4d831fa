"""CPU tests of the product's host logic and of the C-ABI library surface (no GPU compute)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from golden_util import O, ROOT, cfg_of, check, load, prior_inputs, regen_noise

from recombiner_amd import _lib, config, ops, utils
from recombiner_amd import prior_model as PM


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rcb.h")).read()
    declared = sorted(set(re.findall(r"\b(rcb_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 14
    assert os.path.exists(_lib.LIB_PATH), "librcb_hip.so not built (python -m recombiner_amd.build)"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/rcb.h but not exported"
    assert sorted(_lib.EXPORTS) == declared
    assert _lib.load().rcb_version() == _lib.ABI_VERSION == int(re.search(r"#define RCB_VERSION (\d+)", hdr).group(1))


def test_descriptor_mirrors_have_the_library_sizes():
    """every ctypes mirror of a descriptor is as large as the structure the library was compiled with (rcb_struct_bytes) --
    _lib.load() refuses to bind otherwise -- and an unknown selector is rejected"""
    lib = _lib.load()
    for which, cls in _lib._MIRRORS.items():
        assert lib.rcb_struct_bytes(which) == ctypes.sizeof(cls), cls.__name__
    assert lib.rcb_struct_bytes(99) == -1


def test_product_refuses_cpu_tensors():
    with pytest.raises(_lib.RcbError):
        ops.softplus_scale(torch.zeros(4))


def test_configs_match_reference_presets():
    for name in ("cifar", "protein"):
        assert config.configs[name] == cfg_of(load(f"prior_{name}.npz"))
    assert config.configs["kodak"]["hierarchical_patch_nums"] == {"level2": [4, 4], "level3": [8, 12]}
    assert config.configs["video"]["layerwise_scale_factors"] == [(6, 4, 4), 2, 2] or \
        config.configs["video"]["layerwise_scale_factors"] == [[6, 4, 4], 2, 2]
    assert config.configs["audio"]["patch_nums"] == [60]


def test_grouping_host_logic_exact():
    d = load("grouping.npz")
    names = ["group_idx", "start", "end", "group2param", "param2group", "n_groups", "group_kls", "weights"]
    for tag in "abc":
        r = PM.get_grouping_by_kl(d[f"{tag}_in"].copy())
        for k, v in zip(names, r):
            if k in ("group_kls", "weights"):
                np.testing.assert_allclose(np.asarray(v), d[f"{tag}_{k}"], rtol=1e-6)
            else:
                assert np.array_equal(np.asarray(v), d[f"{tag}_{k}"]), (tag, k)


def test_metrics_and_synthetic_inputs():
    d = load("metrics.npz")
    for ds in ["cifar", "kodak", "video", "audio", "protein"]:
        np.testing.assert_allclose(np.asarray(utils.metric(d["a"], d["b"], ds)), d["m_" + ds], rtol=1e-6)
    s = load("synthetic.npz")
    for name in ["cifar", "audio", "protein"]:
        X, Y = utils.synthetic_inputs(list(s[f"{name}_pixel_sizes"]), int(s[f"{name}_fourier_dim"]), 2, 3)
        np.testing.assert_allclose(X.numpy(), s[f"{name}_X"], atol=1e-6)
        assert Y.shape == (2, X.shape[0], 3) and float(Y.min()) >= 0 and float(Y.max()) < 1


@pytest.mark.parametrize("name", ["cifar", "protein", "patch2d", "patch1d", "patch3d"])
def test_patch_stitching_and_row_maps(name):
    d = load(f"prior_{name}.npz")
    cfg, geo, n, p, A, up, X, Y, pri = prior_inputs(d)
    eps = regen_noise(d, "fwd_eps")
    lpe = p["lpe_loc"] + O.st(p["lpe_log_scale"]) * eps[0]
    torch.manual_seed(124)
    net = PM.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"])
    with torch.no_grad():
        pe = utils.map_lpe_to_inr_inputs(net, lpe[None], cfg["latent_dim"], cfg["pixel_sizes"], cfg["upsample_factors"],
                                         cfg["patch"], cfg["patch_nums"], cfg["data_dim"])[:, 0]
    check(d, "fwd_pe", pe, rtol=1e-5, atol=1e-6)
    if cfg["patch"]:
        m2, m3 = utils.hierarchy_row_maps(n, cfg["patch_nums"], cfg["hierarchical_patch_nums"], cfg["data_dim"])
        o2, o3 = geo.level_maps(n)
        assert np.array_equal(m2, o2.numpy()) and np.array_equal(m3, o3.numpy())


def test_level_spec_inverse_maps():
    rows, cols, n = 4, 7, 12
    rm = np.repeat(np.arange(rows), 3)
    perm = np.stack([np.random.RandomState(c).permutation(rows) for c in range(cols)], 1)
    cm = np.random.RandomState(1).permutation(cols)[:5]
    lv = ops.LevelSpec(torch.zeros(rows, cols), torch.zeros(rows, cols), 5, n, row_map=rm, row_perm=perm, col_map=cm)
    mp, mi = lv.member_ptr.numpy(), lv.member_idx.numpy()
    for r in range(rows):
        assert set(mi[mp[r]:mp[r + 1]]) == set(np.flatnonzero(rm == r))
    inv = lv.row_perm_inv.numpy()
    for j in range(cols):
        assert np.array_equal(perm[inv[:, j], j], np.arange(rows))
    ci = lv.col_inv.numpy()
    for dcol, j in enumerate(cm):
        assert ci[j] == dcol
    assert (ci[np.setdiff1d(np.arange(cols), cm)] >= 5).all()


# ---------------------------------------------------------------------------------------------------
# N4: bitstream container (pure host logic)
# ---------------------------------------------------------------------------------------------------
def test_bitstream_container_round_trip_and_rejects_corruption():
    from recombiner_amd import bitstream as B
    rng = np.random.RandomState(0)
    for shapes in ([(5, 7)], [(6, 11), (3, 4), (1, 9)], [(0, 3)], [(2, 0), (1, 1), (1, 1)]):
        levels = [rng.randint(0, 65536, size=s) for s in shapes]
        blob = B.pack_indices(levels)
        out = B.unpack_indices(blob)
        assert len(out) == len(levels) and all(np.array_equal(a, b) and b.dtype == np.int64 for a, b in zip(levels, out))
        n_idx = sum(int(np.prod(s)) for s in shapes)
        assert B.payload_bits(blob) == 16 * n_idx                      # 16 bits per (row, group): test_model.py:245-250
        assert len(blob) == 8 + 8 * len(shapes) + 2 * n_idx + 4        # header + indices + CRC, nothing else
    blob = B.pack_indices([np.array([[0, 65535, 258]])])
    assert blob[16:22] == bytes([0, 0, 255, 255, 2, 1])               # little-endian uint16
    bad = bytearray(blob)
    bad[17] ^= 1
    for broken in (bytes(bad), blob[:-1], blob + b"\0", b"XXXX" + blob[4:], blob[:4] + b"\x02\x00" + blob[6:]):
        with pytest.raises(ValueError):
            B.unpack_indices(broken)
    for a in (np.array([[65536]]), np.array([[-1]]), np.array([[1.5]]), np.array([1, 2, 3])):
        with pytest.raises(ValueError):
            B.pack_indices([a])
    with pytest.raises(ValueError):
        B.pack_indices([np.zeros((1, 1))] * 4)                          # at most three levels
    with pytest.raises(ValueError):
        B.unpack_indices(B.MAGIC + b"\x01\x00\x04\x10" + b"\0" * 40)   # header claiming four levels


def test_upsample_pickles_and_deep_copies_without_its_fast_path_caches():
    """the 16-bit modes cache their evaluation wrappers on the Upsample module; a checkpoint pickle (drivers.save_checkpoint
    writes the module itself, as main_prior_training.py:334-335 does) must stay loadable by the reference and a deep copy must
    not keep evaluating the original's parameters"""
    import copy
    import io
    import pickle
    from recombiner_amd import prior_model as PM
    from recombiner_amd.upsample_fast import phase_module, stitched2d_module
    net = PM.Upsample(2, [2, 1, 1], [4, 2, 2])
    assert phase_module(net) is not None and callable(stitched2d_module(net))
    assert any(k.startswith("_rcb_") for k in net.__dict__)
    blob = pickle.dumps(net)
    assert b"upsample_fast" not in blob and b"_rcb_" not in blob
    back = pickle.load(io.BytesIO(blob))
    assert not any(k.startswith("_rcb_") for k in back.__dict__)
    for a, b in zip(net.state_dict().values(), back.state_dict().values()):
        assert torch.equal(a, b)
    twin = copy.deepcopy(net)
    assert not any(k.startswith("_rcb_") for k in twin.__dict__)
    assert phase_module(twin).net is twin and phase_module(net).net is net


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` starts N RCCL ranks itself; with fewer GPUs than ranks it must fail loudly instead of
    silently running one rank and printing n_gpus: 1 (no GPU call happens before the check: device_count() only)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RCB_DIST_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 64 requested" in out.stderr and "{" not in out.stdout
    # a launcher that started a different number of ranks than --gpus says is refused as well
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "launcher started 1 rank" in out.stderr


def test_in_rank_capture_probe_decisions(monkeypatch):
    """bench.py under an external launcher (torchrun ... bench.py --gpus N): every rank probes the captured form of the sharded
    step in a CHILD process before it touches the GPU.  Host logic only: nothing is probed when the caller or bench.py's own
    launcher already chose, or on a gloo rehearsal; a probe that cannot run (here: no GPU for the child) selects the safe
    four-segment form instead of raising."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    sys.modules["bench_under_test"] = bench
    spec.loader.exec_module(bench)
    for k in ("RCB_CAPTURE_COLLECTIVES", "RCB_STEP_FORM_CHOICE", "RCB_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("RCB_CAPTURE_COLLECTIVES", "0")
    assert bench.probe_capture_form_in_rank(2) is None
    monkeypatch.delenv("RCB_CAPTURE_COLLECTIVES")
    monkeypatch.setenv("RCB_STEP_FORM_CHOICE", "four segments (gloo rehearsal: no probe)")
    assert bench.probe_capture_form_in_rank(2) is None
    monkeypatch.delenv("RCB_STEP_FORM_CHOICE")
    monkeypatch.setenv("RCB_DIST_BACKEND", "gloo")
    assert bench.probe_capture_form_in_rank(2) is None
    monkeypatch.delenv("RCB_DIST_BACKEND")
    if not torch.cuda.is_available():                    # the child finds no GPU: exit code != 0 -> the safe form, no exception
        monkeypatch.setenv("RCB_PROBE_TIMEOUT_S", "120")
        monkeypatch.setenv("MASTER_PORT", "29931")
        assert bench.probe_capture_form_in_rank(1) is False


REF_DIR = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF_DIR), reason="needs the reference checkout (build container only)")
def test_reference_unpickles_a_checkpoint_written_by_the_product(tmp_path):
    """N2, product -> reference: a checkpoint written by drivers.save_checkpoint is read by the REFERENCE's own
    main_compression.py load sequence (eight plain pickle.load calls with the reference's prior_model on the path), the
    two modules come back as the reference's classes (without the product's fast-path caches), and the reference's
    TestBNNmodel built from them runs predict().  Runs in a child process (the module name `prior_model` must be the
    reference's there); skipped where the reference is absent."""
    import subprocess
    import sys
    from recombiner_amd import drivers
    cfg = config.configs["cifar"]
    dims = [cfg["input_dim"]] + cfg["hidden_dims"] + [cfg["output_dim"]]
    torch.manual_seed(3)
    lt = PM.LinearTransform(dims)
    up = PM.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"])
    from recombiner_amd.upsample_fast import phase_module
    phase_module(up)                                                  # the fast-path cache must not travel
    D = 3267 + 512
    g1 = PM.get_grouping_by_kl(np.random.RandomState(0).gamma(0.7, 4.0, size=D).astype(np.float32))
    gen = torch.Generator().manual_seed(1)
    p_loc, p_scale = 0.01 * torch.randn(D, generator=gen), 0.02 + 0.01 * torch.rand(D, generator=gen)
    avg_ls = -4 + 0.1 * torch.randn(D, generator=gen)
    ck = [g1, (p_loc, p_scale, 3e-7, avg_ls), (None,) * 8, (None, None, 3e-7, None), (None,) * 8, (None, None, 3e-7, None), lt, up]
    path = os.path.join(tmp_path, "PRIOR.pkl")
    drivers.save_checkpoint(path, ck)
    assert PM.LinearTransform.__module__ == "recombiner_amd.prior_model" and "prior_model" not in sys.modules
    child = r"""
import pickle, sys, numpy as np, torch
sys.path.insert(0, %r)
import prior_model, test_model, config
assert prior_model.__file__.startswith(%r)
with open(%r, "rb") as f:
    group_idx, gs, ge, g2p, p2g, n_groups, gk, w = pickle.load(f)
    prior_loc, prior_scale, kl_beta, avg_ls = pickle.load(f)
    for _ in range(4):
        pickle.load(f)
    linear_transform = pickle.load(f)
    upsample_net = pickle.load(f)
assert type(linear_transform) is prior_model.LinearTransform and type(upsample_net) is prior_model.Upsample
assert not any(k.startswith("_rcb_") for k in upsample_net.__dict__)
c = config.configs["cifar"]
tm = test_model.TestBNNmodel(c["input_dim"], c["hidden_dims"], c["output_dim"], 2, c["upsample_factors"], c["latent_dim"],
                             c["data_dim"], c["pixel_sizes"], False, None, None, "cifar", linear_transform=linear_transform,
                             upsample_net=upsample_net, p_loc=prior_loc.clone()[p2g],
                             p_log_scale=torch.log(torch.exp(prior_scale * 6) - 1).clone()[p2g], init_log_scale=avg_ls[p2g],
                             param_to_group=p2g, group_to_param=g2p, n_groups=n_groups, group_start_index=gs,
                             group_end_index=ge, group_idx=group_idx, device="cpu", initial_beta=kl_beta)
x = torch.from_numpy(np.load(%r))
with torch.no_grad():
    y = tm.predict(x[None].repeat(2, 1, 1), random_seed=3)
np.save(%r, y.numpy())
print("REF LOAD OK", n_groups)
""" % (REF_DIR, REF_DIR, path, os.path.join(tmp_path, "x.npy"), os.path.join(tmp_path, "y.npy"))
    X, _ = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], 1, 3, seed=0)
    np.save(os.path.join(tmp_path, "x.npy"), X.numpy())
    out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600,
                         env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
    assert out.returncode == 0 and "REF LOAD OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    y_ref = np.load(os.path.join(tmp_path, "y.npy"))
    assert y_ref.shape == (2, 1024, 3) and np.isfinite(y_ref).all() and float(np.abs(y_ref).max()) > 0
    # and the same file read back by the product's loader gives the objects that were written
    back = drivers.load_checkpoint(path)
    assert all(torch.equal(a, b) for a, b in zip(back[6].A, lt.A)) and torch.equal(back[7].conv3.weight, up.conv3.weight)
    assert np.array_equal(back[0][4], g1[4]) and back[1][2] == 3e-7
