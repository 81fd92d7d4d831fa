"""Multi-process CPU tests (gloo, world_size 2 and 3) of the datapoint-sharding helpers: the prior
refit from merged moments must equal the single-process refit over all INRs (oracle), and the
grouping weights / scalar KL must reduce correctly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_moments(loc, ls):
    """what rcb_col_moments returns for a shard (fp64 sum, M2, sum sigma^2), computed with torch on CPU."""
    from oracle import ref_cpu as O
    x = loc.double()
    s = x.sum(0)
    m2 = ((x - x.mean(0)) ** 2).sum(0)
    sg = (O.st(ls) ** 2).double().sum(0)
    return s, m2, sg


def _worker(rank, ws, port, n_dp, ppd, cols, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=ws)
    from oracle import ref_cpu as O
    from recombiner_amd import dist
    g = torch.Generator().manual_seed(0)
    loc = 0.3 + 0.05 * torch.randn(n_dp * ppd, cols, generator=g)
    ls = -4 + torch.randn(n_dp * ppd, cols, generator=g)
    lo, hi = dist.shard_range(n_dp, rank, ws)
    sl = slice(lo * ppd, hi * ppd)                      # whole datapoints per rank
    s, m2, sg = _local_moments(loc[sl], ls[sl])
    cnt = torch.tensor(float((hi - lo) * ppd), dtype=torch.float64)
    n, S, M2, SG = dist.merge_moments(cnt, s, m2, sg)
    mu, sig = dist.prior_from_moments(n, S, M2, SG)
    mu_ref, sig_ref = O.refit_prior(loc, ls)            # single-process reference over ALL rows
    np.testing.assert_allclose(mu.numpy(), mu_ref.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(sig.numpy(), sig_ref.numpy(), rtol=1e-5)
    # scalar KL and grouping weights
    pl, ps = torch.zeros(cols), torch.full((cols,), 0.05)
    kl_el = O.gauss_kl_elem(loc, O.st(ls), pl, ps).double()
    part = kl_el[sl].sum()
    tot = dist.allreduce_scalar(part.reshape(1))
    np.testing.assert_allclose(tot.item(), kl_el.sum().item(), rtol=1e-12)
    w = dist.grouping_weights(kl_el[sl].sum(0), (hi - lo) * ppd)
    w_ref = (kl_el / np.log(2.)).mean(0).float().numpy()
    np.testing.assert_allclose(w, w_ref, rtol=1e-6)
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    td.destroy_process_group()


@pytest.mark.parametrize("ws,n_dp", [(2, 5), (3, 7)])
def test_prior_aggregation_matches_single_process(tmp_path, ws, n_dp):
    port = _free_port()
    mp.spawn(_worker, args=(ws, port, n_dp, 4, 37, str(tmp_path)), nprocs=ws, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(ws))


def test_shard_range_covers_everything():
    from recombiner_amd import dist
    for n in (1, 7, 8, 500):
        for ws in (1, 2, 3, 8):
            seen = []
            for r in range(ws):
                lo, hi = dist.shard_range(n, r, ws)
                seen += list(range(lo, hi))
            assert seen == list(range(n))


def _mapping_grad_worker(rank, ws, port, out_dir):
    """the per-step collective of sharded prior training: sum of the shared-mapping gradients."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=ws)
    g = torch.Generator().manual_seed(1)
    full = torch.randn(ws, 11, generator=g)
    mine = full[rank].clone()
    flat = torch.cat([mine[:4], mine[4:]])
    td.all_reduce(flat)
    np.testing.assert_allclose(flat.numpy(), full.sum(0).numpy(), rtol=1e-6)
    open(os.path.join(out_dir, f"g{rank}"), "w").write("ok")
    td.destroy_process_group()


def test_mapping_gradient_allreduce(tmp_path):
    port = _free_port()
    mp.spawn(_mapping_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "g0") and os.path.exists(tmp_path / "g1")
