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


def _bucket_worker(rank, ws, port, out_dir):
    """PRODUCT code of the per-step collective (recombiner_amd.dist.GradBuckets, what PriorBNNmodel.train packs its mapping
    gradients into and reduces between the captured segments) on CPU tensors over gloo: flat layout [A matrices | upsampling
    net], the views Adam later reads alias the flat buffer in parameter order, bucket 0 can be reduced while bucket 1 is
    still being packed (the segment order of the step), and the result is the sum over ranks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=ws)
    from recombiner_amd import dist
    a_shapes = [(7, 7), (5, 5), (3, 3)]                       # stand-ins for A[l] [L_l, L_l]
    c_shapes = [(4, 2, 3), (4,), (2, 4, 3), (2,)]             # conv weights / biases of the upsampling net
    gen = torch.Generator().manual_seed(100)
    every = [[torch.randn(s_, generator=gen) for s_ in a_shapes + c_shapes] for _ in range(ws)]      # every rank's gradients
    mine = every[rank]
    b = dist.GradBuckets(a_shapes, c_shapes, "cpu", None)
    assert b.flat.numel() == 83 + 24 + 4 + 24 + 2 and b.n_a == 83
    va = b.pack(0, mine[:3])                                  # end of segment 1a: the A gradients are ready
    h0 = b.reduce(0, async_op=True)                           # ... and travel
    vc = b.pack(1, mine[3:])                                  # segment 1b: the conv gradients, packed while bucket 0 is in flight
    h1 = b.reduce(1, async_op=True)
    h0.wait()
    h1.wait()
    want = [sum(every[r][i] for r in range(ws)) for i in range(len(mine))]
    for v, w_ in zip(va + vc, want):
        np.testing.assert_allclose(v.numpy(), w_.numpy(), rtol=1e-6, atol=1e-6)
    # the views ARE the flat buffer, A matrices first, each bucket in parameter order
    off = 0
    for v in va + vc:
        assert v.data_ptr() == b.flat.data_ptr() + 4 * off and v.is_contiguous()
        off += v.numel()
    assert b.bucket(0).data_ptr() == b.flat.data_ptr() and b.bucket(1).data_ptr() == b.flat.data_ptr() + 4 * b.n_a
    try:
        b.pack(1, mine[:3])
        raise AssertionError("a gradient list of the wrong bucket must be refused")
    except ValueError:
        pass
    # the EXACT cross-rank sums of the prior refit / grouping (what the product's kernels return as int64 fixed point, here
    # built with the same splitting rule on the CPU): an integer all-reduce makes the refit independent of the sharding
    from recombiner_amd import ops
    g2 = torch.Generator().manual_seed(7)
    rows, cols = 11 * ws + 3, 29
    loc = 0.3 + 0.05 * torch.randn(rows, cols, generator=g2)
    lo, hi = dist.shard_range(rows, rank, ws)

    def fx(x):                       # the rule of rcb_col_moments for one quantity: hi = floor(v 2^30), lo = rint(frac 2^32)
        v = x.double() * ops.MOM_FX
        f = torch.floor(v)
        return torch.stack([f.sum(0), torch.round((v - f) * ops.MOM_FX_LO).sum(0)]).to(torch.int64)
    part = fx(loc[lo:hi])
    td.all_reduce(part)
    assert torch.equal(part, fx(loc)), "integer sums must not depend on the sharding"
    open(os.path.join(out_dir, f"g{rank}"), "w").write("ok")
    td.destroy_process_group()


@pytest.mark.parametrize("ws", [2, 3])
def test_gradient_buckets_of_the_sharded_step(tmp_path, ws):
    port = _free_port()
    mp.spawn(_bucket_worker, args=(ws, port, str(tmp_path)), nprocs=ws, join=True)
    assert all(os.path.exists(tmp_path / f"g{r}") for r in range(ws))
