"""Datapoint sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on ROCm, "gloo" in CPU tests).

The per-INR losses are independent given the shared (A, Upsample, prior), so INRs shard
embarrassingly: whole datapoints (all patches of an image / clip) stay on one rank.  The only
cross-INR reductions of the reference are
  * the closed-form prior refit, main_prior_training.py:157-172  (mean / unbiased var over INRs),
  * the scalar KL that drives the beta rule, main_prior_training.py:136-154,
  * the per-parameter mean KL behind get_grouping, prior_model.py:268-270,
and, only when the shared mappings are trained, the per-step gradient sum of A / Upsample
(handled inside PriorBNNmodel.train).  Each is a small all-reduce of fp64 sufficient statistics.
"""
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as td


def world(group=None) -> Tuple[int, int]:
    if td.is_available() and td.is_initialized():
        return td.get_rank(group), td.get_world_size(group)
    return 0, 1


def is_dist() -> bool:
    return bool(td.is_available() and td.is_initialized())


def shard_range(n_datapoints: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block of whole datapoints for `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_datapoints, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def merge_moments(count: torch.Tensor, total: torch.Tensor, m2: torch.Tensor, sig2: torch.Tensor, group=None):
    """Chan et al. pairwise merge of per-shard (n, sum, M2, sum sigma^2) via one all-gather-free
    all-reduce: with mean_k = sum_k / n_k,
        N = sum n_k ; S = sum sum_k ; M2 = sum M2_k + sum n_k (mean_k - S/N)^2
    computed as  sum M2_k + sum sum_k^2 / n_k  -  S^2 / N  in fp64."""
    rank, ws = world(group)
    if ws == 1:
        return count, total, m2, sig2
    n = count.to(torch.float64).reshape(1).to(total.device)
    pack = torch.cat([n, total, m2 + total * total / n, sig2])
    td.all_reduce(pack, group=group)
    cols = total.numel()
    N = pack[0]
    S = pack[1:1 + cols]
    M2 = pack[1 + cols:1 + 2 * cols] - S * S / N
    SG = pack[1 + 2 * cols:]
    return N.reshape(()), S, M2, SG


def prior_from_moments(count, total, m2, sig2, out_shape=None):
    """mu_p = mean(mu_q); sigma_p = sqrt(mean(sigma_q^2) + var_unbiased(mu_q))  (main_prior_training.py:157-172)."""
    n = float(count)
    mu = (total / n).to(torch.float32)
    sig = torch.sqrt(sig2 / n + m2 / (n - 1.0)).to(torch.float32)
    if out_shape is not None:
        mu, sig = mu.reshape(out_shape), sig.reshape(out_shape)
    return mu, sig


def refit_prior(loc: torch.Tensor, log_scale: torch.Tensor, group=None):
    """Prior refit over *all* ranks' INRs for one parameter tensor [rows, ...].  The column sums are exact fixed-point
    integers (ops.col_moments_fx) and the all-reduce over ranks adds integers: the refit is bitwise the same however the
    rows are sharded (and from run to run)."""
    from . import ops
    fx = ops.col_moments_fx(loc, log_scale)
    pack = torch.cat([fx.reshape(-1), torch.tensor([loc.shape[0]], dtype=torch.int64, device=loc.device)])
    rank, ws = world(group)
    if ws > 1:
        td.all_reduce(pack, group=group)
    n = int(pack[-1])
    s, m2, sg = ops.moments_from_fx(pack[:-1], n)
    return prior_from_moments(torch.tensor(float(n), dtype=torch.float64), s, m2, sg, loc.shape[1:])


class GradBuckets:
    """The per-step collective of sharded prior training: the gradients of the shared mappings, summed over the ranks in TWO
    buckets of ONE flat fp32 buffer -- [ A matrices (layer order) | parameters of the upsampling net (module order) ] -- so
    that the large first bucket (13.4 of the 14.4 MB on the CIFAR preset) can travel while the upsampling net's backward
    is still running (PriorBNNmodel.train: segment 1a -> reduce(0) || segment 1b -> reduce(1) || segment 2 -> wait ->
    Adam).  `pack` copies a list of gradients into a bucket (one torch.cat with out=) and returns VIEWS of the flat buffer
    in the gradients' shapes: what Adam reads after the wait.  Works on any device / backend (RCCL in production, gloo on
    CPU tensors in tests/test_dist_gloo.py)."""

    def __init__(self, a_shapes, conv_shapes, device, group=None, dtype=torch.float32):
        self.group = group
        self.shapes = ([tuple(s_) for s_ in a_shapes], [tuple(s_) for s_ in conv_shapes])
        numel = [sum(int(np.prod(s_)) for s_ in part) for part in self.shapes]
        self.n_a = numel[0]
        self.flat = torch.empty(numel[0] + numel[1], device=device, dtype=dtype)
        self.bounds = ((0, numel[0]), (numel[0], numel[0] + numel[1]))

    def bucket(self, part: int) -> torch.Tensor:
        lo, hi = self.bounds[part]
        return self.flat[lo:hi]

    def pack(self, part: int, grads):
        """grads (tensors in the bucket's order and shapes) -> views of the flat buffer holding their values"""
        shapes = self.shapes[part]
        if len(grads) != len(shapes) or any(tuple(g.shape) != s_ for g, s_ in zip(grads, shapes)):
            raise ValueError("GradBuckets.pack: gradients do not match the bucket's parameter list")
        torch.cat([g.reshape(-1) for g in grads], out=self.bucket(part))
        views, k = [], self.bounds[part][0]
        for s_ in shapes:
            n = int(np.prod(s_))
            views.append(self.flat[k:k + n].view(s_))
            k += n
        return views

    def reduce(self, part: int, async_op=True):
        """sum of the bucket over the ranks of the group, in place; returns the work handle (None for a single process
        without a process group)"""
        if not (td.is_available() and td.is_initialized()):
            return None
        return td.all_reduce(self.bucket(part), group=self.group, async_op=async_op)


def allreduce_scalar(v: torch.Tensor, group=None) -> torch.Tensor:
    rank, ws = world(group)
    if ws > 1:
        v = v.clone()
        td.all_reduce(v, group=group)
    return v


def grouping_weights(kl_colsum: torch.Tensor, n_rows_local: int, group=None) -> np.ndarray:
    """mean-over-INRs KL in bits per parameter across ranks -> fp32 numpy weights for get_grouping_by_kl.  kl_colsum: the
    exact fixed-point column sums of ops.gauss_kl_colsum_fx (int64 [cols + 1]: every element is rounded to the integer grid
    on its own and the ranks' results are added as integers, so the grouping does not depend on how the rows are sharded --
    any cut, not only multiples of the kernel's 256-row blocks) or plain fp64 sums."""
    rank, ws = world(group)
    if kl_colsum.dtype == torch.int64:
        from . import ops
        pack = torch.cat([kl_colsum, torch.tensor([n_rows_local], dtype=torch.int64, device=kl_colsum.device)])
        if ws > 1:
            td.all_reduce(pack, group=group)
        return (ops.colsum_from_fx(pack[:-1]) / np.log(2.) / float(pack[-1])).to(torch.float32).cpu().numpy()
    pack = torch.cat([kl_colsum.to(torch.float64), torch.tensor([float(n_rows_local)], dtype=torch.float64,
                                                                device=kl_colsum.device)])
    if ws > 1:
        td.all_reduce(pack, group=group)
    return (pack[:-1] / np.log(2.) / pack[-1]).to(torch.float32).cpu().numpy()
