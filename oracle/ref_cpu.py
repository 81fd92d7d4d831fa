"""CPU oracle for the RECOMBINER per-datapoint INR hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a plain PyTorch/NumPy (CPU) restatement of the arithmetic of the reference
implementation (cambridge-mlg/RECOMBINER, files cited per function as ``file:line`` relative
to the reference root).  It exists to *check* the HIP path: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
Nothing under ``recombiner_amd/`` imports it and the product never falls back to it.

Parity status: PINNED.  Every function below is checked against golden vectors produced by
importing the reference itself in the build container (``oracle/make_golden.py`` ->
``tests/golden/*.npz``; see ``tests/test_oracle_golden.py``).  The reference ships no tests
or fixtures of its own for this path.

Design notes (how this differs from the reference's code while keeping its arithmetic):
  * functional style over plain tensors / small state objects instead of nn.Module classes;
  * hierarchical (patched) models are expressed through integer row maps (patch -> level-2
    row, patch -> level-3 row) instead of repeat/reshape chains -- same values, same noise
    shapes and draw order;
  * noise comes from a ``Noise`` source which either draws ``torch.randn`` from the global
    CPU generator in the reference's order (lpe, level-1, level-2, level-3) or replays a
    recorded list, so the HIP path can be fed bit-identical epsilons.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch.distributions import Normal, kl_divergence

LN2 = math.log(2.0)


# ----------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------
def st(x: torch.Tensor) -> torch.Tensor:
    """std transform softplus(x)/6   (prior_model.py:88, test_model.py:101)."""
    return F.softplus(x, beta=1, threshold=20) / 6


def inv_st(s: torch.Tensor) -> torch.Tensor:
    """log(exp(6 s) - 1)   (main_compression.py:51)."""
    return torch.log(torch.exp(s * 6) - 1)


def layer_param_counts(in_dim: int, hidden_dims: Sequence[int], out_dim: int):
    """Per-layer parameter counts out*(in+1) and their cumsum (utils.py:216-232)."""
    dims = [in_dim] + list(hidden_dims) + [out_dim]
    sizes = [dims[i + 1] * (dims[i] + 1) for i in range(len(dims) - 1)]
    return dims, sizes, np.cumsum(sizes)


class Noise:
    """Standard-normal source.  ``replay``: list of tensors consumed in order; otherwise the
    global torch CPU generator is used (``torch.randn`` == the reference's ``randn_like``).
    Everything handed out is appended to ``self.drawn``."""

    def __init__(self, replay: Optional[List[torch.Tensor]] = None):
        self.replay = list(replay) if replay is not None else None
        self.drawn: List[torch.Tensor] = []

    def __call__(self, shape) -> torch.Tensor:
        if self.replay is not None:
            e = self.replay.pop(0)
            assert tuple(e.shape) == tuple(shape), (tuple(e.shape), tuple(shape))
        else:
            e = torch.randn(tuple(shape))
        self.drawn.append(e)
        return e


# ----------------------------------------------------------------------------------------
# A22: synthetic inputs  (utils.py:265-297, data/image.py:24-27)
# ----------------------------------------------------------------------------------------
def coord_grid(pixel_sizes: Sequence[int]) -> torch.Tensor:
    """Pixel-centre coordinates in (-1,1), 'ij' meshgrid, row-major flattened [P, dd]."""
    axes = [-1 + 2 * ((0.5 + torch.arange(s)) / s) for s in pixel_sizes]
    g = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1)
    return g.reshape(-1, len(pixel_sizes))


def fourier_features(pixel_sizes: Sequence[int], fourier_dim: int) -> torch.Tensor:
    """X[P, F] = cat(cos(pi v), sin(pi v)), v = coord (x) w, w = exp(linspace(0, ln 1024, F/(2 dd)))."""
    c = coord_grid(pixel_sizes)
    dd = len(pixel_sizes)
    w = torch.exp(torch.linspace(0, np.log(1024), fourier_dim // (2 * dd)))
    v = torch.matmul(c.unsqueeze(-1), w.unsqueeze(0)).view(c.shape[0], -1)
    return torch.cat([torch.cos(np.pi * v), torch.sin(np.pi * v)], dim=-1)


# ----------------------------------------------------------------------------------------
# model geometry shared by the prior-time and test-time models
# ----------------------------------------------------------------------------------------
@dataclass
class Geometry:
    in_dim: int
    hidden_dims: List[int]
    out_dim: int
    data_dim: int
    pixel_sizes: List[int]
    upsample_factors: List[int]
    latent_dim: int
    patch: bool
    patch_nums: Optional[List[int]]
    hierarchical_patch_nums: Optional[dict]
    paddings: List[int] = field(default_factory=lambda: [2, 1, 1])
    layerwise_scale_factors: list = field(default_factory=lambda: [4, 2, 2])
    w0: float = 30.0
    c: float = 6.0

    @staticmethod
    def from_config(cfg: dict) -> "Geometry":
        return Geometry(cfg["input_dim"], list(cfg["hidden_dims"]), cfg["output_dim"], cfg["data_dim"],
                        list(cfg["pixel_sizes"]), list(cfg["upsample_factors"]), cfg["latent_dim"],
                        bool(cfg["patch"]), cfg["patch_nums"], cfg["hierarchical_patch_nums"],
                        list(cfg["paddings"]), list(cfg["layerwise_scale_factors"]))

    @property
    def dims(self):
        return [self.in_dim] + list(self.hidden_dims) + [self.out_dim]

    @property
    def layer_sizes(self):
        return layer_param_counts(self.in_dim, self.hidden_dims, self.out_dim)[1]

    @property
    def cum(self):
        return layer_param_counts(self.in_dim, self.hidden_dims, self.out_dim)[2]

    @property
    def d_net(self):
        return int(self.cum[-1])

    @property
    def latent_grid(self):
        return [self.pixel_sizes[i] // self.upsample_factors[i] for i in range(self.data_dim)]

    @property
    def n_pix(self):
        return int(np.prod(self.pixel_sizes))

    @property
    def patches_per_datum(self):
        return int(np.prod(self.patch_nums)) if self.patch else 1

    def level_maps(self, n_inr: int):
        """patch index -> level-2 row, level-3 row   (utils.py:151-185 expressed as gathers)."""
        assert self.patch
        ppd = self.patches_per_datum
        l2 = self.hierarchical_patch_nums["level2"]
        ngrp = [self.patch_nums[i] // l2[i] for i in range(self.data_dim)]
        n = np.arange(n_inr)
        datum, local = n // ppd, n % ppd
        pos = np.stack(np.unravel_index(local, self.patch_nums), -1)          # [n, dd]
        grp = np.ravel_multi_index(tuple((pos[:, i] // l2[i]) for i in range(self.data_dim)), ngrp)
        map2 = datum * int(np.prod(ngrp)) + grp
        map3 = datum
        return torch.from_numpy(map2).long(), torch.from_numpy(map3).long()


# ----------------------------------------------------------------------------------------
# A2 / A3: shared mappings
# ----------------------------------------------------------------------------------------
def make_linear_transform(dims: Sequence[int], seed: Optional[int] = None) -> List[torch.Tensor]:
    """A[l] ~ U(-1,1)/L_l, L_l = d_{l+1}(d_l+1)   (prior_model.py:16-21)."""
    if seed is not None:
        torch.manual_seed(seed)
    out = []
    for i in range(1, len(dims)):
        L = dims[i] * (dims[i - 1] + 1)
        out.append((torch.rand(L, L) * 2 - 1) / L)
    return out


class UpsampleNet:
    """up(s0) -> conv k5 -> LeakyReLU -> up(s1) -> conv k3 -> LeakyReLU -> up(s2) -> conv k3
    with channels 128 -> 64 -> 64 -> 16 and nearest upsampling   (prior_model.py:23-59).
    Holds plain weight tensors [w1,b1,w2,b2,w3,b3]; default nn.ConvNd init (RNG order kept)."""

    def __init__(self, kernel_dim: int, paddings, scale_factors, seed: Optional[int] = None):
        if seed is not None:
            torch.manual_seed(seed)
        conv = {1: torch.nn.Conv1d, 2: torch.nn.Conv2d, 3: torch.nn.Conv3d}[kernel_dim]
        mods = [conv(128, 64, 5, padding=paddings[0]), conv(64, 64, 3, padding=paddings[1]),
                conv(64, 16, 3, padding=paddings[2])]
        self.kernel_dim = kernel_dim
        self.paddings = list(paddings)
        self.scale_factors = list(scale_factors)
        self.weights = []
        for m in mods:
            self.weights += [m.weight.detach().clone(), m.bias.detach().clone()]

    def parameters(self):
        return self.weights

    def __call__(self, x):
        conv = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[self.kernel_dim]
        w = self.weights
        x = F.interpolate(x, scale_factor=self.scale_factors[0], mode="nearest")
        x = F.leaky_relu(conv(x, w[0], w[1], padding=self.paddings[0]), 0.01)
        x = F.interpolate(x, scale_factor=self.scale_factors[1], mode="nearest")
        x = F.leaky_relu(conv(x, w[2], w[3], padding=self.paddings[1]), 0.01)
        x = F.interpolate(x, scale_factor=self.scale_factors[2], mode="nearest")
        return conv(x, w[4], w[5], padding=self.paddings[2])


# ----------------------------------------------------------------------------------------
# A4: latent positional encodings -> per-pixel INR inputs   (utils.py:4-120)
# ----------------------------------------------------------------------------------------
def lpe_to_pe(up, lpe: torch.Tensor, geo: Geometry) -> torch.Tensor:
    """lpe [S, N, *lat, C] -> pe [N, S, P, 16].  Patched presets are stitched into one grid per
    datapoint before the conv net and cut back into patches afterwards."""
    S, N = lpe.shape[:2]
    dd = geo.data_dim
    lat = geo.latent_grid
    C = geo.latent_dim
    lpe = lpe.reshape(S, N, *lat, C)
    if not geo.patch:
        z = lpe.reshape(S * N, *lat, C).movedim(-1, 1)                      # channels first
        o = up(z).movedim(1, -1)                                             # [S*N, *px, 16]
        pe = o.reshape(S, N, -1, o.shape[-1])
    else:
        pn = list(geo.patch_nums)
        nd = N // int(np.prod(pn))
        z = lpe.reshape(S, nd, *pn, *lat, C)
        # (pn0, pn1, .., lat0, lat1, ..) -> (pn0, lat0, pn1, lat1, ..)
        order = [0, 1] + [2 + i + j * dd for i in range(dd) for j in range(2)] + [2 + 2 * dd]
        z = z.permute(order).reshape(S * nd, *[pn[i] * lat[i] for i in range(dd)], C)
        o = up(z.movedim(-1, 1)).movedim(1, -1)                              # [S*nd, *(pn*px), 16]
        E = o.shape[-1]
        split = []
        for i in range(dd):
            split += [pn[i], geo.pixel_sizes[i]]
        o = o.reshape(S, nd, *split, E)
        back = [0, 1] + [2 + 2 * i for i in range(dd)] + [3 + 2 * i for i in range(dd)] + [2 + 2 * dd]
        pe = o.permute(back).reshape(S, N, -1, E)
    return pe.permute(1, 0, 2, 3)


# ----------------------------------------------------------------------------------------
# A5: reparameterised sampling of the (1- or 3-level) latent INR weights (utils.py:122-198)
# ----------------------------------------------------------------------------------------
def sample_latent_weights(geo: Geometry, loc, scale, h_loc, h_scale, hh_loc, hh_scale, S: int,
                          noise: Noise) -> torch.Tensor:
    """-> h_w [N, S, D_net].  Noise order: level-1, level-2, level-3; level-2/3 noise is drawn
    per patch *after* broadcasting the shared row (utils.py:181,189)."""
    N, D = loc.shape
    sc1 = scale[:, None, :].repeat(1, S, 1)
    out = loc[:, None, :] + sc1 * noise((N, S, D))
    if geo.patch:
        m2, m3 = geo.level_maps(N)
        sc2 = h_scale[m2][:, None, :].repeat(1, S, 1)
        h = h_loc[m2][:, None, :] + noise((N, S, D)) * sc2
        sc3 = hh_scale[m3][:, None, :].repeat(1, S, 1)
        hh = hh_loc[m3][:, None, :] + sc3 * noise((N, S, D))
        out = out + h + hh
    return out


# ----------------------------------------------------------------------------------------
# A6 / A12: the batched SIREN itself
# ----------------------------------------------------------------------------------------
def siren_apply(geo: Geometry, x: torch.Tensor, h_w: torch.Tensor, A: Sequence[torch.Tensor]):
    """x [N,P,in] with h_w [N,D]  (prior_model.py:168-179)   or
    x [N,S,P,in] with h_w [N,S,D] (test_model.py:347-353).
    Layer vector layout [bias(out) | W(in,out) row-major] (prior_model.py:121-127)."""
    dims, cum = geo.dims, geo.cum
    nl = len(dims) - 1
    for l in range(nl):
        lo = 0 if l == 0 else int(cum[l - 1])
        v = h_w[..., lo:int(cum[l])] @ A[l]
        b = v[..., :dims[l + 1]].unsqueeze(-2)
        W = v[..., dims[l + 1]:].reshape(*v.shape[:-1], dims[l], dims[l + 1])
        x = (x @ W) + b
        if l != nl - 1:
            x = torch.sin(geo.w0 * x)
    return x


def gauss_kl_elem(mu_q, sig_q, mu_p, sig_p):
    """Elementwise KL(N(mu_q,sig_q) || N(mu_p,sig_p)) through torch.distributions, as the
    reference does (prior_model.py:191-199; torch kl.py _kl_normal_normal)."""
    return kl_divergence(Normal(mu_q, sig_q), Normal(mu_p, sig_p))


# ----------------------------------------------------------------------------------------
# A1, A6-A9: prior-time model
# ----------------------------------------------------------------------------------------
LEVEL_KEYS = ["loc", "log_scale", "h_loc", "h_log_scale", "hh_loc", "hh_log_scale", "lpe_loc",
              "lpe_log_scale"]


def init_prior_params(geo: Geometry, n_inr: int, seed: int = 42, init_log_scale: float = -4.0) -> dict:
    """Per-INR variational parameters (prior_model.py:100-110); RNG draw order loc, h_loc,
    hh_loc, lpe_loc under torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    D = geo.d_net
    w_std = np.sqrt(geo.c / geo.hidden_dims[-1]) / geo.w0
    p = {}

    def uni(rows):
        return torch.rand(rows, D) * w_std * 2 - w_std
    p["loc"] = uni(n_inr)
    p["log_scale"] = torch.zeros(n_inr, D) + init_log_scale
    if geo.patch:
        r2 = n_inr // int(np.prod(geo.hierarchical_patch_nums["level2"]))
        r3 = n_inr // int(np.prod(geo.hierarchical_patch_nums["level3"]))
        p["h_loc"] = uni(r2)
        p["h_log_scale"] = torch.zeros(r2, D) + init_log_scale
        p["hh_loc"] = uni(r3)
        p["hh_log_scale"] = torch.zeros(r3, D) + init_log_scale
    lat = geo.latent_grid
    p["lpe_loc"] = torch.randn(n_inr, *lat, geo.latent_dim) * 0.1
    p["lpe_log_scale"] = torch.zeros(n_inr, *lat, geo.latent_dim) + init_log_scale
    return p


def prior_forward(geo: Geometry, p: dict, x: torch.Tensor, A, up, noise: Noise, return_parts=False):
    """PriorBNNmodel.forward (prior_model.py:129-179); x is [N,P,F] (Fourier part only)."""
    lpe = p["lpe_loc"] + st(p["lpe_log_scale"]) * noise(p["lpe_loc"].shape)
    pe = lpe_to_pe(up, lpe[None], geo)[:, 0]
    xin = torch.cat([x, pe], -1)
    hl = p.get("h_loc")
    h_w = sample_latent_weights(geo, p["loc"], st(p["log_scale"]),
                                hl, st(p["h_log_scale"]) if hl is not None else None,
                                p.get("hh_loc"), st(p["hh_log_scale"]) if hl is not None else None,
                                1, noise)[:, 0]
    y = siren_apply(geo, xin, h_w, A)
    return (y, pe, h_w) if return_parts else y


def prior_kl(geo: Geometry, p: dict, priors: Sequence[Optional[torch.Tensor]]) -> torch.Tensor:
    """PriorBNNmodel.calculate_kl (prior_model.py:181-200). priors = (loc, scale, lpe_loc,
    lpe_scale, h_loc, h_scale, hh_loc, hh_scale)."""
    kl = gauss_kl_elem(p["loc"], st(p["log_scale"]), priors[0], priors[1]).sum()
    kl = kl + gauss_kl_elem(p["lpe_loc"], st(p["lpe_log_scale"]), priors[2], priors[3]).sum()
    if geo.patch:
        kl = kl + gauss_kl_elem(p["h_loc"], st(p["h_log_scale"]), priors[4], priors[5]).sum()
        kl = kl + gauss_kl_elem(p["hh_loc"], st(p["hh_log_scale"]), priors[6], priors[7]).sum()
    return kl


def prior_train(geo: Geometry, p: dict, x, y, priors, A, up, n_epoch: int, lr: float, kl_beta: float,
                training_mappings: bool, noise: Noise):
    """PriorBNNmodel.train (prior_model.py:202-262): fresh Adam, loss = mean((yhat-y)^2)*N +
    beta*KL.  Mutates ``p`` (and A / up.weights when ``training_mappings``) in place.
    Returns (mse_last/N, KL/N, ELBO list)."""
    keys = [k for k in LEVEL_KEYS if k in p]
    leaves = [p[k].requires_grad_(True) for k in keys]
    extra = []
    if training_mappings:
        extra = [a.requires_grad_(True) for a in A] + [w.requires_grad_(True) for w in up.weights]
    opt = torch.optim.Adam(leaves + extra, lr)
    N = y.shape[0]
    elbo, mse_v = [], None
    for _ in range(n_epoch):
        yhat = prior_forward(geo, p, x, A if training_mappings else [a.detach() for a in A], up, noise)
        mse = torch.mean((yhat - y) ** 2) * N
        loss = mse + prior_kl(geo, p, priors) * kl_beta
        opt.zero_grad()
        loss.backward()
        opt.step()
        mse_v = mse.item()
        elbo.append(-loss.item())
    for t in leaves + extra:
        t.requires_grad_(False)
    with torch.no_grad():
        klf = prior_kl(geo, p, priors).item()
    return mse_v / N, klf / N, elbo


def refit_prior(loc: torch.Tensor, log_scale: torch.Tensor):
    """Moment matching over the INR axis (main_prior_training.py:157-172):
    mu_p = mean_0(mu_q); sigma_p = sqrt(mean_0(sigma_q^2) + var_0(mu_q)), unbiased var."""
    mu = loc.mean(0)
    sig = ((st(log_scale) ** 2).mean(0) + loc.var(0)) ** 0.5
    return mu, sig


def beta_rule(kl_beta: float, kl_bits_per_inr: float, budget_max: float, budget_min: float) -> float:
    """main_prior_training.py:144-154."""
    if kl_bits_per_inr > budget_max:
        kl_beta *= 1.5
    if kl_bits_per_inr < budget_min:
        kl_beta /= 1.5
    return min(max(kl_beta, 1e-20), 1.0)


# ----------------------------------------------------------------------------------------
# A10: grouping   (prior_model.py:264-316)
# ----------------------------------------------------------------------------------------
def group_by_bits(bits: np.ndarray, max_bits: float = 16):
    """Greedy sequential packing of parameters (in the fixed np.random.seed(0) shuffle order)
    into groups whose running fp32 KL sum stays <= 16 bits.  Returns the reference's 8-tuple
    (group_idx, start, end, group2param, param2group, n_groups, group_kls, weights)."""
    D = bits.shape[0]
    np.random.seed(0)
    order = np.random.choice(D, D, False)
    np.random.seed(None)
    w = bits[order]
    gid = np.empty(D, dtype=np.int64)
    run = w[0]                      # numpy scalar of the input dtype (fp32 in practice)
    g = 0
    gid[0] = 0
    for i in range(1, D):
        if run + w[i] > max_bits:
            g += 1
            run = w[i]
        else:
            run = run + w[i]
        gid[i] = g
    n_groups = g + 1
    param2group = order.copy()                   # parameter ids listed in group order
    group2param = np.argsort(param2group)
    first = np.flatnonzero(np.r_[True, gid[1:] != gid[:-1]])
    start = first
    end = np.r_[first[1:], D]
    group_kls = np.array([sum([bits[j] for j in param2group[s:e]]) for s, e in zip(start, end)])
    return gid.astype(int), start, end, group2param, param2group, n_groups, group_kls, bits


def grouping(q_loc, q_scale, p_loc, p_scale):
    """get_grouping (prior_model.py:264-271): mean-over-rows KL in bits, then group_by_bits."""
    bits = (gauss_kl_elem(q_loc, q_scale, p_loc, p_scale) / np.log(2.)).mean(0).cpu().detach().numpy()
    return group_by_bits(bits)


# ----------------------------------------------------------------------------------------
# A15 / A16: shared candidate tables for A* coding
# ----------------------------------------------------------------------------------------
def gumbel_table(seed: int = 42, K: int = 65536) -> np.ndarray:
    """Decreasing truncated-Gumbel sequence (test_model.py:441-457), fp64."""
    np.random.seed(seed)
    log_u = np.log(np.random.rand(K))
    out = np.empty(K, dtype=np.float64)
    b = -np.log(-log_u[0])
    out[0] = b
    for i in range(1, K):
        b = -np.log(-log_u[i] + np.exp(-b))
        out[i] = b
    return out


def sobol_normal_table(g: int, K: int = 65536, seed: int = 42) -> torch.Tensor:
    """Scrambled Sobol -> scipy norm.ppf (fp32 ufunc loop) -> clamp +-100, fp64 container
    (test_model.py:493-498)."""
    from scipy.stats import norm
    from torch.quasirandom import SobolEngine
    u = SobolEngine(g, scramble=True, seed=seed).draw(K)
    return torch.clamp(torch.from_numpy(norm.ppf(u)), -100, 100)


def rec_score(xi: torch.Tensor, mu_q, sig_q, mu_p, sig_p, gumbel: torch.Tensor):
    """A* scoring of one group (test_model.py:501-533).  xi [K,g] fp64 container; the four
    parameter vectors are fp32 [g].  Returns (index, z_i (fp64 [g]), log_w fp64 [K])."""
    z = mu_p + sig_p * xi
    log_p = Normal(mu_p, sig_p).log_prob(z).sum(-1)
    log_q = Normal(mu_q, sig_q).log_prob(z).sum(-1)
    log_w = log_q - log_p + gumbel[: xi.shape[0]]
    i = int(torch.argmax(log_w).item())
    return i, z[i], log_w


# ----------------------------------------------------------------------------------------
# A11 - A20: test-time model
# ----------------------------------------------------------------------------------------
@dataclass
class Level:
    """One level of the test-time posterior, stored in *group order* (test_model.py:130-180,
    210-237)."""
    loc: torch.Tensor
    log_scale: torch.Tensor
    p_loc: torch.Tensor
    p_log_scale: torch.Tensor
    group_idx: np.ndarray
    start: np.ndarray
    end: np.ndarray
    group_to_param: np.ndarray
    n_groups: int
    kl_beta: torch.Tensor
    row_perm_g2p: Optional[np.ndarray] = None      # [rows, D] per-column row permutation
    done: np.ndarray = None                        # [rows, G] bool
    idx: np.ndarray = None                         # [rows, G] float64 (as in the reference)
    mask: torch.Tensor = None                      # [rows, D]
    sample: torch.Tensor = None
    tables: dict = field(default_factory=dict)

    def __post_init__(self):
        rows = self.loc.shape[0]
        self.done = np.zeros([rows, self.n_groups], dtype=bool)
        self.idx = np.zeros([rows, self.n_groups])
        self.mask = torch.zeros_like(self.loc)
        self.sample = torch.zeros_like(self.loc)

    def effective(self):
        """(mu, sigma) with encoded entries frozen to their sample, sigma=1e-15
        (test_model.py:289-290), then un-permuted to parameter order (:294-298)."""
        m = self.mask
        mu = self.loc * (1 - m) + self.sample * m
        sig = st(self.log_scale) * (1 - m) + (1e-15 + torch.zeros_like(self.loc)) * m
        if self.row_perm_g2p is not None:
            cols = torch.arange(self.loc.shape[1])[None, :].repeat(self.loc.shape[0], 1)
            mu = mu[self.row_perm_g2p, cols]
            sig = sig[self.row_perm_g2p, cols]
        return mu[:, self.group_to_param], sig[:, self.group_to_param]

    def kl_elem(self):
        return gauss_kl_elem(self.loc, st(self.log_scale), self.p_loc[None, :], st(self.p_log_scale)[None, :])

    def weighted_kl(self):
        """calculate_kl (test_model.py:357-377)."""
        fac = self.kl_beta[:, self.group_idx]
        return (self.kl_elem() * fac).sum()

    def group_kls(self) -> np.ndarray:
        """[rows, G] fp64 segment sums via bincount (test_model.py:384-388)."""
        with torch.no_grad():
            kl = self.kl_elem().detach().cpu().numpy()
        return np.stack([np.bincount(self.group_idx, weights=kl[i]) for i in range(kl.shape[0])])

    def anneal(self, kls: np.ndarray, step: float, upper: float, lower: float, bits: float = 16):
        """beta update (test_model.py:404-413)."""
        nb = self.kl_beta.clone()
        m = (kls / np.log(2.) > (bits + upper)).astype(float)
        nb = nb * torch.from_numpy(1 + step * m).float()
        m = (kls / np.log(2.) <= (bits - lower)).astype(float)
        nb = nb / torch.from_numpy(1 + step * m).float()
        nb = torch.clamp(nb, 0., 10000.)
        self.kl_beta = torch.where(torch.from_numpy(~self.done), nb, self.kl_beta)


def column_row_perms(rows: int, cols: int) -> np.ndarray:
    """Per-column row permutation np.random.seed(col); choice(rows, rows, False)
    (test_model.py:182-208) -> [rows, cols] (g2p direction)."""
    out = np.empty([rows, cols], dtype=np.int64)
    for c in range(cols):
        np.random.seed(c)
        out[:, c] = np.random.choice(rows, rows, False)
        np.random.seed(None)
    return out


class TestTimeModel:
    """Functional restatement of TestBNNmodel (test_model.py:33-856)."""
    __test__ = False  # not a pytest class

    def __init__(self, geo: Geometry, n_inr: int, dataset: str, A, up, lvl_kwargs: dict,
                 h_kwargs: Optional[dict] = None, hh_kwargs: Optional[dict] = None,
                 initial_beta=1e-8, seed: int = 42, kl_upper_buffer=0., kl_lower_buffer=0.4,
                 kl_adjust_gap=10, beta_step_size=0.05):
        self.geo, self.n, self.dataset, self.A, self.up, self.seed = geo, n_inr, dataset, A, up, seed
        self.upper, self.lower, self.gap, self.step = kl_upper_buffer, kl_lower_buffer, kl_adjust_gap, beta_step_size
        self.bits = 16

        def mk(kw, rows, perm):
            D = kw["p_loc"].shape[0]
            return Level(loc=kw["p_loc"][None, :].repeat(rows, 1).clone(),
                         log_scale=torch.zeros(rows, D) + kw["init_log_scale"],
                         p_loc=kw["p_loc"].detach().clone(), p_log_scale=kw["p_log_scale"].detach().clone(),
                         group_idx=np.asarray(kw["group_idx"]), start=np.asarray(kw["group_start_index"]),
                         end=np.asarray(kw["group_end_index"]), group_to_param=np.asarray(kw["group_to_param"]),
                         n_groups=int(kw["n_groups"]),
                         kl_beta=torch.zeros(rows, int(kw["n_groups"])) + initial_beta,
                         row_perm_g2p=column_row_perms(rows, D) if perm else None)
        self.l1 = mk(lvl_kwargs, n_inr, geo.patch)
        self.levels = [self.l1]
        if geo.patch:
            r2 = n_inr // int(np.prod(geo.hierarchical_patch_nums["level2"]))
            r3 = n_inr // int(np.prod(geo.hierarchical_patch_nums["level3"]))
            self.l2 = mk(h_kwargs, r2, True)
            self.l3 = mk(hh_kwargs, r3, False)
            self.levels += [self.l2, self.l3]
        P = geo.n_pix
        bpp = self.l1.n_groups * self.bits / P
        if geo.patch:
            bpp += self.l2.n_groups * self.bits / P / np.prod(geo.hierarchical_patch_nums["level2"])
            bpp += self.l3.n_groups * self.bits / P / np.prod(geo.hierarchical_patch_nums["level3"])
        if dataset == "audio":
            bpp = bpp / (3 / 48000) / 1000
        self.bpp = bpp
        self.gumbel = None

    # -- A12 ------------------------------------------------------------------------------
    def predict(self, x, random_seed=None, S: int = 1, noise: Optional[Noise] = None):
        if random_seed is not None:
            torch.manual_seed(random_seed)
        noise = noise or Noise()
        geo = self.geo
        mu, sig = self.l1.effective()
        D = geo.d_net
        lpe_sc = sig[None, :, D:].repeat(S, 1, 1)
        lpe = mu[None, :, D:] + lpe_sc * noise(lpe_sc.shape)
        pe = lpe_to_pe(self.up, lpe, geo)
        xin = torch.cat([x[:, None].repeat(1, S, 1, 1), pe], -1)
        if geo.patch:
            hm, hs = self.l2.effective()
            hhm, hhs = self.l3.effective()
        else:
            hm = hs = hhm = hhs = None
        h_w = sample_latent_weights(geo, mu[:, :D], sig[:, :D], hm, hs, hhm, hhs, S, noise)
        y = siren_apply(geo, xin, h_w, self.A)
        return y[:, 0] if S == 1 else y

    # -- A13 / A14 -------------------------------------------------------------------------
    def weighted_kl(self):
        t = self.l1.weighted_kl()
        if self.geo.patch:
            t = t + self.l2.weighted_kl() + self.l3.weighted_kl()
        return t

    def update_annealing(self, update=True):
        out = []
        for lv in self.levels:
            out.append(lv.group_kls())
        if update:
            for lv, k in zip(self.levels, out):
                lv.anneal(k, self.step, self.upper, self.lower, self.bits)
        return out if self.geo.patch else out[0]

    # -- A19 ------------------------------------------------------------------------------
    def leaves(self):
        out = []
        for lv in self.levels:
            out += [lv.log_scale, lv.loc]
        return out

    def train(self, x, y, n_epochs: int, lr: float, S: int = 5, noises: Optional[List[Noise]] = None):
        """fresh Adam + TestBNNmodel.train (test_model.py:621-635): reseed with the epoch
        number, S samples, loss = mean(.)*N + sum(beta*KL); beta update every ``gap`` epochs
        (incl. epoch 0) after the loss is formed and before the step."""
        for t in self.leaves():
            t.requires_grad_(True)
        opt = torch.optim.Adam(self.leaves(), lr=lr)
        N = y.shape[0]
        for ep in range(n_epochs):
            yp = self.predict(x, random_seed=ep, S=S, noise=None if noises is None else noises[ep])
            tgt = y[:, None] if S != 1 else y
            loss = torch.mean((yp - tgt) ** 2) * N + self.weighted_kl()
            if ep % self.gap == 0:
                self.update_annealing(True)
            opt.zero_grad()
            loss.backward()
            opt.step()
        for t in self.leaves():
            t.requires_grad_(False)

    # -- A15-A18 ---------------------------------------------------------------------------
    def encode_group(self, lv: Level, row: int, grp: int, K: int = 65536):
        s, e = int(lv.start[grp]), int(lv.end[grp])
        g = e - s
        if g not in lv.tables:
            lv.tables[g] = sobol_normal_table(g, K, self.seed)
        if self.gumbel is None:
            self.gumbel = torch.from_numpy(gumbel_table(self.seed, K))
        with torch.no_grad():
            i, z, lw = rec_score(lv.tables[g], lv.loc[row, s:e], st(lv.log_scale[row, s:e]),
                                 lv.p_loc[s:e], st(lv.p_log_scale[s:e]), self.gumbel)
            lv.idx[row, grp] = i
            lv.done[row, grp] = True
            lv.sample[row, s:e] = z.to(lv.sample.dtype)
            lv.mask[row, s:e] = 1
            lv.kl_beta[row, grp] = 0
        return i, z, lw

    # -- A20 ------------------------------------------------------------------------------
    def compress(self, x, y, n_ft: int, h_n_ft: int, hh_n_ft: int, lr: float, metric_name: str):
        """compress_posteriors (test_model.py:687-856) with fine_tune_gap=1 and largest-KL-first;
        order level-3, level-2, level-1; one group per row per round, fine-tune after every round."""
        plan = []
        if self.geo.patch:
            plan = [(self.l3, hh_n_ft), (self.l2, h_n_ft)]
        plan.append((self.l1, n_ft))
        for lv, nft in plan:
            for _ in range(lv.n_groups):
                for row in range(lv.loc.shape[0]):
                    bits = lv.group_kls()[row] / np.log(2.)
                    bits[lv.done[row]] = -1e10
                    self.encode_group(lv, row, int(bits.argmax()))
                self.train(x, y, nft, lr)
        with torch.no_grad():
            yp = self.predict(x)
        return metric(y.numpy(), yp.numpy(), metric_name)


# ----------------------------------------------------------------------------------------
# A21: metrics (utils.py:200-260)
# ----------------------------------------------------------------------------------------
def _quant(c):
    return np.round(np.clip(c, 0, 1) * 255) / 255


def psnr(orig, comp, rnd: bool, max_value=1.0) -> float:
    if rnd:
        comp = _quant(comp)
    return float(20 * np.log10(max_value / np.sqrt(np.mean((orig - comp) ** 2))))


def batch_psnr(orig, comp, rnd: bool, max_value=1.0) -> np.ndarray:
    b = orig.shape[0]
    if rnd:
        comp = _quant(comp)
    mse = np.mean((orig.reshape(b, -1) - comp.reshape(b, -1)) ** 2, axis=-1)
    return 20 * np.log10(max_value / np.sqrt(mse))


def batch_rmsd(orig, comp, scale: float) -> np.ndarray:
    b = orig.shape[0]
    return (((orig * scale - comp * scale) ** 2).reshape(b, -1).mean(-1) * 3) ** 0.5


def metric(orig, comp, dataset: str):
    if dataset == "cifar":
        return batch_psnr(orig, comp, True)
    if dataset in ("kodak", "video"):
        return psnr(orig, comp, True)
    if dataset == "audio":
        return psnr(orig, comp, False)
    if dataset == "protein":
        return batch_rmsd(orig, comp, 25)
    raise ValueError(dataset)
