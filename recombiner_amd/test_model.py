"""Test-time (compression) model on MI355X: posterior optimisation with per-group beta annealing,
A* relative-entropy coding of one group per row per round, progressive fine-tuning -- the call
surface of the reference's test_model.py (TestBNNmodel, same constructor/method names, cited per
member) on hand-written HIP kernels.

Differences in *how* (results are the same):
  * per-group beta, the encoded-sample buffers and masks live on the GPU; the reference keeps beta
    on the CPU and re-uploads an [N, D] gather every step (test_model.py:359);
  * per-(row, group) KL sums are a segmented reduction on the device (parameters are stored in
    group order), not a D2H copy + Python bincount loop (test_model.py:384-388);
  * one encode round scores all rows in ONE batched fp64 launch (rcb_rec_score_argmax) instead of
    N Python calls each recomputing the full KL matrix (test_model.py:806-818): posteriors do not
    change inside the row loop, so the selections are identical.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.optim import Adam

from . import ops
from .ops import LevelSpec, SirenMeta
from .upsample_fast import (hip_path_supported, tiled_2d_preferred, phase_form_preferred, phase_module,
                            stitched2d_module, upsample_cifar_hip)
from .utils import count_net_params, hierarchy_row_maps, map_lpe_to_inr_inputs, metric
from .utils import map_hierarchical_model_to_int_weights  # noqa: F401  (module-level name, as upstream imports it)

LN2 = np.log(2.)


class Sine(nn.Module):
    """sin(w0 x)  (test_model.py:19-29); the HIP MLP kernel fuses it, kept for API parity."""

    def __init__(self, w0=1.):
        super().__init__()
        self.w0 = w0

    def forward(self, x):
        return torch.sin(self.w0 * x)


def _column_row_perms(rows, cols):
    """per-column row permutation: np.random.seed(col); choice(rows, rows, False) (test_model.py:182-208)."""
    out = np.empty([rows, cols], dtype=np.int64)
    for c in range(cols):
        np.random.seed(c)
        out[:, c] = np.random.choice(rows, rows, False)
        np.random.seed(None)
    return out


class _Lvl:
    """Book-keeping of one posterior level (parameters in group order)."""

    def __init__(self, owner, pre, loc, log_scale, p_loc, p_log_scale, group_idx, start, end, group_to_param,
                 param_to_group, n_groups, initial_beta, perm, dev):
        self.pre = pre
        self.loc, self.log_scale = loc, log_scale
        rows, D = loc.shape
        self.rows, self.D = rows, D
        self.p_loc = p_loc.detach().clone().to(dev, torch.float32).contiguous()
        self.p_log_scale = p_log_scale.detach().clone().to(dev, torch.float32).contiguous()
        self.group_idx = np.asarray(group_idx).astype(np.int64)
        self.start = np.asarray(start).astype(np.int64)
        self.end = np.asarray(end).astype(np.int64)
        self.group_to_param = np.asarray(group_to_param).astype(np.int64)
        self.param_to_group = np.asarray(param_to_group).astype(np.int64)
        self.n_groups = int(n_groups)
        self.d_group_idx = torch.from_numpy(self.group_idx.astype(np.int32)).to(dev)
        self.d_start = torch.from_numpy(self.start.astype(np.int32)).to(dev)
        self.d_end = torch.from_numpy(self.end.astype(np.int32)).to(dev)
        self.kl_beta = (torch.zeros(rows, self.n_groups) + initial_beta).to(dev, torch.float32).contiguous()
        self.perm_g2p = _column_row_perms(rows, D) if perm else None
        self.perm_p2g = np.argsort(self.perm_g2p, axis=0) if perm else None
        # encode progress lives on the device (an encode round never waits for the host); the reference's numpy views
        # (compressed_mask_groupwise: bool, compressed_idx_groupwise: float64, test_model.py:220-237) are materialised on access
        self.d_done = torch.zeros(rows, self.n_groups, dtype=torch.uint8, device=dev)
        self.d_idx = torch.zeros(rows, self.n_groups, dtype=torch.int32, device=dev)
        self.mask = torch.zeros(rows, D, device=dev)
        self.sample = torch.zeros(rows, D, device=dev)
        self.sample_std = 1e-15 + torch.zeros(rows, D, device=dev)
        self.tables = {}             # (g, K) -> fp64 [K, g] host tensor, as the reference computes it
        self.rec_tables = None       # ops.RecTables: the same tables on the device, fp32 transposed
        self.d_glen = self.d_end - self.d_start
        self.compressed_num = None

    @property
    def mask_groupwise(self):
        return self.d_done.cpu().numpy().astype(bool)

    @property
    def idx_groupwise(self):
        return self.d_idx.cpu().numpy().astype(np.float64)

    def spec(self, n_inr, cols_out, row_map):
        return LevelSpec(self.loc, self.log_scale, cols_out, n_inr, row_map=row_map, row_perm=self.perm_g2p,
                         col_map=self.group_to_param[:cols_out] if cols_out <= self.D else None,
                         enc_sample=self.sample, enc_mask=self.mask)


class TestBNNmodel(nn.Module):
    """RECOMBINER model for compression (test_model.py:33-856)."""
    __test__ = False

    def __init__(self, in_dim, hidden_dims, out_dim, number_of_datapoints, upsample_factors, latent_dim, data_dim,
                 pixel_sizes, patch, patch_nums, hierarchical_patch_nums, dataset,
                 linear_transform=None, upsample_net=None,
                 p_loc=None, p_log_scale=None, init_log_scale=-4., param_to_group=None, group_to_param=None,
                 n_groups=None, group_start_index=None, group_end_index=None, group_idx=None,
                 h_p_loc=None, h_p_log_scale=None, h_init_log_scale=-4., h_param_to_group=None,
                 h_group_to_param=None, h_n_groups=None, h_group_start_index=None, h_group_end_index=None,
                 h_group_idx=None,
                 hh_p_loc=None, hh_p_log_scale=None, hh_init_log_scale=-4., hh_param_to_group=None,
                 hh_group_to_param=None, hh_n_groups=None, hh_group_start_index=None, hh_group_end_index=None,
                 hh_group_idx=None,
                 w0=30., c=6., random_seed=42, device='cuda', kl_upper_buffer=0., kl_lower_buffer=0.4,
                 kl_adjust_gap=10, initial_beta=1e-8, beta_step_size=0.05):
        super().__init__()
        self.bit_per_group = 16
        self.n_layers = len(hidden_dims) + 1
        self.dims = [in_dim] + list(hidden_dims) + [out_dim]
        self.st = lambda x: F.softplus(x, beta=1, threshold=20) / 6
        self.upsample_factors, self.latent_dim, self.data_dim = upsample_factors, latent_dim, data_dim
        self.patch, self.patch_nums, self.pixel_sizes = patch, patch_nums, pixel_sizes
        self.linear_transform, self.upsample_net = linear_transform, upsample_net
        self.hierarchical_patch_nums = hierarchical_patch_nums
        self.device, self.dataset, self.random_seed, self.w0 = device, dataset, random_seed, float(w0)
        for mod in (self.linear_transform, self.upsample_net):          # frozen mappings (test_model.py:117-126)
            if mod is not None:
                for prm in mod.parameters():
                    prm.requires_grad = False
        _, self.cum_param_sizes = count_net_params(in_dim, hidden_dims, out_dim)
        self._d_net = int(self.cum_param_sizes[-1])
        N = number_of_datapoints
        self._n = N
        dev = device

        def mk_params(p_loc_, init_ls, rows):
            D = p_loc_.shape[0]
            loc = nn.Parameter(p_loc_[None, :].repeat([rows, 1]).to(dev).clone().float())
            ls = nn.Parameter(((torch.zeros([rows, D]) + init_ls).to(dev)).float())
            return loc, ls
        # level 1: latent INR weights + latent positional encodings, concatenated, group order
        self.param_to_group, self.group_to_param, self.n_groups = param_to_group, group_to_param, n_groups
        self.group_start_index, self.group_end_index, self.group_idx = group_start_index, group_end_index, group_idx
        self.loc, self.log_scale = mk_params(p_loc, init_log_scale, N)
        self._l1 = _Lvl(self, "", self.loc, self.log_scale, p_loc, p_log_scale, group_idx, group_start_index,
                        group_end_index, group_to_param, param_to_group, n_groups, initial_beta, patch, dev)
        self._levels = [self._l1]
        if patch:
            r2 = N // int(np.prod(hierarchical_patch_nums['level2']))
            r3 = N // int(np.prod(hierarchical_patch_nums['level3']))
            self.h_param_to_group, self.h_group_to_param, self.h_n_groups = h_param_to_group, h_group_to_param, h_n_groups
            self.h_group_start_index, self.h_group_end_index, self.h_group_idx = h_group_start_index, h_group_end_index, h_group_idx
            self.h_loc, self.h_log_scale = mk_params(h_p_loc, h_init_log_scale, r2)
            self.hh_param_to_group, self.hh_group_to_param, self.hh_n_groups = hh_param_to_group, hh_group_to_param, hh_n_groups
            self.hh_group_start_index, self.hh_group_end_index, self.hh_group_idx = hh_group_start_index, hh_group_end_index, hh_group_idx
            self.hh_loc, self.hh_log_scale = mk_params(hh_p_loc, hh_init_log_scale, r3)
            self._l2 = _Lvl(self, "h_", self.h_loc, self.h_log_scale, h_p_loc, h_p_log_scale, h_group_idx,
                            h_group_start_index, h_group_end_index, h_group_to_param, h_param_to_group, h_n_groups,
                            initial_beta, True, dev)
            self._l3 = _Lvl(self, "hh_", self.hh_loc, self.hh_log_scale, hh_p_loc, hh_p_log_scale, hh_group_idx,
                            hh_group_start_index, hh_group_end_index, hh_group_to_param, hh_param_to_group,
                            hh_n_groups, initial_beta, False, dev)
            self._levels += [self._l2, self._l3]
            self._maps = hierarchy_row_maps(N, patch_nums, hierarchical_patch_nums, data_dim)
        self.beta_step_size, self.kl_upper_buffer = beta_step_size, kl_upper_buffer
        self.kl_lower_buffer, self.kl_adjust_gap = kl_lower_buffer, kl_adjust_gap
        self.g_samples = None
        self.act = Sine(w0)
        P = np.prod(pixel_sizes)
        self.bpp = (self.n_groups * self.bit_per_group) / P
        if patch:
            self.bpp += (self.h_n_groups * self.bit_per_group) / P / np.prod(hierarchical_patch_nums['level2'])
            self.bpp += (self.hh_n_groups * self.bit_per_group) / P / np.prod(hierarchical_patch_nums['level3'])
        if self.dataset == 'audio':
            self.bpp = self.bpp / (3 / 48000) / 1000
        print("Model Initialized. Expected bpp is %.2f" % self.bpp, flush=True)
        self.noise_source = None     # optional callable(kind, shape) for eps injection in parity tests
        self.precision = 0
        self.stage1_bf16 = True      # 16-bit mode only: bf16-operand GEMMs for stage 1 of the upsampling net
        self.pe_bf16 = True          # 16-bit mode only: pe / dpe stored as bf16 (bit-identical, half the traffic)
        self.split_gemm = True       # 16-bit mode only: split-bf16 (hi/lo) operands for the A-transform fwd / dgrad GEMMs
        self.split_terms = 2         # as PriorBNNmodel.split_terms
        self.use_graph = True        # replay the fused training step as captured HIP graphs when possible
        self._specs = None
        self._ws = None

    # ---- reference-named views of the per-level state ------------------------------------------------------
    kl_beta = property(lambda s: s._l1.kl_beta)
    h_kl_beta = property(lambda s: s._l2.kl_beta)
    hh_kl_beta = property(lambda s: s._l3.kl_beta)
    compressed_idx_groupwise = property(lambda s: s._l1.idx_groupwise)
    h_compressed_idx_groupwise = property(lambda s: s._l2.idx_groupwise)
    hh_compressed_idx_groupwise = property(lambda s: s._l3.idx_groupwise)
    compressed_mask_groupwise = property(lambda s: s._l1.mask_groupwise)
    h_compressed_mask_groupwise = property(lambda s: s._l2.mask_groupwise)
    hh_compressed_mask_groupwise = property(lambda s: s._l3.mask_groupwise)
    compressed_mask = property(lambda s: s._l1.mask)
    compressed_sample = property(lambda s: s._l1.sample)
    p_loc = property(lambda s: s._l1.p_loc)
    p_log_scale = property(lambda s: s._l1.p_log_scale)
    permute_patch_x_g2p = property(lambda s: s._l1.perm_g2p)
    h_permute_patch_x_g2p = property(lambda s: s._l2.perm_g2p)

    # ---- sampling --------------------------------------------------------------------------------------------
    def _level_specs(self):
        if self._specs is None:
            N, D = self._n, self._d_net
            sp = [self._l1.spec(N, self._l1.D, None)]
            if self.patch:
                sp.append(self._l2.spec(N, D, self._maps[0]))
                sp.append(self._l3.spec(N, D, self._maps[1]))
            self._specs = sp
        return self._specs

    def _draw(self, kind, shape):
        if self.noise_source is not None:
            return self.noise_source(kind, tuple(shape)).to(self.loc.device, torch.float32).reshape(shape)
        return torch.randn(shape, device=self.loc.device, dtype=torch.float32)

    def _draw_all(self, S):
        """noise in the reference's draw order: lpe [S,N,Dlpe], level-1 [N,S,Dnet], level-2, level-3."""
        N, D, Dt = self._n, self._d_net, self._l1.D
        if self.noise_source is None:
            # production: the GPU generator's stream differs from the reference's CPU stream anyway, so the level-1 and
            # lpe noise is drawn as ONE tensor in the layout the kernels read (no concatenation copy)
            eps1 = self._draw("l1", (N, S, Dt))
        else:
            e_lpe = self._draw("lpe", (S, N, Dt - D))
            e1 = self._draw("l1", (N, S, D))
            eps1 = torch.cat([e1, e_lpe.permute(1, 0, 2)], -1).contiguous()
        eps = [eps1]
        if self.patch:
            eps.append(self._draw("l2", (N, S, D)).contiguous())
            eps.append(self._draw("l3", (N, S, D)).contiguous())
        return eps

    def _meta(self, x, S):
        return SirenMeta(samples=S, n_pix=x.shape[-2], fourier_dim=x.shape[-1], pe_dim=16,
                         n_hidden=self.n_layers - 1, hidden=max(self.dims[1:-1]), out_dim=self.dims[-1], w0=self.w0,
                         precision=self.precision,
                         hidden_dims=tuple(self.dims[1:-1]) if len(set(self.dims[1:-1])) > 1 else None)

    def _layer_slices(self):
        cum = self.cum_param_sizes
        return [(0 if i == 0 else int(cum[i - 1]), int(cum[i])) for i in range(self.n_layers)]

    # ---- reference API: helpers (test_model.py:260-281) ------------------------------------------------------------
    def group_to_layer(self, param, layer_idx):
        """the slice of a layer-vector row (.. x D_net) that belongs to layer `layer_idx`"""
        lo, hi = self._layer_slices()[layer_idx]
        return param[..., lo:hi]

    def layer_to_weight(self, in_dim, out_dim, layer_param):
        """layer vectors [N, L] or [N, S, L] -> (weights [.., in, out], bias [.., 1, out]): `[bias | W row-major]`"""
        if layer_param.ndim == 2:
            return layer_param[:, out_dim:].reshape(-1, in_dim, out_dim), layer_param[:, :out_dim][:, None, :]
        if layer_param.ndim == 3:
            return (layer_param[:, :, out_dim:].reshape(layer_param.shape[0], layer_param.shape[1], in_dim, out_dim),
                    layer_param[:, :, :out_dim][:, :, None, :])
        raise ValueError("layer_to_weight: [N, L] or [N, S, L] expected")

    def _pe_layout(self):
        """as PriorBNNmodel._pe_layout: the patched presets in the 16-bit modes keep pe / dpe on the stitched grids"""
        if self.precision != 0 and self.patch and getattr(self, "stitched_pe", True):
            return ops.PeLayout(self.patch_nums[:self.data_dim], self.pixel_sizes[:self.data_dim])
        return None

    def _pe(self, lpe, stitched=False):
        if self.precision != 0 and hip_path_supported(self.upsample_net, self.pixel_sizes, self.upsample_factors,
                                                      self.patch, self.data_dim):
            if not hasattr(self, "_weff_cache"):
                self._weff_cache = {}        # the mappings are frozen for this model's lifetime (test_model.py:117-126)
            return upsample_cifar_hip(self.upsample_net, lpe, self.stage1_bf16, self.pe_bf16, self._weff_cache)
        net = self.upsample_net
        if self.precision != 0 and tiled_2d_preferred(self.upsample_net, self.patch, self.data_dim):
            net = stitched2d_module(self.upsample_net)
        elif self.precision != 0 and phase_form_preferred(self.data_dim, self.patch):
            net = phase_module(self.upsample_net) or self.upsample_net
        return map_lpe_to_inr_inputs(net, lpe, self.latent_dim, self.pixel_sizes, self.upsample_factors,
                                     self.patch, self.patch_nums, self.data_dim, stitched=stitched)

    def _pe_from_sample(self, sample, S):
        N, D = self._n, self._d_net
        lat = [self.pixel_sizes[i] // self.upsample_factors[i] for i in range(self.data_dim)]
        lpe = sample[..., D:].permute(1, 0, 2).reshape(S, N, *lat, self.latent_dim)
        return lpe

    def _forward_parts(self, x, S, eps, differentiable):
        specs = self._level_specs()
        if differentiable:
            sample = ops.sample_levels(specs, eps, S)
        else:
            sample = ops.reparam_fwd(specs, eps, S)                          # [N,S,Dtot]
        return sample

    # ---- A12 -------------------------------------------------------------------------------------------------
    def predict(self, x, random_seed=None, sample_size=1):
        """x [N,P,F] -> [N,P,C] (S=1) or [N,S,P,C]  (test_model.py:283-355), autograd-capable."""
        if random_seed is not None:
            torch.manual_seed(random_seed)
        S, N, D = sample_size, self._n, self._d_net
        x = x.to(self.loc.device)
        eps = self._draw_all(S)
        sample = self._forward_parts(x, S, eps, torch.is_grad_enabled())
        lpe = self._pe_from_sample(sample, S)
        pe = self._pe(lpe)                                                              # [N,S,P,16]
        h_w = sample[..., :D].reshape(N * S, D)
        parts = [h_w[:, lo:hi] @ self.linear_transform.A[i] for i, (lo, hi) in enumerate(self._layer_slices())]
        wvec = torch.cat(parts, -1)
        P = pe.shape[2]
        y = ops.SirenFn.apply(x, pe.reshape(N * S, P, pe.shape[-1]).contiguous(), wvec, self._meta(x, S))
        y = y.reshape(N, S, P, -1)
        return y[:, 0] if S == 1 else y

    # ---- A13 -------------------------------------------------------------------------------------------------
    def calculate_kl(self):
        """sum_j beta[row, group(j)] * KL_j over all levels (test_model.py:357-377) -> 0-d tensor."""
        tot = None
        for lv in self._levels:
            v = ops.GaussKLFn.apply(lv.loc, lv.log_scale, lv.p_loc, lv.p_log_scale, True, lv.kl_beta,
                                    lv.d_group_idx, lv.d_start, lv.d_end)
            tot = v if tot is None else tot + v
        return tot

    # ---- A14 -------------------------------------------------------------------------------------------------
    def _group_kls(self, lv):
        _, grp = ops.gauss_kl(lv.loc, lv.log_scale, lv.p_loc, lv.p_log_scale, True, None, None, lv.d_start, lv.d_end,
                              want_rows=False, want_groups=True)
        return grp

    def update_annealing_factors(self, update=True):
        """per-(row, group) KL (fp64) and, if `update`, the beta step for not-yet-encoded groups
        (test_model.py:379-439).  Returns numpy arrays like the reference."""
        out = []
        for lv in self._levels:
            grp = self._group_kls(lv)
            if update:
                ops.beta_update(grp, lv.kl_beta, lv.d_done, float(self.bit_per_group), float(self.kl_upper_buffer),
                                float(self.kl_lower_buffer), float(self.beta_step_size))
            out.append(grp.cpu().numpy())
        return tuple(out) if self.patch else out[0]

    # ---- A15 / A16 ----------------------------------------------------------------------------------------------
    def get_gumbel_sample(self):
        """decreasing truncated-Gumbel sequence shared by all groups (test_model.py:441-457), fp64."""
        K = int(np.ceil(2 ** self.bit_per_group))
        np.random.seed(self.random_seed)
        log_u = np.log(np.random.rand(K))
        out = np.empty(K)
        b = -np.log(-log_u[0])
        out[0] = b
        for i in range(1, K):
            b = -np.log(-log_u[i] + np.exp(-b))
            out[i] = b
        self.g_samples = torch.from_numpy(out).to(self.loc.device)
        self._g_absmax = float(np.abs(out).max())       # (the fast scorer's error bound uses it: never stale)

    def get_sobol_normal_sample(self, param_size, sample_size):
        """scrambled Sobol -> scipy norm.ppf -> clamp +-100 (test_model.py:493-498), fp64 container."""
        from scipy.stats import norm
        from torch.quasirandom import SobolEngine
        u = SobolEngine(param_size, scramble=True, seed=self.random_seed).draw(sample_size)
        return torch.clamp(torch.from_numpy(norm.ppf(u)), -100, 100)

    def _table(self, lv, g, K):
        """fp64 [K, g] table of the reference (host tensor), cached per (g, K); also registered with the level's device tables"""
        key = (int(g), int(K))
        if key not in lv.tables:
            lv.tables[key] = self.get_sobol_normal_sample(int(g), int(K)).contiguous()
        if lv.rec_tables is None or lv.rec_tables.K != int(K):
            lv.rec_tables = ops.RecTables(self.loc.device, int(K))
        if int(g) not in lv.rec_tables:
            lv.rec_tables.add(int(g), lv.tables[key])
        return lv.tables[key]

    def _rec_tables(self, lv, lens, K):
        for g in np.unique(np.asarray(lens)):
            self._table(lv, int(g), K)
        return lv.rec_tables

    def _gumbel(self, K):
        if self.g_samples is None:
            self.get_gumbel_sample()
        if getattr(self, "_g_absmax", None) is None:
            self._g_absmax = float(self.g_samples.abs().max())
        return self.g_samples[:K], self._g_absmax

    def get_sample(self, group_idx, group_sample_size):
        lv = self._l1
        return self._table(lv, lv.end[group_idx] - lv.start[group_idx], group_sample_size).to(self.loc.device)

    def h_get_sample(self, group_idx, group_sample_size):
        lv = self._l2
        return self._table(lv, lv.end[group_idx] - lv.start[group_idx], group_sample_size).to(self.loc.device)

    def hh_get_sample(self, group_idx, group_sample_size):
        lv = self._l3
        return self._table(lv, lv.end[group_idx] - lv.start[group_idx], group_sample_size).to(self.loc.device)

    # ---- A17 / A18 ----------------------------------------------------------------------------------------------
    def _encode_jobs(self, lv, rows, groups, K, want_z=False):
        """score + commit a batch of (row, group) encodes of one level: rows / groups are host arrays or device int tensors.
        Everything runs on the device and nothing is read back (-> idx int32 device tensor in job order after sorting by
        group length, z or None)."""
        dev = lv.loc.device
        tables = self._rec_tables(lv, lv.end - lv.start, K)      # every length of the level: all of them get encoded
        gum, gmax = self._gumbel(K)
        if not torch.is_tensor(groups):
            groups = torch.as_tensor(np.asarray(groups, dtype=np.int64), device=dev)
        if not torch.is_tensor(rows):
            rows = torch.as_tensor(np.asarray(rows, dtype=np.int64), device=dev)
        groups = groups.to(torch.int64)
        glen = lv.d_glen[groups]
        order = torch.sort(glen, stable=True)[1]          # eight jobs of one group length share the fast scorer's table loads
        groups, glen = groups[order], glen[order]
        jobs = ops.RecJobs(rows.to(torch.int32)[order].contiguous(), lv.d_start[groups].contiguous(), glen.contiguous(),
                           groups.to(torch.int32).contiguous())
        scale = ops.softplus_scale(lv.log_scale)
        p_scale = ops.softplus_scale(lv.p_log_scale)
        idx, _, _, _ = ops.rec_score(lv.loc, scale, lv.p_loc, p_scale, tables, gum, jobs, ops.REC_FAST, gumbel_absmax=gmax)
        # commit: index, masks, encoded sample (fp32), beta = 0  (test_model.py:586-595)
        z = ops.rec_commit(lv.p_loc, p_scale, tables, jobs, idx, n_groups=lv.n_groups, want_z=want_z, enc_sample=lv.sample,
                           enc_mask=lv.mask, done=lv.d_done, beta=lv.kl_beta, idx_groupwise=lv.d_idx)
        return idx, z

    def _sample_group(self, lv, row_idx, group_idx, group_sample_size):
        """A* scoring of one group without committing it -> (i, z_i fp64, log_w fp64 [K]); the reference arithmetic op for op."""
        s, e = int(lv.start[group_idx]), int(lv.end[group_idx])
        K = int(group_sample_size)
        tables = self._rec_tables(lv, [e - s], K)
        gum, _ = self._gumbel(K)
        idx, z, best, logw = ops.rec_score_argmax(lv.loc, ops.softplus_scale(lv.log_scale), lv.p_loc,
                                                  ops.softplus_scale(lv.p_log_scale), tables, gum, [row_idx], [s], [e - s],
                                                  want_logw0=True, mode=ops.REC_EXACT)
        return int(idx.item()), z[0, :e - s], logw

    def sample_group(self, row_idx, group_idx, group_sample_size):
        return self._sample_group(self._l1, row_idx, group_idx, group_sample_size)

    def h_sample_group(self, row_idx, group_idx, group_sample_size):
        return self._sample_group(self._l2, row_idx, group_idx, group_sample_size)

    def hh_sample_group(self, row_idx, group_idx, group_sample_size):
        return self._sample_group(self._l3, row_idx, group_idx, group_sample_size)

    def _compress_group(self, lv, row_idx, group_idx):
        K = int(np.ceil(2 ** self.bit_per_group))
        idx, z = self._encode_jobs(lv, [row_idx], [group_idx], K, want_z=True)
        s, e = int(lv.start[group_idx]), int(lv.end[group_idx])
        return int(idx[0].item()), z[0, :e - s]

    def compress_group(self, row_idx, group_idx):
        return self._compress_group(self._l1, row_idx, group_idx)

    def h_compress_group(self, row_idx, group_idx):
        return self._compress_group(self._l2, row_idx, group_idx)

    def hh_compress_group(self, row_idx, group_idx):
        return self._compress_group(self._l3, row_idx, group_idx)

    # ---- A19 -------------------------------------------------------------------------------------------------
    def _fresh(self, optimizer):
        return isinstance(optimizer, Adam) and len(optimizer.state) == 0 and len(optimizer.param_groups) == 1 and \
            optimizer.param_groups[0].get("betas", (0.9, 0.999)) == (0.9, 0.999) and \
            not optimizer.param_groups[0].get("amsgrad", False) and optimizer.param_groups[0].get("weight_decay", 0) == 0

    def train(self, x, y, n_epochs, optimizer, verbose, sample_size=5):
        """n_epochs of: seeded S-sample prediction, loss = mean(.)*N + sum beta*KL, beta update every
        kl_adjust_gap epochs (after the loss, before the step), Adam step  (test_model.py:621-635).
        A freshly constructed torch Adam (what every caller in the reference passes) selects the fused
        pipeline with the optimizer's lr/eps; anything else runs through autograd + optimizer.step()."""
        rng = range(n_epochs)
        if verbose:
            from tqdm import tqdm
            rng = tqdm(rng)
        if not self._fresh(optimizer):
            for epoch in rng:
                y_pred = self.predict(x=x, random_seed=epoch, sample_size=sample_size)
                tgt = y[:, None, :, :] if sample_size != 1 else y
                elbo = torch.mean((y_pred - tgt) ** 2) * y.shape[0] + self.calculate_kl()
                if epoch % self.kl_adjust_gap == 0:
                    self.update_annealing_factors(update=True)
                optimizer.zero_grad()
                elbo.backward()
                optimizer.step()
            return
        lr = optimizer.param_groups[0]["lr"]
        eps_adam = optimizer.param_groups[0].get("eps", 1e-8)
        dev = self.loc.device
        x = x.to(dev)
        y = y.to(dev).contiguous()
        S, N, D = sample_size, self._n, self._d_net
        P, Cc = y.shape[1], y.shape[2]
        specs = self._level_specs()
        A = [a.detach() for a in self.linear_transform.A]
        slices = self._layer_slices()
        meta = self._meta(x, S)
        # workspace that survives across calls (Adam state is re-zeroed = a fresh optimiser, the step counter
        # restarts): its stable addresses let the two captured step graphs be reused by every fine-tune call
        key = (x.data_ptr(), y.data_ptr(), tuple(x.shape), S, float(lr), float(eps_adam), self.precision,
               self.loc.data_ptr(), self.log_scale.data_ptr(), bool(self.split_gemm), self.split_terms,
               tuple(q.data_ptr() for q in A), tuple(q.data_ptr() for q in self.upsample_net.parameters()))
        ws = self._ws
        if ws is None or ws["key"] != key or ws["tab"].shape[0] < n_epochs:
            # the Adam moments of all levels as views of one buffer: a fine-tune call (a few steps each, thousands of them
            # per compression run) resets them with one launch
            shapes = [lv.loc.shape for lv in self._levels for _ in range(4)]
            offs, tot = [], 0
            for shp in shapes:
                offs.append(tot)
                tot += (int(np.prod(shp)) + 63) // 64 * 64
            state_flat = torch.zeros(tot, device=dev, dtype=torch.float32)
            views = iter([state_flat[o:o + int(np.prod(shp))].view(shp) for o, shp in zip(offs, shapes)])
            ws = dict(key=key, tab=ops.adam_table(lr, max(n_epochs, 2048)).to(dev), state_flat=state_flat,
                      dyn=torch.zeros(2, device=dev), step_t=torch.zeros(1, device=dev, dtype=torch.long),
                      states=[{k: next(views) for k in ("m_loc", "v_loc", "m_ls", "v_ls")} for lv in self._levels],
                      graphs={}, warm=0,
                      # bf16 copy of the coordinate grid: owned by the workspace that owns the graphs reading its address
                      xf16=ops.xf_bf16(x, self.precision) if (self.precision in (1, 2) and dev.type == "cuda") else None)
            self._ws = ws
        else:
            ws["state_flat"].zero_()
        ws["step_t"].zero_()
        tab, dyn, step_t, states = ws["tab"], ws["dyn"], ws["step_t"], ws["states"]
        cfg = ops.adam_cfg(lr, 1, eps=eps_adam, dyn=dyn)
        split = None
        if self.split_gemm and self.precision != 0:
            split = ws.get("split")
            if split is None:                       # the mappings are fixed at test time: packed once per workspace
                split = ops.ATransform(slices, dev, self.split_terms)
                split.prepare(A)
                ws["split"] = split

        pe_lay = self._pe_layout()

        def body(adjust):
            ops.step_begin(tab, step_t, dyn)
            eps = self._draw_all(S)
            sample = ops.reparam_fwd(specs, eps, S)                                   # [N,S,Dtot]
            lpe_t = self._pe_from_sample(sample, S).contiguous().requires_grad_(True)
            with torch.enable_grad():
                if pe_lay is not None:
                    pe_c = self._pe(lpe_t, stitched=True)
                else:
                    pe = self._pe(lpe_t)
                    pe_c = pe.reshape(N * S, P, pe.shape[-1]).contiguous()
            h_w = sample[..., :D].reshape(N * S, D)
            if split is not None:
                wvec = split.forward(h_w, split.new_rows(N * S))
            else:
                wvec = torch.empty(N * S, D, device=dev, dtype=torch.float32)
                for (lo, hi), a in zip(slices, A):
                    torch.mm(h_w[:, lo:hi], a, out=wvec[:, lo:hi])
            sse, dw, dpe = ops.siren_loss_bwd(x, pe_c.detach(), wvec, y, 1.0 / (S * P * Cc), meta, pe_layout=pe_lay,
                                              xf16=ws["xf16"])
            (d_lpe,) = torch.autograd.grad(pe_c, [lpe_t], dpe)                        # [S,N,*lat,C]
            Dt = self._l1.D
            if self.patch:        # levels 2 and 3 need the contiguous [N, S, D] gradient as well
                dh = torch.empty(N * S, D, device=dev, dtype=torch.float32)
                d_full = None
            else:                 # the GEMMs write straight into the level-1 gradient buffer [N, S, D + Dlpe]
                d_full = torch.empty(N, S, Dt, device=dev, dtype=torch.float32)
                dh = d_full.view(N * S, Dt)[:, :D]
            if split is not None:
                split.dgrad(dw, dh)
            else:
                for (lo, hi), a in zip(slices, A):
                    torch.mm(dw[:, lo:hi], a.t(), out=dh[:, lo:hi])
            if d_full is None:
                dh3 = dh.view(N, S, D)
                d_full = torch.cat([dh3, d_lpe.reshape(S, N, -1).permute(1, 0, 2)], -1).contiguous()
            else:
                dh3 = None
                d_full[:, :, D:].copy_(d_lpe.reshape(S, N, -1).permute(1, 0, 2))
            grp = [self._group_kls(lv) for lv in self._levels] if adjust else None
            for li, (lv, sp, e, stt) in enumerate(zip(self._levels, specs, eps, states)):
                ops.posterior_bwd(sp, lv.p_loc, lv.p_log_scale, True, 1.0, d_full if li == 0 else dh3, e, S,
                                  beta=lv.kl_beta, group_idx=lv.d_group_idx, n_groups=lv.n_groups, adam=cfg, state=stt)
            if adjust:   # the old beta shaped this step's gradient; the new one acts from the next epoch on
                for lv, gk in zip(self._levels, grp):
                    ops.beta_update(gk, lv.kl_beta, lv.d_done, float(self.bit_per_group), float(self.kl_upper_buffer),
                                    float(self.kl_lower_buffer), float(self.beta_step_size))
            ops.step_end(step_t)

        use_graph = self.use_graph and self.noise_source is None and not verbose and n_epochs >= 8 and dev.type == "cuda"
        if use_graph:
            # one noise stream per call (the reference reseeds with the epoch index before every step,
            # test_model.py:285,623 -- an un-capturable host action; either way every call sees the same stream)
            torch.manual_seed(0)
        for epoch in rng:
            adjust = (epoch % self.kl_adjust_gap == 0)
            if not use_graph:
                torch.manual_seed(epoch)
                body(adjust)
                continue
            g = ws["graphs"].get(adjust)
            if g is None:
                if ws["warm"] < 3:                     # eager warm-up steps (they are real steps)
                    body(adjust)
                    ws["warm"] += 1
                    continue
                try:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):          # records; executes nothing
                        body(adjust)
                    ws["graphs"][adjust] = g
                except Exception as exc:
                    import warnings
                    warnings.warn(f"HIP graph capture of the test-time step failed ({exc}); running eagerly")
                    use_graph = False
                    self.use_graph = False
                    torch.cuda.synchronize()
                    body(adjust)
                    continue
            g.replay()

    # ---- A20 -------------------------------------------------------------------------------------------------
    def _report(self, x, y):
        with torch.no_grad():
            y_pred = self.predict(x.to(self.loc.device)).cpu()
        return metric(y.cpu().numpy(), y_pred.numpy(), self.dataset)

    def _bits_summary(self):
        r = self.update_annealing_factors(False)
        arrs = r if self.patch else (r,)
        bits = np.concatenate([a.reshape(-1) for a in arrs]) / LN2
        print("Bits per group: ave %.2f" % bits.mean() + " max %.2f" % bits.max(), flush=True)

    def optimize_posteriors(self, x, y, n_epochs, lr, verbose):
        """test_model.py:637-685."""
        if verbose:
            print("Initialization: Average Distortion %.4f" % np.mean(self._report(x, y)), flush=True)
            self._bits_summary()
            print(' ')
            print("Start to optimize posteriors...", flush=True)
        optimizer = Adam(self.parameters(), lr=lr)
        self.train(x=x, y=y, n_epochs=n_epochs, optimizer=optimizer, verbose=verbose)
        if verbose:
            print("Optimization Finished. Average Distortion %.4f" % np.mean(self._report(x, y)), flush=True)
            self._bits_summary()

    def _encode_round(self, lv, largest_kl_first, round_idx):
        """one group for every row of the level, scored in one batched launch."""
        rows = torch.arange(lv.rows, device=lv.loc.device)
        if largest_kl_first:
            bits = self._group_kls(lv) / LN2
            bits = torch.where(lv.d_done.bool(), torch.full_like(bits, -1e10), bits)
            groups = torch.argmax(bits, dim=1)                  # stays on the device: the round needs no host round trip
        else:
            groups = torch.full((lv.rows,), int(round_idx), device=lv.loc.device, dtype=torch.int64)
        K = int(np.ceil(2 ** self.bit_per_group))
        self._encode_jobs(lv, rows, groups, K)

    def compress_posteriors(self, x, y, n_epochs_finetune, h_n_epochs_finetune, hh_n_epochs_finetune, verbose, lr,
                            fine_tune_gap, compress_from_group_with_largest_kl=True):
        """A* coding of every group, level 3 -> 2 -> 1, fine-tuning in between (test_model.py:687-856)."""
        if verbose:
            print("Start to compress posteriors by A* coding...", flush=True)
        plan = []
        if self.patch:
            plan += [(self._l3, hh_n_epochs_finetune, True), (self._l2, h_n_epochs_finetune, True)]
        plan.append((self._l1, n_epochs_finetune, False))
        for lv, n_ft, per_row_counter in plan:
            if lv.compressed_num is None:
                lv.compressed_num = 0
            first = lv.compressed_num if not per_row_counter else lv.compressed_num // max(lv.rows, 1)
            steps = set(np.round(np.linspace(0, lv.n_groups, 10)).astype(int).tolist())
            for _i in range(first, lv.n_groups):
                self._encode_round(lv, compress_from_group_with_largest_kl, _i)
                # the reference counts level-2/3 encodes per row and level-1 per round (:719,768,819)
                lv.compressed_num += lv.rows if per_row_counter else 1
                if lv.compressed_num % fine_tune_gap == 0:
                    optimizer = Adam(self.parameters(), lr=lr)
                    self.train(x, y, n_epochs=n_ft, optimizer=optimizer, verbose=False)
                if verbose and _i in steps:
                    kb = (self._group_kls(lv) / LN2).cpu().numpy()
                    left = kb[~lv.mask_groupwise]
                    if left.size:
                        print("Compress progress: %d; " % (100 * (_i + 1) / lv.n_groups),
                              "Average Distortion %.4f; " % np.mean(self._report(x, y)),
                              "KL in uncompressed groups: MAX %.3f" % left.max(), "AVE %.3f. " % left.mean(), flush=True)
            if verbose and lv is not self._l1:
                print(' ')
            # the kernels reject a malformed job by returning index -1 and rcb_rec_commit skips it: an encode round must
            # not be able to leave a group behind silently (checked once per level: one device reduction, one host read)
            if not bool(lv.d_done.all().item()):
                left = torch.nonzero(~lv.d_done.bool().reshape(lv.rows, -1))
                raise ops.RcbError("compress_posteriors: %d (row, group) pairs of a level were not encoded, first at row %d "
                                   "group %d" % (left.shape[0], int(left[0, 0]), int(left[0, 1])))
        distortion = self._report(x, y)
        if verbose:
            print("Optimization Finished. Average Distortion %.4f" % np.mean(distortion), flush=True)
        return distortion
