"""Prior-learning model on MI355X: per-INR variational posteriors, batched SIREN forward/backward,
ELBO/KL, Adam -- the call surface of the reference's prior_model.py (same class / function names,
argument order and return values, cited per member) on top of hand-written HIP kernels.

Where the reference chains torch ops through autograd, `train()` here runs one fused pipeline per
step: sample (K1, in-kernel noise) -> upsampling net (hand-written phase-conv kernels in the 16-bit modes,
MIOpen only in the fp32 parity mode) -> A transform (atrans.hip) -> fused SIREN fwd+MSE+bwd (K3/K4) -> A-transform
backward -> upsampling-net backward -> fused reparam-bwd + KL-bwd + Adam (K1'/K5'/K11), replayed as one HIP graph.
`forward()` / `calculate_kl()` stay autograd-capable through custom Functions so user code written
against the reference API keeps working.  There is no CPU fallback.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import dist as rdist
from . import ops
from .ops import LevelSpec, SirenMeta
from .upsample_fast import (hip_path_supported, tiled_2d_preferred, phase_form_preferred, phase_module,
                            stitched2d_module, upsample_cifar_hip)
from .utils import count_net_params, hierarchy_row_maps, map_lpe_to_inr_inputs
from .utils import map_hierarchical_model_to_int_weights  # noqa: F401  (module-level name, as upstream imports it)


class LinearTransform(nn.Module):
    """Shared per-layer square maps A[l] (L_l x L_l), init U(-1,1)/L_l  (prior_model.py:16-21)."""

    def __init__(self, net_dims):
        super().__init__()
        mats = []
        for i in range(1, len(net_dims)):
            L = net_dims[i] * (net_dims[i - 1] + 1)
            mats.append((torch.rand(L, L) * 2 - 1) / L)
        self.A = nn.ParameterList(mats)


class Upsample(nn.Module):
    """Latent grid -> per-pixel encodings: 3 x (nearest upsample, conv), 128 -> 64 -> 64 -> 16 channels,
    kernels 5/3/3, LeakyReLU(0.01) after the first two convs  (prior_model.py:23-59).  Submodule names
    match the reference so state dicts / pickles interchange."""

    def __init__(self, kernel_dim, paddings, layerwise_scale_factors):
        super().__init__()
        conv = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[kernel_dim]
        layerwise_scale_factors = [tuple(s) if isinstance(s, (list, tuple)) else s for s in layerwise_scale_factors]
        self.up1 = nn.Upsample(scale_factor=layerwise_scale_factors[0])
        self.conv1 = conv(128, 64, 5, padding=paddings[0])
        self.conv2 = conv(64, 64, 3, padding=paddings[1])
        self.conv3 = conv(64, 16, 3, padding=paddings[2])
        self.act1 = nn.LeakyReLU()
        self.up2 = nn.Upsample(scale_factor=layerwise_scale_factors[1])
        self.act2 = nn.LeakyReLU()
        self.up3 = nn.Upsample(scale_factor=layerwise_scale_factors[2])

    def forward(self, x):
        x = self.act1(self.conv1(self.up1(x)))
        x = self.act2(self.conv2(self.up2(x)))
        return self.conv3(self.up3(x))

    def __getstate__(self):
        # the fast evaluation paths cache their wrappers on the module (`_rcb_*`, upsample_fast.py); they share this module's
        # parameters and must neither travel in a checkpoint pickle (the reference could not load it) nor be carried over
        # by copy.deepcopy (the copy would keep evaluating the ORIGINAL's parameters)
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_rcb_")}


class PriorBNNmodel(nn.Module):
    """Training-set posteriors + batched INR evaluation (prior_model.py:62-262)."""

    def __init__(self, in_dim, hidden_dims, out_dim, train_size, data_dim, pixel_sizes, upsample_factors,
                 latent_dim, patch, patch_nums, hierarchical_patch_nums, random_seed=42, device="cuda",
                 init_log_scale=-4, c=6., w0=30.):
        super().__init__()
        self.random_seed, self.device = random_seed, device
        self.n_layers = len(hidden_dims) + 1
        self.dims = [in_dim] + list(hidden_dims) + [out_dim]
        self.patch, self.w0 = patch, float(w0)
        self.st = lambda x: F.softplus(x, beta=1, threshold=20) / 6
        self.data_dim, self.train_size, self.latent_dim = data_dim, train_size, latent_dim
        self.pixel_sizes, self.upsample_factors = pixel_sizes, upsample_factors
        self.patch_nums, self.hierarchical_patch_nums = patch_nums, hierarchical_patch_nums
        self.net_params_list, self.cum_param_sizes = count_net_params(in_dim, hidden_dims, out_dim)
        D = int(self.cum_param_sizes[-1])
        # (any list of hidden widths, as in the reference, prior_model.py:84-85: widths that differ run in the fp32 parity
        # mode on the plain-FMA kernel; the 16-bit modes take one width -- 32, 48 or 64 -- and say so when asked otherwise)

        # A1: same CPU-generator draw order as the reference (loc, h_loc, hh_loc, lpe_loc), then moved
        torch.manual_seed(random_seed)
        w_std = np.sqrt(c / hidden_dims[-1]) / w0

        def uni(rows):
            return nn.Parameter((torch.rand(rows, D) * w_std * 2 - w_std).to(device))

        def const(rows):
            return nn.Parameter((torch.zeros(rows, D) + init_log_scale).to(device))
        self.loc, self.log_scale = uni(train_size), const(train_size)
        if patch:
            r2 = train_size // int(np.prod(hierarchical_patch_nums["level2"]))
            r3 = train_size // int(np.prod(hierarchical_patch_nums["level3"]))
            self.h_loc, self.h_log_scale = uni(r2), const(r2)
            self.hh_loc, self.hh_log_scale = uni(r3), const(r3)
        lat = [pixel_sizes[i] // upsample_factors[i] for i in range(data_dim)]
        self.lpe_loc = nn.Parameter((torch.randn(train_size, *lat, latent_dim) * 0.1).to(device))
        self.lpe_log_scale = nn.Parameter((torch.zeros(train_size, *lat, latent_dim) + init_log_scale).to(device))
        self._lat = lat
        self._d_net, self._d_lpe = D, int(np.prod(lat)) * latent_dim
        self._maps = hierarchy_row_maps(train_size, patch_nums, hierarchical_patch_nums, data_dim) if patch else None
        self._levels_cache = None
        self.noise_source = None     # optional callable(shape) -> standard normal GPU tensor (eps injection)
        self.precision = 0           # 0 = fp32 MFMA, 1 = bf16 operands (throughput mode)
        self.dp_group = None         # torch.distributed group for sharded training of the shared mappings
        self.lowp_gemm = False       # 16-bit mode only: bf16-operand hipBLASLt GEMMs for the A transform
        self.stage1_bf16 = True      # 16-bit mode only: bf16-operand GEMMs for stage 1 of the upsampling net
        self.pe_bf16 = True          # 16-bit mode only: pe / dpe stored as bf16 (bit-identical, half the traffic)
        self.stitched_pe = True      # 16-bit mode, patched presets: the SIREN kernel reads pe / writes dpe inside the stitched grids
        self.split_gemm = True       # 16-bit mode only: split-bf16 (hi/lo) operands for the A-transform fwd / dgrad GEMMs
        self.split_terms = 2         # with split_gemm: 3 = both operands split; 2 = the mappings enter as bf16 (ops.ATransform)
        self.split_dgrad_terms = None  # with split_gemm: terms of the data-gradient GEMM (None = split_terms)
        self.wgrad_bf16 = True       # with split_gemm: bf16 high parts for the A weight-gradient GEMMs (sum over INRs)
        self.fused_noise = True      # draw eps inside the reparam kernel (Philox) when no noise_source is injected
        # with fused_noise: the posterior update of step t also draws the sample of step t + 1 (rcb_level_bwd.next_*), so
        # the sampling kernels -- a second read of every loc / log_scale -- run once per train() call instead of once per step
        self.fuse_next_sample = os.environ.get("RCB_FUSE_NEXT", "1") != "0"      # (the switch is for same-box A/B runs)
        # with split_gemm + fused_noise: the A transform's per-INR operands (h_w, and the SIREN gradient dwvec) may travel as the
        # (hi, lo) bf16 planes their producers write (ops.Planes) instead of fp32 rows + a bf16 copy: identical bits, no
        # conversion work in the A-transform kernels, 4 instead of 6 bytes per element written.  OFF by default: measured on
        # MI355X (round 4, same-box A/B) the plane kernels are no faster (forward 84.9 vs 80.6 us: the kernel is bound by the
        # operand stream into LDS and the lockstep of its eight waves, not by the conversion) and the producers' 2-byte plane
        # stores cost more than the bytes they save: 1.120 vs 1.095 ms per step.  Kept as a tested option (RCB_PLANES=1).
        self.operand_planes = os.environ.get("RCB_PLANES", "0") != "0"
        # with fuse_next_sample: the posterior update re-draws its step's noise from the counter instead of reading the copy
        # the sampler stored (rcb_level_bwd.eps_from_rng: same bits, 8 bytes per element less traffic)
        self.redraw_noise = os.environ.get("RCB_REDRAW_EPS", "1") != "0"
        # three-level presets: the latent weights' noise drawn in the sampling kernel (RCB_HIER_RNG=0: three randn launches; A/B)
        self.hier_rng = os.environ.get("RCB_HIER_RNG", "1") != "0"
        # concurrent streams inside the captured step (bit mask, see train(); 0 = one stream).  Measured on MI355X, same-box A/B of
        # the 4096-INR CIFAR step (gpurun_out r04_ab5): one stream 1.162 ms; A-transform backward beside the upsampling net's
        # backward (2) 1.125; + the net's weight-gradient side on a third stream (6) 1.109; + the network level's posterior
        # update (HBM-bound) straight behind the A backward, i.e. beside the LDS / MFMA-bound upsampling kernels (14) 1.089:
        # -6.2 %.  The forward fork (1) gains nothing (7: 1.113); the update EARLIER, right behind the data gradient and beside
        # the stage-3 / stage-2 backward kernels (32; needs a second bf16 sample buffer) LOSES 7 % (46: 1.152 vs 1.079): the
        # HBM-bound update slows the LDS-bound kernels more than it gains; the wide weight-gradient GEMM on the third stream with
        # the update behind it and the data gradient (64) loses 2 % (78: 1.142 vs 1.116).  Same kernels, same operands: bit-identical.
        self.stream_forks = 14
        # measurement aid: take the SHARDED form of the step (four captured segments around two host-enqueued all-reduces)
        # even when `dp_group` has a single rank -- what the segmentation costs on the host, without a second GPU
        self.force_segments = False
        # sharded step: None = capture the all-reduces inside the step graph when the backend is RCCL ("nccl"), else four
        # segments around host-enqueued collectives; True / False force one form
        # None: automatic (inside the step graph when the backend is RCCL); RCB_CAPTURE_COLLECTIVES=0 / 1 forces the four-segment
        # form / the captured form without a code change (the escape hatch of a multi-GPU run)
        self.capture_collectives = {"0": False, "1": True}.get(os.environ.get("RCB_CAPTURE_COLLECTIVES", ""), None)
        # in-kernel noise: first row of this model's INRs inside a larger (virtual) batch.  Row r of the model draws the noise of
        # row rng_row_offset + r of that batch (ops.rng_group_offset): a shard or a sub-batch trained on its own then sees
        # exactly the noise of the unsharded run, given the same seed (`rng_seed_override`, else derived per model and rank)
        self.rng_row_offset = 0
        self.rng_seed_override = None
        self._train_calls = 0
        self._ws = None              # persistent training workspace (captured graphs + everything they reference)
        self._rng_ctr_init = 0       # first value of the noise counter of a new workspace (tests build exact twins)
        self.use_graph = True        # replay the training step as one captured HIP graph when possible

    # ---- level descriptions ------------------------------------------------------------------------
    def _levels(self):
        """LevelSpecs of the latent-weight hierarchy [level1, (level2, level3)] and of the lpe."""
        c = self._levels_cache
        if c is None or c["dev"] != self.loc.device:
            N, D = self.train_size, self._d_net
            net = [LevelSpec(self.loc, self.log_scale, D, N)]
            if self.patch:
                net.append(LevelSpec(self.h_loc, self.h_log_scale, D, N, row_map=self._maps[0]))
                net.append(LevelSpec(self.hh_loc, self.hh_log_scale, D, N, row_map=self._maps[1]))
            c = {"dev": self.loc.device, "net": net}
            self._levels_cache = c
        lpe = LevelSpec(self.lpe_loc.view(self.train_size, -1), self.lpe_log_scale.view(self.train_size, -1),
                        self._d_lpe, self.train_size)
        return c["net"], lpe

    def _pe_layout(self):
        """the patched presets in the 16-bit modes hand the SIREN kernel the upsampling net's output on the stitched grids
        as it is (ops.PeLayout); None: pe is cut back into [N, S, P, 16]"""
        if self.precision != 0 and self.patch and self.stitched_pe:
            return ops.PeLayout(self.patch_nums[:self.data_dim], self.pixel_sizes[:self.data_dim])
        return None

    def _pe(self, upsample_net, lpe, stitched=False, lpe16=None):
        """lpe [S, N, *lat, C] -> pe [N, S, P, 16]: hand-written phase-conv kernels in the 16-bit modes
        where the geometry is instantiated, the nn.Module (MIOpen) otherwise.  `stitched`: see _pe_layout.
        lpe16: the producer's bf16 copy of lpe [N, Dlpe] (stage-1 operand; saves the cast pass)."""
        if self.precision != 0 and hip_path_supported(upsample_net, self.pixel_sizes, self.upsample_factors, self.patch,
                                                      self.data_dim):
            return upsample_cifar_hip(upsample_net, lpe, self.stage1_bf16, self.pe_bf16, lpe16=lpe16)
        net = upsample_net
        if self.precision != 0 and tiled_2d_preferred(upsample_net, self.patch, self.data_dim):
            net = stitched2d_module(upsample_net)                  # stitched 2-D grid: phase-conv kernels on overlapping tiles
        elif self.precision != 0 and phase_form_preferred(self.data_dim, self.patch):
            net = phase_module(upsample_net) or upsample_net       # torch-level phase form: same function, fewer flops
        return map_lpe_to_inr_inputs(net, lpe, self.latent_dim, self.pixel_sizes, self.upsample_factors,
                                     self.patch, self.patch_nums, self.data_dim, stitched=stitched)

    def _noise(self, shape):
        if self.noise_source is not None:
            e = self.noise_source(tuple(shape))
            return e.to(self.loc.device, torch.float32).reshape(shape).contiguous()
        return torch.randn(shape, device=self.loc.device, dtype=torch.float32)

    def _meta(self, x, pe_dim, samples=1):
        return SirenMeta(samples=samples, n_pix=x.shape[-2], fourier_dim=x.shape[-1], pe_dim=pe_dim,
                         n_hidden=self.n_layers - 1, hidden=max(self.dims[1:-1]), out_dim=self.dims[-1], w0=self.w0,
                         precision=self.precision,
                         hidden_dims=tuple(self.dims[1:-1]) if len(set(self.dims[1:-1])) > 1 else None)

    def _layer_slices(self):
        cum = self.cum_param_sizes
        return [(0 if i == 0 else int(cum[i - 1]), int(cum[i])) for i in range(self.n_layers)]

    # ---- reference API: helpers ----------------------------------------------------------------------
    def group_to_layer(self, params, layer_idx):
        lo, hi = self._layer_slices()[layer_idx]
        return params[..., lo:hi]

    def layer_to_weight(self, in_dim, out_dim, layer_param):
        bias = layer_param[:, :out_dim]
        weights = layer_param[:, out_dim:].reshape(-1, in_dim, out_dim)
        return weights, bias

    # ---- reference API: forward (autograd-capable) -----------------------------------------------------
    def forward(self, x, linear_transform, upsample_net, gradient_through_A=True):
        """x [N, P, F] (or [P, F] shared grid) -> y_hat [N, P, C]  (prior_model.py:129-179).
        Noise draw order as in the reference: lpe, level 1, level 2, level 3."""
        assert x.shape[0] == self.train_size or x.dim() == 2
        N = self.train_size
        net, lpe_lv = self._levels()
        e_lpe = self._noise((N, 1, self._d_lpe))
        lpe = ops.sample_levels([lpe_lv], [e_lpe], 1)                       # [N,1,Dlpe]
        lpe = lpe.reshape(N, *self._lat, self.latent_dim)[None]
        pe = self._pe(upsample_net, lpe)[:, 0]
        eps = [self._noise((N, 1, self._d_net)) for _ in net]
        h_w = ops.sample_levels(net, eps, 1)[:, 0]                          # [N, D_net]
        parts = []
        for idx, (lo, hi) in enumerate(self._layer_slices()):
            A = linear_transform.A[idx] if gradient_through_A else linear_transform.A[idx].detach()
            parts.append(h_w[:, lo:hi] @ A)
        wvec = torch.cat(parts, -1)
        return ops.SirenFn.apply(x, pe.contiguous(), wvec, self._meta(x, pe.shape[-1]))

    def calculate_kl(self, prior_loc, prior_scale, prior_lpe_loc, prior_lpe_scale, prior_h_loc, prior_h_scale,
                     prior_hh_loc, prior_hh_scale):
        """Sum of elementwise Gaussian KLs over all levels (prior_model.py:181-200) -> 0-d tensor."""
        def one(loc, ls, pl, ps):
            return ops.GaussKLFn.apply(loc, ls, pl.contiguous(), ps.contiguous(), False, None, None, None, None)
        kl = one(self.loc, self.log_scale, prior_loc, prior_scale)
        kl = kl + one(self.lpe_loc, self.lpe_log_scale, prior_lpe_loc, prior_lpe_scale)
        if self.patch:
            kl = kl + one(self.h_loc, self.h_log_scale, prior_h_loc, prior_h_scale)
            kl = kl + one(self.hh_loc, self.hh_log_scale, prior_hh_loc, prior_hh_scale)
        return kl

    def _kl_value(self, priors):
        """fp64 0-d KL on device, no autograd."""
        tot = None
        pairs = [(self.loc, self.log_scale, priors[0], priors[1]),
                 (self.lpe_loc, self.lpe_log_scale, priors[2], priors[3])]
        if self.patch:
            pairs += [(self.h_loc, self.h_log_scale, priors[4], priors[5]),
                      (self.hh_loc, self.hh_log_scale, priors[6], priors[7])]
        for loc, ls, pl, ps in pairs:
            r, _ = ops.gauss_kl(loc, ls, pl.contiguous(), ps.contiguous())
            tot = r.sum() if tot is None else tot + r.sum()
        return tot

    # ---- reference API: train (fused pipeline) -----------------------------------------------------------
    def train(self, n_epoch, lr, x, y, prior_loc, prior_scale, prior_lpe_loc, prior_lpe_scale, prior_h_loc,
              prior_h_scale, prior_hh_loc, prior_hh_scale, linear_transform, upsample_net, kl_beta,
              training_mappings=True, verbose=False):
        """n_epoch Adam steps on loss = mean((y_hat-y)^2)*N + kl_beta*KL with a fresh Adam
        (prior_model.py:202-262).  Returns (MSE_last/N, KL/N, ELBO list).  NB: shadows nn.Module.train
        exactly like the reference does."""
        dev = self.loc.device
        x = x.to(dev)
        y = y.to(dev).contiguous()
        N = y.shape[0]
        P, Cc = y.shape[1], y.shape[2]
        priors = [prior_loc, prior_scale, prior_lpe_loc, prior_lpe_scale, prior_h_loc, prior_h_scale,
                  prior_hh_loc, prior_hh_scale]
        priors = [None if p is None else p.detach().to(dev, torch.float32).contiguous() for p in priors]
        net, lpe_lv = self._levels()
        A = [a for a in linear_transform.A]
        conv = [p for p in upsample_net.parameters()]
        slices = self._layer_slices()
        D = self._d_net
        # Gradients of the shared mappings are summed over ranks only when the caller asked for it by setting `dp_group`
        # (drivers.train_prior and bench.py do): an initialised process group alone must not couple models that merely
        # live in the same job (several priors / bit-rates trained side by side).
        world, rank_id = 1, 0
        if self.dp_group is not None:
            world = torch.distributed.get_world_size(self.dp_group)
            rank_id = torch.distributed.get_rank(self.dp_group)      # ranks must not draw the same noise

        # ---- workspace -----------------------------------------------------------------------------------------
        # Everything a captured step references lives in a workspace that survives across train() calls: Adam state
        # (re-zeroed per call = the reference's fresh optimiser), step counter, logs, and device copies of what changes
        # between calls (priors, beta, the noise counter).  The EM loop calls train() 550 times with 100 steps each;
        # re-capturing the graph every call costs ~14 ms = 9 % of such a call, replaying a cached one nothing.
        graphable = bool(self.use_graph and self.noise_source is None and dev.type == "cuda" and not verbose)
        key = (N, P, Cc, x.data_ptr(), tuple(x.shape), tuple(x.stride()), y.data_ptr(), float(lr), bool(training_mappings),
               world, id(linear_transform), id(upsample_net), self.precision, self.lowp_gemm, self.split_gemm, self.split_terms, self.split_dgrad_terms,
               self.wgrad_bf16, self.stage1_bf16, self.pe_bf16, self.fused_noise, self.fuse_next_sample, self.patch,
               self.operand_planes, self.redraw_noise, self.hier_rng, self.rng_row_offset, self.rng_seed_override,
               os.environ.get("RCB_FORK", str(self.stream_forks)), self.force_segments, self.capture_collectives,
               tuple(None if q is None else tuple(q.shape) for q in priors),
               # the captured kernels read and Adam-update these STORAGES: Module.cpu()/.to() (a checkpoint written the
               # reference's way, main_prior_training.py:334-338) re-allocates param.data while id() stays equal
               tuple(q.data_ptr() for q in A + conv), tuple(q.data_ptr() for q in self.parameters()))
        ws = self._ws if graphable else None
        if ws is not None and (ws["key"] != key or ws["rows"] < n_epoch):
            ws = None
        if ws is None:
            # all Adam moments (per-INR levels and shared mappings) as views of ONE zero-filled buffer: a train() call resets
            # them with one launch instead of ~30 (the reference creates a fresh optimiser per call: main_prior_training.py)
            shapes = [lv.loc.shape for lv in net for _ in range(4)] + [lpe_lv.loc.shape] * 4
            if training_mappings:
                shapes += [q.shape for q in A + conv for _ in range(2)]
            offs, tot = [], 0
            for shp in shapes:
                offs.append(tot)
                tot += (int(np.prod(shp)) + 63) // 64 * 64          # every view starts on a 256-byte boundary
            state_flat = torch.zeros(tot, device=dev, dtype=torch.float32)
            views = iter([state_flat[o:o + int(np.prod(shp))].view(shp) for o, shp in zip(offs, shapes)])

            def st4(lv):
                return {k: next(views) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
            rows = max(n_epoch, 256) if graphable else n_epoch
            self._train_calls += 1
            ws = dict(key=key, rows=rows, state_flat=state_flat, net_state=[st4(lv) for lv in net], lpe_state=st4(lpe_lv),
                      map_state=[(next(views), next(views)) for q in A + conv] if training_mappings else None,
                      tab=ops.adam_table(lr, rows).to(dev), dyn=torch.zeros(2, device=dev, dtype=torch.float32),
                      step_t=torch.zeros(1, device=dev, dtype=torch.long),
                      kl_slots=torch.zeros(1024, device=dev, dtype=torch.int64),       # fixed point, 2^-24 nats
                      mse_buf=torch.zeros(rows, device=dev, dtype=torch.float64),
                      kl_buf=torch.zeros(rows, device=dev, dtype=torch.float64),
                      pri=[None if q is None else torch.empty_like(q) for q in priors],
                      beta_dev=torch.zeros(1, device=dev, dtype=torch.float32),
                      rng_ctr=torch.full((1,), int(self._rng_ctr_init), device=dev, dtype=torch.long), graphs=None,
                      seed=(int(self.random_seed) * 0x9E3779B97F4A7C15 + int(torch.initial_seed()) * 0xBF58476D1CE4E5B9
                            + self._train_calls * 0x94D049BB133111EB + rank_id * 0xD6E8FEB86659FD93) & (2 ** 64 - 1),
                      # sharded: the mapping gradients of a step travel in two buckets of one flat buffer (dist.GradBuckets)
                      flat=(rdist.GradBuckets([q.shape for q in A], [q.shape for q in conv], dev, self.dp_group)
                            if (training_mappings and (world > 1 or (self.force_segments and self.dp_group is not None))) else None),
                      # the bf16 copy of the coordinate grid the captured SIREN launches read: owned by the workspace,
                      # i.e. alive exactly as long as the graphs that reference its address
                      xf16=ops.xf_bf16(x, self.precision) if (self.precision in (1, 2) and dev.type == "cuda") else None)
            # 16-bit modes: the A transform on the hand-written kernels of atrans.hip (every geometry)
            ws["split"] = (ops.ATransform(slices, dev, self.split_terms, self.split_dgrad_terms)
                           if (self.split_gemm and self.precision != 0 and not self.lowp_gemm) else None)
            if graphable:
                self._ws = ws
        else:
            ws["state_flat"].zero_()
        n_a = sum(q.numel() for q in A)                     # the A matrices come first in the gradient bucket
        net_state, lpe_state, map_state = ws["net_state"], ws["lpe_state"], ws["map_state"]
        tab, dyn, step_t, kl_slots = ws["tab"], ws["dyn"], ws["step_t"], ws["kl_slots"]
        mse_buf, kl_buf, beta_dev, rng_ctr, flat, split = (ws["mse_buf"], ws["kl_buf"], ws["beta_dev"], ws["rng_ctr"],
                                                         ws["flat"], ws["split"])
        step_t.zero_()
        beta_dev.fill_(float(kl_beta))
        for dst, src in zip(ws["pri"], priors):
            if dst is not None:
                dst.copy_(src)
        pri_d = ws["pri"]              # what the kernels read (stable addresses); `priors` = this call's values
        net_priors = [(pri_d[0], pri_d[1])] + ([(pri_d[4], pri_d[5]), (pri_d[6], pri_d[7])] if self.patch else [])
        # per-step Adam scalars and the step counter live on the device, so that one captured HIP graph of the
        # step body can be replayed for every step (the host launch cost of ~60 kernels is ~2 ms per step)
        cfg = ops.adam_cfg(lr, 1, dyn=dyn)

        # The step runs as four segments so that, under data-parallel sharding, the gradient bucket of the shared
        # mappings is all-reduced asynchronously BETWEEN captured graphs while work that does not depend on it proceeds:
        #   seg1a: sample .. SIREN .. A-transform backward | all-reduce(A grads, 13.4 MB) ||
        #   seg1b: upsampling-net backward | all-reduce(conv grads, 1 MB) || seg2: posterior update | wait |
        #   seg3: Adam on the mappings + bookkeeping.
        st = {}
        # in-kernel noise: a pure function of (seed, stream, counter, element); the counter lives in the workspace and is
        # never reset, so every step of every train() call draws fresh noise, replayed or not
        # (the lpe is a single plain level in every preset; the latent weights are one only without patches: the three-level
        # presets keep stored noise for the weights and draw the lpe's -- two thirds of what they sample -- in the kernels)
        rng_lpe = bool(self.fused_noise and self.noise_source is None and ops.rng_eligible(lpe_lv))
        use_rng = rng_lpe and len(net) == 1 and ops.rng_eligible(net[0])
        rng_seed = ws["seed"] if self.rng_seed_override is None else int(self.rng_seed_override) & (2 ** 64 - 1)
        # three levels of latent weights (patched presets): every level's noise drawn in the sampling kernel (Philox streams 0, 2, 3
        # at the element index of the [N, D] noise arrays: independent of torch's generator and of how the rows are sharded) and
        # written for the posterior updates, instead of three randn launches read back by the kernel
        rng_hier = bool(self.hier_rng and rng_lpe and not use_rng and ops.hier_rng_eligible(net) and (N * D) % 4 == 0
                        and (self.rng_row_offset * D) % 4 == 0)
        goff_net = ops.rng_group_offset(self.rng_row_offset, D) if (use_rng or rng_hier) else 0
        goff_lpe = ops.rng_group_offset(self.rng_row_offset, self._d_lpe) if rng_lpe else 0

        want16 = split is not None and training_mappings and self.wgrad_bf16     # bf16 operands of the weight gradient
        lpe16_want = bool(use_rng and self.stage1_bf16 and self.precision != 0 and not self.patch)
        # (the fused form lives on the posterior update's flat path: element counts divisible by 4)
        fuse_next = bool(use_rng and self.fuse_next_sample and (N * D) % 4 == 0 and (N * self._d_lpe) % 4 == 0)
        # the A transform's operands as (hi, lo) planes written by their producers (single-level presets with in-kernel noise;
        # layers must start at multiples of 8 columns for the 16-byte LDS-DMA pieces)
        planes = bool(split is not None and use_rng and self.operand_planes
                      and all(lo % 8 == 0 for (lo, hi) in slices if (hi - lo) % 8 == 0))
        fuse_lpe = bool(fuse_next or (rng_lpe and self.fuse_next_sample and (N * self._d_lpe) % 4 == 0))
        redraw = bool(fuse_next and self.redraw_noise)
        redraw_lpe = bool(fuse_lpe and self.redraw_noise)
        smp_net = smp_lpe = None
        if fuse_next:
            if "smp_net" not in ws:
                ws["smp_net"] = ops.sample_buffers(net[0], True, planes=planes, want_eps=not redraw)
                ws["smp_lpe"] = ops.sample_buffers(lpe_lv, True, want_eps=not redraw)
            smp_net, smp_lpe = ws["smp_net"], ws["smp_lpe"]
        elif fuse_lpe:
            if "smp_lpe" not in ws:
                ws["smp_lpe"] = ops.sample_buffers(lpe_lv, lpe16_want, want_eps=not redraw_lpe)
            smp_lpe = ws["smp_lpe"]
        elif planes and "smp_net" not in ws:
            ws["smp_net"] = ops.sample_buffers(net[0], planes=True)
        if fuse_next and not planes and "h16_next" not in ws and smp_net[2] is not None:
            ws["h16_next"] = torch.empty_like(smp_net[2])

        def prime():
            """the sample of the call's first step (every later one comes out of the previous step's posterior update)"""
            if fuse_next:
                ops.reparam_rng(net[0], rng_seed, 0, rng_ctr, want_bf16=want16, buffers=smp_net, group_offset=goff_net)
            ops.reparam_rng(lpe_lv, rng_seed, 1, rng_ctr, want_bf16=lpe16_want, buffers=smp_lpe, group_offset=goff_lpe)

        pe_lay = self._pe_layout()
        # sharded step: with RCCL the two all-reduces are CAPTURED inside the one step graph (measured on this stack: an
        # all-reduce captures and replays correctly from the capturing stream and from a forked one, tools/rccl_capture_probe.py;
        # the four-segment form costs 142 us of host work per 1.09 ms step, bench.py `sharded_step_host_cost`); other
        # backends (gloo: the CPU rehearsals) keep the four captured segments around host-enqueued collectives
        # Default (capture_collectives None): captured on a ONE-rank RCCL communicator only -- that is all this form has ever
        # run on (the pool has one GPU per box).  With more ranks the four-segment form runs unless RCB_CAPTURE_COLLECTIVES=1
        # (or capture_collectives=True) opts in; bench.py's launcher sets it after an N-rank tools/rccl_capture_probe.py run
        # succeeded within its timeout, so that the first N > 1 execution of the captured form is never the measured one.
        capture_coll = bool(flat is not None and self.capture_collectives is not False
                            and (self.capture_collectives is True
                                 or (torch.distributed.get_backend(self.dp_group) == "nccl"
                                     and torch.distributed.get_world_size(self.dp_group) == 1)))
        # stream forks inside the (captured) step -- bit mask, same kernels on the same operands, so results are identical:
        #   1: the A transform's forward beside the upsampling net's forward;  2: the A transform's backward beside the
        #   upsampling net's backward;  4: the weight-gradient side of the upsampling net's backward beside its data path
        #   (upsample_fast.WEIGHT_SIDE_STREAM);  8: the network level's posterior update right behind the A transform's backward
        #   on the forked stream (it only needs dh);  16: Adam on the mappings on the third stream beside the lpe level's update;
        #   32 (with 2 and 8, one rank): that update right behind the data gradient, the wide weight-gradient GEMM on the third stream;
        #   64 (with 2 and 8, one rank): the wide GEMM on the third stream, the update behind it AND the data gradient
        fork_mask = int(os.environ.get("RCB_FORK", str(self.stream_forks))) if dev.type == "cuda" else 0
        fork = side = None
        if fork_mask:
            if "fork_stream" not in ws:
                ws["fork_stream"], ws["side_stream"] = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            fork, side = ws["fork_stream"], ws["side_stream"]
        from . import upsample_fast as _uf
        _uf.WEIGHT_SIDE_STREAM = side if (fork_mask & 4) else None
        if split is not None and not training_mappings:
            split.prepare(A)                          # fixed mappings: packed once per call, outside the captured step

        def seg1a():
            ops.step_begin(tab, step_t, dyn, kl_slots)
            # ---- sample + the two forward maps.  With the fused next sample h_w was written by the previous step's posterior
            # update (the step's LAST big kernel, see body()): its A transform goes first, while h_w is still in the Infinity Cache
            # (measured: read cold, half a step later, the forward A transform takes 10 us longer)
            def pe_forward():
                # ---- sample ---------------------------------------------------------------------------------
                lpe16 = None
                if fuse_lpe:                 # written by the previous step's posterior update (or by prime() before the first)
                    lpe, e_lpe, lpe16 = smp_lpe[0], smp_lpe[1], (smp_lpe[2][:, :lpe_lv.cols] if lpe16_want else None)
                elif rng_lpe and lpe16_want:
                    lpe, e_lpe, lpe16 = ops.reparam_rng(lpe_lv, rng_seed, 1, rng_ctr, want_bf16=True, group_offset=goff_lpe)   # bf16 copy: stage-1 operand
                elif rng_lpe:
                    lpe, e_lpe = ops.reparam_rng(lpe_lv, rng_seed, 1, rng_ctr, group_offset=goff_lpe)
                else:
                    e_lpe = self._noise((N, 1, self._d_lpe))
                    lpe = ops.reparam_fwd([lpe_lv], [e_lpe], 1)
                lpe_t = lpe.view(1, N, *self._lat, self.latent_dim).requires_grad_(True)
                with torch.enable_grad():
                    if pe_lay is not None:                                   # stitched grids, addressed in place by the kernel
                        pe_c = self._pe(upsample_net, lpe_t, stitched=True)
                    else:
                        pe = self._pe(upsample_net, lpe_t, lpe16=lpe16)      # [N, 1, P, E]
                        pe_c = pe.reshape(N, pe.shape[2], pe.shape[3]).contiguous()   # (not pe[:, 0]: select's backward
                        #                                                               materialises zeros + a copy)
                return lpe_t, e_lpe, pe_c

            def net_forward():
                h16 = None
                if planes:                   # h_w as its (hi, lo) planes: from the previous step's update, or sampled here
                    if not fuse_next:
                        ops.reparam_rng(net[0], rng_seed, 0, rng_ctr, buffers=ws["smp_net"], group_offset=goff_net)
                    h_w, eps = ws["smp_net"][2], [ws["smp_net"][1]]
                elif fuse_next:
                    h_w, e0, h16 = smp_net[0], smp_net[1], (smp_net[2][:, :D] if want16 else None)
                    eps, h_w = [e0], h_w.view(N, D)
                elif use_rng:
                    if want16:
                        h_w, e0, h16 = ops.reparam_rng(net[0], rng_seed, 0, rng_ctr, want_bf16=True, group_offset=goff_net)
                    else:
                        h_w, e0 = ops.reparam_rng(net[0], rng_seed, 0, rng_ctr, group_offset=goff_net)
                    eps, h_w = [e0], h_w.view(N, D)
                elif rng_hier:
                    if "hier_eps" not in ws:
                        ws["hier_eps"] = [torch.empty(N, 1, D, device=dev, dtype=torch.float32) for _ in net]
                    eps = ws["hier_eps"]
                    h_w = ops.reparam_hier_rng(net, eps, rng_seed, (0, 2, 3)[:len(net)], rng_ctr, goff_net).view(N, D)
                else:
                    eps = [self._noise((N, 1, D)) for _ in net]
                    h_w = ops.reparam_fwd(net, eps, 1).view(N, D)
                # ---- A transform (dense GEMMs) --------------------------------------------------------------
                lowp = self.lowp_gemm and self.precision != 0
                if lowp:
                    # forward: f16 operands (11-bit mantissa; A pre-scaled by 2^10 so that the ~1e-4 products of
                    # h_w @ A stay in f16's normal range) -- finer than the bf16 rounding the MLP kernel applies to
                    # its weights anyway; backward GEMMs: bf16 operands (range-safe, gradients only)
                    hf = h_w.to(torch.float16)
                    wvec = torch.cat([torch.mm(hf[:, lo:hi], (a.detach() * 1024.0).to(torch.float16))
                                      for (lo, hi), a in zip(slices, A)], 1).float() * (1.0 / 1024.0)
                    h16 = h_w.to(torch.bfloat16)
                    A16 = [a.detach().to(torch.bfloat16) for a in A]
                elif split is not None:
                    if training_mappings:
                        split.prepare(A)                  # the mappings change every step when they are trained
                    wvec = split.forward(h_w, split.new_rows(N))
                else:
                    wvec = torch.empty(N, D, device=dev, dtype=torch.float32)
                    for (lo, hi), a in zip(slices, A):
                        torch.mm(h_w[:, lo:hi], a.detach(), out=wvec[:, lo:hi])
                return h_w, h16, eps, wvec, (A16 if lowp else None)

            if fuse_next and fork is not None and (fork_mask & 1):
                # experiment (RCB_FORK bit 1, same-box A/B only): the A transform of the sample on a second stream beside the
                # upsampling net's forward -- the two are independent until the SIREN kernel
                cur = torch.cuda.current_stream()
                fork.wait_stream(cur)
                with torch.cuda.stream(fork):
                    h_w, h16, eps, wvec, A16 = net_forward()
                lpe_t, e_lpe, pe_c = pe_forward()
                cur.wait_stream(fork)
            elif fuse_next:
                h_w, h16, eps, wvec, A16 = net_forward()
                lpe_t, e_lpe, pe_c = pe_forward()
            else:
                lpe_t, e_lpe, pe_c = pe_forward()
                h_w, h16, eps, wvec, A16 = net_forward()
            lowp = self.lowp_gemm and self.precision != 0
            # ---- fused SIREN forward + MSE + backward ---------------------------------------------------
            meta = self._meta(x, pe_c.shape[-1])
            dw16 = None
            if planes:        # the gradient leaves the kernel as its (hi, lo) planes: the A transform's operand form
                sse, _, dpe, dw = ops.siren_loss_bwd(x, pe_c.detach(), wvec, y, 1.0 / (P * Cc), meta, want_planes=True,
                                                    pe_layout=pe_lay, xf16=ws["xf16"])
            elif want16:        # the kernel's epilogue also writes the bf16 copy of the gradient
                sse, dw, dpe, dw16 = ops.siren_loss_bwd(x, pe_c.detach(), wvec, y, 1.0 / (P * Cc), meta, want_bf16=True,
                                                        pe_layout=pe_lay, xf16=ws["xf16"])
            else:
                sse, dw, dpe = ops.siren_loss_bwd(x, pe_c.detach(), wvec, y, 1.0 / (P * Cc), meta, pe_layout=pe_lay,
                                                  xf16=ws["xf16"])
            # ---- backward through the A transform (first: its gradients are the bulk of the all-reduce bucket) -------
            gA = []
            fork_bwd = fork is not None and (fork_mask & 2) and split is not None and (flat is None or capture_coll) and fuse_next
            early = bool(fork_bwd and (fork_mask & 32) and (fork_mask & 8) and flat is None and training_mappings and want16 and not planes
                         and self.wgrad_bf16 and side is not None and h16 is not None and dw16 is not None)
            split_w = bool(fork_bwd and (fork_mask & 64) and not (fork_mask & 32) and (fork_mask & 8) and flat is None and training_mappings
                           and want16 and not planes and self.wgrad_bf16 and side is not None and h16 is not None and dw16 is not None)
            if split_w:
                # bit 64: the wide layers' weight-gradient GEMM on the third stream right behind the SIREN kernel; the forked
                # stream runs the narrow layers' kernels and the data gradient, then WAITS for that GEMM (it reads the bf16 sample
                # the update overwrites) before the network level's posterior update: the update starts as soon as both are done
                # instead of behind the GEMM on its own stream
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                fork.wait_stream(cur)
                with torch.cuda.stream(side):
                    g_wide = split.wgrad(h_w, dw, h16, dw16, True, part="wide")
                    st["ev_wide"] = torch.cuda.Event()
                    st["ev_wide"].record(side)
                with torch.cuda.stream(fork):
                    g_rest = split.wgrad(h_w, dw, h16, dw16, True, part="rest")
                    dh = split.dgrad(dw, torch.empty(N, D, device=dev, dtype=torch.float32))
                gA = [a_ if a_ is not None else b_ for a_, b_ in zip(g_wide, g_rest)]
                st["fork_pending"] = st["side_pending"] = True
                st["fork_keep"] = (dw, dw16, h_w, h16)
            elif early:
                # bit 32: the network level's posterior update as EARLY as possible -- right behind the data gradient -- so that
                # this HBM-bound kernel runs beside the LDS / MFMA-bound upsampling backward instead of beside the step's last
                # small GEMMs (measured: two 30 us GEMMs took 170 us each next to it).  What kept it late was a hazard: it
                # writes the next step's sample over the buffers the weight gradient still reads.  The wide layers' GEMM
                # (bf16 copies only) goes to the third stream and the update writes the next bf16 copy into a SECOND buffer,
                # copied over at the end of the step (27 MB); the narrow layers' kernels (fp32 rows) run in front of the update.
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                fork.wait_stream(cur)
                with torch.cuda.stream(side):
                    g_wide = split.wgrad(h_w, dw, h16, dw16, True, part="wide")
                with torch.cuda.stream(fork):
                    g_rest = split.wgrad(h_w, dw, h16, dw16, True, part="rest")
                    dh = split.dgrad(dw, torch.empty(N, D, device=dev, dtype=torch.float32))
                gA = [a_ if a_ is not None else b_ for a_, b_ in zip(g_wide, g_rest)]
                st["fork_pending"] = st["early"] = True
                st["fork_keep"] = (dw, dw16, h_w, h16)
            elif fork_bwd:      # the A transform's backward on the second stream beside the upsampling net's backward
                fork.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(fork):
                    dh = split.dgrad(dw, torch.empty(N, D, device=dev, dtype=torch.float32))
                    if training_mappings:
                        gA = split.wgrad(h_w, dw, h16, dw16, self.wgrad_bf16)
                        if capture_coll:                # bucket 0 (13.4 of the 14.4 MB) leaves from this stream, inside the graph
                            gA = flat.pack(0, gA)
                            st["h0"] = flat.reduce(0, async_op=True)
                    if fork_mask & 16:                  # the mapping gradients are complete here (Adam may start on another stream)
                        st["ev_gA"] = torch.cuda.Event()
                        st["ev_gA"].record(fork)
                st["fork_pending"] = True
                # dw (and its bf16 copy) were allocated on the main stream and are READ on the forked one: they must stay
                # allocated until the streams have joined, or the caching allocator hands their memory to the next main-stream
                # allocation (the upsampling net's backward) while the A transform is still reading it
                st["fork_keep"] = (dw, dw16, h_w, h16)
            elif lowp:
                dw16 = dw.to(torch.bfloat16)
                dh = torch.cat([torch.mm(dw16[:, lo:hi], a16.t()) for (lo, hi), a16 in zip(slices, A16)], 1).float()
                if training_mappings:
                    gA = [torch.mm(h16[:, lo:hi].t(), dw16[:, lo:hi]).float() for (lo, hi) in slices]
            elif split is not None:
                dh = split.dgrad(dw, torch.empty(N, D, device=dev, dtype=torch.float32))
                if training_mappings:
                    gA = split.wgrad(h_w, dw, h16, dw16, self.wgrad_bf16)
            else:
                dh = torch.empty(N, D, device=dev, dtype=torch.float32)
                for (lo, hi), a in zip(slices, A):
                    torch.mm(dw[:, lo:hi], a.detach().t(), out=dh[:, lo:hi])
                    if training_mappings:
                        gA.append(torch.mm(h_w[:, lo:hi].t(), dw[:, lo:hi]))
            st.update(sse=sse, dh3=dh.view(N, 1, D), eps=eps, e_lpe=e_lpe, pe_c=pe_c, lpe_t=lpe_t, dpe=dpe)
            if training_mappings:
                if flat is not None and "h0" not in st:
                    gA = flat.pack(0, gA)
                    if capture_coll:
                        st["h0"] = flat.reduce(0, async_op=True)
                st["gA"] = gA

        def seg1b():
            # ---- backward through the upsampling net -----------------------------------------------------------
            inputs = [st["lpe_t"]] + (conv if training_mappings else [])
            g_in = torch.autograd.grad(st["pe_c"], inputs, st["dpe"])
            st["d_lpe"] = g_in[0].reshape(N, 1, self._d_lpe).contiguous()
            if training_mappings:
                gc = [g.contiguous() for g in g_in[1:]]
                if flat is not None:
                    gc = flat.pack(1, gc)
                    if capture_coll:
                        st["h1"] = flat.reduce(1, async_op=True)
                st["grads"] = st["gA"] + gc
            # the autograd graph of this step must not outlive it: a graph kept alive from an eager warm-up step (default
            # stream) into the capture (side stream) makes its AccumulateGrad nodes cross streams and breaks the capture
            for k_ in ("pe_c", "lpe_t", "dpe"):
                st.pop(k_, None)

        def comm(part):
            """sum of the mapping gradients over the ranks (the only per-step collectives): part 0 = the A matrices (13.4 of
            the 14.4 MB, ready before the upsampling backward starts), part 1 = the conv weights; returns the async handle"""
            if flat is None:
                return None
            return flat.reduce(part, async_op=True)

        def seg2_net():
            # fused posterior update (also accumulates the pre-update KL for the ELBO log)
            nxt_net = None
            if fuse_next:            # the next step sees the noise counter + 1 (rcb_step_end increments it after this segment)
                o16_ = smp_net[2] if (want16 or planes) else None
                if st.get("early"):                  # (bit 32: the next bf16 copy goes to the second buffer, see seg1a)
                    o16_ = ws["h16_next"]
                nxt_net = ops.NextSample((smp_net[0], smp_net[1], o16_), rng_seed, 0, rng_ctr, 1,
                                         redraw_eps=redraw, group_offset=goff_net)
            for lv, (pl, ps), e, stt in zip(net, net_priors, st["eps"], net_state):
                ops.posterior_bwd(lv, pl, ps, False, 1.0, st["dh3"], e, 1, adam=cfg, state=stt, kl_accum=kl_slots,
                                  kl_scalar_dev=beta_dev, next_sample=nxt_net)

        def seg2_lpe():
            nxt_lpe = None
            if fuse_lpe:
                nxt_lpe = ops.NextSample((smp_lpe[0], smp_lpe[1], smp_lpe[2] if lpe16_want else None), rng_seed, 1, rng_ctr, 1,
                                         redraw_eps=redraw_lpe, group_offset=goff_lpe)
            ops.posterior_bwd(lpe_lv, pri_d[2].reshape(-1), pri_d[3].reshape(-1), False, 1.0, st["d_lpe"],
                              st["e_lpe"], 1, adam=cfg, state=lpe_state, kl_accum=kl_slots, kl_scalar_dev=beta_dev,
                              next_sample=nxt_lpe)

        def seg2():
            seg2_net()
            seg2_lpe()

        def seg3_adam():
            for k_ in ("h0", "h1"):             # captured collectives: the stream that runs Adam waits for both buckets
                h_ = st.pop(k_, None)
                if h_ is not None:
                    h_.wait()
            if training_mappings:
                ops.adam_multi([p.data for p in A + conv], [g.contiguous() for g in st["grads"]],
                               [m for m, _ in map_state], [v for _, v in map_state], cfg)

        def seg3_end():
            ops.step_end(step_t, st["sse"], 1.0 / (P * Cc), kl_slots, mse_buf, kl_buf, aux_counter=rng_ctr)

        def seg3():
            seg3_adam()
            seg3_end()

        def body():
            if (fuse_next and flat is None) or capture_coll:
                # one rank (or collectives inside the graph): the network level's update -- which writes the next step's h_w -- is the step's last big kernel, so
                # that the next step's A transform (its first) finds h_w in the Infinity Cache
                seg1a()
                pending = st.pop("fork_pending", False)
                if pending and (fork_mask & 8):
                    with torch.cuda.stream(fork):      # the network level's update needs dh only: straight behind the A backward
                        ev_w = st.pop("ev_wide", None)
                        if ev_w is not None:
                            fork.wait_event(ev_w)
                        seg2_net()
                seg1b()
                if st.pop("side_pending", False):
                    torch.cuda.current_stream().wait_stream(side)
                if st.pop("early", False):
                    # join both streams, then the next step's bf16 copy moves into the buffer the step reads (see seg1a)
                    torch.cuda.current_stream().wait_stream(fork)
                    torch.cuda.current_stream().wait_stream(side)
                    smp_net[2].copy_(ws["h16_next"])
                ev = st.pop("ev_gA", None)
                if pending and ev is not None and side is not None:
                    # bit 16: Adam on the mappings on the third stream (it needs the A gradients -- the event -- and the
                    # upsampling net's weight gradients, which that stream produced itself) beside the lpe level's update
                    side.wait_stream(torch.cuda.current_stream())
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        seg3_adam()
                    seg2_lpe()
                    torch.cuda.current_stream().wait_stream(fork)
                    torch.cuda.current_stream().wait_stream(side)
                    st.pop("fork_keep", None)
                else:
                    if pending:
                        torch.cuda.current_stream().wait_stream(fork)
                        st.pop("fork_keep", None)
                    seg2_lpe()
                    seg3_adam()
                if not (pending and (fork_mask & 8)):
                    seg2_net()
                seg3_end()
                return
            seg1a()
            w0 = comm(0)
            seg1b()
            w1 = comm(1)
            seg2()
            for w_ in (w0, w1):
                if w_ is not None:
                    w_.wait()
            seg3()

        n_warm = 3

        def replay(k):
            kind, gr = ws["graphs"]
            for _ in range(k):
                if kind == "one":
                    gr.replay()
                else:
                    gr[0].replay()
                    w0 = comm(0)
                    gr[1].replay()
                    w1 = comm(1)
                    gr[2].replay()
                    w0.wait()
                    w1.wait()
                    gr[3].replay()

        if fuse_lpe:
            prime()
        if graphable and (ws["graphs"] is not None or n_epoch > n_warm):
            left = n_epoch
            failure = None
            if ws["graphs"] is None:               # first call with this workspace: warm up eagerly, then capture
                for _ in range(n_warm):
                    body()
                left -= n_warm
                torch.cuda.synchronize()
                try:
                    if flat is None or capture_coll:   # one rank / frozen mappings / captured collectives: the whole step is one graph
                        graph = torch.cuda.CUDAGraph()
                        # (thread_local: the process group's watchdog thread must not be able to invalidate the capture)
                        with torch.cuda.graph(graph, capture_error_mode="thread_local" if flat is not None else "global"):
                            body()                         # records the step; nothing executes during capture
                        ws["graphs"] = ("one", graph)
                    else:                               # sharded: four graphs around the two eager, asynchronous all-reduces
                        pool = torch.cuda.graph_pool_handle()
                        graphs = []
                        for seg in (seg1a, seg1b, seg2, seg3):
                            g = torch.cuda.CUDAGraph()
                            # thread_local: the process group's own threads (RCCL watchdog: event queries) must not be able
                            # to invalidate a capture in progress on this thread
                            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                                seg()
                            graphs.append(g)
                        ws["graphs"] = ("segments", graphs)
                except Exception as exc:     # capture is an optimisation: fall back to eager stepping
                    failure = exc
                    torch.cuda.synchronize()
                    # a body that aborted in mid-capture leaves its per-step state behind (pending fork / side-stream
                    # flags, the Work handles of collectives that were only RECORDED): an eager step that found them would
                    # skip its own all-reduce and wait on a handle of the dead capture
                    st.clear()
                if flat is not None:
                    # every rank must take the same route from here on: replayed segments and eager steps issue the same
                    # collectives, but a rank that alone drops to eager would also alone flip `use_graph` for later calls.
                    # No collective has EXECUTED during the capture attempt -- in the four-segment form they sit between the
                    # segments, in the captured form they were recorded into a graph that is dropped -- so the ranks are
                    # still aligned and can agree here.
                    ok = torch.tensor([0 if failure is not None else 1], device=dev, dtype=torch.int32)
                    torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=self.dp_group)
                    if int(ok.item()) == 0 and failure is None:
                        failure = RuntimeError("capture failed on another rank")
            if failure is None:
                replay(left)
            else:
                import warnings
                warnings.warn(f"HIP graph capture of the training step failed ({failure}); running eagerly")
                ws["graphs"] = None
                self._ws = None
                self.use_graph = False
                for _ in range(left):
                    body()
        else:
            it = range(n_epoch)
            if verbose:
                from tqdm import tqdm
                it = tqdm(it)
            for _ in it:
                body()
        _uf.WEIGHT_SIDE_STREAM = None       # (a module-level switch: never left set behind this call)
        mse_buf, kl_buf = mse_buf[:n_epoch], kl_buf[:n_epoch]
        kl_final = self._kl_value(priors)
        # one transfer (one synchronisation) for everything the call returns: [elbo per step ..., last MSE, final KL]
        out_h = torch.cat([-(mse_buf + kl_buf * float(kl_beta)), mse_buf[-1:], kl_final.reshape(1)]).cpu()
        return float(out_h[-2]) / N, float(out_h[-1]) / N, out_h[:-2].tolist()


# ------------------------------------------------------------------------------------------------------
# grouping (prior_model.py:264-316) -- host side, as in the reference
# ------------------------------------------------------------------------------------------------------
def get_grouping(q_loc, q_scale, prior_loc, prior_scale):
    """Mean-over-rows KL in bits per parameter (HIP reduction), then greedy packing on the host."""
    rows = q_loc.shape[0]
    colsum = ops.gauss_kl_colsum(q_loc, q_scale, prior_loc, prior_scale, q_is_log=False)
    weights = (colsum / np.log(2.) / rows).to(torch.float32).cpu().numpy()
    return get_grouping_by_kl(weights)


def group_parameters(parameters, weights, max_weight=16):
    """Sequential greedy packing; the running sum keeps the dtype of `weights` (fp32 in practice)."""
    groups = [[parameters[0]]]
    run = weights[0]
    for i in range(1, len(parameters)):
        if run + weights[i] > max_weight:
            groups.append([parameters[i]])
            run = weights[i]
        else:
            groups[-1].append(parameters[i])
            run = run + weights[i]
    return groups


def get_grouping_by_kl(kls_bits):
    """-> (group_idx, group_start_index, group_end_index, group2param, param2group, n_groups,
    group_kls, weights) with the fixed np.random.seed(0) shuffle (prior_model.py:273-299)."""
    D = kls_bits.shape[0]
    np.random.seed(0)
    order = np.random.choice(D, D, False)
    np.random.seed(None)
    result = group_parameters(np.arange(D)[order], kls_bits[order])
    sizes = np.array([len(g) for g in result])
    n_groups = len(result)
    param2group = np.concatenate([np.asarray(g) for g in result])
    group2param = np.argsort(param2group)
    group_idx = np.repeat(np.arange(n_groups), sizes).astype(int)
    end = np.cumsum(sizes)
    start = end - sizes
    group_kls = np.array([sum(kls_bits[j] for j in g) for g in result])
    return group_idx, start, end, group2param, param2group, n_groups, group_kls, kls_bits
