"""Sub-pixel ("phase") decomposition of the upsampling net (prior_model.py:23-59).

Every stage of the reference net is  nearest-upsample(f) -> conv(k, pad) .  For output pixel f*i + a
(phase a in [0, f)) kernel tap kk reads up-sampled coordinate f*i + a + kk - pad, i.e. SOURCE pixel
i + floor((a + kk - pad) / f).  For the reference's (f, k, pad) = (4,5,2), (6,5,2), (2,3,1) every phase
touches only TWO neighbouring source pixels per axis, so each stage is, per phase, a dense 2^d-tap
convolution on the *small* source grid with pre-summed weights

    Weff[phase][tap][ci][co] = sum_{kernel taps that land on that source pixel} W[co][ci][taps].

This is exact (only the fp32 summation order changes) and needs 21 instead of 64 MFLOP per CIFAR
INR; the up-sampled intermediates are never materialised.  Implemented here with plain tensor ops
(window gathers + GEMMs) so that autograd supplies the backward; it works for 1-D, 2-D and 3-D nets.
"""
import itertools
import math

import numpy as np
import torch
import torch.nn.functional as F


def _axis_plan(f: int, k: int, pad: int):
    """per-axis phase plan: window width w, left/right zero padding, per-phase window shift s[a]
    and the 0/1 matrix R[a, t, kk] that folds kernel tap kk of phase a onto window tap t."""
    lo = [min(math.floor((a + kk - pad) / f) for kk in range(k)) for a in range(f)]
    hi = [max(math.floor((a + kk - pad) / f) for kk in range(k)) for a in range(f)]
    w = max(h - l + 1 for l, h in zip(lo, hi))
    pad_l = -min(lo)
    pad_r = max(l + w - 1 for l in lo)
    R = np.zeros([f, w, k], dtype=np.float32)
    for a in range(f):
        for kk in range(k):
            o = math.floor((a + kk - pad) / f)
            R[a, o - lo[a], kk] = 1.0
    shift = [l + pad_l for l in lo]
    return w, pad_l, max(pad_r, 0), shift, R


class PhaseStage:
    """one  nearest-upsample(f) -> conv(k, pad)  stage in phase form (channel-last tensors)."""

    def __init__(self, factors, k, pad, dd):
        self.dd = dd
        self.f = [int(v) for v in (factors if isinstance(factors, (tuple, list)) else [factors] * dd)]
        self.plans = [_axis_plan(f, k, pad) for f in self.f]
        self.k, self.pad = k, pad

    def eff_weight(self, W):
        """W [Cout, Cin, *k] -> Weff [*taps, Cin, *phases, Cout]."""
        dd = self.dd
        Rs = [_phase_R(W.device, f, self.k, self.pad).to(W.dtype) for f in self.f]
        letters = "abcdefghijklmnopqrstuvwxyz"
        ph, tp, kk = letters[0:dd], letters[dd:2 * dd], letters[2 * dd:3 * dd]
        o, i = "O", "I"
        terms = [f"{ph[d]}{tp[d]}{kk[d]}" for d in range(dd)] + [f"{o}{i}{kk}"]
        out = f"{tp}{i}{ph}{o}"
        return torch.einsum(",".join(terms) + "->" + out, *Rs, W)

    def forward(self, x, W, b):
        """x [B, *g, Cin] -> [B, *(f*g), Cout]"""
        dd = self.dd
        B, g, Cin = x.shape[0], list(x.shape[1:-1]), x.shape[-1]
        Cout = W.shape[0]
        Weff = self.eff_weight(W)                                   # [*taps, Cin, *phases, Cout]
        ws = [p[0] for p in self.plans]
        pads = []
        for p in reversed(self.plans):                              # F.pad lists the last dim first
            pads += [p[1], p[2]]
        xp = F.pad(x, [0, 0] + pads)                                # zero halo
        for d in range(dd):
            xp = xp.unfold(1 + d, ws[d], 1)                         # [B, *g', Cin, w0, w1, ..]
        # -> [B, *g', w0.., Cin]
        nd = xp.dim()
        perm = [0] + list(range(1, 1 + dd)) + list(range(2 + dd, nd)) + [1 + dd]
        col = xp.permute(perm)
        K = int(np.prod(ws)) * Cin
        out = x.new_empty([B] + [g[d] * self.f[d] for d in range(dd)] + [Cout])
        # phases that share the same window shift along every axis are served by one GEMM
        shifts = [sorted(set(p[3])) for p in self.plans]
        for combo in itertools.product(*shifts):
            sel = [[a for a in range(self.f[d]) if self.plans[d][3][a] == combo[d]] for d in range(dd)]
            sl = (slice(None),) + tuple(slice(combo[d], combo[d] + g[d]) for d in range(dd))
            A = col[sl].reshape(B * int(np.prod(g)), K)
            Wsel = Weff
            for d in range(dd):      # the phases of a shift group are a contiguous range: a view, no index tensor (an
                Wsel = Wsel.narrow(dd + 1 + d, sel[d][0], len(sel[d]))       # H2D copy would break graph capture)
            nph = [len(s) for s in sel]
            Y = (A @ Wsel.reshape(K, -1)).reshape([B] + g + nph + [Cout]) + b
            # scatter the phases of this shift group into the interleaved output
            ov = out.view([B] + [v for d in range(dd) for v in (g[d], self.f[d])] + [Cout])
            # Y dims: B, g0..g_{d-1}, p0..p_{d-1}, C  ->  B, g0, p0, g1, p1, .., C
            perm2 = [0] + [v for d in range(dd) for v in (1 + d, 1 + dd + d)] + [1 + 2 * dd]
            Yp = Y.permute(perm2)
            index = [slice(None)]
            for d in range(dd):
                index += [slice(None), slice(sel[d][0], sel[d][-1] + 1)]
                assert sel[d] == list(range(sel[d][0], sel[d][-1] + 1))
            index.append(slice(None))
            ov[tuple(index)] = Yp
        return out


class UpsampleFast(torch.nn.Module):
    """Drop-in evaluation of an `Upsample` module (same parameters, shared storage) in phase form.
    Input/outputs are channel-FIRST like the reference module so it can replace it anywhere."""

    def __init__(self, net):
        super().__init__()
        self.net = net
        dd = net.conv1.weight.dim() - 2
        self.dd = dd
        sf = [net.up1.scale_factor, net.up2.scale_factor, net.up3.scale_factor]
        self.stages = [PhaseStage(sf[0], 5, int(net.conv1.padding[0]), dd),
                       PhaseStage(sf[1], 3, int(net.conv2.padding[0]), dd),
                       PhaseStage(sf[2], 3, int(net.conv3.padding[0]), dd)]
        for st, conv in zip(self.stages, [net.conv1, net.conv2, net.conv3]):
            for (w, pl, pr, shift, R), f in zip(st.plans, st.f):
                if any(shift[a] > shift[a + 1] for a in range(f - 1)):
                    raise ValueError("unsupported upsample/conv geometry")

    def forward_channel_last(self, x):
        n = self.net
        x = F.leaky_relu(self.stages[0].forward(x, n.conv1.weight, n.conv1.bias), 0.01)
        x = F.leaky_relu(self.stages[1].forward(x, n.conv2.weight, n.conv2.bias), 0.01)
        return self.stages[2].forward(x, n.conv3.weight, n.conv3.bias)

    def forward(self, x):
        return self.forward_channel_last(x.movedim(1, -1)).movedim(-1, 1)


# ---------------------------------------------------------------------------------------------------
# HIP path for the CIFAR-shaped net: 2x2 latent grid -> 8x8 -> 16x16 -> 32x32, 128 -> 64 -> 64 -> 16
# ---------------------------------------------------------------------------------------------------
_CONST_CACHE = {}


def _dev_const(key, dev, make):
    """small constant tensors are uploaded once per device (an H2D copy is not allowed inside graph capture)."""
    k = (key, str(dev))
    if k not in _CONST_CACHE:
        _CONST_CACHE[k] = make().to(dev)
    return _CONST_CACHE[k]


def _stage1_maps(dev, dtype):
    """My[y, s, k] = 1 iff kernel tap k of output row y (up-sampled 8x8 grid, pad 2) reads source row s."""
    def make():
        M = np.zeros([8, 2, 5], dtype=np.float32)
        for y in range(8):
            for k in range(5):
                u = y + k - 2
                if 0 <= u < 8:
                    M[y, u // 4, k] = 1.0
        return torch.from_numpy(M)
    return _dev_const("stage1_M", dev, make).to(dtype)


def _phase_R(dev, f, k, pad):
    return _dev_const(("R", f, k, pad), dev, lambda: torch.from_numpy(_axis_plan(f, k, pad)[4]))


def hip_path_supported(net, pixel_sizes, upsample_factors, patch, data_dim):
    try:
        ok = (data_dim == 2 and not patch and list(pixel_sizes) == [32, 32] and list(upsample_factors) == [16, 16]
              and isinstance(net.conv1, torch.nn.Conv2d) and tuple(net.conv1.weight.shape) == (64, 128, 5, 5)
              and tuple(net.conv2.weight.shape) == (64, 64, 3, 3) and tuple(net.conv3.weight.shape) == (16, 64, 3, 3)
              and tuple(net.conv1.padding) == (2, 2) and tuple(net.conv2.padding) == (1, 1)
              and tuple(net.conv3.padding) == (1, 1)
              and [float(net.up1.scale_factor), float(net.up2.scale_factor), float(net.up3.scale_factor)] == [4., 2., 2.])
    except (AttributeError, TypeError):
        ok = False
    return ok


class _UpsampleCifarFn(torch.autograd.Function):
    """lpe [B, 512] (channel-last 2x2x128) -> pe [B, 1024, 16].  Stage 1 is one dense GEMM against the
    pre-summed 512 x 4096 weight (hipBLASLt); stages 2 and 3 are the rcb_upconv_* kernels."""

    @staticmethod
    def forward(ctx, lpe, W1, b1, W2, b2, W3, b3, stage1_bf16, pe_bf16):
        from . import ops
        B = lpe.shape[0]
        # effective (phase-form) weights of all three stages in one launch; bf16 stage 1 = bf16 operands and a bf16 z1
        # (stage 2 rounds z1 to bf16 for its MFMA operand anyway)
        Weff1, b1rep, Weff2, Weff3, pack = ops.upconv_weff_build(W1, b1, W2, W3, stage1_bf16)
        if stage1_bf16:
            lpe = lpe.to(torch.bfloat16)
        z1 = torch.addmm(b1rep, lpe, Weff1).view(B, 8, 8, 64)
        h2 = ops.upconv_fwd(z1, Weff2, b2.contiguous(), 8, 64, out_f32=False, preact=True, pack=pack)
        pe = ops.upconv_fwd(h2, Weff3, b3.contiguous(), 16, 16, out_f32=not pe_bf16, linear_bf16=pe_bf16, pack=pack)
        ctx.save_for_backward(lpe, Weff1, z1, Weff2, h2, Weff3, pack)
        return pe.view(B, 1024, 16)

    @staticmethod
    def backward(ctx, dpe):
        from . import ops
        lpe, Weff1, z1, Weff2, h2, Weff3, pack = ctx.saved_tensors
        B = lpe.shape[0]
        need_w = any(ctx.needs_input_grad[1:7])
        dpe = dpe.contiguous().view(B, 32, 32, 16)
        fused3 = need_w and dpe.dtype == torch.bfloat16
        if fused3:      # data and weight gradient of stage 3 from one pass over dpe and h2
            dz2, dWeff3, db3 = ops.upconv_bwd_fused(dpe, Weff3, h2, 16, 16, pack=pack)
        else:
            dz2 = ops.upconv_dgrad(dpe, Weff3, h2, 16, 16, pack=pack)      # bf16 [B,16,16,64]
        dz1, db1_part = ops.upconv_dgrad(dz2, Weff2, z1, 8, 64, preact=True, want_dbias=True, pack=pack)   # [B,8,8,64]
        dz1f = dz1.view(B, 4096)
        dlpe = (dz1f @ Weff1.t()).float() if ctx.needs_input_grad[0] else None
        if not need_w:
            return dlpe, None, None, None, None, None, None, None, None
        if not fused3:
            dWeff3, db3 = ops.upconv_wgrad(h2, dpe, 16, 16)
        dWeff2, db2 = ops.upconv_wgrad(z1, dz2, 8, 64, preact=True)
        dWeff1 = lpe.t() @ dz1f                                            # [512, 4096], dtype of the stage-1 operands
        dW1, dW2, dW3, db1 = ops.upconv_weff_grad(dWeff1, dWeff2, dWeff3, db1_part)
        return dlpe, dW1, db1, dW2, db2, dW3, db3, None, None


def upsample_cifar_hip(net, lpe, stage1_bf16=True, pe_bf16=True):
    """lpe [S, N, 2, 2, 128] -> pe [N, S, 1024, 16] (contiguous) through the HIP phase-conv kernels.
    `stage1_bf16` runs the stage-1 library GEMMs (fwd, dgrad, wgrad) with bf16 operands / fp32 accumulation and keeps
    z1 in bf16.  `pe_bf16` stores pe (and hence its gradient) as bf16: the 16-bit SIREN kernels and the stage-3
    gradient kernels round both to bf16 for their MFMA operands anyway, so the results are bit-identical to fp32
    storage at half the traffic.  The (small) lpe is reordered to INR-major so that pe needs no transpose."""
    S, N = lpe.shape[:2]
    pe = _UpsampleCifarFn.apply(lpe.permute(1, 0, 2, 3, 4).reshape(N * S, 512), net.conv1.weight, net.conv1.bias,
                                net.conv2.weight, net.conv2.bias, net.conv3.weight, net.conv3.bias, bool(stage1_bf16),
                                bool(pe_bf16))
    return pe.view(N, S, 1024, 16)


def phase_form_preferred(data_dim, patch):
    """Measured on MI355X (tools/bench_upsample_presets.py, fwd + bwd): against torch.nn -> MIOpen the torch-level phase
    form is 13x faster on the 3-D video geometry (22 vs 289 ms) and 3.5x faster on many small un-patched signals
    (protein: 3.5 vs 12 ms); on the large stitched 1-D / 2-D grids of the patched audio / Kodak presets MIOpen is 2-3x
    faster.  Used only where the hand-written kernels do not apply."""
    return data_dim == 3 or not patch


def phase_module(net):
    """UpsampleFast wrapper of `net`, built once and cached on the module (shares its parameters); None if the
    geometry has no phase plan."""
    fast = getattr(net, "_rcb_phase_form", None)
    if fast is None:
        try:
            fast = UpsampleFast(net)
        except ValueError:
            fast = False
        object.__setattr__(net, "_rcb_phase_form", fast)       # not a registered sub-module: no parameter duplication
    return fast or None
