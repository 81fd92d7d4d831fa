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
import os
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
        self.hip_weight = True       # GPU: build the window-GEMM weight with rcb_phase_bigweight (False: einsums, for A/B and tests)

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

    def window3(self):
        """True if every phase of every axis reads inside the 3-pixel window around its source pixel (all reference nets)"""
        return all(p[0] + max(p[3]) == 3 and p[1] == 1 and p[2] == 1 for p in self.plans)

    def big_weight(self, W):
        """W [Cout, Cin, *k] -> [3^d * Cin, prod(f) * Cout]: rows (window offsets lexicographic, ci), columns (phases, co);
        phase a of an axis uses the window offsets shift[a] + t.  Differentiable (einsums of W with 0/1 tensors)."""
        dd = self.dd
        Weff = self.eff_weight(W)                                   # [*taps, Cin, *phases, Cout]
        letters = "abcdefghijklmnopqrstuvwxyz"
        tp, wn, ph = letters[0:dd], letters[dd:2 * dd], letters[2 * dd:3 * dd]
        sels = []
        for d, (w, _, _, shift, _) in enumerate(self.plans):
            def make(w=w, shift=shift, f=self.f[d]):
                m = torch.zeros(f, 3, w)
                for a in range(f):
                    for t_ in range(w):
                        m[a, shift[a] + t_, t_] = 1.0
                return m
            sels.append(_dev_const(("win3", self.f[d], self.k, self.pad), W.device, make).to(W.dtype))
        terms = [f"{ph[d]}{wn[d]}{tp[d]}" for d in range(dd)] + [f"{tp}I{ph}O"]
        big = torch.einsum(",".join(terms) + f"->{wn}I{ph}O", *sels, Weff)
        return big.reshape(3 ** dd * W.shape[1], -1)

    def forward_gemm(self, x, W, b, dtype=None):
        """the stage as ONE GEMM over 3^d-pixel windows (all phases at once) + pixel shuffle: x [B, *g, Cin] -> [B, *(f*g), Cout].
        `dtype` (bf16 in the 16-bit modes) is the operand / result type; accumulation is fp32 in the library GEMM."""
        dd, g = self.dd, list(x.shape[1:-1])
        dt = x.dtype if dtype is None else dtype
        nph = int(np.prod(self.f))
        if x.is_cuda and W.dtype == torch.float32 and dt in (torch.bfloat16, torch.float32) and self.window3() and self.hip_weight:
            Wbig = _BigWeightFn.apply(W, self, dt)               # rcb_phase_bigweight: one gather-sum kernel each way
        else:
            Wbig = self.big_weight(W).to(dt)
        y = _WindowGemmFn.apply(x.to(dt), Wbig, b.to(dt).repeat(nph))
        y = y.view([x.shape[0]] + g + self.f + [W.shape[0]])
        perm = [0] + [v for d in range(dd) for v in (1 + d, 1 + dd + d)] + [1 + 2 * dd]
        return y.permute(perm).reshape([x.shape[0]] + [g[d] * self.f[d] for d in range(dd)] + [W.shape[0]])

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


class _BigWeightFn(torch.autograd.Function):
    """PhaseStage.big_weight(W).to(dtype) through rcb_phase_bigweight / _grad (the same 0 / 1 linear map, fp32 sums)"""

    @staticmethod
    def forward(ctx, W, stage, dtype):
        from . import ops
        ctx.stage, ctx.w_shape = stage, tuple(W.shape)
        return ops.phase_bigweight(W.detach(), stage.f, stage.k, stage.pad, dtype)

    @staticmethod
    def backward(ctx, dbig):
        from . import ops
        st = ctx.stage
        return ops.phase_bigweight_grad(dbig.contiguous(), ctx.w_shape, st.f, st.k, st.pad), None, None


def _column_sum(t, dtype):
    """t [rows, C] -> [C]: sum over the rows.  For a tall matrix with few columns torch's reduction runs on ONE 64-thread
    workgroup (0.86 ms for the 3 M x 256 gradient of the audio shard's stage 1): two steps instead -- 2^k row blocks, each
    summed by its own threads, then the blocks -- in a fixed association."""
    rows, parts = t.shape[0], 1
    while parts < 1024 and rows % (2 * parts) == 0 and rows // (2 * parts) >= 64:
        parts *= 2
    if parts == 1:
        return t.sum(0, dtype=dtype)
    return t.view(parts, rows // parts, t.shape[1]).sum(1, dtype=dtype).sum(0)


class _WindowGemmFn(torch.autograd.Function):
    """y[rows, Nout] = cols(x) @ Wbig + bias, cols = the 3^d-pixel window around every grid position (zero halo), channel-last.
    Backward: dW = cols^T dy (fp32), dcols = dy Wbig^T folded back onto the grid by 3^d shifted in-place adds (fp32) --
    instead of autograd's generic window backward (one zero-filled full-size tensor per tap)."""

    @staticmethod
    def forward(ctx, x, Wbig, brep):
        dd = x.dim() - 2
        g = list(x.shape[1:-1])
        offs = list(itertools.product(range(3), repeat=dd))
        hip = x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 8 == 0
        if hip:           # rcb_window_gather: one pass instead of 3^d slice copies
            from . import ops
            cols = ops.window_gather(x.contiguous())
        else:
            xp = F.pad(x, [0, 0] + [1, 1] * dd)
            cols = torch.cat([xp[(slice(None),) + tuple(slice(o[d], o[d] + g[d]) for d in range(dd))] for o in offs], dim=-1)
            cols = cols.reshape(-1, cols.shape[-1])
        ctx.hip = hip
        ctx.save_for_backward(cols, Wbig)
        ctx.geo = (tuple(x.shape), offs)
        return torch.addmm(brep, cols, Wbig)

    @staticmethod
    def backward(ctx, dy):
        cols, Wbig = ctx.saved_tensors
        shape, offs = ctx.geo
        dd, g, cin = len(shape) - 2, list(shape[1:-1]), shape[-1]
        dy = dy.contiguous()
        lowp = dy.dtype in (torch.bfloat16, torch.float16)
        if lowp:
            # the contraction runs over ALL grid positions of the batch (10^5 rows) while the result is small: split K into
            # batched chunks so that the library GEMM fills the chip (one 192 x 32 result = 12 workgroups otherwise)
            rows, parts = cols.shape[0], 1
            while parts < 64 and rows % (2 * parts) == 0 and rows // (2 * parts) >= 2048 and \
                    parts * (cols.shape[1] // 64 + 1) * (dy.shape[1] // 64 + 1) < 512:
                parts *= 2
            if parts > 1:
                dW = torch.bmm(cols.view(parts, rows // parts, -1).transpose(1, 2), dy.view(parts, rows // parts, -1),
                               out_dtype=torch.float32).sum(0).to(Wbig.dtype)
            else:
                dW = torch.mm(cols.t(), dy, out_dtype=torch.float32).to(Wbig.dtype)
        else:
            dW = cols.t() @ dy
        # (the two-step sum only in the 16-bit mode: its association follows the row count, and the fp32 parity mode keeps the
        # single reduction its goldens were made with)
        db = (_column_sum(dy, torch.float32) if lowp else dy.sum(0, dtype=dy.dtype)).to(dy.dtype)
        dx = None
        if ctx.needs_input_grad[0] and ctx.hip:
            from . import ops
            dx = ops.window_fold(dy @ Wbig.t(), shape)      # rcb_window_fold: fp32 sums over the taps in one pass
        elif ctx.needs_input_grad[0]:
            dcols = (dy @ Wbig.t()).view(shape[0], *g, len(offs), cin)
            dxp = torch.zeros([shape[0]] + [v + 2 for v in g] + [cin], device=dy.device,
                              dtype=torch.float32 if lowp else dy.dtype)
            for k, o in enumerate(offs):
                dxp[(slice(None),) + tuple(slice(o[d], o[d] + g[d]) for d in range(dd))] += dcols.select(-2, k)
            dx = dxp[(slice(None),) + tuple(slice(1, 1 + g[d]) for d in range(dd))].to(dy.dtype)
        return dx, dW, db


def _unshuffle(t, nd):
    """[B, 2 g..., C] -> [B * prod(g), 2^nd * C]: the phases of every source pixel side by side (axis 0 most significant),
    i.e. the column order of PhaseStage.big_weight"""
    B, C = t.shape[0], t.shape[-1]
    g = [v // 2 for v in t.shape[1:-1]]
    v = t.view([B] + [u for k in range(nd) for u in (g[k], 2)] + [C])
    perm = [0] + [1 + 2 * k for k in range(nd)] + [2 + 2 * k for k in range(nd)] + [1 + 2 * nd]
    return v.permute(perm).reshape(B * int(np.prod(g)), (2 ** nd) * C)


def _window_wgrad(x_act, dy, stage, W):
    """weight and bias gradient of one (x2, 3, pad 1) stage from its input activations and the gradient of its linear output:
    dWbig = cols(x)^T @ dy over 3^d-pixel windows (rcb_window_gather; the contraction over all grid positions split into
    batched chunks so that the library GEMM fills the chip), mapped back to the conv weight through the transpose of
    PhaseStage.big_weight"""
    from . import ops
    nd = x_act.dim() - 2
    cols = ops.window_gather(x_act)
    dyu = _unshuffle(dy, nd)
    rows, parts = cols.shape[0], 1
    while parts < 64 and rows % (2 * parts) == 0 and rows // (2 * parts) >= 2048 and \
            parts * (cols.shape[1] // 64 + 1) * (dyu.shape[1] // 64 + 1) < 512:
        parts *= 2
    if parts > 1:
        dWbig = torch.bmm(cols.view(parts, rows // parts, -1).transpose(1, 2), dyu.view(parts, rows // parts, -1),
                          out_dtype=torch.float32).sum(0)
    else:
        dWbig = torch.mm(cols.t(), dyu, out_dtype=torch.float32)
    with torch.enable_grad():
        Wd = W.detach().requires_grad_(True)
        (dW,) = torch.autograd.grad(stage.big_weight(Wd), [Wd], dWbig)
    db = dy.reshape(-1, dy.shape[-1]).sum(0, dtype=torch.float32)
    return dW, db


class _PhaseConv23Fn(torch.autograd.Function):
    """stages 2 and 3 of the upsampling net on a grid of any dimension through the direct sub-pixel kernels
    (rcb_phaseconv_fwd / _dgrad): z1 [B, *g, 64] (bf16 PRE-activation of stage 1) -> pe [B, *(4 g), 16] (bf16, linear).
    Activations are stored post-LeakyReLU (x1, h2); the data-gradient kernels multiply by LeakyReLU' = sign of the stored
    value.  Weight gradients: rcb_phaseconv_wgrad (contraction over all positions, both operands read transposed from LDS
    images, deterministic slab sums)."""
    direct_wgrad = True

    @staticmethod
    def forward(ctx, z1, W2, b2, W3, b3, stage2, stage3, post_act=False):
        from . import ops
        # post_act: the input IS x1 (stage 1's direct kernel applies the LeakyReLU itself, _Stage1Direct1dFn); the gradient
        # returned for it is still that of the pre-activation
        x1 = z1.contiguous() if post_act else F.leaky_relu(z1, 0.01).contiguous()
        f2, d2 = ops.phaseconv_pack(W2)
        f3, d3 = ops.phaseconv_pack(W3)
        h2 = ops.phaseconv_fwd(x1, f2, b2, 64, True)
        pe = ops.phaseconv_fwd(h2, f3, b3, 16, False)
        ctx.save_for_backward(x1, h2, d2, d3, W2, W3)
        ctx.stages = (stage2, stage3)
        return pe

    @staticmethod
    def backward(ctx, dpe):
        from . import ops
        x1, h2, d2, d3, W2, W3 = ctx.saved_tensors
        stage2, stage3 = ctx.stages
        dpe = dpe.to(torch.bfloat16).contiguous()
        dh2 = ops.phaseconv_dgrad(dpe, d3, h2)                   # gradient of stage 2's PRE-activation (LeakyReLU' inside)
        dz1 = ops.phaseconv_dgrad(dh2, d2, x1) if ctx.needs_input_grad[0] else None
        dW2 = db2 = dW3 = db3 = None
        if any(ctx.needs_input_grad[1:5]):
            if _PhaseConv23Fn.direct_wgrad:
                dW3, db3 = ops.phaseconv_wgrad(h2, dpe)
                dW2, db2 = ops.phaseconv_wgrad(x1, dh2)
            else:                                      # (A/B: GEMMs over 3^d-pixel windows)
                dW3, db3 = _window_wgrad(h2, dpe, stage3, W3)
                dW2, db2 = _window_wgrad(x1, dh2, stage2, W2)
        return dz1, dW2, db2, dW3, db3, None, None, None


# RCB_STAGE1_DIRECT=0: the window-GEMM form of stage 1 also on 1-D grids (same-box A/B)
STAGE1_DIRECT = os.environ.get("RCB_STAGE1_DIRECT", "1") != "0"


class _Stage1Direct1dFn(torch.autograd.Function):
    """stage 1 of the 1-D net through rcb_stage1_1d_*: x [B, g, 128] (fp32 latent grid, read in place) -> x1 [B, 4 g, 64] (bf16,
    POST-LeakyReLU).  Contract with _PhaseConv23Fn(post_act=True): the gradient arriving here is that of the pre-activation
    (the stage-2 data-gradient kernel multiplies by LeakyReLU' from the sign of x1).  dx comes back in fp32 (no cast pass), the
    weight gradient as fp32 dWbig mapped onto conv1.weight by rcb_phase_bigweight_grad."""

    @staticmethod
    def forward(ctx, x, W, b, stage):
        from . import ops
        x = x.contiguous()
        Wbig = ops.phase_bigweight(W.detach(), stage.f, stage.k, stage.pad, torch.bfloat16)
        x1 = ops.stage1_1d_fwd(x, Wbig, b)
        ctx.save_for_backward(x, Wbig)
        ctx.stage, ctx.w_shape = stage, tuple(W.shape)
        return x1

    @staticmethod
    def backward(ctx, dz):
        from . import ops
        x, Wbig = ctx.saved_tensors
        st = ctx.stage
        dz = dz.contiguous()
        dx = ops.stage1_1d_dgrad(dz, Wbig) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dwbig, db = ops.stage1_1d_wgrad(x, dz)
            dW = ops.phase_bigweight_grad(dwbig, ctx.w_shape, st.f, st.k, st.pad)
        return dx, dW, db, None


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
        self.window_gemm = True      # one GEMM per stage over 3^d-pixel windows (False: one GEMM per window-shift group)
        self.direct = True           # stages 2 and 3 through the direct sub-pixel kernels (rcb_phaseconv_*) where they apply
        self.gemm_dtype = None       # operand / activation type of the window GEMMs (None: the input's)

    def _direct_ok(self, x):
        """stages 2 and 3 are (x2, 3, pad 1) with 64 -> 64 -> 16 channels (every reference net) and the tensors are bf16 GPU"""
        n = self.net
        return (self.direct and x.is_cuda and self.gemm_dtype == torch.bfloat16 and all(f == 2 for st in self.stages[1:] for f in st.f)
                and all(st.k == 3 and st.pad == 1 for st in self.stages[1:]) and n.conv2.weight.shape[:2] == (64, 64)
                and n.conv3.weight.shape[:2] == (16, 64))

    def forward_channel_last(self, x):
        n = self.net
        if self.window_gemm and self.stages[0].window3() and self._direct_ok(x):
            st0 = self.stages[0]
            if (STAGE1_DIRECT and self.dd == 1 and x.dtype == torch.float32 and tuple(n.conv1.weight.shape) == (64, 128, 5)
                    and list(st0.f) == [4] and st0.k == 5 and st0.pad == 2 and n.conv1.bias is not None):
                x1 = _Stage1Direct1dFn.apply(x, n.conv1.weight, n.conv1.bias, st0)
                return _PhaseConv23Fn.apply(x1, n.conv2.weight, n.conv2.bias, n.conv3.weight, n.conv3.bias,
                                            self.stages[1], self.stages[2], True)
            z1 = self.stages[0].forward_gemm(x, n.conv1.weight, n.conv1.bias, torch.bfloat16)
            return _PhaseConv23Fn.apply(z1.contiguous(), n.conv2.weight, n.conv2.bias, n.conv3.weight, n.conv3.bias,
                                        self.stages[1], self.stages[2])
        if self.window_gemm and all(st.window3() for st in self.stages):
            dt = self.gemm_dtype
            x = F.leaky_relu(self.stages[0].forward_gemm(x, n.conv1.weight, n.conv1.bias, dt), 0.01)
            x = F.leaky_relu(self.stages[1].forward_gemm(x, n.conv2.weight, n.conv2.bias, dt), 0.01)
            return self.stages[2].forward_gemm(x, n.conv3.weight, n.conv3.bias, dt)
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


WEIGHT_SIDE_STREAM = None        # see _UpsampleCifarFn.backward


class _UpsampleCifarFn(torch.autograd.Function):
    """lpe [B, 512] (channel-last 2x2x128) -> pe [B, 1024, 16].  Stage 1 is one dense GEMM against the
    pre-summed 512 x 4096 weight (hipBLASLt); stages 2 and 3 are the rcb_upconv_* kernels."""

    @staticmethod
    def forward(ctx, lpe, W1, b1, W2, b2, W3, b3, stage1_bf16, pe_bf16, cache=None, lpe16=None):
        from . import ops
        B = lpe.shape[0]
        # effective (phase-form) weights of all three stages in one launch; bf16 stage 1 = bf16 operands and a bf16 z1
        # (stage 2 rounds z1 to bf16 for its MFMA operand anyway).  `cache`: a dict owned by a caller whose mappings are
        # frozen for its lifetime (TestBNNmodel): built once instead of every step
        key = ("weff", bool(stage1_bf16))
        if cache is not None and key in cache:
            Weff1, b1rep, Weff2, Weff3, pack = cache[key]
        else:
            Weff1, b1rep, Weff2, Weff3, pack = ops.upconv_weff_build(W1, b1, W2, W3, stage1_bf16)
            if cache is not None:
                cache[key] = (Weff1, b1rep, Weff2, Weff3, pack)
        if stage1_bf16:
            # the stage-1 GEMM operand: the producer's bf16 copy (reparam kernel) when there is one, else a cast pass
            lpe = lpe16 if (lpe16 is not None and tuple(lpe16.shape) == tuple(lpe.shape)) else lpe.to(torch.bfloat16)
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
        # WEIGHT_SIDE_STREAM (set by PriorBNNmodel.train when its stream fork is on): the weight-gradient side of the backward
        # (stage-2 weight gradient, stage-1 weight-gradient GEMM, the map back onto the conv weights) runs on a second stream
        # beside the data path (stage-2 data gradient -> stage-1 data-gradient GEMM): the two only share their inputs, and
        # the library GEMMs (160 macro-tiles) leave a third of the chip idle on their own.  Same kernels, same operands:
        # identical results.
        side = WEIGHT_SIDE_STREAM if (need_w and fused3) else None
        if side is not None and side.device != dz2.device:      # (a stream left behind by a model on another device)
            side = None
        main = torch.cuda.current_stream() if side is not None else None
        if side is not None:
            side.wait_stream(main)                       # dz2 is complete
            with torch.cuda.stream(side):
                dWeff2, db2 = ops.upconv_wgrad(z1, dz2, 8, 64, preact=True)
        dz1, db1_part = ops.upconv_dgrad(dz2, Weff2, z1, 8, 64, preact=True, want_dbias=True, pack=pack)   # [B,8,8,64]
        dz1f = dz1.view(B, 4096)
        if side is not None:
            side.wait_stream(main)                       # dz1 / db1_part are complete
            with torch.cuda.stream(side):
                dWeff1 = torch.mm(lpe.t(), dz1f, out_dtype=torch.float32) if dz1f.dtype == torch.bfloat16 else lpe.t() @ dz1f
                dW1, dW2, dW3, db1 = ops.upconv_weff_grad(dWeff1, dWeff2, dWeff3, db1_part)
        dlpe = None
        if ctx.needs_input_grad[0]:      # fp32 result straight from the GEMM (no cast pass)
            dlpe = torch.mm(dz1f, Weff1.t(), out_dtype=torch.float32) if dz1f.dtype == torch.bfloat16 else dz1f @ Weff1.t()
        if side is not None:
            main.wait_stream(side)
            return dlpe, dW1, db1, dW2, db2, dW3, db3, None, None, None, None
        if not need_w:
            return dlpe, None, None, None, None, None, None, None, None, None, None
        if not fused3:
            dWeff3, db3 = ops.upconv_wgrad(h2, dpe, 16, 16)
        dWeff2, db2 = ops.upconv_wgrad(z1, dz2, 8, 64, preact=True)
        # [512, 4096]; bf16 operands: fp32 straight from the GEMM (no rounding of the sum over the batch, and the fp32 form of
        # rcb_upconv_weff_grad is the faster one: 7.7 vs 11.1 us)
        dWeff1 = (torch.mm(lpe.t(), dz1f, out_dtype=torch.float32) if dz1f.dtype == torch.bfloat16 else lpe.t() @ dz1f)
        dW1, dW2, dW3, db1 = ops.upconv_weff_grad(dWeff1, dWeff2, dWeff3, db1_part)
        return dlpe, dW1, db1, dW2, db2, dW3, db3, None, None, None, None


def upsample_cifar_hip(net, lpe, stage1_bf16=True, pe_bf16=True, frozen_cache=None, lpe16=None):
    """lpe [S, N, 2, 2, 128] -> pe [N, S, 1024, 16] (contiguous) through the HIP phase-conv kernels.
    `stage1_bf16` runs the stage-1 library GEMMs (fwd, dgrad, wgrad) with bf16 operands / fp32 accumulation and keeps
    z1 in bf16.  `pe_bf16` stores pe (and hence its gradient) as bf16: the 16-bit SIREN kernels and the stage-3
    gradient kernels round both to bf16 for their MFMA operands anyway, so the results are bit-identical to fp32
    storage at half the traffic.  The (small) lpe is reordered to INR-major so that pe needs no transpose."""
    S, N = lpe.shape[:2]
    pe = _UpsampleCifarFn.apply(lpe.permute(1, 0, 2, 3, 4).reshape(N * S, 512), net.conv1.weight, net.conv1.bias,
                                net.conv2.weight, net.conv2.bias, net.conv3.weight, net.conv3.bias, bool(stage1_bf16),
                                bool(pe_bf16), frozen_cache, lpe16 if S == 1 else None)
    return pe.view(N, S, 1024, 16)


def phase_form_preferred(data_dim, patch):
    """16-bit modes, geometries without hand-written phase-conv kernels: the torch-level phase form (one bf16 GEMM per stage
    over 3^d-pixel windows, rcb_window_gather / _fold around it) beats torch.nn -> MIOpen on every reference geometry
    measured on MI355X (prior-training step, tools/bench_presets.py): video 3-D 2.9 vs 19 ms with the first phase form
    (MIOpen's conv3d: 289 ms fwd + bwd), protein 1.4 vs 3.6 ms, audio (stitched 1-D) 1.09 vs 1.75 ms."""
    return True


def phase_module(net):
    """UpsampleFast wrapper of `net`, built once and cached on the module (shares its parameters); None if the
    geometry has no phase plan."""
    fast = getattr(net, "_rcb_phase_form", None)
    if fast is None:
        try:
            fast = UpsampleFast(net)
            fast.gemm_dtype = torch.bfloat16       # only the 16-bit modes route here: bf16 operands, fp32 accumulation
        except ValueError:
            fast = False
        object.__setattr__(net, "_rcb_phase_form", fast)       # not a registered sub-module: no parameter duplication
    return fast or None


# ---------------------------------------------------------------------------------------------------
# HIP path for the STITCHED 2-D grids of the patched presets (Kodak: 8 x 12 patches -> one 32 x 48 latent grid per photo
# -> 128 x 192 -> 256 x 384 -> 512 x 768).  The rcb_upconv_* kernels work on small images with a ZERO halo; a large image is
# cut into tiles that OVERLAP by one source pixel: tile t holds source rows t (G-1) - 1 .. t (G-1) + G - 2 (row -1 and
# rows >= H are the conv's zero padding), so of its 2G output rows all but the first and the last see their whole 2 x 2
# window inside the tile.  Those 2G - 2 valid rows of consecutive tiles abut exactly; the invalid ring is dropped in the
# forward pass and gets a zero upstream gradient in the backward pass, which makes every sum (data, weight and bias
# gradients) exact: overlapping source pixels collect their gradient from both tiles (fold), nothing is counted twice.
# ---------------------------------------------------------------------------------------------------
class _UpsampleStitched23Fn(torch.autograd.Function):
    """stages 2 and 3 of the upsampling net on a stitched 2-D grid: z1 [n, H, W, 64] (bf16 pre-activation of stage 1)
    -> pe [n, 4H, 4W, 16] (bf16, linear) through the rcb_upconv_* kernels on overlapping tiles (8 x 8 source pixels
    for the 64 -> 64 stage, 16 x 16 for the 64 -> 16 stage); rcb_tile_gather / _crop / _fold move between images and tiles."""

    @staticmethod
    def forward(ctx, z1, W1, b1, W2, b2, W3, b3):
        from . import ops
        n, H, W, _ = z1.shape
        _, _, Weff2, Weff3, pack = ops.upconv_weff_build(W1, b1, W2, W3, True)      # (the stage-1 part is the CIFAR map: unused)
        Ty2, Tx2, Ty3, Tx3 = ops.tile_count(H, 8), ops.tile_count(W, 8), ops.tile_count(2 * H, 16), ops.tile_count(2 * W, 16)
        t1 = ops.tile_gather(z1.contiguous(), Ty2, Tx2, 8, 7, 1, 0)
        h2 = ops.upconv_fwd(t1, Weff2, b2.contiguous(), 8, 64, out_f32=False, preact=True, pack=pack)
        h2_img = ops.tile_crop(h2, n, 2 * H, 2 * W, Ty2, Tx2, 1)
        t2 = ops.tile_gather(h2_img, Ty3, Tx3, 16, 15, 1, 0)
        pe = ops.upconv_fwd(t2, Weff3, b3.contiguous(), 16, 16, out_f32=False, linear_bf16=True, pack=pack)
        ctx.save_for_backward(t1, Weff2, t2, Weff3, pack)
        ctx.geo = (n, H, W, Ty2, Tx2, Ty3, Tx3)
        return ops.tile_crop(pe, n, 4 * H, 4 * W, Ty3, Tx3, 1)

    @staticmethod
    def backward(ctx, dpe):
        from . import ops
        t1, Weff2, t2, Weff3, pack = ctx.saved_tensors
        n, H, W, Ty2, Tx2, Ty3, Tx3 = ctx.geo
        dy3 = ops.tile_gather(dpe.to(torch.bfloat16).contiguous(), Ty3, Tx3, 32, 30, 2, 1)
        dz2_t, dWeff3, db3 = ops.upconv_bwd_fused(dy3, Weff3, t2, 16, 16, pack=pack)
        dz2 = ops.tile_fold(dz2_t, n, 2 * H, 2 * W, Ty3, Tx3, 1)
        dy2 = ops.tile_gather(dz2, Ty2, Tx2, 16, 14, 2, 1)
        dz1_t = ops.upconv_dgrad(dy2, Weff2, t1, 8, 64, preact=True, pack=pack)
        dz1 = ops.tile_fold(dz1_t, n, H, W, Ty2, Tx2, 1)
        dWeff2, db2 = ops.upconv_wgrad(t1, dy2, 8, 64, preact=True)
        zero1 = _dev_const("dweff1_zero", dpe.device, lambda: torch.zeros(512 * 4096, dtype=torch.bfloat16))
        _, dW2, dW3 = ops.upconv_weff_grad(zero1, dWeff2, dWeff3)
        return dz1, None, None, dW2, db2, dW3, db3


# Stitched 2-D grids (Kodak): the direct sub-pixel kernels of any dimension (rcb_phaseconv_*, via phase_module) work on the
# whole grid and need no tile copies -- same-box 0.93 vs 0.98 ms per step for two photos, 1.29 vs 1.35 at width 48 -- so they
# are the default; the overlapping-tile route through the fixed-size CIFAR kernels stays selectable (and tested).
PREFER_TILED_2D = False


def tiled_2d_preferred(net, patch, data_dim):
    return PREFER_TILED_2D and hip_stitched_supported(net, patch, data_dim)


def hip_stitched_supported(net, patch, data_dim):
    try:
        ok = (data_dim == 2 and bool(patch) and isinstance(net.conv1, torch.nn.Conv2d)
              and tuple(net.conv1.weight.shape) == (64, 128, 5, 5) and tuple(net.conv2.weight.shape) == (64, 64, 3, 3)
              and tuple(net.conv3.weight.shape) == (16, 64, 3, 3) and tuple(net.conv1.padding) == (2, 2)
              and tuple(net.conv2.padding) == (1, 1) and tuple(net.conv3.padding) == (1, 1)
              and [float(net.up1.scale_factor), float(net.up2.scale_factor), float(net.up3.scale_factor)] == [4., 2., 2.])
    except (AttributeError, TypeError):
        ok = False
    return ok


def stitched2d_module(net):
    """channel-first callable (drop-in for `net` inside map_lpe_to_inr_inputs) evaluating the upsampling net on a
    stitched 2-D latent grid: stage 1 (x4, 5x5, 128 -> 64) in phase form as bf16 GEMMs on the small latent grid, stages 2
    and 3 through the hand-written phase-conv kernels.  Returns bf16 (the 16-bit SIREN kernels read pe as bf16)."""
    fast = getattr(net, "_rcb_stitched2d", None)
    if fast is None:
        stage1 = PhaseStage(4, 5, 2, 2)

        def fast(z_cf):
            # stage 1 (x4, 5x5, 128 -> 64) on the small latent grid: ONE bf16 GEMM over 3x3 windows against the phase-form
            # weight [1152, 1024] (rcb_window_gather / _fold around it), pixel shuffle -> z1 [n, 4h, 4w, 64] pre-activation
            z1 = stage1.forward_gemm(z_cf.movedim(1, -1), net.conv1.weight, net.conv1.bias, torch.bfloat16)
            pe = _UpsampleStitched23Fn.apply(z1, net.conv1.weight, net.conv1.bias, net.conv2.weight,
                                             net.conv2.bias, net.conv3.weight, net.conv3.bias)
            return pe.movedim(-1, 1)
        object.__setattr__(net, "_rcb_stitched2d", fast)
    return fast
