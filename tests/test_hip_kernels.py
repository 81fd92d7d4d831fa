"""GPU parity tests: every C-ABI kernel against the CPU oracle on seeded inputs (run with -m gpu)."""
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, O, cfg_of, check, load, prior_inputs, regen_noise, t

pytestmark = pytest.mark.gpu

from recombiner_amd import _lib, ops  # noqa: E402
from recombiner_amd.ops import LevelSpec, SirenMeta  # noqa: E402

DEV = "cuda"


def g(x):
    return None if x is None else x.to(DEV).contiguous()


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ---------------------------------------------------------------------------------------------------
# SIREN MLP
# ---------------------------------------------------------------------------------------------------
def _siren_case(F, E, n_hidden, C, P, N, S, seed, hidden=32):
    gen = torch.Generator().manual_seed(seed)
    in0 = F + E
    dims = [in0] + [hidden] * n_hidden + [C]
    D = sum(dims[i + 1] * (dims[i] + 1) for i in range(len(dims) - 1))
    xf = torch.rand(P, F, generator=gen) * 2 - 1
    pe = torch.randn(N * S, P, E, generator=gen) * 0.5
    # weights with the magnitude the A-transform produces for SIREN-initialised latents
    wv = (torch.rand(N * S, D, generator=gen) * 2 - 1) * (np.sqrt(6 / hidden) / 30) * 3.0
    y = torch.rand(N, P, C, generator=gen)
    return dims, D, xf, pe, wv, y


def _oracle_mlp(dims, xf, pe, wv, S):
    """plain fp32 torch: x @ W + b, sin(30 x)  (same layer-vector layout)."""
    G, P = pe.shape[0], pe.shape[1]
    x = torch.cat([xf[None].expand(G, -1, -1), pe], -1)
    lo = 0
    nl = len(dims) - 1
    for l in range(nl):
        n = dims[l + 1] * (dims[l] + 1)
        v = wv[:, lo:lo + n]
        b = v[:, :dims[l + 1]].unsqueeze(1)
        W = v[:, dims[l + 1]:].reshape(G, dims[l], dims[l + 1])
        x = x @ W + b
        if l != nl - 1:
            x = torch.sin(30.0 * x)
        lo += n
    return x


SIREN_CASES = [
    dict(F=16, E=16, n_hidden=3, C=3, P=1024, N=3, S=1),     # cifar
    dict(F=16, E=16, n_hidden=3, C=1, P=800, N=2, S=1),      # audio (P = 25 tiles)
    dict(F=18, E=16, n_hidden=3, C=3, P=192, N=2, S=1),      # video input width 34
    dict(F=16, E=16, n_hidden=3, C=3, P=96, N=2, S=5),       # protein, S = 5 samples
    dict(F=16, E=16, n_hidden=2, C=3, P=100, N=2, S=1),      # ragged P (not a multiple of 32), 2 hidden
    dict(F=16, E=16, n_hidden=1, C=3, P=64, N=1, S=1),
    dict(F=16, E=16, n_hidden=4, C=3, P=64, N=1, S=1),
]


@pytest.mark.parametrize("case", SIREN_CASES)
def test_siren_fwd_bwd_loss(case):
    S, N, P, C = case["S"], case["N"], case["P"], case["C"]
    dims, D, xf, pe, wv, y = _siren_case(seed=1, **case)
    meta = SirenMeta(samples=S, n_pix=P, fourier_dim=case["F"], pe_dim=case["E"], n_hidden=case["n_hidden"], hidden=32,
                     out_dim=C)
    assert meta.d_net == D
    # ---- oracle with autograd
    pe_r, wv_r = pe.clone().requires_grad_(True), wv.clone().requires_grad_(True)
    y_ref = _oracle_mlp(dims, xf, pe_r, wv_r, S)
    tgt = y.repeat_interleave(S, 0)
    scale = 1.0 / (S * P * C)
    loss = ((y_ref - tgt) ** 2).sum() * scale
    loss.backward()
    # ---- forward
    y_hip = ops.siren_fwd(g(xf), g(pe), g(wv), meta)
    assert rel_err(y_hip, y_ref.detach()) < 2e-5
    # ---- fused loss + backward
    sse, dw, dpe = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), scale, meta)
    sse_ref = ((y_ref.detach() - tgt) ** 2).sum((1, 2))
    assert rel_err(sse, sse_ref) < 2e-5
    assert rel_err(dw, wv_r.grad) < 1e-4, rel_err(dw, wv_r.grad)
    assert rel_err(dpe, pe_r.grad) < 1e-4
    # ---- plain backward with an arbitrary dy
    gen = torch.Generator().manual_seed(3)
    dy = torch.randn(N * S, P, C, generator=gen)
    pe_r.grad = None
    wv_r.grad = None
    y_ref2 = _oracle_mlp(dims, xf, pe_r, wv_r, S)
    y_ref2.backward(dy)
    dw2, dpe2 = ops.siren_bwd(g(xf), g(pe), g(wv), g(dy), meta)
    assert rel_err(dw2, wv_r.grad) < 1e-4
    assert rel_err(dpe2, pe_r.grad) < 1e-4
    # determinism: two launches give bitwise identical gradients
    sse_b, dw_b, dpe_b = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), scale, meta)
    assert torch.equal(dw, dw_b) and torch.equal(dpe, dpe_b) and torch.equal(sse, sse_b)


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("case", [SIREN_CASES[0], SIREN_CASES[1], SIREN_CASES[2], SIREN_CASES[3], SIREN_CASES[4]])
def test_siren_16bit_operands(case, prec):
    """bf16-operand / fp32-accumulate MFMA path against the fp32 oracle: bounded relative error
    (operands carry 8 mantissa bits; accumulation, biases, loss and reductions are fp32)."""
    S, N, P, C = case["S"], case["N"], case["P"], case["C"]
    dims, D, xf, pe, wv, y = _siren_case(seed=1, **case)
    meta = SirenMeta(samples=S, n_pix=P, fourier_dim=case["F"], pe_dim=case["E"], n_hidden=case["n_hidden"], hidden=32,
                     out_dim=C, precision=prec)
    pe_r, wv_r = pe.clone().requires_grad_(True), wv.clone().requires_grad_(True)
    y_ref = _oracle_mlp(dims, xf, pe_r, wv_r, S)
    tgt = y.repeat_interleave(S, 0)
    scale = 1.0 / (S * P * C)
    (((y_ref - tgt) ** 2).sum() * scale).backward()
    y_hip = ops.siren_fwd(g(xf), g(pe), g(wv), meta)
    sse, dw, dpe = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), scale, meta)
    e_y, e_w, e_p = rel_err(y_hip, y_ref.detach()), rel_err(dw, wv_r.grad), rel_err(dpe, pe_r.grad)
    e_s = rel_err(sse, ((y_ref.detach() - tgt) ** 2).sum((1, 2)))
    print("prec %d rel err: y %.2e  sse %.2e  dW %.2e  dpe %.2e" % (prec, e_y, e_s, e_w, e_p))
    # the test weights are 3x the SIREN init scale (phases up to ~10 rad), a stress case for short mantissas
    lim = {1: (0.12, 1e-2, 0.12, 0.15), 2: (0.02, 2e-3, 0.02, 0.03)}[prec]
    assert e_y < lim[0] and e_s < lim[1] and e_w < lim[2] and e_p < lim[3]
    sse_b, dw_b, dpe_b = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), scale, meta)
    assert torch.equal(dw, dw_b) and torch.equal(dpe, dpe_b) and torch.equal(sse, sse_b)


@pytest.mark.parametrize("hidden", [32, 64])
@pytest.mark.parametrize("prec", [1, 2])
def test_siren_16bit_operands_at_init_scale(prec, hidden):
    """the same comparison at the scale training actually runs at (SIREN initialisation, sqrt(6 / hidden) / 30), with limits
    set at 3x the error MEASURED on MI355X (tools/siren_err_probe.py, round 4: bf16 y 6.4e-3, sse 1.9e-4, dW 2.9e-3, dpe
    1.0e-2; f16 y 9.0e-4, sse 2.0e-5, dW 4.4e-4, dpe 1.2e-3 -- worst over the cases and widths 32 / 48 / 64): a kernel that
    lost a factor of three in accuracy fails here, where the 3x-weight stress case above would still pass."""
    lim = {1: (2.0e-2, 6e-4, 9e-3, 3.0e-2), 2: (2.7e-3, 6e-5, 1.4e-3, 3.6e-3)}[prec]
    worst = np.zeros(4)
    for case in [SIREN_CASES[0], SIREN_CASES[3]] + ([SIREN_CASES[1], SIREN_CASES[2], SIREN_CASES[4]] if hidden == 32 else []):
        S, N, P, C = case["S"], case["N"], case["P"], case["C"]
        dims, D, xf, pe, wv, y = _siren_case(seed=1, hidden=hidden, **case)
        wv = wv / 3.0                                             # _siren_case draws 3x the init scale
        meta = SirenMeta(samples=S, n_pix=P, fourier_dim=case["F"], pe_dim=case["E"], n_hidden=case["n_hidden"], hidden=hidden,
                         out_dim=C, precision=prec)
        pe_r, wv_r = pe.clone().requires_grad_(True), wv.clone().requires_grad_(True)
        y_ref = _oracle_mlp(dims, xf, pe_r, wv_r, S)
        tgt = y.repeat_interleave(S, 0)
        scale = 1.0 / (S * P * C)
        (((y_ref - tgt) ** 2).sum() * scale).backward()
        y_hip = ops.siren_fwd(g(xf), g(pe), g(wv), meta)
        sse, dw, dpe = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), scale, meta)
        e = np.array([rel_err(y_hip, y_ref.detach()), rel_err(sse, ((y_ref.detach() - tgt) ** 2).sum((1, 2))),
                      rel_err(dw, wv_r.grad), rel_err(dpe, pe_r.grad)])
        worst = np.maximum(worst, e)
    print("init scale, prec %d, width %d: worst rel err y %.2e sse %.2e dW %.2e dpe %.2e (limits %s)" % (prec, hidden, *worst, lim))
    assert (worst < np.array(lim)).all(), (worst, lim)


@pytest.mark.parametrize("F", [16, 18])
@pytest.mark.parametrize("prec", [1, 2])
def test_siren_bf16_pe_storage_is_bit_identical(prec, F):
    """pe / dpe held as bf16 arrays: the 16-bit kernels round pe to the operand type on load and the consumers of
    dpe round it to bf16 on load, so bf16 storage must reproduce fp32 storage exactly (dpe: after that rounding).
    F = 18 is the video geometry (34 inputs, the Fourier half not a multiple of 8)."""
    case = dict(F=F, E=16, n_hidden=3, C=3, P=1000, N=3, S=2)       # ragged last tile
    dims, D, xf, pe, wv, y = _siren_case(seed=11, **case)
    meta = SirenMeta(2, 1000, F, 16, 3, 32, 3, precision=prec)
    pe16 = g(pe).bfloat16()
    pe32 = pe16.float()
    scale = 1.0 / (2 * 1000 * 3)
    y32 = ops.siren_fwd(g(xf), pe32, g(wv), meta)
    y16 = ops.siren_fwd(g(xf), pe16, g(wv), meta)
    assert torch.equal(y32, y16)
    s32, w32, d32 = ops.siren_loss_bwd(g(xf), pe32, g(wv), g(y), scale, meta)
    # (bf16 pe + a shared grid select the variant that loads both input halves as 16-bit rows and, with one input block
    # (F % 8 == 0), keeps the cosines unrounded in fp32 -- same forward, slightly MORE accurate gradients; the storage statement
    # is about one arithmetic, so that variant is switched off here and compared separately below)
    os.environ["RCB_SIREN_NO_XF16"] = "1"
    try:
        s16, w16, d16 = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta)
    finally:
        del os.environ["RCB_SIREN_NO_XF16"]
    assert d16.dtype == torch.bfloat16
    assert torch.equal(s32, s16) and torch.equal(w32, w16) and torch.equal(d32.bfloat16(), d16)
    if F % 8 != 0:
        # two input blocks (F = 18): the 16-bit-row variant (xf copy padded to 24 features, in the operand format) keeps the packed
        # cosines -- the same arithmetic on the same operand bits
        s16b, w16b, d16b = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta)
        assert torch.equal(s16b, s16) and torch.equal(w16b, w16) and torch.equal(d16b, d16)
        x16 = ops.xf_bf16(g(xf), prec)
        assert x16.shape == (1000, 24) and x16.dtype == (torch.bfloat16 if prec == 1 else torch.float16) and not x16[:, F:].any()
        with pytest.raises(ops.RcbError):                              # an unpadded copy is refused, not read out of bounds
            ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta, xf16=g(xf).to(x16.dtype))
    else:
        s16b, w16b, d16b = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta)
        assert torch.equal(s16b, s16)                                     # the forward pass is the same arithmetic
        assert not torch.equal(w16b, w16)                                 # (the variant really ran)
        assert rel_err(w16b, w16) < 5e-3 and rel_err(d16b.float(), d16.float()) < 1e-2
        # unrounded cosines: at least as close to the fp32 oracle as the packed-bf16 ones
        po, wo = pe16.float().cpu().requires_grad_(True), wv.clone().requires_grad_(True)
        yo = _oracle_mlp(dims, xf, po, wo, 2)
        loss = (scale * (yo - y.repeat_interleave(2, 0)) ** 2).sum()
        gw, = torch.autograd.grad(loss, [wo])
        assert rel_err(w16b, gw) <= rel_err(w16, gw) * 1.05, (rel_err(w16b, gw), rel_err(w16, gw))
    dy = 1e-3 * torch.randn(6, 1000, 3, device=DEV)       # gradient-sized: unit-scale dy overflows the f16 operands
    wb32, db32 = ops.siren_bwd(g(xf), pe32, g(wv), dy, meta)
    wb16, db16 = ops.siren_bwd(g(xf), pe16, g(wv), dy, meta)
    assert torch.isfinite(wb32).all() and torch.isfinite(db32).all()
    assert torch.equal(wb32, wb16) and torch.equal(db32.bfloat16(), db16)
    with pytest.raises(ops.RcbError):                                  # the fp32 kernel has no bf16-storage variant
        ops.siren_fwd(g(xf), pe16, g(wv), SirenMeta(2, 1000, F, 16, 3, 32, 3, precision=0))


@pytest.mark.parametrize("prec", [0, 2])
def test_siren_model_scale_weights(prec):
    """the regime of the real model: effective weights (h_w @ A) ~1e-4, biases ~1e-2.  f16 operands
    must not lose them (they are carried scaled by 2^10 inside the kernel)."""
    case = dict(F=16, E=16, n_hidden=3, C=3, P=256, N=3, S=1)
    dims, D, xf, pe, wv, y = _siren_case(seed=7, **case)
    gen = torch.Generator().manual_seed(8)
    wv = (torch.rand(3, D, generator=gen) * 2 - 1) * 3e-4
    lo = 0
    for l in range(4):                                   # biases are the first `out` entries of each layer vector
        n = dims[l + 1] * (dims[l] + 1)
        wv[:, lo:lo + dims[l + 1]] = (torch.rand(3, dims[l + 1], generator=gen) * 2 - 1) * 0.03
        lo += n
    meta = SirenMeta(1, 256, 16, 16, 3, 32, 3, precision=prec)
    pe_r, wv_r = pe.clone().requires_grad_(True), wv.clone().requires_grad_(True)
    y_ref = _oracle_mlp(dims, xf, pe_r, wv_r, 1)
    scale = 1.0 / (256 * 3)
    (((y_ref - y) ** 2).sum() * scale).backward()
    sse, dw, dpe = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), scale, meta)
    e_w, e_p = rel_err(dw, wv_r.grad), rel_err(dpe, pe_r.grad)
    print("prec %d model-scale: dW %.2e dpe %.2e" % (prec, e_w, e_p))
    tol = 1e-4 if prec == 0 else 5e-3
    assert e_w < tol and e_p < tol


# hidden widths 48 and 64 (BASELINE.json configs "width-48" / "width-64 fp16": siren_mlp_wide.hip, 16-bit modes only)
WIDE_CASES = [
    dict(F=16, E=16, n_hidden=3, C=3, P=1000, N=2, S=1, hidden=48),    # kodak geometry, ragged last tile
    dict(F=16, E=16, n_hidden=3, C=3, P=256, N=2, S=2, hidden=64),
    dict(F=18, E=16, n_hidden=3, C=3, P=192, N=2, S=1, hidden=64),     # video: input width 34 (two input blocks)
    dict(F=16, E=16, n_hidden=3, C=1, P=800, N=2, S=1, hidden=64),     # audio
]


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("case", WIDE_CASES)
def test_siren_wide_16bit_operands(case, prec):
    """the wide kernel against the fp32 torch restatement of the same MLP: forward, fused loss + backward, plain
    backward; error bounds of the 16-bit operand types as for width 32; bitwise deterministic."""
    S, N, P, C, W = case["S"], case["N"], case["P"], case["C"], case["hidden"]
    dims, D, xf, pe, wv, y = _siren_case(seed=21, **case)
    meta = SirenMeta(samples=S, n_pix=P, fourier_dim=case["F"], pe_dim=case["E"], n_hidden=case["n_hidden"], hidden=W,
                     out_dim=C, precision=prec)
    assert meta.d_net == D
    pe_r, wv_r = pe.clone().requires_grad_(True), wv.clone().requires_grad_(True)
    y_ref = _oracle_mlp(dims, xf, pe_r, wv_r, S)
    tgt = y.repeat_interleave(S, 0)
    scale = 1.0 / (S * P * C)
    (((y_ref - tgt) ** 2).sum() * scale).backward()
    y_hip = ops.siren_fwd(g(xf), g(pe), g(wv), meta)
    sse, dw, dpe = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), scale, meta)
    e_y, e_w, e_p = rel_err(y_hip, y_ref.detach()), rel_err(dw, wv_r.grad), rel_err(dpe, pe_r.grad)
    e_s = rel_err(sse, ((y_ref.detach() - tgt) ** 2).sum((1, 2)))
    print("width %d prec %d rel err: y %.2e  sse %.2e  dW %.2e  dpe %.2e" % (W, prec, e_y, e_s, e_w, e_p))
    lim = {1: (0.12, 1e-2, 0.12, 0.15), 2: (0.02, 2e-3, 0.02, 0.03)}[prec]
    assert e_y < lim[0] and e_s < lim[1] and e_w < lim[2] and e_p < lim[3]
    # per-layer check of the weight gradient (a mis-indexed small layer would hide behind the global maximum)
    lo = 0
    for l in range(len(dims) - 1):
        n = dims[l + 1] * (dims[l] + 1)
        assert rel_err(dw[:, lo:lo + n], wv_r.grad[:, lo:lo + n]) < lim[2] * 1.5, l
        lo += n
    sse_b, dw_b, dpe_b = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), scale, meta)
    assert torch.equal(dw, dw_b) and torch.equal(dpe, dpe_b) and torch.equal(sse, sse_b)
    # plain backward with an arbitrary (gradient-sized) dy
    gen = torch.Generator().manual_seed(3)
    dy = 1e-3 * torch.randn(N * S, P, C, generator=gen)
    pe_r.grad = None
    wv_r.grad = None
    _oracle_mlp(dims, xf, pe_r, wv_r, S).backward(dy)
    dw2, dpe2 = ops.siren_bwd(g(xf), g(pe), g(wv), g(dy), meta)
    assert rel_err(dw2, wv_r.grad) < lim[2] and rel_err(dpe2, pe_r.grad) < lim[3]


@pytest.mark.parametrize("case", WIDE_CASES + [dict(F=16, E=16, n_hidden=2, C=3, P=100, N=2, S=2, hidden=40)])
def test_siren_other_widths_fp32(case):
    """fp32 parity mode at hidden widths other than 32 (siren_mlp_generic.hip): forward, fused loss + backward and plain
    backward against the fp32 torch restatement at fp32 tolerance; bitwise deterministic."""
    S, N, P, C, W = case["S"], case["N"], case["P"], case["C"], case["hidden"]
    dims, D, xf, pe, wv, y = _siren_case(seed=31, **case)
    meta = SirenMeta(samples=S, n_pix=P, fourier_dim=case["F"], pe_dim=case["E"], n_hidden=case["n_hidden"], hidden=W,
                     out_dim=C, precision=0)
    assert meta.d_net == D
    pe_r, wv_r = pe.clone().requires_grad_(True), wv.clone().requires_grad_(True)
    y_ref = _oracle_mlp(dims, xf, pe_r, wv_r, S)
    tgt = y.repeat_interleave(S, 0)
    scale = 1.0 / (S * P * C)
    (((y_ref - tgt) ** 2).sum() * scale).backward()
    y_hip = ops.siren_fwd(g(xf), g(pe), g(wv), meta)
    sse, dw, dpe = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), scale, meta)
    e_y, e_w, e_p = rel_err(y_hip, y_ref.detach()), rel_err(dw, wv_r.grad), rel_err(dpe, pe_r.grad)
    e_s = rel_err(sse, ((y_ref.detach() - tgt) ** 2).sum((1, 2)))
    print("width %d fp32 rel err: y %.2e  sse %.2e  dW %.2e  dpe %.2e" % (W, e_y, e_s, e_w, e_p))
    assert e_y < 3e-5 and e_s < 3e-6 and e_w < 3e-5 and e_p < 3e-5
    lo = 0
    for l in range(len(dims) - 1):
        n = dims[l + 1] * (dims[l] + 1)
        assert rel_err(dw[:, lo:lo + n], wv_r.grad[:, lo:lo + n]) < 5e-5, l
        lo += n
    sse_b, dw_b, dpe_b = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), scale, meta)
    assert torch.equal(dw, dw_b) and torch.equal(dpe, dpe_b) and torch.equal(sse, sse_b)
    gen = torch.Generator().manual_seed(3)
    dy = 1e-3 * torch.randn(N * S, P, C, generator=gen)
    pe_r.grad = None
    wv_r.grad = None
    _oracle_mlp(dims, xf, pe_r, wv_r, S).backward(dy)
    dw2, dpe2 = ops.siren_bwd(g(xf), g(pe), g(wv), g(dy), meta)
    assert rel_err(dw2, wv_r.grad) < 3e-5 and rel_err(dpe2, pe_r.grad) < 3e-5


@pytest.mark.parametrize("W", [48, 64])
def test_siren_wide_model_scale_weights_f16(W):
    """the regime of the real model (effective weights ~1e-4, biases ~1e-2) in f16: 5e-3 of the fp32 restatement, which
    no indexing error in the fragment builders, the image transposes or the layer-wise reduction could meet"""
    case = dict(F=16, E=16, n_hidden=3, C=3, P=288, N=3, S=1, hidden=W)
    dims, D, xf, pe, wv, y = _siren_case(seed=7, **case)
    gen = torch.Generator().manual_seed(8)
    wv = (torch.rand(3, D, generator=gen) * 2 - 1) * 3e-4
    lo = 0
    for l in range(4):
        n = dims[l + 1] * (dims[l] + 1)
        wv[:, lo:lo + dims[l + 1]] = (torch.rand(3, dims[l + 1], generator=gen) * 2 - 1) * 0.03
        lo += n
    meta = SirenMeta(1, 288, 16, 16, 3, W, 3, precision=2)
    pe_r, wv_r = pe.clone().requires_grad_(True), wv.clone().requires_grad_(True)
    y_ref = _oracle_mlp(dims, xf, pe_r, wv_r, 1)
    scale = 1.0 / (288 * 3)
    (((y_ref - y) ** 2).sum() * scale).backward()
    y_hip = ops.siren_fwd(g(xf), g(pe), g(wv), meta)
    sse, dw, dpe = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), scale, meta)
    e_y, e_w, e_p = rel_err(y_hip, y_ref.detach()), rel_err(dw, wv_r.grad), rel_err(dpe, pe_r.grad)
    print("width %d model-scale f16: y %.2e dW %.2e dpe %.2e" % (W, e_y, e_w, e_p))
    assert e_y < 5e-3 and e_w < 5e-3 and e_p < 5e-3
    lo = 0
    for l in range(4):
        n = dims[l + 1] * (dims[l] + 1)
        assert rel_err(dw[:, lo:lo + n], wv_r.grad[:, lo:lo + n]) < 5e-3, l
        lo += n


def test_siren_fp32_with_different_hidden_widths():
    """the reference builds its INR from any hidden_dims list (prior_model.py:84-85): widths that differ from layer to layer
    run on the plain-FMA fp32 kernel (rcb_siren_desc.hidden_dims) and match the fp32 restatement and its autograd"""
    gen = torch.Generator().manual_seed(3)
    F, E, C, P, N = 16, 16, 3, 70, 3
    hid = (24, 40, 16)
    dims = [F + E] + list(hid) + [C]
    D = sum(dims[i + 1] * (dims[i] + 1) for i in range(len(dims) - 1))
    xf = torch.rand(P, F, generator=gen) * 2 - 1
    pe = torch.randn(N, P, E, generator=gen) * 0.5
    wv = (torch.rand(N, D, generator=gen) * 2 - 1) * 0.05
    y = torch.rand(N, P, C, generator=gen)
    meta = SirenMeta(1, P, F, E, 3, max(hid), C, hidden_dims=hid)
    assert meta.d_net == D
    out = ops.siren_fwd(g(xf), g(pe), g(wv), meta)
    wv_r, pe_r = wv.clone().requires_grad_(True), pe.clone().requires_grad_(True)
    ref = _oracle_mlp(dims, xf, pe_r, wv_r, 1)
    assert rel_err(out, ref) < 1e-5             # (fp32 on both sides; sin(30 z) amplifies the last-bit differences of z)
    loss = ((ref - y) ** 2).sum() / (P * C)
    loss.backward()
    sse, dw, dpe = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), 1.0 / (P * C), meta)
    assert rel_err(sse.sum() / (P * C), loss.detach()) < 1e-5
    assert rel_err(dw, wv_r.grad) < 2e-5 and rel_err(dpe, pe_r.grad) < 2e-5
    with pytest.raises(ops.RcbError):                                     # the 16-bit kernels take one width
        ops.siren_fwd(g(xf), g(pe), g(wv), SirenMeta(1, P, F, E, 3, max(hid), C, precision=1, hidden_dims=hid))


def test_siren_wide_bf16_pe_storage_and_split_output():
    """width 64: bf16-stored pe / dpe reproduce fp32 storage exactly, and the bf16 copy of the gradient the epilogue
    writes (rcb_siren_desc.dw_bf16) is the rounded fp32 gradient (as for width 32)"""
    case = dict(F=16, E=16, n_hidden=3, C=3, P=200, N=3, S=1, hidden=64)
    dims, D, xf, pe, wv, y = _siren_case(seed=13, **case)
    meta = SirenMeta(1, 200, 16, 16, 3, 64, 3, precision=1)
    pe16 = g(pe).bfloat16()
    pe32 = pe16.float()
    scale = 1.0 / (200 * 3)
    assert torch.equal(ops.siren_fwd(g(xf), pe32, g(wv), meta), ops.siren_fwd(g(xf), pe16, g(wv), meta))
    s32, w32, d32 = ops.siren_loss_bwd(g(xf), pe32, g(wv), g(y), scale, meta)
    s16, w16, d16, wcopy = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta, want_bf16=True)
    assert d16.dtype == torch.bfloat16
    assert torch.equal(s32, s16) and torch.equal(w32, w16) and torch.equal(d32.bfloat16(), d16)
    assert ops.siren_wide_layers(meta) == (2, 64 * 65)
    assert wcopy.shape == w32.shape and wcopy.stride(0) % 8 == 0 and torch.equal(wcopy, w32.bfloat16())


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("case", WIDE_CASES)
def test_siren_wide_16bit_input_rows_equal_fp32_input_rows(case, prec):
    """width 48 / 64, both operand formats, one and two input blocks: the instances that load both input halves as 16-bit rows
    (a copy of the coordinate grid in the operand format, rows padded to 8 features; bf16 pe converted to f16 in registers) run
    the arithmetic of the instances that load fp32 rows and round them per tile -- bit for bit."""
    S, P, C, W = case["S"], case["P"], case["C"], case["hidden"]
    dims, D, xf, pe, wv, y = _siren_case(seed=23, **case)
    meta = SirenMeta(samples=S, n_pix=P, fourier_dim=case["F"], pe_dim=case["E"], n_hidden=case["n_hidden"], hidden=W,
                     out_dim=C, precision=prec)
    pe16, scale = g(pe).bfloat16(), 1.0 / (S * P * C)
    os.environ["RCB_SIREN_NO_XF16"] = "1"
    try:
        ref = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta, want_bf16=True)
    finally:
        del os.environ["RCB_SIREN_NO_XF16"]
    out = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta, want_bf16=True, xf16=ops.xf_bf16(g(xf), prec))
    for a_, b_ in zip(out, ref):
        assert torch.equal(a_, b_)
    assert torch.isfinite(out[1]).all() and float(out[1].abs().max()) > 0


@pytest.mark.parametrize("width,prec", [(32, 1), (32, 2), (48, 1), (64, 2)])
def test_siren_pixel_chunks_equal_whole_row_launch(width, prec):
    """rcb_siren_desc.pixel_chunks (launches with few rows): outputs and the pe gradient are bit-identical to the one-
    workgroup-per-row launch (disjoint pixels); the weight gradient and the loss are the same sums in a different
    (fixed) association; the reduction also emits the split-bf16 form."""
    case = dict(F=16, E=16, n_hidden=3, C=3, P=1000, N=3, S=2, hidden=width)
    dims, D, xf, pe, wv, y = _siren_case(seed=17, **case)
    meta = SirenMeta(2, 1000, 16, 16, 3, width, 3, precision=prec)
    assert ops.siren_pixel_chunks(6, meta) == 1                       # opt-in (keeps bitwise batch invariance by default)
    ops.PIXEL_CHUNKS_AUTO = True
    try:
        assert ops.siren_pixel_chunks(6, meta) == (4 if width == 32 else 1) and ops.siren_pixel_chunks(4096, meta) == 1
        assert ops.siren_pixel_chunks(6, SirenMeta(2, 1000, 16, 16, 3, 32, 3, precision=0)) == 1
    finally:
        ops.PIXEL_CHUNKS_AUTO = False
    pe16 = g(pe).bfloat16()
    scale = 1.0 / 6000
    y1 = ops.siren_fwd(g(xf), pe16, g(wv), meta, pixel_chunks=1)
    for c in (2, 4, 32):
        assert torch.equal(ops.siren_fwd(g(xf), pe16, g(wv), meta, pixel_chunks=c), y1)
    s1, w1, d1, sp1 = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta, want_bf16=True, pixel_chunks=1)
    assert torch.equal(sp1, w1.bfloat16())
    for c in (2, 5):
        sc, wc, dc, spc = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta, want_bf16=True, pixel_chunks=c)
        assert torch.equal(dc, d1)
        assert rel_err(sc, s1) < 1e-6 and rel_err(wc, w1) < 2e-6
        assert torch.equal(spc, wc.bfloat16())                                   # (written by the chunk reduction)
        s2, w2, d2 = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta, pixel_chunks=c)
        assert torch.equal(w2, wc) and torch.equal(s2, sc)                       # deterministic
    dy = 1e-3 * torch.randn(6, 1000, 3, device=DEV)
    wb1, db1 = ops.siren_bwd(g(xf), pe16, g(wv), dy, meta, pixel_chunks=1)
    wb4, db4 = ops.siren_bwd(g(xf), pe16, g(wv), dy, meta, pixel_chunks=4)
    assert torch.equal(db4, db1) and rel_err(wb4, wb1) < 2e-6
    with pytest.raises(ops.RcbError):
        ops.siren_fwd(g(xf), pe16, g(wv), meta, pixel_chunks=33)                 # more chunks than 32-pixel tiles
    with pytest.raises(ops.RcbError):
        ops.siren_fwd(g(xf), g(pe), g(wv), SirenMeta(2, 1000, 16, 16, 3, 32, 3, precision=0), pixel_chunks=2)


def test_siren_per_inr_coordinates_and_strided_rows():
    """xf given per INR ([N,P,F]) and wvec rows with a padded stride."""
    case = dict(F=16, E=16, n_hidden=3, C=3, P=64, N=3, S=1)
    dims, D, xf, pe, wv, y = _siren_case(seed=5, **case)
    meta = SirenMeta(1, 64, 16, 16, 3, 32, 3)
    xfn = torch.stack([xf, xf * 0.5, -xf])
    ref = torch.stack([_oracle_mlp(dims, xfn[i], pe[i:i + 1], wv[i:i + 1], 1)[0] for i in range(3)])
    wpad = torch.zeros(3, D + 5)
    wpad[:, :D] = wv
    y_hip = ops.siren_fwd(g(xfn), g(pe), g(wpad)[:, :D], meta)
    assert rel_err(y_hip, ref) < 2e-5


def _stitch(t, S, pn, ps):
    """[N * S rows (row = n * S + s), P, E] -> the stitched channel-last grids [S * nd, *(pn_i * ps_i), E] of utils.map_lpe_to_inr_inputs"""
    G, P, E = t.shape
    dd, N = len(pn), G // S
    nd = N // int(np.prod(pn))
    v = t.view(nd, *pn, S, *ps, E)                                   # [nd, pn.., S, ps.., E]
    perm = [1 + dd, 0] + [x for i in range(dd) for x in (1 + i, 2 + dd + i)] + [2 + 2 * dd]
    return v.permute(perm).reshape(S * nd, *[a * b for a, b in zip(pn, ps)], E).contiguous()


def _unstitch(t, S, pn, ps):
    dd, E = len(pn), t.shape[-1]
    nd = t.shape[0] // S
    v = t.view(S, nd, *[x for i in range(dd) for x in (pn[i], ps[i])], E)
    perm = [1] + [2 + 2 * i for i in range(dd)] + [0] + [3 + 2 * i for i in range(dd)] + [2 + 2 * dd]
    return v.permute(perm).reshape(nd * int(np.prod(pn)) * S, int(np.prod(ps)), E).contiguous()


@pytest.mark.parametrize("pn,ps,S,width,prec", [((2, 3), (8, 16), 2, 32, 1), ((3,), (200,), 1, 32, 1), ((2, 1, 2), (4, 6, 10), 1, 64, 2),
                                                ((2, 2), (16, 16), 1, 48, 1), ((2, 2, 2), (2, 8, 8), 3, 32, 2)])
def test_siren_stitched_pe_layout_is_bit_identical(pn, ps, S, width, prec):
    """rcb_siren_desc.pe_grid_dims: pe / dpe addressed inside the upsampling net's stitched output grids (the patched presets,
    reference utils.py:60-116) == the same launch on patch-major copies, bit for bit (forward, loss + backward)."""
    P, per = int(np.prod(ps)), int(np.prod(pn))
    N = 2 * per
    case = dict(F=16, E=16, n_hidden=3, C=3, P=P, N=N, S=S, hidden=width)
    dims, D, xf, pe, wv, y = _siren_case(seed=41, **case)
    meta = SirenMeta(S, P, 16, 16, 3, width, 3, precision=prec)
    pe16 = g(pe).bfloat16()                                        # [N * S, P, E]
    lay = ops.PeLayout(pn, ps)
    st = _stitch(pe16, S, list(pn), list(ps))
    assert torch.equal(_unstitch(st, S, list(pn), list(ps)), pe16)
    assert torch.equal(ops.siren_fwd(g(xf), st, g(wv), meta, pe_layout=lay), ops.siren_fwd(g(xf), pe16, g(wv), meta))
    scale = 1.0 / (S * P * 3)
    # (bit identity holds inside one kernel family: the stitched layouts run on the workgroup kernel, so the patch-major launch
    # is pinned to it as well; the two families against each other: test_siren_wave_family_against_the_workgroup_family)
    lib = _lib.load()
    fam = lib.rcb_debug_siren_wave_tiles(0)
    try:
        s1, w1, d1 = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta)
        s2, w2, d2 = ops.siren_loss_bwd(g(xf), st, g(wv), g(y), scale, meta, pe_layout=lay)
    finally:
        lib.rcb_debug_siren_wave_tiles(fam)
    assert d2.shape == st.shape and torch.equal(s1, s2) and torch.equal(w1, w2)
    assert torch.equal(_unstitch(d2, S, list(pn), list(ps)), d1)
    with pytest.raises(ops.RcbError):
        ops.siren_loss_bwd(g(xf), st.float(), g(wv), g(y), scale, SirenMeta(S, P, 16, 16, 3, 32, 3, precision=0), pe_layout=lay)


@pytest.mark.parametrize("N,S,P,dpe", [(37, 1, 1024, True), (6, 5, 1024, True), (9, 1, 96, False), (3, 2, 4096, True)])
def test_siren_wave_family_against_the_workgroup_family(N, S, P, dpe):
    """The two width-32 bf16 loss / backward families (siren_mlp_wave.hip: one wave per row, siren_mlp_bf16.hip: one workgroup
    per row; rcb_debug_siren_wave_tiles) evaluate the same bf16 products with fp32 accumulation: loss to 1e-5, weight gradient
    to 5e-4 of its largest entry (tile order, bias as two bf16 halves), input gradient to one bf16 ulp (1 / 128 of the largest
    entry), each also against the fp32 kernel at the 16-bit-operand tolerance; the wave family is bitwise reproducible, takes
    pixel-chunked launches and rows that are not on 16-byte boundaries, and leaves what it has no instance for (rows of
    partial tiles, fp32 pe) to the workgroup family."""
    lib = _lib.load()
    case = dict(F=16, E=16, n_hidden=3, C=3, P=P, N=N, S=S)
    dims, D, xf, pe, wv, y = _siren_case(seed=23, **case)
    wv = wv / 3.0
    meta = SirenMeta(S, P, 16, 16, 3, 32, 3, precision=1)
    pe16, xf16 = g(pe).bfloat16(), ops.xf_bf16(g(xf))
    scale = 1.0 / (S * P * 3)
    ref = ops.siren_loss_bwd(g(xf), pe16.float(), g(wv), g(y), scale, SirenMeta(S, P, 16, 16, 3, 32, 3, precision=0), want_dpe=dpe)
    out = {}
    fam = lib.rcb_debug_siren_wave_tiles(-1)
    try:
        for f in (0, 1):
            lib.rcb_debug_siren_wave_tiles(2 * f)          # 2: the wave family whatever the number of rows
            out[f] = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta, want_dpe=dpe, want_bf16=True, xf16=xf16)
        again = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta, want_dpe=dpe, want_bf16=True, xf16=xf16)
        ch = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), scale, meta, want_dpe=dpe, xf16=xf16, pixel_chunks=2 if P >= 128 else 1)
    finally:
        lib.rcb_debug_siren_wave_tiles(fam)
    for a_, b_ in zip(out[1], again):                      # bitwise reproducible
        assert (a_ is None and b_ is None) or torch.equal(a_, b_)
    wmax = float(ref[1].abs().max())
    for f in (0, 1):
        s_, w_, d_, w16 = out[f]
        torch.testing.assert_close(s_, ref[0], rtol=3e-3, atol=0)
        assert float((w_ - ref[1]).abs().max()) < 2e-2 * wmax
        assert float((w16.float() - w_).abs().max()) <= 2.0 ** -8 * wmax
        if dpe:
            assert float((d_.float() - ref[2]).abs().max()) < 3e-2 * float(ref[2].abs().max())
    torch.testing.assert_close(out[1][0], out[0][0], rtol=1e-5, atol=0)
    assert float((out[1][1] - out[0][1]).abs().max()) < 5e-4 * wmax
    if dpe:
        assert float((out[1][2].float() - out[0][2].float()).abs().max()) <= 2.0 ** -7 * float(out[0][2].float().abs().max())
    # chunked launch (partials summed in chunk order) == the unchunked one to fp32 rounding
    assert float((ch[1] - out[1][1]).abs().max()) < 1e-5 * wmax and torch.allclose(ch[0], out[1][0], rtol=1e-5)


def test_siren_rejects_bad_arguments():
    meta = SirenMeta(1, 64, 16, 16, 3, 32, 3)
    xf = torch.zeros(64, 16, device=DEV)
    pe = torch.zeros(2, 64, 16, device=DEV)
    with pytest.raises(ops.RcbError):
        ops.siren_fwd(xf, pe, torch.zeros(2, 100, device=DEV), meta)             # wrong d_net
    with pytest.raises(ops.RcbError):
        ops.siren_fwd(xf.cpu(), pe, torch.zeros(2, meta.d_net, device=DEV), meta)  # CPU tensor: no fallback
    bad = SirenMeta(1, 64, 16, 16, 3, 72, 3)
    with pytest.raises(ops.RcbError):
        ops.siren_fwd(xf, pe, torch.zeros(2, bad.d_net, device=DEV), bad)          # fp32 mode: widths up to 64
    bad = SirenMeta(1, 64, 16, 16, 3, 40, 3, precision=1)
    with pytest.raises(ops.RcbError):
        ops.siren_fwd(xf, pe, torch.zeros(2, bad.d_net, device=DEV), bad)          # unsupported width


# ---------------------------------------------------------------------------------------------------
# reparam / posterior
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["patch2d", "patch1d", "patch3d", "cifar"])
def test_reparam_fwd_and_posterior_grads(name):
    d = load(f"prior_{name}.npz")
    cfg, geo, n, p, A, up, X, Y, pri = prior_inputs(d)
    D = geo.d_net
    S = 3
    torch.manual_seed(9)
    noise = O.Noise()
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    hl = pr.get("h_loc")
    hw_ref = O.sample_latent_weights(geo, pr["loc"], O.st(pr["log_scale"]), hl,
                                     O.st(pr["h_log_scale"]) if hl is not None else None, pr.get("hh_loc"),
                                     O.st(pr["hh_log_scale"]) if hl is not None else None, S, noise)
    levels = [LevelSpec(g(p["loc"]), g(p["log_scale"]), D, n)]
    if geo.patch:
        m2, m3 = geo.level_maps(n)
        levels.append(LevelSpec(g(p["h_loc"]), g(p["h_log_scale"]), D, n, row_map=m2.numpy()))
        levels.append(LevelSpec(g(p["hh_loc"]), g(p["hh_log_scale"]), D, n, row_map=m3.numpy()))
    eps = [g(e) for e in noise.drawn]
    out = ops.reparam_fwd(levels, eps, S)
    assert rel_err(out, hw_ref.detach()) < 1e-6     # un-fused fp32; softplus differs by <= 1 ulp CPU vs GPU
    # gradients of sum(out * G) + beta*KL
    gen = torch.Generator().manual_seed(2)
    Gm = torch.randn(n, S, D, generator=gen)
    beta = 0.37
    keys = [("loc", "log_scale", 0, 1)] + ([("h_loc", "h_log_scale", 4, 5), ("hh_loc", "hh_log_scale", 6, 7)] if geo.patch else [])
    kl = sum(O.gauss_kl_elem(pr[a], O.st(pr[b]), pri[i], pri[j]).sum() for a, b, i, j in keys)
    ((hw_ref * Gm).sum() + beta * kl).backward()
    for lv, (a, b, i, j), e in zip(levels, keys, eps):
        gl, gs = ops.posterior_bwd(lv, g(pri[i]), g(pri[j]), False, beta, g(Gm), e, S, want_grads=True)
        assert rel_err(gl, pr[a].grad) < 2e-5
        assert rel_err(gs, pr[b].grad) < 2e-5


@pytest.mark.parametrize("N,D", [(3, 3267), (5, 512), (1, 7)])
def test_reparam_flat_fast_path_equals_generic(N, D):
    """one level / one sample / no maps takes the 16-byte flat kernel: same bits as the generic kernel (forced here
    by an identity column map), including the tail when N * D is not a multiple of 4."""
    gen = torch.Generator().manual_seed(12)
    loc = g(torch.randn(N, D, generator=gen))
    ls = g(torch.randn(N, D, generator=gen) * 3 - 2)
    eps = g(torch.randn(N, 1, D, generator=gen))
    fast = ops.reparam_fwd([LevelSpec(loc, ls, D, N)], [eps], 1)
    slow = ops.reparam_fwd([LevelSpec(loc, ls, D, N, col_map=np.arange(D))], [eps], 1)
    assert torch.equal(fast, slow)
    ref = loc.double() + torch.nn.functional.softplus(ls.double()) / 6 * eps[:, 0].double()
    assert rel_err(fast[:, 0], ref) < 1e-6


def test_siren_bf16_gradient_copy_equals_rounded_dw():
    """rcb_siren_desc.dw_bf16: the epilogue's bf16 copy of the gradient == dwvec rounded to bf16, the fp32 results untouched"""
    case = dict(F=16, E=16, n_hidden=3, C=3, P=256, N=5, S=1)
    dims, D, xf, pe, wv, y = _siren_case(seed=13, **case)
    meta = SirenMeta(1, 256, 16, 16, 3, 32, 3, precision=1)
    assert ops.siren_wide_layers(meta) == (3, 1056)
    sse, dw, dpe, d16 = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), 1.0 / 768, meta, want_bf16=True)
    sse2, dw2, dpe2 = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), 1.0 / 768, meta)
    assert torch.equal(dw, dw2) and torch.equal(dpe, dpe2) and torch.equal(sse, sse2)
    assert d16.shape == (5, 3267) and d16.stride(0) == 3272 and torch.equal(d16, dw.bfloat16())
    with pytest.raises(ops.RcbError):
        ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), 1.0 / 768, SirenMeta(1, 256, 16, 16, 3, 32, 3), want_bf16=True)


@pytest.mark.parametrize("rows,sizes", [(37, [1056, 1056, 1056, 99]), (300, [1056, 1056, 1056, 99]), (4096, [1056, 1056, 1056, 99]),
                                        (192, [1584, 2352, 2352, 147]), (130, [420, 441, 63]), (5, [33]), (129, [64, 8, 40])])
def test_atrans_kernels_forward_and_data_gradient(rows, sizes):
    """atrans.hip (K2, prior_model.py:173-174): out[:, lo:hi] = x[:, lo:hi] @ A[l] and x @ A[l]^T for all layers in one launch.
    Asymmetric random mappings (a transposed or mis-indexed operand cannot pass); held to the fp64 product with the operands
    each number of terms keeps: 3 -> both to ~16 bits, 2 -> the mapping rounded to bf16, 1 -> both rounded to bf16."""
    gen = torch.Generator().manual_seed(rows * 7 + len(sizes))
    D = sum(sizes)
    cum = np.cumsum([0] + sizes)
    slices = [(int(cum[i]), int(cum[i + 1])) for i in range(len(sizes))]
    big = torch.randn(rows + 3, D + 5, generator=gen) * 0.03          # rows with a stride that is not a multiple of 4 floats
    x = g(big)[1:rows + 1, 2:D + 2]
    A = [g(torch.randn(n, n, generator=gen) / n ** 0.5) for n in sizes]
    for terms in (3, 2, 1):
        tr = ops.ATransform(slices, DEV, terms=terms)
        tr.prepare(A)
        w = tr.forward(x, torch.full((rows, D), float("nan"), device=DEV))
        outbuf = torch.full((rows + 2, D + 3), 7.0, device=DEV)        # strided output: nothing outside the view is touched
        dh = tr.dgrad(x, outbuf[1:rows + 1, 1:D + 1])
        assert torch.isfinite(w).all() and torch.isfinite(dh).all()
        assert float(outbuf[0].min()) == 7.0 and float(outbuf[-1].min()) == 7.0 and float(outbuf[:, 0].min()) == 7.0 \
            and float(outbuf[:, D + 1:].min()) == 7.0
        for (a, b), m in zip(slices, A):
            xs = x[:, a:b].double() if terms >= 2 else x[:, a:b].bfloat16().double()
            md = m.double() if terms == 3 else m.bfloat16().double()
            tol = 2e-5 if terms >= 2 else 1e-5
            assert rel_err(w[:, a:b], xs @ md) < tol, (terms, a, b)
            assert rel_err(dh[:, a:b], xs @ md.t()) < tol, (terms, a, b)
            if terms == 3 and b - a >= 64:
                assert rel_err(x[:, a:b].bfloat16().float() @ m.bfloat16().float(), x[:, a:b].double() @ m.double()) > 3e-4
    # bitwise reproducible (no atomics, fixed decomposition)
    tr = ops.ATransform(slices, DEV, terms=2)
    tr.prepare(A)
    w1 = tr.forward(x, torch.empty(rows, D, device=DEV))
    w2 = tr.forward(x, torch.empty(rows, D, device=DEV))
    assert torch.equal(w1, w2)
    # identity mappings with an asymmetric probe: the forward returns bf16-exact inputs untouched, column for column
    eye = [g(torch.eye(n)) for n in sizes]
    tr.prepare(eye)
    xi = g(torch.arange(rows * D, dtype=torch.float32).reshape(rows, D) % 251 - 125.0)
    assert torch.equal(tr.forward(xi, torch.empty(rows, D, device=DEV)), xi)
    assert torch.equal(tr.dgrad(xi, torch.empty(rows, D, device=DEV)), xi)


@pytest.mark.parametrize("rows,sizes", [(300, [1056, 1056, 1056, 99]), (4096, [1056, 1056, 1056, 99]), (192, [1584, 2352, 2352, 147]),
                                        (70, [64, 8, 40]), (1000, [33])])
def test_atrans_weight_gradient(rows, sizes):
    """dA[l] = h_w[:, lo:hi]^T @ dw[:, lo:hi]: the batched GEMM on bf16 operands (the producers' copies, or cast on the
    spot) agrees with the fp64 product of the rounded operands; the narrow layers' kernel (fp32 MFMA: exact products,
    fixed order) with the fp64 product of the fp32 ones."""
    gen = torch.Generator().manual_seed(rows + 11)
    D = sum(sizes)
    cum = np.cumsum([0] + sizes)
    slices = [(int(cum[i]), int(cum[i + 1])) for i in range(len(sizes))]
    h = g(torch.randn(rows, D, generator=gen) * 0.03)
    dw = g(torch.randn(rows, D, generator=gen) * 1e-3)
    tr = ops.ATransform(slices, DEV, terms=2)
    ld = (D + 7) // 8 * 8
    h16 = torch.zeros(rows, ld, device=DEV, dtype=torch.bfloat16)[:, :D]
    h16.copy_(h)
    gA = tr.wgrad(h, dw, h16, None, True)               # one operand handed over, the other cast inside
    gA32 = tr.wgrad(h, dw, None, None, False)
    big = max(sizes)
    for (a, b), g16, g32 in zip(slices, gA, gA32):
        ref = h[:, a:b].double().t() @ dw[:, a:b].double()
        assert rel_err(g32, ref) < 1e-5
        if b - a == big and big >= 256:
            ref16 = h[:, a:b].bfloat16().double().t() @ dw[:, a:b].bfloat16().double()
            assert rel_err(g16, ref16) < 1e-5
        else:
            assert torch.equal(g16, g32)
    g2 = tr.wgrad(h, dw, h16, None, True)
    assert all(torch.equal(x1, x2) for x1, x2 in zip(gA, g2))            # reproducible
    # narrow kernel = a chain of fmaf in row order per (wave, slab): bit-exact restatement for one small case
    if rows <= 300:
        n = sizes[-1]
        if n <= 256:
            hh, dd = h[:, slices[-1][0]:].double(), dw[:, slices[-1][0]:].double()
            assert rel_err(gA[-1], hh.t() @ dd) < 2e-6


def test_philox_noise_stream_and_fused_reparam():
    """in-kernel noise: N(0,1) statistics, a pure function of (seed, stream, step, index), and the fused kernel equals
    the explicit-noise reparam on the materialised stream bit for bit (including a tail that is not a multiple of 4)."""
    n = 1 << 22
    e = ops.philox_normal(n, 1234, 0, 7).double()
    assert abs(float(e.mean())) < 3e-3 and abs(float(e.var()) - 1) < 5e-3
    assert abs(float((e ** 4).mean()) - 3) < 0.05 and abs(float((e ** 3).mean())) < 0.02
    assert 4.5 < float(e.abs().max()) < 6.5                              # 24-bit uniforms: tails reach ~5.9 sigma
    assert abs(float((e[:-1] * e[1:]).mean())) < 3e-3 and abs(float((e[:-4] * e[4:]).mean())) < 3e-3
    assert torch.equal(ops.philox_normal(1000, 1234, 0, 7), ops.philox_normal(n, 1234, 0, 7)[:1000])   # index-addressed
    for other in ((1235, 0, 7), (1234, 1, 7), (1234, 0, 8)):            # seed / stream / step all change the values
        o = ops.philox_normal(4096, *other)
        assert float((o == e[:4096].float()).float().mean()) < 0.01 and abs(float((o.double() * e[:4096]).mean())) < 0.08
    step = torch.tensor([7], device=DEV, dtype=torch.int64)
    assert torch.equal(ops.philox_normal(4096, 1234, 0, step), e[:4096].float())                     # device-side step
    lvb = LevelSpec(g(torch.randn(5, 3267) * 0.05), g(torch.full((5, 3267), -4.0)), 3267, 5)
    ob, eb, o16 = ops.reparam_rng(lvb, 99, 0, step, want_bf16=True)                                  # bf16 copy on the side
    o2, e2 = ops.reparam_rng(lvb, 99, 0, step)
    assert torch.equal(ob, o2) and torch.equal(eb, e2) and o16.stride(0) == 3272 and torch.equal(o16, ob.view(5, 3267).bfloat16())
    gen = torch.Generator().manual_seed(31)
    for N, D in ((5, 3267), (4, 512), (3, 7)):
        loc = g(torch.randn(N, D, generator=gen))
        ls = g(torch.randn(N, D, generator=gen) * 2 - 3)
        lv = LevelSpec(loc, ls, D, N)
        out, eps = ops.reparam_rng(lv, 99, 3, step)
        ref_eps = ops.philox_normal(N * D, 99, 3, 7).view(N, 1, D)
        assert torch.equal(eps, ref_eps)
        assert torch.equal(out, ops.reparam_fwd([lv], [ref_eps], 1))
    step += 1
    out2, eps2 = ops.reparam_rng(lv, 99, 3, step)
    assert not torch.equal(eps2, eps)                                    # next step: fresh noise


def test_adam_matches_torch():
    gen = torch.Generator().manual_seed(4)
    p0 = torch.randn(1000, generator=gen)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=2e-4)
    p = g(p0.clone())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        gr = torch.randn(1000, generator=gen) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=gen)))
        ref.grad = gr.clone()
        opt.step()
        ops.adam_flat(p, g(gr), m, v, ops.adam_cfg(2e-4, step))
    np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-9)


def test_adam_multi_equals_adam_flat():
    """one launch over a ragged list of tensors == one adam_flat launch per tensor, bit for bit"""
    gen = torch.Generator().manual_seed(9)
    sizes = [1056 * 1056, 99 * 99, 73728, 64, 1, 1023, 1024, 1025, 0, 36864, 16]
    ps = [g(torch.randn(n, generator=gen)) for n in sizes]
    gs = [g(torch.randn(n, generator=gen) * 1e-3) for n in sizes]
    ms, vs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    ps2, ms2, vs2 = [p.clone() for p in ps], [m.clone() for m in ms], [v.clone() for v in vs]
    for step in (1, 2, 3):
        cfg = ops.adam_cfg(2e-4, step)
        ops.adam_multi(ps, gs, ms, vs, cfg)
        for p, gr, m, v in zip(ps2, gs, ms2, vs2):
            if p.numel():
                ops.adam_flat(p, gr, m, v, cfg)
    for a, b in zip(ps + ms + vs, ps2 + ms2 + vs2):
        assert torch.equal(a, b)
    # more tensors than one launch holds: chunked
    many = [g(torch.randn(37, generator=gen)) for _ in range(20)]
    many2 = [p.clone() for p in many]
    gm = [g(torch.randn(37, generator=gen)) for _ in range(20)]
    z = lambda: [torch.zeros(37, device=DEV) for _ in range(20)]  # noqa: E731
    m1, v1, m2, v2 = z(), z(), z(), z()
    ops.adam_multi(many, gm, m1, v1, ops.adam_cfg(1e-3, 1))
    for p, gr, m, v in zip(many2, gm, m2, v2):
        ops.adam_flat(p, gr, m, v, ops.adam_cfg(1e-3, 1))
    assert all(torch.equal(a, b) for a, b in zip(many, many2))


def test_posterior_adam_step_matches_autograd_adam():
    """fused grad + KL + Adam == torch autograd + torch.optim.Adam on the same loss."""
    gen = torch.Generator().manual_seed(6)
    n, D, S = 5, 300, 2
    loc = 0.02 * torch.randn(n, D, generator=gen)
    ls = -4 + 0.5 * torch.randn(n, D, generator=gen)
    pl = 0.01 * torch.randn(D, generator=gen)
    ps = 0.02 + 0.01 * torch.rand(D, generator=gen)
    eps = torch.randn(n, S, D, generator=gen)
    Gm = torch.randn(n, S, D, generator=gen) * 1e-3
    rl, rs = loc.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    opt = torch.optim.Adam([rl, rs], lr=2e-4)
    dl, ds = g(loc.clone()), g(ls.clone())
    lv = LevelSpec(dl, ds, D, n)
    state = {k: torch.zeros_like(dl) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
    for step in range(1, 4):
        out = rl[:, None] + O.st(rs)[:, None] * eps
        loss = (out * Gm).sum() + 1e-3 * O.gauss_kl_elem(rl, O.st(rs), pl, ps).sum()
        opt.zero_grad()
        loss.backward()
        opt.step()
        ops.posterior_bwd(lv, g(pl), g(ps), False, 1e-3, g(Gm), g(eps), S, adam=ops.adam_cfg(2e-4, step), state=state)
    np.testing.assert_allclose(dl.cpu().numpy(), rl.detach().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ds.cpu().numpy(), rs.detach().numpy(), rtol=1e-5, atol=1e-7)


def test_posterior_flat_fast_path_equals_generic_kernel():
    """plain case (one sample, no maps): the 16-byte flat kernel must update loc / log_scale / Adam state bit for bit
    like the generic kernel (forced here through an identity column map) and accumulate the same KL."""
    gen = torch.Generator().manual_seed(17)
    n, D = 8, 3267
    loc = 0.02 * torch.randn(n, D, generator=gen)
    ls = -4 + 0.5 * torch.randn(n, D, generator=gen)
    pl = 0.01 * torch.randn(D, generator=gen)
    ps = 0.02 + 0.01 * torch.rand(D, generator=gen)
    eps = g(torch.randn(n, 1, D, generator=gen))
    Gm = g(torch.randn(n, 1, D, generator=gen) * 1e-3)
    res = []
    for col_map in (None, np.arange(D)):
        dl, ds = g(loc.clone()), g(ls.clone())
        lv = LevelSpec(dl, ds, D, n, col_map=col_map)
        state = {k: torch.zeros_like(dl) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
        slots = torch.zeros(1024, device=DEV, dtype=torch.int64)          # fixed point, ops.KL_FX units per nat
        for step in (1, 2, 3):
            ops.posterior_bwd(lv, g(pl), g(ps), False, 1e-3, Gm, eps, 1, adam=ops.adam_cfg(2e-4, step), state=state,
                              kl_accum=slots)
        res.append((dl, ds, state, float(slots.sum()) / ops.KL_FX))
    (l0, s0, st0, k0), (l1, s1, st1, k1) = res
    assert torch.equal(l0, l1) and torch.equal(s0, s1)
    assert all(torch.equal(st0[k], st1[k]) for k in st0)
    assert k0 == pytest.approx(k1, rel=1e-9) and k0 > 0       # (each workgroup's partial sum is rounded to 2^-24 nats once)


def test_three_level_sample_with_noise_drawn_in_the_kernel():
    """rcb_reparam_hier_rng_fwd: every level's noise is the Philox stream rcb_philox_normal materialises (streams 0, 2, 3 at the
    element index of the [N, D] arrays, group offset honoured), written out for the posterior update, and the sample equals
    rcb_reparam_fwd on that noise bit for bit; a launch over a row range with its group offset draws the rows' own noise."""
    gen = torch.Generator().manual_seed(31)
    n, D = 24, 3201                              # n * D is a multiple of 4; rows are not 16-byte aligned
    maps = [None, np.repeat(np.arange(n // 4), 4).astype(np.int32), np.repeat(np.arange(n // 12), 12).astype(np.int32)]
    levels = []
    for mp in maps:
        rows = n if mp is None else int(mp.max()) + 1
        levels.append(LevelSpec(g(0.02 * torch.randn(rows, D, generator=gen)), g(-4 + 0.5 * torch.randn(rows, D, generator=gen)), D, n,
                                row_map=mp))
    assert ops.hier_rng_eligible(levels)
    step = torch.tensor([7], device=DEV, dtype=torch.long)
    seed, streams = 0x1234_5678_9ABC_DEF0, (0, 2, 3)
    eps = [torch.empty(n, 1, D, device=DEV) for _ in levels]
    out = ops.reparam_hier_rng(levels, eps, seed, streams, step)
    want_eps = [ops.philox_normal(n * D, seed, st_, step).view(n, 1, D) for st_ in streams]
    for e, w in zip(eps, want_eps):
        assert torch.equal(e, w)
    assert torch.equal(out, ops.reparam_fwd(levels, want_eps, 1))
    assert float(torch.stack([e.flatten() for e in eps]).std()) == pytest.approx(1.0, abs=0.02)
    assert not torch.equal(eps[0], eps[1]) and not torch.equal(eps[1], eps[2])
    # rows 8 .. 23 as a launch of their own (a shard): the same noise and the same sample
    r0 = 8
    sub = [LevelSpec(levels[0].loc[r0:], levels[0].log_scale[r0:], D, n - r0)]
    for lv, mp in zip(levels[1:], maps[1:]):
        sub.append(LevelSpec(lv.loc, lv.log_scale, D, n - r0, row_map=mp[r0:].copy()))
    eps_s = [torch.empty(n - r0, 1, D, device=DEV) for _ in levels]
    out_s = ops.reparam_hier_rng(sub, eps_s, seed, streams, step, group_offset=ops.rng_group_offset(r0, D))
    assert torch.equal(out_s, out[r0:]) and all(torch.equal(a_, b_[r0:]) for a_, b_ in zip(eps_s, eps))


def test_coarse_level_of_the_test_time_layout_indexed_by_produced_column():
    """test-time layout of a coarse level (column permutation, encode mask, per-group beta, S = 5, members behind a row map): with
    rcb_level_bwd.col_map the threads are indexed by the produced column (coalesced gradient gathers) -- the same bits as with the
    threads indexed by the parameter column (rcb_debug_generic_kernels_only drops the hint)."""
    from recombiner_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(37)
    n, D, S, G, group = 24, 3201, 5, 300, 6
    rows = n // group
    row_map = np.repeat(np.arange(rows), group).astype(np.int32)
    perm = torch.randperm(D, generator=gen).numpy()
    loc = 0.02 * torch.randn(rows, D, generator=gen)
    ls = -4 + 0.5 * torch.randn(rows, D, generator=gen)
    mask = (torch.rand(rows, D, generator=gen) < 0.3).float()
    samp = 0.02 * torch.randn(rows, D, generator=gen)
    pl = 0.01 * torch.randn(D, generator=gen)
    pls = -3 + 0.2 * torch.randn(D, generator=gen)
    gidx = torch.sort(torch.randint(0, G, (D,), generator=gen)).values.int()
    beta = torch.rand(rows, G, generator=gen) * 1e-3
    eps = g(torch.randn(n, S, D, generator=gen))
    Gm = g(torch.randn(n, S, D, generator=gen) * 1e-3)
    outs = []
    for generic in (1, 0):
        lib.rcb_debug_generic_kernels_only(generic)
        try:
            dl, ds = g(loc.clone()), g(ls.clone())
            lv = LevelSpec(dl, ds, D, n, row_map=row_map, col_map=perm, enc_sample=g(samp), enc_mask=g(mask))
            state = {k: torch.zeros_like(dl) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
            slots = torch.zeros(1024, device=DEV, dtype=torch.int64)
            for step in (1, 2):
                ops.posterior_bwd(lv, g(pl), g(pls), True, 1.0, Gm, eps, S, beta=g(beta), group_idx=g(gidx), n_groups=G,
                                  adam=ops.adam_cfg(2e-4, step), state=state, kl_accum=slots)
            outs.append((dl, ds, state, float(slots[:-1].sum()) / ops.KL_FX))
        finally:
            lib.rcb_debug_generic_kernels_only(0)
    (l0, s0, st0, k0), (l1, s1, st1, k1) = outs
    assert torch.equal(l0, l1) and torch.equal(s0, s1) and all(torch.equal(st0[k], st1[k]) for k in st0)
    assert k0 == pytest.approx(k1, rel=1e-9) and k0 > 0
    assert float((l0 - g(loc)).abs().max()) > 0


def test_level_one_of_the_patched_test_time_layout_with_presummed_samples():
    """level 1 of a patched preset at test time: per-column row permutation, group-order column map, encode mask, per-group beta,
    S = 5, and a row too long for the LDS-staged kernel (the lpe is part of it): the sums over the samples are formed in a contiguous
    pass first (rcb_level_bwd.sample_sum_ws) -- the same bits as the generic kernel's 2 S gathers per parameter."""
    from recombiner_amd import _lib
    from recombiner_amd.test_model import _column_row_perms
    lib = _lib.load()
    gen = torch.Generator().manual_seed(41)
    n, D, S, G = 12, 9601, 5, 500
    perm = torch.randperm(D, generator=gen).numpy()
    rperm = _column_row_perms(n, D)
    loc = 0.02 * torch.randn(n, D, generator=gen)
    ls = -4 + 0.5 * torch.randn(n, D, generator=gen)
    mask = (torch.rand(n, D, generator=gen) < 0.3).float()
    samp = 0.02 * torch.randn(n, D, generator=gen)
    pl = 0.01 * torch.randn(D, generator=gen)
    pls = -3 + 0.2 * torch.randn(D, generator=gen)
    gidx = torch.sort(torch.randint(0, G, (D,), generator=gen)).values.int()
    beta = torch.rand(n, G, generator=gen) * 1e-3
    eps = g(torch.randn(n, S, D, generator=gen))
    Gm = g(torch.randn(n, S, D, generator=gen) * 1e-3)
    outs = []
    for generic in (1, 0):
        lib.rcb_debug_generic_kernels_only(generic)
        try:
            dl, ds = g(loc.clone()), g(ls.clone())
            lv = LevelSpec(dl, ds, D, n, row_perm=rperm, col_map=perm, enc_sample=g(samp), enc_mask=g(mask))
            state = {k: torch.zeros_like(dl) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
            slots = torch.zeros(1024, device=DEV, dtype=torch.int64)
            for step in (1, 2):
                ops.posterior_bwd(lv, g(pl), g(pls), True, 1.0, Gm, eps, S, beta=g(beta), group_idx=g(gidx), n_groups=G,
                                  adam=ops.adam_cfg(2e-4, step), state=state, kl_accum=slots)
            outs.append((dl, ds, state, float(slots[:-1].sum()) / ops.KL_FX))
        finally:
            lib.rcb_debug_generic_kernels_only(0)
    (l0, s0, st0, k0), (l1, s1, st1, k1) = outs
    assert torch.equal(l0, l1) and torch.equal(s0, s1) and all(torch.equal(st0[k], st1[k]) for k in st0)
    assert k0 == k1 and k0 > 0                   # (the same kernel, the same workgroups: the KL log is the same too)
    assert float((l0 - g(loc)).abs().max()) > 0


def test_three_gathered_levels_sampled_from_packed_mu_sigma_records():
    """test-time sample of a patched preset (three levels, group-order column maps, per-column row permutations on levels 1 and 2,
    encode masks, S = 5): with the (mu, sigma) records packed in a contiguous pass first (rcb_level.mu_sigma_ws) the sampler
    gathers one 8-byte record per level and element -- the same samples bit for bit as the generic kernel's 4-byte gathers."""
    from recombiner_amd import _lib
    from recombiner_amd.test_model import _column_row_perms
    lib = _lib.load()
    gen = torch.Generator().manual_seed(43)
    n, D, S = 24, 3301, 5
    maps = [None, np.repeat(np.arange(n // 4), 4).astype(np.int32), np.repeat(np.arange(n // 12), 12).astype(np.int32)]
    outs = []
    for generic in (1, 0):
        lib.rcb_debug_generic_kernels_only(generic)
        try:
            gen.manual_seed(43)
            levels, eps = [], []
            for li, mp in enumerate(maps):
                rows = n if mp is None else int(mp.max()) + 1
                levels.append(LevelSpec(g(0.02 * torch.randn(rows, D, generator=gen)), g(-4 + 0.5 * torch.randn(rows, D, generator=gen)), D, n,
                                        row_map=mp, row_perm=_column_row_perms(rows, D) if li < 2 else None,
                                        col_map=torch.randperm(D, generator=gen).numpy(),
                                        enc_sample=g(0.02 * torch.randn(rows, D, generator=gen)),
                                        enc_mask=g((torch.rand(rows, D, generator=gen) < 0.3).float())))
                eps.append(g(torch.randn(n, S, D, generator=gen)))
            outs.append(ops.reparam_fwd(levels, eps, S))
        finally:
            lib.rcb_debug_generic_kernels_only(0)
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().max()) > 0


def test_four_column_member_kernel_equals_generic_kernel():
    """training update of a coarse level (members behind a row map, one sample, rows of 3201 floats: only 4-byte aligned): the
    four-columns-per-thread kernel against the generic one (rcb_debug_generic_kernels_only): identical bits; the KL log differs
    by the rounding of its per-workgroup partial sums only."""
    from recombiner_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(29)
    D = 3201
    for n, group in ((36, 4), (36, 12), (1024, 4)):       # members per coarse row; few rows (64-thread workgroups) and many (256)
        rows = n // group
        row_map = np.repeat(np.arange(rows), group).astype(np.int32)
        loc = 0.02 * torch.randn(rows, D, generator=gen)
        ls = -4 + 0.5 * torch.randn(rows, D, generator=gen)
        pl = 0.01 * torch.randn(D, generator=gen)
        pls = -3 + 0.2 * torch.randn(D, generator=gen)
        eps = g(torch.randn(n, 1, D, generator=gen))
        Gm = g(torch.randn(n, 1, D, generator=gen) * 1e-3)
        outs = []
        for generic in (1, 0):
            lib.rcb_debug_generic_kernels_only(generic)
            try:
                dl, ds = g(loc.clone()), g(ls.clone())
                lv = LevelSpec(dl, ds, D, n, row_map=row_map)
                state = {k: torch.zeros_like(dl) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
                slots = torch.zeros(1024, device=DEV, dtype=torch.int64)
                for step in (1, 2):
                    ops.posterior_bwd(lv, g(pl), g(pls), True, 0.37, Gm, eps, 1, adam=ops.adam_cfg(2e-4, step), state=state,
                                      kl_accum=slots)
                outs.append((dl, ds, state, float(slots[:-1].sum()) / ops.KL_FX, int(slots[-1])))
            finally:
                lib.rcb_debug_generic_kernels_only(0)
        (l0, s0, st0, k0, b0), (l1, s1, st1, k1, b1) = outs
        assert torch.equal(l0, l1) and torch.equal(s0, s1) and all(torch.equal(st0[k], st1[k]) for k in st0)
        assert k0 == pytest.approx(k1, rel=1e-9) and k0 > 0 and b0 == b1 == 0
        assert float((l0 - g(loc)).abs().max()) > 0          # the update did something


def test_lds_staged_gather_kernels_equal_generic_kernels():
    """test-time layout (column permutation, encode mask, per-group beta, S = 5): the LDS-staged reparam and posterior
    kernels against the generic ones (rcb_debug_generic_kernels_only) on identical inputs: identical bits."""
    from recombiner_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(23)
    n, D, S, G = 6, 3779, 5, 400
    perm = torch.randperm(D, generator=gen).numpy()
    loc = 0.02 * torch.randn(n, D, generator=gen)
    ls = -4 + 0.5 * torch.randn(n, D, generator=gen)
    mask = (torch.rand(n, D, generator=gen) < 0.3).float()
    samp = 0.02 * torch.randn(n, D, generator=gen)
    pl = 0.01 * torch.randn(D, generator=gen)
    pls = -3 + 0.2 * torch.randn(D, generator=gen)
    gidx = torch.sort(torch.randint(0, G, (D,), generator=gen)).values.int()
    beta = torch.rand(n, G, generator=gen) * 1e-3
    eps = g(torch.randn(n, S, D, generator=gen))
    Gm = g(torch.randn(n, S, D, generator=gen) * 1e-3)
    outs = []
    for generic in (1, 0):
        lib.rcb_debug_generic_kernels_only(generic)
        try:
            dl, ds = g(loc.clone()), g(ls.clone())
            lv = LevelSpec(dl, ds, D, n, col_map=perm, enc_sample=g(samp), enc_mask=g(mask))
            h = ops.reparam_fwd([lv], [eps], S)
            state = {k: torch.zeros_like(dl) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
            slots = torch.zeros(1024, device=DEV, dtype=torch.int64)
            for step in (1, 2):
                ops.posterior_bwd(lv, g(pl), g(pls), True, 1.0, Gm, eps, S, beta=g(beta), group_idx=g(gidx), n_groups=G,
                                  adam=ops.adam_cfg(2e-4, step), state=state, kl_accum=slots)
            outs.append((h, dl, ds, state, float(slots.sum()) / ops.KL_FX))
        finally:
            lib.rcb_debug_generic_kernels_only(0)
    (h0, l0, s0, st0, k0), (h1, l1, s1, st1, k1) = outs
    assert torch.equal(h0, h1)
    assert torch.equal(l0, l1) and torch.equal(s0, s1) and all(torch.equal(st0[k], st1[k]) for k in st0)
    assert k0 == pytest.approx(k1, rel=1e-9) and k0 > 0
    assert float((l0 - g(loc)).abs().max()) > 0          # the update did something


# ---------------------------------------------------------------------------------------------------
# KL family, annealing, moments
# ---------------------------------------------------------------------------------------------------
def test_gauss_kl_rows_groups_and_beta_update():
    d = load("test_cifar.npz")
    loc, ls = t(d, "t_loc"), t(d, "t_log_scale")
    pl, pls = t(d, "kw_p_loc"), t(d, "kw_p_log_scale")
    gi = d["G_group_idx"].astype(np.int32)
    st_, en_ = d["G_start"].astype(np.int32), d["G_end"].astype(np.int32)
    beta = t(d, "beta_before")
    rows, groups = ops.gauss_kl(g(loc), g(ls), g(pl), g(pls), True, beta=g(beta), group_idx=g(torch.from_numpy(gi)),
                                seg_start=g(torch.from_numpy(st_)), seg_end=g(torch.from_numpy(en_)), want_groups=True)
    np.testing.assert_allclose(groups.cpu().numpy(), d["kls"], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(rows.sum().item(), float(d["kl_beta_weighted"]), rtol=2e-5)
    # unweighted, no segments
    rows2, _ = ops.gauss_kl(g(loc), g(ls), g(pl), g(pls), True)
    np.testing.assert_allclose(rows2.cpu().numpy(), d["kls"].sum(1), rtol=2e-5)
    # beta update from the golden KLs (identical inputs -> identical decisions)
    b = g(beta.clone())
    done = torch.zeros_like(b, dtype=torch.uint8)
    done[0, 3] = 1
    ops.beta_update(g(torch.from_numpy(d["kls"])), b, done)
    exp = d["beta_after"].copy()
    exp[0, 3] = d["beta_before"][0, 3]
    np.testing.assert_array_equal(b.cpu().numpy(), exp)


def test_col_moments_and_kl_colsum():
    gen = torch.Generator().manual_seed(8)
    rows, cols = 700, 517
    loc = 0.3 + 0.02 * torch.randn(rows, cols, generator=gen)
    ls = -4 + torch.randn(rows, cols, generator=gen)
    s, m2, sg = ops.col_moments(g(loc), g(ls))
    np.testing.assert_allclose((s / rows).cpu().numpy(), loc.double().mean(0).numpy(), rtol=1e-9)
    np.testing.assert_allclose((m2 / (rows - 1)).cpu().numpy(), loc.double().var(0).numpy(), rtol=1e-7)
    np.testing.assert_allclose((sg / rows).cpu().numpy(), (O.st(ls) ** 2).double().mean(0).numpy(), rtol=1e-5)
    mu, sig = O.refit_prior(loc, ls)
    sig_h = torch.sqrt(sg / rows + m2 / (rows - 1)).float().cpu()
    np.testing.assert_allclose(sig_h.numpy(), sig.numpy(), rtol=1e-5)
    pl = 0.3 + 0.01 * torch.randn(cols, generator=gen)
    ps = 0.02 + 0.01 * torch.rand(cols, generator=gen)
    cs = ops.gauss_kl_colsum(g(loc), g(O.st(ls)), g(pl), g(ps))
    ref = O.gauss_kl_elem(loc, O.st(ls), pl, ps).double().sum(0)
    np.testing.assert_allclose(cs.cpu().numpy(), ref.numpy(), rtol=1e-5)
    # the sums are exact fixed-point integers: bitwise the same from call to call, and the same when the rows are summed in
    # shards and the integers added -- for ANY cut, of the moments and of the KL sums alike (every element is rounded to the
    # integer grid on its own; round 3 rounded 256-row partial sums, which tied the KL sums to multiples of 256 rows)
    fx = ops.col_moments_fx(g(loc), g(ls))
    assert fx.shape == (6 * cols + 1,) and int(fx[-1]) == 0
    assert torch.equal(fx, ops.col_moments_fx(g(loc), g(ls)))
    kfx = ops.gauss_kl_colsum_fx(g(loc), g(ls), g(pl), g(ps), q_is_log=True)
    assert kfx.shape == (cols + 1,) and int(kfx[-1]) == 0
    assert torch.equal(kfx, ops.gauss_kl_colsum_fx(g(loc), g(ls), g(pl), g(ps), q_is_log=True))
    for cut in (1, 256, 333, 512):
        parts = ops.col_moments_fx(g(loc[:cut]), g(ls[:cut])) + ops.col_moments_fx(g(loc[cut:]), g(ls[cut:]))
        assert torch.equal(fx, parts)
        kparts = (ops.gauss_kl_colsum_fx(g(loc[:cut]), g(ls[:cut]), g(pl), g(ps), q_is_log=True)
                  + ops.gauss_kl_colsum_fx(g(loc[cut:]), g(ls[cut:]), g(pl), g(ps), q_is_log=True))
        assert torch.equal(kfx, kparts), cut
    # a diverged posterior must not turn into a finite prior or a finite grouping statistic: NaN / Inf / out-of-range terms
    # are counted, and the wrappers turn a non-zero count back into NaN (the reference propagates NaN by itself)
    for poison in (float("nan"), float("inf"), 100.0):          # 100^2 is beyond the documented |x| < 2^6
        bad = loc.clone()
        bad[5, 7] = poison
        fxb = ops.col_moments_fx(g(bad), g(ls))
        assert int(fxb[-1]) >= 1
        s_b, m2_b, sg_b = ops.moments_from_fx(fxb, rows)
        assert torch.isnan(m2_b).all() and (torch.isnan(s_b).all() or poison == 100.0)
    bad = loc.clone()
    bad[3, 11] = float("nan")
    kb = ops.gauss_kl_colsum_fx(g(bad), g(ls), g(pl), g(ps), q_is_log=True)
    assert int(kb[-1]) == 1 and torch.equal(kb[:11], kfx[:11]) and torch.equal(kb[12:-1], kfx[12:-1])
    assert torch.isnan(ops.gauss_kl_colsum(g(bad), g(O.st(ls)), g(pl), g(ps))).all()


def test_kl_log_of_a_diverged_step_is_nan():
    """the per-step KL accumulators are fixed-point integers; a workgroup sum that is NaN / Inf (a diverged posterior) is
    counted in the last slot and rcb_step_end logs NaN for that step, as the reference's ELBO list would show"""
    gen = torch.Generator().manual_seed(3)
    rows, cols = 64, 300
    loc = torch.nn.Parameter(g(0.05 * torch.randn(rows, cols, generator=gen)))
    ls = torch.nn.Parameter(g(-4 + 0.1 * torch.randn(rows, cols, generator=gen)))
    lv = LevelSpec(loc, ls, cols, rows)
    pl, ps = g(torch.zeros(cols)), g(torch.full((cols,), 0.02))
    d, e = g(torch.randn(rows, 1, cols, generator=gen)), g(torch.randn(rows, 1, cols, generator=gen))
    state = {k: torch.zeros(rows, cols, device=DEV) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
    tab, dyn = ops.adam_table(1e-3, 4).to(DEV), torch.zeros(2, device=DEV)
    step, slots = torch.zeros(1, device=DEV, dtype=torch.long), torch.zeros(1024, device=DEV, dtype=torch.int64)
    kl_log = torch.zeros(4, device=DEV, dtype=torch.float64)
    for it in range(2):
        if it == 1:
            with torch.no_grad():
                loc[2, 5] = float("nan")
        ops.step_begin(tab, step, dyn, slots)
        ops.posterior_bwd(lv, pl, ps, False, 1e-4, d, e, 1, adam=ops.adam_cfg(1e-3, 1, dyn=dyn), state=state, kl_accum=slots)
        ops.step_end(step, None, 1.0, slots, None, kl_log)
    out = kl_log.cpu().numpy()
    assert np.isfinite(out[0]) and out[0] > 0 and np.isnan(out[1]), out


# ---------------------------------------------------------------------------------------------------
# REC scoring: exact index selection
# ---------------------------------------------------------------------------------------------------
def test_rec_score_exact_against_golden_encodes():
    d = load("test_cifar.npz")
    loc, ls = t(d, "t_loc"), t(d, "t_log_scale")
    pl, pls = t(d, "kw_p_loc"), t(d, "kw_p_log_scale")
    st_, en_ = d["G_start"], d["G_end"]
    gum = torch.from_numpy(np.load(os.path.join(GOLDEN, "tables", "gumbel_seed42_f64.npy")))
    enc = d["enc_table"]
    rows = enc[:, 0].astype(int)
    grps = enc[:, 1].astype(int)
    starts = st_[grps]
    lens = en_[grps] - st_[grps]
    tables = {}
    for gl in np.unique(lens):
        tables[int(gl)] = g(O.sobol_normal_table(int(gl)))
    scale = O.st(ls)            # identical fp32 inputs for both sides
    pscale = O.st(pls)
    # put the first golden encode as job 0 so that its log-weights can be compared too
    idx, z, best, lw0 = ops.rec_score_argmax(g(loc), g(scale), g(pl), g(pscale), tables, g(gum), rows, starts, lens,
                                             want_logw0=True, mode=ops.REC_EXACT)
    idx = idx.cpu().numpy()
    assert np.array_equal(idx, enc[:, 2].astype(int)), (idx, enc[:, 2])
    # the fast scorer (quadratic form + certificate, what compress_posteriors runs): same indices, same samples
    idx_f, z_f, best_f, _ = ops.rec_score_argmax(g(loc), g(scale), g(pl), g(pscale), tables, g(gum), rows, starts, lens)
    assert np.array_equal(idx_f.cpu().numpy(), idx) and torch.equal(z_f, z)
    np.testing.assert_allclose((best_f[:, 0] - best_f[:, 1]).cpu().numpy(), enc[:, 3], rtol=1e-6, atol=1e-9)
    margins = (best[:, 0] - best[:, 1]).cpu().numpy()
    np.testing.assert_allclose(margins, enc[:, 3], rtol=1e-6, atol=1e-9)
    r0, g0 = rows[0], grps[0]
    np.testing.assert_allclose(lw0[:256].cpu().numpy(), d[f"enc_{r0}_{g0}_lw_head"], rtol=0, atol=2e-6)
    for b, (r, gr) in enumerate(zip(rows, grps)):
        zz = z[b, :lens[b]].cpu().numpy()
        np.testing.assert_allclose(zz, d[f"enc_{r}_{gr}_z"], rtol=1e-15, atol=0)


def test_rec_score_batch_vs_oracle_random():
    """many (row, group) jobs of mixed length against the fp64 oracle: all indices equal."""
    gen = torch.Generator().manual_seed(10)
    N, D = 16, 400
    loc = 0.02 * torch.randn(N, D, generator=gen)
    scale = 0.002 + 0.004 * torch.rand(N, D, generator=gen)
    pl = 0.01 * torch.randn(D, generator=gen)
    ps = 0.015 + 0.01 * torch.rand(D, generator=gen)
    gum = torch.from_numpy(O.gumbel_table(42))
    lens_all = [3, 5]
    tabs_cpu = {3: torch.from_numpy(np.load(os.path.join(GOLDEN, "tables", "sobol_normal_g3_seed42_f32.npy")).astype(np.float64)),
                5: torch.from_numpy(np.load(os.path.join(GOLDEN, "tables", "sobol_normal_g5_seed42_f32.npy")).astype(np.float64))}
    jobs = []
    rs = np.random.RandomState(0)
    for _ in range(48):
        gl = lens_all[rs.randint(2)]
        jobs.append((rs.randint(N), rs.randint(D - gl), gl))
    jr, js, jg = map(np.array, zip(*jobs))
    idx, z, best, _ = ops.rec_score_argmax(g(loc), g(scale), g(pl), g(ps), {k: g(v) for k, v in tabs_cpu.items()}, g(gum),
                                           jr, js, jg)
    idx = idx.cpu().numpy()
    for b, (r, s, gl) in enumerate(jobs):
        i, zi, lw = O.rec_score(tabs_cpu[gl], loc[r, s:s + gl], scale[r, s:s + gl], pl[s:s + gl], ps[s:s + gl], gum)
        assert idx[b] == i, (b, idx[b], i)
        np.testing.assert_allclose(z[b, :gl].cpu().numpy(), zi.numpy(), rtol=1e-15)


def _rec_problem(seed, N, D, lens_pool, n_jobs, ratio):
    """random posteriors / priors with posterior-to-prior scale ratio around `ratio` and (row, start, len) jobs"""
    gen = torch.Generator().manual_seed(seed)
    pl = 0.01 * torch.randn(D, generator=gen)
    ps = 0.015 + 0.01 * torch.rand(D, generator=gen)
    loc = pl[None] + ps[None] * 1.5 * torch.randn(N, D, generator=gen)
    scale = ps[None] * ratio * (0.5 + torch.rand(N, D, generator=gen))
    rs = np.random.RandomState(seed)
    jr = rs.randint(N, size=n_jobs)
    jg = np.asarray(lens_pool)[rs.randint(len(lens_pool), size=n_jobs)]
    js = np.array([rs.randint(D - gl + 1) for gl in jg])
    return loc, scale, pl, ps, jr, js, jg


def test_rec_fast_scorer_equals_exact_scorer_on_many_random_jobs():
    """12 288 random jobs (group lengths 1..12, posterior scales from 0.03x to 1x the prior's): the certified fast scorer
    must return exactly the indices of the op-for-op scorer, and must have had to fall back only rarely."""
    K = 65536
    gum = g(torch.from_numpy(O.gumbel_table(42)))
    tabs = ops.RecTables("cuda", K)
    for gl in range(1, 13):
        tabs.add(gl, O.sobol_normal_table(gl))
    n_unc = 0
    for seed, ratio in ((1, 0.03), (2, 0.2), (3, 1.0)):
        loc, scale, pl, ps, jr, js, jg = _rec_problem(seed, 64, 600, list(range(1, 13)), 4096, ratio)
        args = (g(loc), g(scale), g(pl), g(ps), tabs, gum)
        jobs, _ = ops.RecJobs.from_host("cuda", jr, js, jg, rows=64, cols=600)
        i_exact, b_exact, _, _ = ops.rec_score(*args, jobs, ops.REC_EXACT)
        i_fast, b_fast, unc, _ = ops.rec_score(*args, jobs, ops.REC_FAST)
        assert torch.equal(i_exact, i_fast)
        assert int(i_exact.min()) >= 0 and int(i_exact.max()) < K
        # the fast scores themselves agree with the exact ones far inside the certified bound
        err = float((b_fast[:, 0] - b_exact[:, 0]).abs().max())
        assert err < 1e-7, err
        n_unc += int(unc.sum())
    assert n_unc <= 12, n_unc        # <= 0.1 %: the certificate is tight enough to be useful


def test_rec_long_groups_against_oracle():
    """groups of several hundred parameters (low bit-rates pack that many: prior_model.py:301-316 has no size cap),
    a length that is no multiple of anything, and a batch mixing lengths 1 .. 333 -- against the fp64 oracle"""
    K = 65536
    gum_c = torch.from_numpy(O.gumbel_table(42))
    lens = [1, 33, 64, 200, 333]
    tabs_c = {gl: O.sobol_normal_table(gl) for gl in lens}
    tabs = ops.RecTables.from_dict(tabs_c, "cuda", K)
    loc, scale, pl, ps, jr, js, jg = _rec_problem(5, 6, 700, lens, 40, 0.9)
    # long groups only make sense when every element carries a fraction of a bit: posteriors close to the prior
    loc = pl[None] + 0.05 * (loc - pl[None])
    for mode in (ops.REC_EXACT, ops.REC_FAST):
        idx, z, best, _ = ops.rec_score_argmax(g(loc), g(scale), g(pl), g(ps), tabs, g(gum_c), jr, js, jg, mode=mode)
        idx = idx.cpu().numpy()
        for b, (r, s, gl) in enumerate(zip(jr, js, jg)):
            i, zi, lw = O.rec_score(tabs_c[gl], loc[r, s:s + gl], scale[r, s:s + gl], pl[s:s + gl], ps[s:s + gl], gum_c)
            assert idx[b] == i, (mode, b, gl, idx[b], i)
            np.testing.assert_allclose(z[b, :gl].cpu().numpy(), zi.numpy(), rtol=1e-15)
            # log(sigma) is taken in fp32 (torch Normal.log_prob on fp32 scales) and device and host logf differ by an ulp here
            # and there: a candidate-INDEPENDENT shift of ~1e-7 per element that cannot move the arg-max.  Differences between
            # candidates are free of it: the gap to the runner-up agrees to fp64 rounding.
            top2 = torch.topk(lw, 2).values
            assert float(best[b, 0] - best[b, 1]) == pytest.approx(float(top2[0] - top2[1]), abs=2e-10 * gl)
            assert float(best[b, 0]) == pytest.approx(float(lw.max()), abs=2e-6 + 2e-7 * gl)


def test_rec_rejects_bad_jobs_on_the_device_and_refuses_inexact_tables():
    K = 4096
    gum = g(torch.from_numpy(O.gumbel_table(42)[:K].copy()))
    t3 = O.sobol_normal_table(3)[:K].contiguous()
    tabs = ops.RecTables.from_dict({3: t3}, "cuda", K)
    with pytest.raises(ops.RcbError):
        tabs.add(2, torch.full((K, 2), 0.1, dtype=torch.float64))            # 0.1 is no fp32 number
    loc, scale, pl, ps, _, _, _ = _rec_problem(7, 4, 50, [3], 1, 0.5)
    i32 = torch.int32
    # device-built job list with a row / a window outside the matrix and a length without table: idx = -1, nothing written
    jobs = ops.RecJobs(torch.tensor([0, 9, 1, 2], dtype=i32, device="cuda"), torch.tensor([0, 0, 49, 5], dtype=i32, device="cuda"),
                       torch.tensor([3, 3, 3, 2], dtype=i32, device="cuda"), torch.tensor([0, 0, 0, 0], dtype=i32, device="cuda"))
    for mode in (ops.REC_EXACT, ops.REC_FAST):
        idx, _, _, _ = ops.rec_score(g(loc), g(scale), g(pl), g(ps), tabs, gum, jobs, mode)
        idx = idx.cpu().numpy()
        assert idx[0] >= 0 and (idx[1:] == -1).all(), idx
    sample = torch.zeros(4, 50, device="cuda")
    idx_t = torch.tensor([5, 5, 5, 5], dtype=i32, device="cuda")
    ops.rec_commit(g(pl), g(ps), tabs, jobs, idx_t, n_groups=1, enc_sample=sample)
    assert int((sample != 0).sum()) == 3 and int((sample[0, :3] != 0).sum()) == 3
    with pytest.raises(ops.RcbError):                                          # host job lists are validated on the host
        ops.rec_score_argmax(g(loc), g(scale), g(pl), g(ps), tabs, gum, [9], [0], [3])


def test_posterior_update_draws_the_next_sample_bit_for_bit():
    """rcb_level_bwd.next_*: the posterior update's fused sample of the next step equals rcb_reparam_rng_fwd run afterwards
    at step counter + 1 on the updated parameters (sample, noise and bf16 copy), and the update itself is unchanged."""
    torch.manual_seed(5)
    rows, cols = 8, 3267
    def level():
        loc = (torch.randn(rows, cols) * 0.05).to(DEV)
        ls = (torch.randn(rows, cols) * 0.3 - 4).to(DEV)
        return ops.LevelSpec(loc, ls, cols, rows)
    torch.manual_seed(6)
    lv_a = level()
    lv_b = ops.LevelSpec(lv_a.loc.clone(), lv_a.log_scale.clone(), cols, rows)
    p_loc, p_scale = torch.zeros(cols, device=DEV), torch.full((cols,), 0.02, device=DEV)
    d = (torch.randn(rows, 1, cols) * 1e-3).to(DEV)
    step = torch.tensor([41], device=DEV, dtype=torch.int64)
    seed = 0x1234567890ABCDEF
    cfg = ops.adam_cfg(2e-4, 3)
    def state(lv):
        return {k: torch.zeros_like(lv.loc) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
    # a: update, then the stand-alone sampling kernel at step 42
    _, eps_a = ops.reparam_rng(lv_a, seed, 0, step)[:2]
    ops.posterior_bwd(lv_a, p_loc, p_scale, False, 1.0, d, eps_a, 1, adam=cfg, state=state(lv_a))
    step42 = step + 1
    out_a, eps_a2, o16_a = ops.reparam_rng(lv_a, seed, 0, step42, want_bf16=True)
    # b: update with the fused next sample
    buf = ops.sample_buffers(lv_b, True)
    ops.reparam_rng(lv_b, seed, 0, step, want_bf16=True, buffers=buf)
    ops.posterior_bwd(lv_b, p_loc, p_scale, False, 1.0, d, buf[1], 1, adam=cfg, state=state(lv_b),
                      next_sample=ops.NextSample(buf, seed, 0, step, 1))
    assert torch.equal(lv_a.loc, lv_b.loc) and torch.equal(lv_a.log_scale, lv_b.log_scale)
    assert torch.equal(out_a, buf[0]) and torch.equal(eps_a2, buf[1]) and torch.equal(o16_a, buf[2][:, :cols])
    # levels that do not take the flat path refuse
    lv_c = ops.LevelSpec(lv_a.loc[:3].contiguous(), lv_a.log_scale[:3].contiguous(), cols, 3)     # 3 * 3267 is odd
    bc = ops.sample_buffers(lv_c, False)
    with pytest.raises(ops.RcbError):
        ops.posterior_bwd(lv_c, p_loc, p_scale, False, 1.0, d[:3].contiguous(), bc[1], 1, adam=cfg, state=state(lv_c),
                          next_sample=ops.NextSample(bc, seed, 0, step, 1))


def test_integration_md_binding_stub_runs_and_matches_the_oracle():
    """The ctypes stub INTEGRATION.md shows a maintainer is EXTRACTED from that file and executed as written (working
    directory = the repository root, as its relative library path assumes): its structure mirror must have the library's
    size (the stub asserts rcb_struct_bytes itself) and its results must equal the product binding's bit for bit and the
    fp32 oracle's within the fp32 kernel's tolerance."""
    import re
    from golden_util import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "class SirenDesc" in b]
    assert len(blocks) == 1, "INTEGRATION.md must hold exactly one SirenDesc stub"
    ns, cwd = {}, os.getcwd()
    os.chdir(ROOT)
    try:
        exec(compile(blocks[0], "INTEGRATION.md:stub", "exec"), ns)
    finally:
        os.chdir(cwd)
    from recombiner_amd import _lib
    import ctypes
    assert ctypes.sizeof(ns["SirenDesc"]) == ctypes.sizeof(_lib.SirenDesc)
    assert [f[0] for f in ns["SirenDesc"]._fields_] == [f[0] for f in _lib.SirenDesc._fields_]
    case = dict(F=16, E=16, n_hidden=3, C=3, P=1024, N=5, S=1)
    dims, D, xf, pe, wv, y = _siren_case(seed=21, **case)
    sse, dw, dpe = ns["siren_loss_bwd"](g(xf), g(pe), g(wv), g(y))
    torch.cuda.synchronize()
    meta = SirenMeta(1, 1024, 16, 16, 3, 32, 3)
    s2, w2, p2 = ops.siren_loss_bwd(g(xf), g(pe), g(wv), g(y), 1.0 / (1024 * 3), meta)
    assert torch.equal(sse, s2) and torch.equal(dw, w2) and torch.equal(dpe, p2)
    pe_r, wv_r = pe.clone().requires_grad_(True), wv.clone().requires_grad_(True)
    y_ref = _oracle_mlp(dims, xf, pe_r, wv_r, 1)
    (((y_ref - y) ** 2).sum() / (1024 * 3)).backward()
    assert rel_err(sse, ((y_ref.detach() - y) ** 2).sum((1, 2))) < 2e-5
    assert rel_err(dw, wv_r.grad) < 1e-4 and rel_err(dpe, pe_r.grad) < 1e-4


# ---------------------------------------------------------------------------------------------------
# round 4: the A transform's per-row operands as (hi, lo) bf16 planes written by their producers
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,sizes", [(37, [1056, 1056, 1056, 99]), (4096, [1056, 1056, 1056, 99]), (192, [1584, 2352, 2352, 147]),
                                        (130, [424, 440, 63]), (129, [64, 8, 40])])
def test_atrans_plane_operands_are_bit_identical_to_fp32_rows(rows, sizes):
    """rcb_atrans_apply with x_hi / x_lo (ops.Planes) instead of fp32 rows: hi = bf16(x), lo = bf16(x - hi) is exactly what the
    kernel forms from an fp32 row, and the plane kernel multiplies the same fragments in the same order -- forward and data
    gradient must equal the fp32-row launch BIT FOR BIT for every number of terms, incl. the ragged 99- / 147- / 63-wide
    layers, K-split launches (few rows) and layer sizes that are not a multiple of 32 (424, 440: a partial last chunk)."""
    gen = torch.Generator().manual_seed(rows * 3 + len(sizes))
    D = sum(sizes)
    cum = np.cumsum([0] + sizes)
    slices = [(int(cum[i]), int(cum[i + 1])) for i in range(len(sizes))]
    x = g(torch.randn(rows, D, generator=gen) * 0.03)
    A = [g(torch.randn(n, n, generator=gen) / n ** 0.5) for n in sizes]
    xp = ops.Planes.from_float(x)
    assert xp.ld % 32 == 0 and torch.equal(xp.hi, x.bfloat16()) and rel_err(xp.float(), x) < 2.0 ** -15
    for terms in (2, 3, 1):
        tr = ops.ATransform(slices, DEV, terms=terms)
        tr.prepare(A)
        w32 = tr.forward(x, torch.empty(rows, D, device=DEV))
        w16 = tr.forward(xp, torch.full((rows, D), float("nan"), device=DEV))
        assert torch.equal(w32, w16), (terms, "forward")
        d32 = tr.dgrad(x, torch.empty(rows, D, device=DEV))
        d16 = tr.dgrad(xp, torch.full((rows, D), float("nan"), device=DEV))
        assert torch.equal(d32, d16), (terms, "data gradient")
    # weight gradient: wide layers on the hi planes == the bf16 copies; narrow layers read float(hi) + float(lo)
    dw = g(torch.randn(rows, D, generator=gen) * 1e-3)
    dp = ops.Planes.from_float(dw)
    tr = ops.ATransform(slices, DEV, terms=2)
    gp = tr.wgrad(xp, dp)
    gr = tr.wgrad(xp.float(), dp.float(), xp.hi, dp.hi, True)
    for a_, b_ in zip(gp, gr):
        assert torch.equal(a_, b_)


def test_producers_write_the_operand_planes():
    """the three producers of plane operands write hi = bf16(v), lo = bf16(v - hi) of exactly the fp32 value they would have
    stored: rcb_reparam_rng_fwd (out_bf16 + out_lo, fp32 output optional), the fused next sample of rcb_posterior_bwd
    (next_out_bf16 + next_out_lo) together with the re-drawn noise (eps_from_rng: eps = NULL, the update is bit-identical to
    the one that reads the stored noise), and the SIREN gradient epilogue (dw_bf16 + dw_lo, dwvec = NULL), chunked or not."""
    torch.manual_seed(5)
    rows, cols = 8, 3267
    loc = (torch.randn(rows, cols) * 0.05).to(DEV)
    ls = (torch.randn(rows, cols) * 0.3 - 4).to(DEV)
    lv_a = ops.LevelSpec(loc.clone(), ls.clone(), cols, rows)
    lv_b = ops.LevelSpec(loc.clone(), ls.clone(), cols, rows)
    step = torch.tensor([41], device=DEV, dtype=torch.int64)
    seed = 0x1234567890ABCDEF
    # 1. sampler
    out_a, eps_a = ops.reparam_rng(lv_a, seed, 0, step)
    buf = ops.sample_buffers(lv_b, planes=True, want_eps=False)
    assert buf[0] is None and buf[1] is None
    ops.reparam_rng(lv_b, seed, 0, step, buffers=buf)
    pl = buf[2]
    ref = ops.Planes.from_float(out_a.view(rows, cols))
    assert torch.equal(pl.hi, ref.hi) and torch.equal(pl.lo, ref.lo)
    assert float(pl.buf[:, :, cols:].abs().max()) == 0.0                  # the row padding is never written
    # 2. posterior update: stored noise + fp32 next sample  vs  re-drawn noise + planes
    p_loc, p_scale = torch.zeros(cols, device=DEV), torch.full((cols,), 0.02, device=DEV)
    d = (torch.randn(rows, 1, cols) * 1e-3).to(DEV)
    cfg = ops.adam_cfg(2e-4, 3)
    st_a = {k: torch.zeros_like(loc) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
    st_b = {k: torch.zeros_like(loc) for k in ("m_loc", "v_loc", "m_ls", "v_ls")}
    buf_a = ops.sample_buffers(lv_a, True)
    ops.posterior_bwd(lv_a, p_loc, p_scale, False, 1.0, d, eps_a, 1, adam=cfg, state=st_a,
                      next_sample=ops.NextSample(buf_a, seed, 0, step, 1))
    ops.posterior_bwd(lv_b, p_loc, p_scale, False, 1.0, d, None, 1, adam=cfg, state=st_b,
                      next_sample=ops.NextSample(buf, seed, 0, step, 1, redraw_eps=True))
    assert torch.equal(lv_a.loc, lv_b.loc) and torch.equal(lv_a.log_scale, lv_b.log_scale)
    for k in st_a:
        assert torch.equal(st_a[k], st_b[k]), k
    ref = ops.Planes.from_float(buf_a[0].view(rows, cols))
    assert torch.equal(pl.hi, ref.hi) and torch.equal(pl.lo, ref.lo) and torch.equal(pl.hi, buf_a[2][:, :cols])
    with pytest.raises(ops.RcbError):                                       # re-drawn noise and a stored copy exclude each other
        ops.posterior_bwd(lv_b, p_loc, p_scale, False, 1.0, d, eps_a, 1, adam=cfg, state=st_b,
                          next_sample=ops.NextSample(buf, seed, 0, step, 1, redraw_eps=True))
    # 3. SIREN gradient epilogue (width 32 bf16 kernel, a wide kernel, and a chunked launch through rcb_siren_reduce_chunks)
    for hidden, P, chunks in ((32, 256, None), (64, 256, None), (32, 2048, 2)):
        case = dict(F=16, E=16, n_hidden=3, C=3, P=P, N=5, S=1, hidden=hidden)
        dims, D, xf, pe, wv, y = _siren_case(seed=33, **case)
        meta = SirenMeta(1, P, 16, 16, 3, hidden, 3, precision=1)
        pe16 = g(pe).bfloat16()
        sse, dw, dpe = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), 1.0 / (P * 3), meta, pixel_chunks=chunks)
        sse2, none, dpe2, dwp = ops.siren_loss_bwd(g(xf), pe16, g(wv), g(y), 1.0 / (P * 3), meta, pixel_chunks=chunks, want_planes=True)
        assert none is None and torch.equal(sse, sse2) and torch.equal(dpe, dpe2)
        ref = ops.Planes.from_float(dw.contiguous())
        assert torch.equal(dwp.hi, ref.hi) and torch.equal(dwp.lo, ref.lo), (hidden, P, chunks)
