"""GPU parity of the hand-written phase-conv upsampling kernels against the plain fp32 nn.Module
(nearest-upsample + conv) and of the torch-level phase form (exactness of the decomposition)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from recombiner_amd import prior_model as PM  # noqa: E402
from recombiner_amd.upsample_fast import UpsampleFast, hip_path_supported, upsample_cifar_hip  # noqa: E402
from recombiner_amd.utils import map_lpe_to_inr_inputs  # noqa: E402

DEV = "cuda"


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("S,N,stage1_bf16,pe_bf16", [(1, 6, False, False), (5, 3, False, False), (2, 5, True, False),
                                                     (3, 4, True, True)])
def test_upsample_hip_forward_backward(S, N, stage1_bf16, pe_bf16):
    torch.manual_seed(0)
    net = PM.Upsample(2, [2, 1, 1], [4, 2, 2]).to(DEV)
    assert hip_path_supported(net, [32, 32], [16, 16], False, 2)
    lpe = (0.1 * torch.randn(S, N, 2, 2, 128, device=DEV)).requires_grad_(True)
    g = torch.randn(N, S, 1024, 16, device=DEV)
    params = list(net.parameters())
    # fp64 reference of the plain module (MIOpen's fp32 algorithms disagree among themselves by ~2e-2 on
    # the gradients, so fp32 is not a usable yardstick here)
    import copy
    net64 = copy.deepcopy(net).double()
    lpe64 = lpe.detach().double().requires_grad_(True)
    ref = map_lpe_to_inr_inputs(net64, lpe64, 128, [32, 32], [16, 16], False, None, 2)
    gr = torch.autograd.grad(ref, [lpe64] + list(net64.parameters()), g.double())
    out = upsample_cifar_hip(net, lpe, stage1_bf16, pe_bf16)
    assert out.dtype == (torch.bfloat16 if pe_bf16 else torch.float32) and out.is_contiguous()
    go = torch.autograd.grad(out, [lpe] + params, g.to(out.dtype))
    e_fwd = rel(out, ref)
    errs = [rel(a, b) for a, b in zip(go, gr)]
    print("upsample hip: fwd %.2e  dlpe %.2e  dW1 %.2e db1 %.2e dW2 %.2e db2 %.2e dW3 %.2e db3 %.2e" % (e_fwd, *errs))
    # bf16 operands (8-bit mantissa), fp32 accumulation
    # bf16 operands and bf16 intermediate images (h2, dz2): max-norm errors of a few 1e-2 on gradients
    assert e_fwd < (2e-2 if stage1_bf16 else 1e-2) + (4e-3 if pe_bf16 else 0)
    assert max(errs) < (8e-2 if stage1_bf16 else 6e-2)


def _bf(x):
    """round to bf16, keep fp64"""
    return x.to(torch.bfloat16).double()


def _phase_eval(x, We, b):
    """`nearest-upsample(2) -> conv3x3(pad 1)` in sub-pixel form, written from the definition: x [B, G, G, Ci] (fp64),
    We [ty, tx, ci, a, c, co], b [co] -> [B, 2G, 2G, co]; output pixel (2i+a, 2j+c) reads source pixels (i+a+ty-1, j+c+tx-1)"""
    B, G = x.shape[0], x.shape[1]
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1))
    out = torch.zeros(B, 2 * G, 2 * G, We.shape[-1], dtype=x.dtype, device=x.device)
    for a in range(2):
        for c in range(2):
            acc = 0
            for ty in range(2):
                for tx in range(2):
                    acc = acc + xp[:, a + ty:a + ty + G, c + tx:c + tx + G, :] @ We[ty, tx, :, a, c, :]
            out[:, a::2, c::2, :] = acc + b
    return out


def _eff23(W):
    """effective weights of a (x2, 3x3, pad 1) stage from the conv weight [co, ci, 3, 3]: kernel tap k of output phase a lands
    on source tap t with  a = 0: {0} -> t 0, {1, 2} -> t 1;  a = 1: {0, 1} -> t 0, {2} -> t 1"""
    R = torch.zeros(2, 2, 3, dtype=W.dtype, device=W.device)      # [a, t, k]
    R[0, 0, 0] = R[0, 1, 1] = R[0, 1, 2] = 1
    R[1, 0, 0] = R[1, 0, 1] = R[1, 1, 2] = 1
    return torch.einsum("ayk,cxl,oikl->yxiaco", R, R, W)


def _eff1(W1):
    """stage 1 (x4, 5x5, pad 2) on the 2 x 2 latent grid as one dense map [2*2*128, 8*8*64]"""
    M = torch.zeros(8, 2, 5, dtype=W1.dtype, device=W1.device)       # [y, s, k]: tap k of output row y reads source row s
    for y in range(8):
        for k in range(5):
            u = y + k - 2
            if 0 <= u < 8:
                M[y, u // 4, k] = 1
    return torch.einsum("ysk,xtl,oikl->stiyxo", M, M, W1).reshape(512, 4096)


def test_phase_conv_kernels_against_rounded_operand_reference():
    """The hand-written phase-conv kernels (shipped 16-bit mode: bf16 stage 1, bf16 intermediates) against an fp64
    evaluation of the SAME arithmetic -- operands rounded to bf16 exactly where the kernels round them (inputs, effective
    weights after the tap sums, z1, the activations a1 / h2, the gradients dz2 / dz1 and the bf16 GEMM results of stage 1)
    -- so that only the fp32 accumulation order differs: tolerances at rounding level instead of the 2e-2 / 8e-2 a
    full-precision yardstick needs.  The evaluator itself is tied to the reference: without any rounding it reproduces the
    oracle's UpsampleNet (nearest-upsample + conv, fp64) to 1e-12, forward and gradients."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_cpu as O
    torch.manual_seed(11)
    B = 6
    net = PM.Upsample(2, [2, 1, 1], [4, 2, 2]).to(DEV)
    W1, b1, W2, b2, W3, b3 = [p_.detach().double() for p_ in (net.conv1.weight, net.conv1.bias, net.conv2.weight, net.conv2.bias,
                                                               net.conv3.weight, net.conv3.bias)]
    lpe = 0.1 * torch.randn(1, B, 2, 2, 128, device=DEV)
    g16 = (0.01 * torch.randn(B, 1, 1024, 16, device=DEV)).bfloat16()
    lk = lambda t_: torch.where(t_ > 0, t_, 0.01 * t_)       # noqa: E731
    dlk = lambda t_: torch.where(t_ > 0, torch.ones_like(t_), torch.full_like(t_, 0.01))      # noqa: E731

    # ---- 1. the evaluator, unrounded, IS the reference's upsampling net ------------------------------------------------
    W1r, W2r, W3r = (w.clone().requires_grad_(True) for w in (W1, W2, W3))
    x64 = lpe[0].double().clone().requires_grad_(True)
    z1 = (x64.reshape(B, 512) @ _eff1(W1r) + b1.repeat(64)).view(B, 8, 8, 64)
    pe_ev = _phase_eval(lk(_phase_eval(lk(z1), _eff23(W2r), b2)), _eff23(W3r), b3)
    up = O.UpsampleNet(2, [2, 1, 1], [4, 2, 2])
    up.weights = [w.cpu().clone().requires_grad_(True) for w in (W1, b1, W2, b2, W3, b3)]
    xo = lpe[0].double().cpu().clone().requires_grad_(True)
    pe_or = up(xo.movedim(-1, 1)).movedim(1, -1)
    assert rel(pe_ev.cpu(), pe_or) < 1e-12
    ge = torch.autograd.grad(pe_ev, [x64, W1r, W2r, W3r], g16.double().view(B, 32, 32, 16))
    go_ = torch.autograd.grad(pe_or, [xo, up.weights[0], up.weights[2], up.weights[4]], g16.double().cpu().view(B, 32, 32, 16))
    for a_, b_ in zip(ge, go_):
        assert rel(a_.cpu(), b_) < 1e-11

    # ---- 2. the kernels' arithmetic in fp64, rounded where they round ------------------------------------------------------
    We1, We2, We3 = _eff1(W1), _eff23(W2), _eff23(W3)
    lpe_r, We1_r = _bf(lpe[0].double().reshape(B, 512)), _bf(We1)
    z1 = _bf(lpe_r @ We1_r + _bf(b1).repeat(64)).view(B, 8, 8, 64)
    a1 = _bf(lk(z1))
    We2_r, We3_r = _bf(We2).requires_grad_(True), _bf(We3).requires_grad_(True)
    a1g = a1.clone().requires_grad_(True)
    b2g, b3g = b2.clone().requires_grad_(True), b3.clone().requires_grad_(True)
    pre2 = _phase_eval(a1g, We2_r, b2g)
    h2 = _bf(lk(pre2)).detach()
    h2g = h2.clone().requires_grad_(True)
    pe_ref = _phase_eval(h2g, We3_r, b3g)
    # backward, stage by stage
    dh2, dWe3, db3 = torch.autograd.grad(pe_ref, [h2g, We3_r, b3g], g16.double().view(B, 32, 32, 16))
    dz2 = _bf(dh2 * dlk(h2))
    da1, dWe2, db2 = torch.autograd.grad(pre2, [a1g, We2_r, b2g], dz2)
    dz1 = _bf(da1 * dlk(z1)).reshape(B, 4096)
    dlpe_ref = _bf(dz1 @ We1_r.t())
    dWe1 = _bf(lpe_r.t() @ dz1)
    db1_ref = dz1.view(B, 64, 64).sum((0, 1))
    W1g, W2g, W3g = (w.clone().requires_grad_(True) for w in (W1, W2, W3))
    dW1_ref, = torch.autograd.grad(_eff1(W1g), [W1g], dWe1)
    dW2_ref, = torch.autograd.grad(_eff23(W2g), [W2g], dWe2)
    dW3_ref, = torch.autograd.grad(_eff23(W3g), [W3g], dWe3)

    # ---- 3. the kernels ----------------------------------------------------------------------------------------------------
    lpe_k = lpe.clone().requires_grad_(True)
    params = [net.conv1.weight, net.conv1.bias, net.conv2.weight, net.conv2.bias, net.conv3.weight, net.conv3.bias]
    pe32 = upsample_cifar_hip(net, lpe_k, True, False)                       # fp32 pe: the last rounding is not in the way
    assert rel(pe32.view(B, 32, 32, 16), pe_ref) < 1.5e-3, rel(pe32.view(B, 32, 32, 16), pe_ref)      # (a few z1 / h2 elements land on the other side of a bf16 rounding boundary)
    pe16 = upsample_cifar_hip(net, lpe_k, True, True)
    assert rel(pe16.view(B, 32, 32, 16), pe_ref) < 4.5e-3                  # one bf16 ulp of the largest element
    gk = torch.autograd.grad(pe16, [lpe_k] + params, g16)
    names = ["dlpe", "dW1", "db1", "dW2", "db2", "dW3", "db3"]
    refs = [dlpe_ref.view(1, B, 2, 2, 128), dW1_ref, db1_ref, dW2_ref, db2, dW3_ref, db3]
    # dlpe and dW1 pass through a bf16-output GEMM: one bf16 ulp; the others are fp32 sums of identical products
    tols = [5e-3, 2e-3, 1e-3, 1e-3, 5e-4, 1.5e-3, 1e-5]        # measured: 1.8e-3 4.6e-4 1.8e-4 2.0e-4 2.7e-5 3.5e-4 5.0e-8
    errs = [rel(a_, b_) for a_, b_ in zip(gk, refs)]
    print("phase-conv kernels vs rounded-operand fp64 reference:", " ".join("%s %.1e" % (n_, e_) for n_, e_ in zip(names, errs)))
    for n_, e_, t_ in zip(names, errs, tols):
        assert e_ < t_, (n_, e_, t_)


def test_phase_form_is_exact_on_gpu():
    torch.manual_seed(1)
    net = PM.Upsample(2, [2, 1, 1], [4, 2, 2]).to(DEV).double()
    fast = UpsampleFast(net)
    x = torch.randn(3, 128, 2, 2, device=DEV, dtype=torch.double)
    assert rel(fast(x), net(x)) < 1e-12


def test_bf16_pe_storage_is_bit_identical():
    """pe / dpe stored as bf16 == fp32 storage rounded where the consumers round it anyway."""
    torch.manual_seed(3)
    S, N = 2, 7
    net = PM.Upsample(2, [2, 1, 1], [4, 2, 2]).to(DEV)
    lpe = (0.1 * torch.randn(S, N, 2, 2, 128, device=DEV)).requires_grad_(True)
    g16 = torch.randn(N, S, 1024, 16, device=DEV).bfloat16()
    params = list(net.parameters())
    pe32 = upsample_cifar_hip(net, lpe, True, False)
    pe16 = upsample_cifar_hip(net, lpe, True, True)
    assert torch.equal(pe32.bfloat16(), pe16)
    g32 = torch.autograd.grad(pe32, [lpe] + params, g16.float())
    g16r = torch.autograd.grad(pe16, [lpe] + params, g16)
    assert torch.equal(g32[0], g16r[0])                      # data gradient: deterministic kernels, same operands
    for a, b in zip(g32[1:], g16r[1:]):                      # weight gradients: fp32 atomics, order-dependent sums
        assert rel(a, b) < 1e-5


def test_effective_weight_kernels_match_einsum():
    """rcb_upconv_weff_build / _grad against the einsum definition of the phase form (fp64)."""
    from recombiner_amd import ops
    from recombiner_amd.upsample_fast import PhaseStage, _phase_R, _stage1_maps
    torch.manual_seed(5)
    W1 = torch.randn(64, 128, 5, 5, device=DEV)
    b1 = torch.randn(64, device=DEV)
    W2 = torch.randn(64, 64, 3, 3, device=DEV)
    W3 = torch.randn(16, 64, 3, 3, device=DEV)
    M = _stage1_maps(DEV, torch.float64)
    st = PhaseStage(2, 3, 1, 2)
    ref1 = torch.einsum("ysk,xtl,oikl->stiyxo", M, M, W1.double()).reshape(512, 4096)
    ref2, ref3 = st.eff_weight(W2.double()), st.eff_weight(W3.double())
    for bf in (False, True):
        e1, b1rep, e2, e3, pack = ops.upconv_weff_build(W1, b1, W2, W3, bf)
        assert e1.dtype == (torch.bfloat16 if bf else torch.float32)
        assert rel(e1, ref1) < (5e-3 if bf else 1e-6) and rel(e2, ref2) < 1e-6 and rel(e3, ref3) < 1e-6
        assert torch.equal(b1rep.float().view(64, 64), b1.to(b1rep.dtype).float().expand(64, 64))
    # gradient map = transpose of the build map
    g1 = torch.randn(512, 4096, device=DEV)
    g2 = torch.randn(2, 2, 64, 2, 2, 64, device=DEV)
    g3 = torch.randn(2, 2, 64, 2, 2, 16, device=DEV)
    R = _phase_R(DEV, 2, 3, 1).double()
    r1 = torch.einsum("ysk,xtl,stiyxo->oikl", M, M, g1.double().view(2, 2, 128, 8, 8, 64))
    r2 = torch.einsum("atk,bul,tuiabo->oikl", R, R, g2.double())
    r3 = torch.einsum("atk,bul,tuiabo->oikl", R, R, g3.double())
    d1, d2, d3 = ops.upconv_weff_grad(g1, g2, g3)
    assert rel(d1, r1) < 1e-6 and rel(d2, r2) < 1e-6 and rel(d3, r3) < 1e-6
    d1b, _, _ = ops.upconv_weff_grad(g1.bfloat16(), g2, g3)
    r1b = torch.einsum("ysk,xtl,stiyxo->oikl", M, M, g1.bfloat16().double().view(2, 2, 128, 8, 8, 64))
    assert rel(d1b, r1b) < 1e-6


def test_packed_fragments_give_the_same_kernels_results():
    """rcb_upconv_weff_build's fragment pack vs fragments built inside the kernels from the fp32 effective weights:
    both round the same fp32 values to bf16 (sums of the same taps in a different order: equal to 1 bf16 ulp)"""
    from recombiner_amd import ops
    torch.manual_seed(8)
    W1 = torch.randn(64, 128, 5, 5, device=DEV) * 0.05
    b1 = torch.randn(64, device=DEV)
    W2 = torch.randn(64, 64, 3, 3, device=DEV) * 0.05
    W3 = torch.randn(16, 64, 3, 3, device=DEV) * 0.05
    _, _, Weff2, Weff3, pack = ops.upconv_weff_build(W1, b1, W2, W3, True)
    B = 37
    z1 = torch.randn(B, 8, 8, 64, device=DEV).bfloat16()
    h2 = torch.randn(B, 16, 16, 64, device=DEV).bfloat16()
    dz2 = torch.randn(B, 16, 16, 64, device=DEV).bfloat16()
    dpe = torch.randn(B, 32, 32, 16, device=DEV).bfloat16()
    b2, b3 = torch.randn(64, device=DEV), torch.randn(16, device=DEV)
    pairs = [(ops.upconv_fwd(z1, Weff2, b2, 8, 64, out_f32=False, preact=True, pack=pk),
              ops.upconv_fwd(h2, Weff3, b3, 16, 16, out_f32=False, linear_bf16=True, pack=pk),
              ops.upconv_dgrad(dz2, Weff2, z1, 8, 64, preact=True, pack=pk),
              ops.upconv_dgrad(dpe, Weff3, h2, 16, 16, pack=pk)) for pk in (None, pack)]
    for a, b in zip(*pairs):
        assert rel(a, b) < 1e-2
        assert float((a.float() - b.float()).abs().mean() / b.float().abs().mean()) < 1e-3


@pytest.mark.parametrize("name", ["protein", "video", "audio"])
def test_phase_form_routing_for_other_geometries(name):
    """16-bit mode, geometries without hand-written kernels: the 3-D and the un-patched 1-D presets go through the
    torch-level phase form, which must reproduce the plain module (forward and gradients) to fp32 rounding."""
    from recombiner_amd import config
    from recombiner_amd.upsample_fast import phase_form_preferred, phase_module
    c = config.configs[name]
    assert phase_form_preferred(c["data_dim"], c["patch"])
    torch.manual_seed(2)
    net = PM.Upsample(c["data_dim"], c["paddings"], c["layerwise_scale_factors"]).to(DEV)
    fast = phase_module(net)
    assert fast is not None and phase_module(net) is fast                   # cached, parameters shared
    assert len(list(net.parameters())) == 6
    lat = [c["pixel_sizes"][i] // c["upsample_factors"][i] * (c["patch_nums"][i] if c["patch"] else 1) for i in range(c["data_dim"])]
    x = torch.randn(3 if name == "protein" else 1, 128, *lat, device=DEV, requires_grad=True)   # (MIOpen's conv3d is slow)
    assert fast.window_gemm and fast.gemm_dtype == torch.bfloat16           # one bf16 GEMM per stage over 3^d-pixel windows
    y0 = net(x)
    g = torch.randn_like(y0)
    g0 = torch.autograd.grad(y0, [x] + list(net.parameters()), g)
    # the window-GEMM form itself, in the input's precision: the plain module to fp32 rounding
    fast.gemm_dtype = None
    try:
        y1 = fast(x)
        assert rel(y1, y0) < 1e-4
        g1 = torch.autograd.grad(y1, [x] + list(net.parameters()), g)
        assert max(rel(a, b) for a, b in zip(g1, g0)) < 3e-2                # MIOpen's own fp32 algorithms differ by ~2e-2
    finally:
        fast.gemm_dtype = torch.bfloat16
    # as routed in the 16-bit modes: bf16 operands and activations, fp32 accumulation
    y2 = fast(x)
    assert y2.dtype == torch.bfloat16 and rel(y2, y0) < 2e-2
    g2 = torch.autograd.grad(y2, [x] + list(net.parameters()), g.to(y2.dtype))
    # yardstick: the fp32 window-GEMM gradients (MIOpen's fp32 gradients are themselves 2-3 % off, see above)
    errs = [rel(a, b) for a, b in zip(g2, g1)]
    l2 = [float((a.double() - b.double()).norm() / b.double().norm()) for a, b in zip(g2, g1)]
    print(name, "bf16 window GEMMs: max-norm", ["%.2e" % e for e in errs], "L2", ["%.2e" % e for e in l2])
    # bf16 pre-activations flip the LeakyReLU branch of the ~0.3 % of elements that sit within rounding of zero (slope 1 vs
    # 0.01): an L2 error of ~sqrt(0.003) on the gradients that pass through an activation, as in any bf16 training
    assert max(errs) < 0.25 and max(l2) < 0.1 and max(l2[-2:]) < 1e-2


def test_fused_stage3_backward_equals_separate_kernels():
    """rcb_upconv_bwd_fused == rcb_upconv_dgrad + rcb_upconv_wgrad on the same inputs: dx and the bias gradient bit-identical
    (same MFMA sequence / same sums), the weight gradient equal to fp32 summation order (the fused kernel uses the 16 x 16 x 32
    MFMA since round 5) and both within fp32 rounding of the fp64 definition."""
    from recombiner_amd import ops
    torch.manual_seed(9)
    for B in (5, 300):
        W3 = torch.randn(2, 2, 64, 2, 2, 16, device=DEV) * 0.05
        h2 = torch.randn(B, 16, 16, 64, device=DEV).bfloat16()
        dpe = torch.randn(B, 32, 32, 16, device=DEV).bfloat16()
        dx0 = ops.upconv_dgrad(dpe, W3, h2, 16, 16)
        dw0, db0 = ops.upconv_wgrad(h2, dpe, 16, 16)
        dx1, dw1, db1 = ops.upconv_bwd_fused(dpe, W3, h2, 16, 16)
        assert torch.equal(dx0, dx1) and torch.equal(db0, db1)
        # weight gradient: the fused kernel contracts 32 positions per v_mfma_f32_16x16x32_bf16, the separate one 16 per
        # 32 x 32 x 16 MFMA -- the same bf16-exact products in another fp32 association.  Both against the fp64 definition
        # dW[ty, tx, ci, pa, pb, co] = sum_{b, i, j} x[b, i + pa + ty - 1, j + pb + tx - 1, ci] dy[b, 2 i + pa, 2 j + pb, co]
        xp = torch.nn.functional.pad(h2.double(), (0, 0, 1, 1, 1, 1))                       # zero halo
        ref = torch.zeros(2, 2, 64, 2, 2, 16, dtype=torch.float64, device=DEV)
        for ty in range(2):
            for tx in range(2):
                for pa in range(2):
                    for pb in range(2):
                        xs = xp[:, pa + ty:pa + ty + 16, pb + tx:pb + tx + 16, :]
                        ref[ty, tx, :, pa, pb, :] = torch.einsum("bijc,bijo->co", xs, dpe.double()[:, pa::2, pb::2, :])
        tol = 3e-6 * float(ref.abs().max())
        assert float((dw0.double() - ref).abs().max()) < tol and float((dw1.double() - ref).abs().max()) < tol
        assert torch.equal(dw1, ops.upconv_bwd_fused(dpe, W3, h2, 16, 16)[1])                 # fixed order: reproducible


@pytest.mark.parametrize("n,grid", [(2, (5, 7)), (1, (8, 12)), (2, (32, 48))])
def test_stitched_2d_grid_through_overlapping_tiles(n, grid):
    """patched 2-D presets (Kodak: the 8 x 12 patches of a photo form one 32 x 48 latent grid): stage 1 in phase form,
    stages 2 / 3 through the phase-conv kernels on tiles overlapping by one source pixel -- against the fp64 nn.Module
    on the whole grid (forward, gradients of the input and of every conv weight / bias).  Odd grid sizes exercise tiles
    that hang over the image border."""
    import copy
    from recombiner_amd.upsample_fast import hip_stitched_supported, stitched2d_module
    torch.manual_seed(4)
    net = PM.Upsample(2, [2, 1, 1], [4, 2, 2]).to(DEV)
    assert hip_stitched_supported(net, True, 2) and not hip_stitched_supported(net, False, 2)
    z = (0.1 * torch.randn(n, 128, *grid, device=DEV)).requires_grad_(True)
    net64 = copy.deepcopy(net).double()
    z64 = z.detach().double().requires_grad_(True)
    ref = net64(z64)
    g = torch.randn_like(ref)
    gr = torch.autograd.grad(ref, [z64] + list(net64.parameters()), g)
    out = stitched2d_module(net)(z)
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (n, 16, 16 * grid[0], 16 * grid[1])
    go = torch.autograd.grad(out, [z] + list(net.parameters()), g.to(out.dtype))
    e_fwd = rel(out, ref)
    errs = [rel(a, b) for a, b in zip(go, gr)]
    print("stitched %s: fwd %.2e  dz %.2e  dW1 %.2e db1 %.2e dW2 %.2e db2 %.2e dW3 %.2e db3 %.2e" % (grid, e_fwd, *errs))
    # bf16 operands and bf16 intermediate images in all three stages (stage 1 under autocast): max-norm errors up to
    # ~1e-1 on the smallest grids, where a handful of elements set the norm
    assert e_fwd < 2.5e-2 and max(errs) < 0.13
    # the tiling itself is exact: no seam between tiles.  Compare in the interior and on tile borders separately
    d = (out.float() - ref.float()).abs()
    seam = d[:, :, 29:33].max() if d.shape[2] > 33 else d.max()      # output rows around the first stage-3 tile border
    assert float(seam) <= float(d.max()) and float(d.max()) < 2.5e-2 * float(ref.abs().max())


@pytest.mark.parametrize("n,H,W,C,G", [(2, 13, 9, 64, 8), (1, 32, 48, 16, 16), (3, 7, 30, 64, 16)])
def test_tile_kernels_match_their_definitions(n, H, W, C, G):
    """rcb_tile_gather / _crop / _fold against plain tensor restatements: exact (pure data movement; the fold's sums are
    exact on small integers), incl. sizes that are no multiple of the tile step and error returns."""
    from recombiner_amd import ops
    gen = torch.Generator(device=DEV).manual_seed(5)
    img = torch.randint(-8, 9, (n, H, W, C), device=DEV, generator=gen).to(torch.bfloat16)
    Ty, Tx = ops.tile_count(H, G), ops.tile_count(W, G)

    def windows(x, T, step, off):
        plane = x.new_zeros(n, (Ty - 1) * step + T, (Tx - 1) * step + T, x.shape[-1])
        hh, ww = min(x.shape[1], plane.shape[1] - off), min(x.shape[2], plane.shape[2] - off)
        plane[:, off:off + hh, off:off + ww] = x[:, :hh, :ww]
        s0, s1, s2, _ = plane.stride()
        return plane.as_strided((n, Ty, Tx, T, T, x.shape[-1]), (s0, step * s1, step * s2, s1, s2, 1)).reshape(n * Ty * Tx, T, T, -1)

    # source tiles: G pixels every G-1, starting at -1
    t = ops.tile_gather(img, Ty, Tx, G, G - 1, 1, 0)
    assert torch.equal(t, windows(img, G, G - 1, 1))
    # upstream-gradient tiles of the 2x output image: 2G pixels every 2G-2 starting at -2, outermost ring zeroed
    out = torch.randint(-8, 9, (n, 2 * H, 2 * W, C), device=DEV, generator=gen).to(torch.bfloat16)
    d = ops.tile_gather(out, Ty, Tx, 2 * G, 2 * G - 2, 2, 1)
    ref = windows(out, 2 * G, 2 * G - 2, 2).clone()
    ref[:, 0] = 0
    ref[:, -1] = 0
    ref[:, :, 0] = 0
    ref[:, :, -1] = 0
    assert torch.equal(d, ref)
    # crop is the inverse of that gather on the valid rows: round trip
    assert torch.equal(ops.tile_crop(d, n, 2 * H, 2 * W, Ty, Tx, 1), out)
    # fold = adjoint of the ring-0 gather: every pixel collects all tile elements that map to it
    tiles = torch.randint(-8, 9, (n * Ty * Tx, G, G, C), device=DEV, generator=gen).to(torch.bfloat16)
    plane = torch.zeros(n, (Ty - 1) * (G - 1) + G, (Tx - 1) * (G - 1) + G, C, device=DEV)
    tv = tiles.view(n, Ty, Tx, G, G, C).float()
    for ty in range(Ty):
        for tx in range(Tx):
            plane[:, ty * (G - 1):ty * (G - 1) + G, tx * (G - 1):tx * (G - 1) + G] += tv[:, ty, tx]
    assert torch.equal(ops.tile_fold(tiles, n, H, W, Ty, Tx, 1), plane[:, 1:H + 1, 1:W + 1].to(torch.bfloat16))
    with pytest.raises(ops.RcbError):
        ops.tile_crop(d, n, 2 * H + 64, 2 * W, Ty, Tx, 1)             # image larger than the tiles cover
    with pytest.raises(ops.RcbError):
        ops.tile_gather(img.float(), Ty, Tx, G, G - 1, 1, 0)          # bf16 only
    with pytest.raises(ops.RcbError):
        ops.tile_fold(tiles, n, (Ty + 1) * G, W, Ty, Tx, 1)


@pytest.mark.parametrize("shape", [(3, 7, 64), (2, 5, 6, 64), (2, 3, 4, 5, 16), (1, 1, 8, 8, 128)])
def test_window_gather_and_fold_kernels(shape):
    """rcb_window_gather == the 3^d shifted slices of the zero-padded grid side by side; rcb_window_fold == its adjoint
    (exact on small integers), for 1, 2 and 3 windowed axes."""
    import itertools
    import torch.nn.functional as F
    from recombiner_amd import ops
    gen = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randint(-8, 9, shape, device=DEV, generator=gen).to(torch.bfloat16)
    dd, g = len(shape) - 2, list(shape[1:-1])
    offs = list(itertools.product(range(3), repeat=dd))
    xp = F.pad(x, [0, 0] + [1, 1] * dd)
    ref = torch.cat([xp[(slice(None),) + tuple(slice(o[d], o[d] + g[d]) for d in range(dd))] for o in offs], dim=-1)
    cols = ops.window_gather(x)
    assert torch.equal(cols, ref.reshape(-1, ref.shape[-1]))
    d = torch.randint(-4, 5, cols.shape, device=DEV, generator=gen).to(torch.bfloat16)
    dxp = torch.zeros_like(xp, dtype=torch.float32)
    dv = d.view(shape[0], *g, len(offs), shape[-1]).float()
    for k, o in enumerate(offs):
        dxp[(slice(None),) + tuple(slice(o[i], o[i] + g[i]) for i in range(dd))] += dv.select(-2, k)
    want = dxp[(slice(None),) + tuple(slice(1, 1 + g[i]) for i in range(dd))].to(torch.bfloat16)
    assert torch.equal(ops.window_fold(d, shape), want)
    with pytest.raises(ops.RcbError):
        ops.window_gather(x.float())
    with pytest.raises(ops.RcbError):
        ops.window_fold(d[:-1], shape)


# ---------------------------------------------------------------------------------------------------
# direct sub-pixel convolutions for 1-D / 3-D (and 2-D) grids: rcb_phaseconv_*
# ---------------------------------------------------------------------------------------------------
def _stage_ref(x, W, b, nd, leaky):
    """the reference's stage, channel-last: nearest-upsample(2) -> conv(3, pad 1) (-> LeakyReLU), fp64"""
    import torch.nn.functional as F
    conv = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[nd]
    y = conv(F.interpolate(x.movedim(-1, 1), scale_factor=2, mode="nearest"), W, b, padding=1).movedim(1, -1)
    return F.leaky_relu(y, 0.01) if leaky else y


@pytest.mark.parametrize("shape,cout,leaky", [((3, 70), 64, True), ((2, 45), 16, False), ((2, 5, 33), 64, True), ((2, 3, 4, 40), 64, True),
                                              ((1, 2, 3, 32), 16, False), ((2, 6, 32, 32), 64, True),
                                              ((300, 200), 64, True), ((211, 400), 16, False),     # audio stages: every workgroup walks
                                              ((3, 9, 70), 16, False), ((5, 37, 96), 64, True), ((2, 64, 40), 16, False)])   # 2-D: ragged row blocks
def test_phaseconv_forward_and_data_gradient(shape, cout, leaky):
    """rcb_phaseconv_fwd / _dgrad on 1-D, 2-D and 3-D grids (ragged last tiles, borders on every axis) against the stage
    as the reference defines it (nearest-upsample + ConvNd, fp64) evaluated on the same bf16-rounded operands: what differs
    is the bf16 rounding of the pre-summed phase weights (<= 2^-9 relative per weight) and the fp32 accumulation order."""
    from recombiner_amd import ops
    torch.manual_seed(sum(shape) + cout)
    nd = len(shape) - 1
    B, g = shape[0], list(shape[1:])
    W = (torch.randn(cout, 64, *([3] * nd), device=DEV) * (0.5 / (64 * 3 ** nd) ** 0.5))
    b = torch.randn(cout, device=DEV) * 0.1
    x = torch.nn.functional.leaky_relu(torch.randn(B, *g, 64, device=DEV), 0.01).bfloat16()      # an activation tensor
    ff, fd = ops.phaseconv_pack(W)
    y = ops.phaseconv_fwd(x, ff, b, cout, leaky)
    assert list(y.shape) == [B] + [2 * v for v in g] + [cout]
    x64 = x.double().requires_grad_(True)
    ref = _stage_ref(x64, W.double(), b.double(), nd, leaky)
    e_fwd = rel(y, ref)
    assert e_fwd < 6e-3, e_fwd                 # bf16 output + bf16 effective weights
    # data gradient of the linear part, times LeakyReLU'(stage input): what the kernel returns for upstream dy
    dy = (torch.randn_like(y.float()) * 0.1).bfloat16()
    lin = _stage_ref(x64, W.double(), b.double(), nd, False)
    gx, = torch.autograd.grad(lin, [x64], dy.double())
    want = gx * torch.where(x.double() > 0, 1.0, 0.01)
    dx = ops.phaseconv_dgrad(dy, fd, x)
    e_bwd = rel(dx, want)
    assert e_bwd < 8e-3, e_bwd
    # weight and bias gradient: contraction over every position of the batch
    Wd, bd = W.double().requires_grad_(True), b.double().requires_grad_(True)
    lin2 = _stage_ref(x.double(), Wd, bd, nd, False)
    gW, gb = torch.autograd.grad(lin2, [Wd, bd], dy.double())
    dW, db = ops.phaseconv_wgrad(x, dy)
    e_w, e_b = rel(dW, gW), rel(db, gb)
    assert e_w < 2e-3 and e_b < 1e-5, (e_w, e_b)             # identical bf16 operands: fp32 accumulation order only
    dW2, db2 = ops.phaseconv_wgrad(x, dy)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)      # no atomics: bitwise reproducible
    print("phaseconv %s cout %d: fwd %.1e dgrad %.1e wgrad %.1e dbias %.1e" % (shape, cout, e_fwd, e_bwd, e_w, e_b))


@pytest.mark.parametrize("B,g", [(3, 70), (2, 33), (5, 6), (4, 1000), (37, 50)])
def test_stage1_1d_direct_kernels(B, g):
    """rcb_stage1_1d_fwd / _dgrad / _wgrad (stage 1 of the 1-D net without the window matrix) against the stage as the reference
    defines it (nearest-upsample(4) -> Conv1d(128 -> 64, 5, pad 2) -> LeakyReLU, fp64) on the same bf16-rounded latent grid:
    what differs is the bf16 rounding of the pre-summed phase weights, of the stored pre-activation, and the fp32 accumulation
    order; and against the window-GEMM form they replace (same Wbig, same roundings)."""
    import torch.nn.functional as F
    from recombiner_amd import ops
    from recombiner_amd.upsample_fast import PhaseStage
    torch.manual_seed(B * 1000 + g)
    W = torch.randn(64, 128, 5, device=DEV) * (0.5 / (128 * 5) ** 0.5)
    b = torch.randn(64, device=DEV) * 0.1
    x = (torch.randn(B, g, 128, device=DEV) * 0.5).bfloat16().float()               # values the kernel's rounding keeps
    st = PhaseStage(4, 5, 2, 1)
    wbig = ops.phase_bigweight(W, st.f, st.k, st.pad, torch.bfloat16)
    x1 = ops.stage1_1d_fwd(x, wbig, b)
    assert tuple(x1.shape) == (B, 4 * g, 64) and x1.dtype == torch.bfloat16

    def ref_lin(x64, W64, b64):
        return F.conv1d(F.interpolate(x64.movedim(-1, 1), scale_factor=4, mode="nearest"), W64, b64, padding=2).movedim(1, -1)
    x64 = x.double().requires_grad_(True)
    Wd, bd = W.double().requires_grad_(True), b.double().requires_grad_(True)
    lin = ref_lin(x64, Wd, bd)
    e_f = rel(x1, F.leaky_relu(lin, 0.01))
    assert e_f < 8e-3, e_f                                   # bf16 pre-activation + bf16 output + bf16 phase weights
    dz = (torch.randn(B, 4 * g, 64, device=DEV) * 0.1).bfloat16()
    gx, gW, gb = torch.autograd.grad(lin, [x64, Wd, bd], dz.double())
    dx = ops.stage1_1d_dgrad(dz, wbig)
    assert dx.dtype == torch.float32 and tuple(dx.shape) == (B, g, 128)
    e_d = rel(dx, gx)
    assert e_d < 6e-3, e_d
    dwbig, db = ops.stage1_1d_wgrad(x, dz)
    dW = ops.phase_bigweight_grad(dwbig, tuple(W.shape), st.f, st.k, st.pad)
    e_w, e_b = rel(dW, gW), rel(db, gb)
    assert e_w < 2e-3 and e_b < 1e-5, (e_w, e_b)             # identical bf16 operands: fp32 accumulation order only
    dwbig2, db2 = ops.stage1_1d_wgrad(x, dz)
    assert torch.equal(dwbig, dwbig2) and torch.equal(db, db2)                  # no atomics: bitwise reproducible
    # the window-GEMM form with the same Wbig: accumulation order and the bias (bf16-rounded there): a bf16 ulp or two of the result
    cols = ops.window_gather(x.bfloat16())
    z_w = torch.addmm(b.bfloat16().repeat(4), cols, wbig).view(B, g, 4, 64).reshape(B, 4 * g, 64)
    x1_w = F.leaky_relu(z_w, 0.01)
    assert rel(x1, x1_w) < 8e-3
    print("stage-1 direct %dx%d: fwd %.1e dgrad %.1e wgrad %.1e dbias %.1e" % (B, g, e_f, e_d, e_w, e_b))


@pytest.mark.parametrize("dd,f,k,pad,cin,cout", [(1, 4, 5, 2, 16, 8), (2, 4, 5, 2, 24, 16), (3, 4, 5, 2, 8, 8), (1, 6, 5, 2, 16, 8),
                                                 (3, 2, 3, 1, 8, 16), (2, 2, 3, 1, 64, 64)])
def test_phase_bigweight_kernels_match_the_einsum_form(dd, f, k, pad, cin, cout):
    """rcb_phase_bigweight / _grad == PhaseStage.big_weight (einsums with 0 / 1 tensors) and its autograd gradient: the window-
    GEMM weight of a nearest-upsample(f) -> conv(k, pad) stage for every stage geometry of the reference nets; and the bf16
    result is the rounded fp32 one."""
    from recombiner_amd import ops
    from recombiner_amd.upsample_fast import PhaseStage
    torch.manual_seed(3)
    st = PhaseStage(f, k, pad, dd)
    W = torch.randn(cout, cin, *([k] * dd), device=DEV)
    Wr = W.clone().requires_grad_(True)
    ref = st.big_weight(Wr)
    got = ops.phase_bigweight(W, st.f, k, pad, torch.float32)
    assert got.shape == ref.shape
    torch.testing.assert_close(got, ref.detach(), rtol=1e-5, atol=1e-5)      # (sums of up to 2^d taps in another order)
    assert torch.equal(ops.phase_bigweight(W, st.f, k, pad, torch.bfloat16), got.bfloat16())
    dbig = torch.randn_like(ref)
    (gref,) = torch.autograd.grad(ref, [Wr], dbig)
    gw = ops.phase_bigweight_grad(dbig, tuple(W.shape), st.f, k, pad)
    torch.testing.assert_close(gw, gref, rtol=2e-5, atol=2e-5)
    gw16 = ops.phase_bigweight_grad(dbig.bfloat16(), tuple(W.shape), st.f, k, pad)
    torch.testing.assert_close(gw16, ops.phase_bigweight_grad(dbig.bfloat16().float(), tuple(W.shape), st.f, k, pad), rtol=0, atol=0)
