"""measured error of the 16-bit SIREN kernels against the fp32 oracle at SIREN-INIT weight scale (the scale training runs at) and
at the 3x stress scale of tests/test_hip_kernels.py::test_siren_16bit_operands -- the numbers the test limits are set from"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_hip_kernels as T
from recombiner_amd import ops
from recombiner_amd.ops import SirenMeta
for scale in (1.0, 3.0):
    for prec in (1, 2):
        for hidden in (32, 48, 64):
            worst = np.zeros(4)
            for case in T.SIREN_CASES[:5]:
                S, N, P, C = case["S"], case["N"], case["P"], case["C"]
                dims, D, xf, pe, wv, y = T._siren_case(seed=1, hidden=hidden, **case)
                wv = wv * (scale / 3.0)
                meta = SirenMeta(samples=S, n_pix=P, fourier_dim=case["F"], pe_dim=case["E"], n_hidden=case["n_hidden"], hidden=hidden, out_dim=C, precision=prec)
                pe_r, wv_r = pe.clone().requires_grad_(True), wv.clone().requires_grad_(True)
                y_ref = T._oracle_mlp(dims, xf, pe_r, wv_r, S)
                tgt = y.repeat_interleave(S, 0)
                sc = 1.0 / (S * P * C)
                (((y_ref - tgt) ** 2).sum() * sc).backward()
                try:
                    y_hip = ops.siren_fwd(T.g(xf), T.g(pe), T.g(wv), meta)
                    sse, dw, dpe = ops.siren_loss_bwd(T.g(xf), T.g(pe), T.g(wv), T.g(y), sc, meta)
                except Exception as e:
                    print("skip", hidden, case, str(e)[:80]); continue
                e = np.array([T.rel_err(y_hip, y_ref.detach()), T.rel_err(sse, ((y_ref.detach() - tgt) ** 2).sum((1, 2))), T.rel_err(dw, wv_r.grad), T.rel_err(dpe, pe_r.grad)])
                worst = np.maximum(worst, e)
            print("scale %.0fx prec %d hidden %d: worst rel err  y %.2e  sse %.2e  dW %.2e  dpe %.2e" % (scale, prec, hidden, *worst), flush=True)
