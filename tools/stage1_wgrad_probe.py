"""Stage-1 weight gradient dWeff1 = lpe^T dz1 (512 x 4096 from 4096 rows): one GEMM vs row-sliced batched GEMM + sum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recombiner_amd import tuning
tuning.enable_tuned_gemms()
lpe = torch.randn(4096, 512, device="cuda").bfloat16()
dz = (torch.randn(4096, 4096, device="cuda") * 1e-3).bfloat16()
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ref = torch.mm(lpe.t(), dz, out_dtype=torch.float32)
print("one GEMM fp32 out: %.1f us" % t(lambda: torch.mm(lpe.t(), dz, out_dtype=torch.float32)))
print("one GEMM bf16 out: %.1f us" % t(lambda: lpe.t() @ dz))
for S in (2, 4, 8):
    a = lpe.view(S, 4096 // S, 512).transpose(1, 2)
    b = dz.view(S, 4096 // S, 4096)
    f = lambda: torch.bmm(a, b, out_dtype=torch.float32)
    g = lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0)
    print("S=%d: bmm %.1f us, bmm + sum %.1f us, max diff %.2e" % (S, t(f), t(g), float((g() - ref).abs().max())))

# ---- the data gradient dlpe = dz1 @ Weff1^T ([4096, 4096] x [4096, 512], fp32 out): K-sliced
W = torch.randn(512, 4096, device="cuda").bfloat16()
ref2 = torch.mm(dz, W.t(), out_dtype=torch.float32)
print("dlpe one GEMM: %.1f us" % t(lambda: torch.mm(dz, W.t(), out_dtype=torch.float32)))
for S in (2, 4):
    a = dz.view(4096, S, 4096 // S).transpose(0, 1)                  # [S, 4096, K/S]
    b = W.view(512, S, 4096 // S).permute(1, 2, 0)                   # [S, K/S, 512]
    f = lambda: torch.bmm(a, b, out_dtype=torch.float32)
    g = lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0)
    print("dlpe S=%d: bmm %.1f us, + sum %.1f us, max diff %.2e" % (S, t(f), t(g), float((g() - ref2).abs().max())))
