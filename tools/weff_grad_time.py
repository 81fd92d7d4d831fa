import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from recombiner_amd import ops
d1 = (torch.randn(512, 4096, device="cuda") * 1e-3).bfloat16()
d1f = d1.float()
d2 = torch.randn(65536, device="cuda"); d3 = torch.randn(16384, device="cuda")
pp = torch.randn(256, 64, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("bf16 + db1 us", t(lambda: ops.upconv_weff_grad(d1, d2, d3, pp)))
print("bf16 no db1 us", t(lambda: ops.upconv_weff_grad(d1, d2, d3)))
print("fp32 no db1 us", t(lambda: ops.upconv_weff_grad(d1f, d2, d3)))
print("empty alloc x3 us", t(lambda: (torch.empty(64,128,5,5,device="cuda"), torch.empty(64,64,3,3,device="cuda"), torch.empty(16,64,3,3,device="cuda"))))
