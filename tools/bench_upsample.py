"""Times fwd+bwd of the upsampling net variants at the bench shape (N=4096 CIFAR)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import prior_model as PM
from recombiner_amd.upsample_fast import UpsampleFast
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
net = PM.Upsample(2, [2, 1, 1], [4, 2, 2]).to(dev)
fast = UpsampleFast(net)
x = torch.randn(N, 128, 2, 2, device=dev, requires_grad=True)
g = torch.randn(N, 16, 32, 32, device=dev)

def run(fn, autocast=None, reps=5, cl=False):
    xi = x.contiguous(memory_format=torch.channels_last) if cl else x
    def once():
        if autocast:
            with torch.autocast("cuda", dtype=autocast):
                y = fn(xi)
        else:
            y = fn(xi)
        return torch.autograd.grad(y, [x] + list(net.parameters()), g.to(y.dtype))
    for _ in range(2):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r

ref_ms, ref = run(net)
print("nn.Module fp32           %.2f ms" % ref_ms)
for name, fn, ac, cl in [("nn.Module fp32 channels_last", net, None, True), ("nn.Module bf16 autocast", net, torch.bfloat16, False),
                         ("phase form fp32", fast, None, False), ("phase form bf16 autocast", fast, torch.bfloat16, False)]:
    ms, r = run(fn, ac, cl=cl)
    err = max(((a.float() - b).abs().max() / (b.abs().max() + 1e-20)).item() for a, b in zip(r, ref))
    print("%-30s %.2f ms   max rel grad err %.2e" % (name, ms, err))
