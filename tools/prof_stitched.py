"""fwd + bwd of the stitched 2-D upsampling path alone (for rocprofv3): python tools/prof_stitched.py [photos]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import prior_model as PM
from recombiner_amd.upsample_fast import stitched2d_module

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(0)
net = PM.Upsample(2, [2, 1, 1], [4, 2, 2]).cuda()
f = stitched2d_module(net)
x = torch.randn(nb, 128, 32, 48, device="cuda", requires_grad=True)
g = torch.randn(nb, 16, 512, 768, device="cuda").bfloat16()
for _ in range(12):
    y = f(x)
    torch.autograd.grad(y, [x] + list(net.parameters()), g)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    y = f(x)
    torch.autograd.grad(y, [x] + list(net.parameters()), g)
e1.record()
torch.cuda.synchronize()
print("stitched fwd+bwd, %d photos: %.3f ms" % (nb, e0.elapsed_time(e1) / 10))
