"""Stage-2 forward / stage-2 data gradient / stage-3 forward, stand-alone launch times (non-pack weight path):
    python tools/ab_pin.py        (RCB_LIB selects another build for a same-box A/B)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from recombiner_amd import ops
B=4096; dev="cuda"
z1=torch.randn(B,8,8,64,device=dev).bfloat16(); w2=torch.randn(2,2,64,2,2,64,device=dev)*0.05; b2=torch.randn(64,device=dev)*0.1
dy2=(torch.randn(B,16,16,64,device=dev)*1e-3).bfloat16()
h2=torch.randn(B,16,16,64,device=dev).bfloat16(); w3=torch.randn(2,2,64,2,2,16,device=dev)*0.05; b3=torch.zeros(16,device=dev)
from recombiner_amd import upsample_fast as UF
def t(fn,n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
# (the pack path, as the training step uses it: fragments pre-ordered by rcb_upconv_weff_build)
pack=None
try:
    pack=ops.upconv_frag_pack_for_test(w2,w3) if hasattr(ops,"upconv_frag_pack_for_test") else None
except Exception as e: print("no pack helper", e)
print("fwd2 %.1f us  dgrad2 %.1f us  fwd3 %.1f us" % (t(lambda: ops.upconv_fwd(z1,w2,b2,8,64,False,preact=True,pack=pack)), t(lambda: ops.upconv_dgrad(dy2,w2,z1,8,64,preact=True,pack=pack)), t(lambda: ops.upconv_fwd(h2,w3,b3,16,16,False,linear_bf16=True,pack=pack))))
