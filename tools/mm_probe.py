"""probe: bf16 x bf16 -> fp32 GEMM written into a strided column slice (the A-transform output layout)"""
import torch
dev = "cuda"
torch.manual_seed(0)
N, D, W = 4096, 3267, 1056
L = (torch.randn(N, 3 * W, device=dev) * 0.03).bfloat16()
R = (torch.randn(3 * W, W, device=dev) / W ** 0.5).bfloat16()
Rt = (torch.randn(W, 3 * W, device=dev) / W ** 0.5).bfloat16()
out = torch.zeros(N, D, device=dev)
ref = (L.double() @ R.double())


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, fn in (("mm(out_dtype, out=slice)", lambda: torch.mm(L, R, out_dtype=torch.float32, out=out[:, 1056:2112])),
                 ("mm(out_dtype) + copy_", lambda: out[:, 1056:2112].copy_(torch.mm(L, R, out_dtype=torch.float32))),
                 ("mm(out_dtype) TN into slice", lambda: torch.mm(L, Rt.t(), out_dtype=torch.float32, out=out[:, 1056:2112]))):
    try:
        out.zero_()
        fn()
        torch.cuda.synchronize()
        err = float((out[:, 1056:2112].double() - ref).abs().max() / ref.abs().max()) if "TN" not in name else -1
        print("%-32s ok  err %.1e  %.1f us" % (name, err, timed(fn)))
    except Exception as e:
        print("%-32s FAILED: %s" % (name, repr(e)[:200]))
