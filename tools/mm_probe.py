"""probe: batched split-bf16 A transform -- [3, N, 3W] @ [3, 3W, W] -> fp32, written into column slices of [N, D]"""
import torch
dev = "cuda"
torch.manual_seed(0)
N, D, W = 4096, 3267, 1056
L = (torch.randn(3, N, 3 * W, device=dev) * 0.03).bfloat16()
R = (torch.randn(3, 3 * W, W, device=dev) / W ** 0.5).bfloat16()
Rt = (torch.randn(3, W, 3 * W, device=dev) / W ** 0.5).bfloat16()
out = torch.zeros(N, D, device=dev)
view = out[:, :3 * W].view(N, 3, W).permute(1, 0, 2)          # [3, N, W], strides (W, D, 1)


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


def per_layer():
    for l in range(3):
        torch.mm(L[l], R[l], out_dtype=torch.float32, out=out[:, l * W:(l + 1) * W])


def per_layer_t():
    for l in range(3):
        torch.mm(L[l], Rt[l].t(), out_dtype=torch.float32, out=out[:, l * W:(l + 1) * W])


ref = torch.stack([L[l].double() @ R[l].double() for l in range(3)])
print("3 x mm into slices (NN): %.1f us" % timed(per_layer))
print("3 x mm into slices (NT): %.1f us" % timed(per_layer_t))
for name, fn in (("bmm -> new tensor", lambda: torch.bmm(L, R, out_dtype=torch.float32)),
                 ("bmm(out=strided view)", lambda: torch.bmm(L, R, out_dtype=torch.float32, out=view)),
                 ("bmm NT -> new tensor", lambda: torch.bmm(L, Rt.transpose(1, 2), out_dtype=torch.float32)),
                 ("bmm NT (out=strided view)", lambda: torch.bmm(L, Rt.transpose(1, 2), out_dtype=torch.float32, out=view))):
    try:
        out.zero_()
        r = fn()
        torch.cuda.synchronize()
        if "NT" not in name:
            got = view if "view" in name else r
            err = float((got.double() - ref).abs().max() / ref.abs().max())
        else:
            err = -1
        print("%-28s ok err %.1e  %.1f us" % (name, err, timed(fn)))
    except Exception as e:
        print("%-28s FAILED %s" % (name, repr(e)[:160]))
