"""Can an RCCL all-reduce sit INSIDE a captured HIP graph on this stack (PyTorch ProcessGroupNCCL on ROCm)?  One rank, one GPU:
capture [kernel, all_reduce(async), kernel on the main stream, wait, kernel], replay it, check the values; also with the
collective issued from a side stream (the forked step's structure).   python tools/rccl_capture_probe.py"""
import os
import sys
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29713")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(1 << 20, device=dev)
y = torch.zeros(1 << 20, device=dev)
dist.all_reduce(x)                       # communicator set-up outside the capture
torch.cuda.synchronize()
side = torch.cuda.Stream()
ok = True
for variant in ("same-stream", "side-stream"):
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            x.mul_(2.0)
            if variant == "same-stream":
                h = dist.all_reduce(x, async_op=True)
                y.add_(1.0)
                h.wait()
            else:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    dist.all_reduce(x)
                y.add_(1.0)
                torch.cuda.current_stream().wait_stream(side)
            y.add_(x)
        x.fill_(1.0)
        y.zero_()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        # x: 1 -> 2 -> 4 -> 8;  y: (1 + 2) + (1 + 4) + (1 + 8) = 17
        print(variant, "captured and replayed:", float(x[0]), float(y[0]), "expected 8.0 17.0", flush=True)
        ok = ok and float(x[0]) == 8.0 and float(y[0]) == 17.0
    except Exception as e:
        ok = False
        print(variant, "FAILED:", repr(e)[:300], flush=True)
        torch.cuda.synchronize()
print("RCCL_CAPTURE_OK" if ok else "RCCL_CAPTURE_NOT_OK")
dist.destroy_process_group()
