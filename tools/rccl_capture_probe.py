"""Can an RCCL all-reduce sit INSIDE a captured HIP graph on this stack (PyTorch ProcessGroupNCCL on ROCm)?  Captures
[kernel, all_reduce(async), kernel on the main stream, wait, kernel], replays it three times and checks the values; also with the
collective issued from a side stream (the forked training step's structure).
    python tools/rccl_capture_probe.py                      one rank, one GPU
    python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node N tools/rccl_capture_probe.py
bench.py's launcher runs the N-rank form under a hard timeout before the measured run and lets the training step capture its
collectives (RCB_CAPTURE_COLLECTIVES=1) only if every rank printed RCCL_CAPTURE_OK; exit code 0 = ok."""
import os
import sys
import torch
import torch.distributed as dist

rank = int(os.environ.get("RANK", "0"))
ws = int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29713")
dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=dev)
n = 1 << 20
x = torch.ones(n, device=dev)
y = torch.zeros(n, device=dev)
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
        # per replay x -> 2 x summed over the ranks;  y += 1 + x
        f = 2.0 * ws
        ex, ey = f ** 3, 3.0 + f + f ** 2 + f ** 3
        print(f"rank {rank}/{ws} {variant} captured and replayed: {float(x[0])} {float(y[0])} expected {ex} {ey}", flush=True)
        ok = ok and float(x[0]) == ex and float(y[0]) == ey and float(x[n - 1]) == ex and float(y[n - 1]) == ey
    except Exception as e:
        ok = False
        print(f"rank {rank}/{ws} {variant} FAILED:", repr(e)[:300], flush=True)
        torch.cuda.synchronize()
flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
ok = bool(int(flag.item()))
print("RCCL_CAPTURE_OK" if ok else "RCCL_CAPTURE_NOT_OK", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
