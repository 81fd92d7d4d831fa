"""Rehearsal of the sharded training step on fewer GPUs than ranks (gloo; the ranks share device 0):
   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dist_rehearsal.py
Each rank trains its own INRs; the shared mappings must end up identical on all ranks, and the segmented-graph replay path
(async all-reduce between captured segments) must reproduce the eager path.  Prints one line 'REHEARSAL OK ...'."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import config, utils  # noqa: E402
from recombiner_amd import prior_model as PM  # noqa: E402


def run(use_graph, rank, n, steps):
    cfg = config.configs["cifar"]
    X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, 3, seed=rank)
    m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                         cfg["upsample_factors"], cfg["latent_dim"], False, None, None, random_seed=42 + rank, device="cuda")
    m.precision = 1
    m.use_graph = use_graph
    torch.manual_seed(123)
    lt = PM.LinearTransform(m.dims).cuda()
    torch.manual_seed(124)
    up = PM.Upsample(2, cfg["paddings"], cfg["layerwise_scale_factors"]).cuda()
    torch.manual_seed(1000 + rank)
    D, s0 = m._d_net, 0.0211547
    pri = [torch.zeros(D).cuda(), torch.full((D,), s0).cuda(), torch.zeros(2, 2, 128).cuda(),
           torch.full((2, 2, 128), s0).cuda(), None, None, None, None]
    mse, kl, elbo = m.train(steps, 1e-3, X.cuda()[None].expand(n, -1, -1), Y.cuda(), *pri, lt, up, 1e-8,
                            training_mappings=True)
    flat = torch.cat([p.detach().reshape(-1) for p in list(lt.parameters()) + list(up.parameters())])
    return flat, m.loc.detach().clone(), mse


if __name__ == "__main__":
    dist.init_process_group(os.environ.get("RCB_DIST_BACKEND", "gloo"))
    rank, ws = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    n, steps = 48, 10
    eager, loc_e, mse_e = run(False, rank, n, steps)
    graph, loc_g, mse_g = run(True, rank, n, steps)
    # 1. mappings identical on every rank (they saw the same summed gradients)
    for name, v in (("eager", eager), ("graph", graph)):
        gathered = [torch.empty_like(v) for _ in range(ws)]
        dist.all_gather(gathered, v)
        for other in gathered[1:]:
            assert torch.equal(gathered[0], other), "%s: mappings differ between ranks" % name
    # 2. segmented-graph replay == eager stepping (same seeds; the bf16 mode is deterministic up to the noise stream, which
    #    is drawn identically in both modes)
    rel = float((eager - graph).abs().max() / eager.abs().max())
    rel_loc = float((loc_e - loc_g).abs().max() / loc_e.abs().max())
    assert rel < 5e-3 and rel_loc < 5e-2, (rel, rel_loc)
    # 3. finite, and both paths report the same loss
    assert torch.isfinite(graph).all() and abs(mse_g - mse_e) <= 2e-2 * abs(mse_e), (mse_e, mse_g)
    if rank == 0:
        print("REHEARSAL OK ws=%d  graph-vs-eager rel %.2e (mappings) %.2e (loc)  mse %.4f / %.4f" % (ws, rel, rel_loc, mse_e, mse_g),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()
