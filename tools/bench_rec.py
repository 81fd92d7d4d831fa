"""A* candidate scoring (rcb_rec_score_argmax) on the CIFAR test batch of main_compression.py: 500 rows, K = 65 536, one
group per row per round.  Times the exact (op-for-op) scorer, the certified fast scorer and a whole encode round
(group selection + scoring + commit, no host round trip), and prices them against the fp64 vector peak with SURVEY
section 8(d)'s algorithmic count of 14 K g flops per job.   python tools/bench_rec.py [rows] [long]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import config, ops, utils  # noqa: E402
from recombiner_amd import prior_model as PM, test_model as TM  # noqa: E402

FP64_PEAK = 78.6e12


def build(N, bits_per_param, dev="cuda"):
    cfg = config.configs["cifar"]
    dims = [cfg["input_dim"]] + cfg["hidden_dims"] + [cfg["output_dim"]]
    torch.manual_seed(123)
    lt = PM.LinearTransform(dims).to(dev)
    torch.manual_seed(124)
    up = PM.Upsample(2, cfg["paddings"], cfg["layerwise_scale_factors"]).to(dev)
    D = 3267 + 512
    rng = np.random.RandomState(0)
    bits = rng.gamma(0.7, bits_per_param / 0.7, size=D).astype(np.float32)
    gi, gs, ge, g2p, p2g, G, gk, w = PM.get_grouping_by_kl(bits)
    p_loc, p_ls = torch.zeros(D), torch.full((D,), -2.0)
    m = TM.TestBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], N, cfg["upsample_factors"], cfg["latent_dim"],
                        2, cfg["pixel_sizes"], False, None, None, "cifar", linear_transform=lt, upsample_net=up,
                        p_loc=p_loc[p2g], p_log_scale=p_ls[p2g], init_log_scale=torch.full((D,), -4.0), param_to_group=p2g,
                        group_to_param=g2p, n_groups=G, group_start_index=gs, group_end_index=ge, group_idx=gi, device=dev,
                        initial_beta=1e-8)
    with torch.no_grad():
        m.loc.add_(0.02 * torch.randn_like(m.loc))
    return m


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    long_groups = len(sys.argv) > 2 and sys.argv[2] == "long"
    m = build(N, 0.05 if long_groups else 4.0)
    lv = m._l1
    K = 65536
    t0 = time.perf_counter()
    tables = m._rec_tables(lv, lv.end - lv.start, K)
    t_tab = time.perf_counter() - t0
    gum, gmax = m._gumbel(K)
    # one round's jobs: the largest-KL group of every row
    bits = m._group_kls(lv)
    groups = torch.argmax(bits, dim=1)
    glen = lv.d_glen[groups]
    order = torch.sort(glen, stable=True)[1]
    groups, glen = groups[order], glen[order]
    rows = torch.arange(N, device="cuda", dtype=torch.int32)[order].contiguous()
    jobs = ops.RecJobs(rows, lv.d_start[groups].contiguous(), glen.contiguous(), groups.to(torch.int32).contiguous())
    scale, p_scale = ops.softplus_scale(lv.log_scale), ops.softplus_scale(lv.p_log_scale)
    args = (lv.loc, scale, lv.p_loc, p_scale, tables, gum, jobs)
    sum_g = int(glen.sum())
    alg = 14.0 * K * sum_g
    res = {"rows": N, "K": K, "groups_in_level": int(lv.n_groups), "mean_group_len": sum_g / N, "max_group_len": int(glen.max()),
           "table_build_s": round(t_tab, 2), "alg_flops_per_round": alg}
    i_e = ops.rec_score(*args, ops.REC_EXACT)[0]
    i_f, _, unc, _ = ops.rec_score(*args, ops.REC_FAST, gumbel_absmax=gmax)
    res["indices_identical"] = bool(torch.equal(i_e, i_f))
    res["uncertified_jobs"] = int(unc.sum())
    for name, mode in (("exact", ops.REC_EXACT), ("fast", ops.REC_FAST)):
        ms = timed(lambda: ops.rec_score(*args, mode, gumbel_absmax=gmax), 20)
        res[name + "_ms"] = round(ms, 4)
        res[name + "_alg_tflops"] = round(alg / (ms * 1e-3) / 1e12, 2)
        res[name + "_frac_fp64_peak"] = round(alg / (ms * 1e-3) / FP64_PEAK, 4)
    # executed flops of the fast form: 2 FMA per candidate-element and job + 1 multiply per candidate-element and batch
    res["fast_executed_tflops"] = round(4.0 * K * sum_g / (res["fast_ms"] * 1e-3) / 1e12, 2)
    # whole encode rounds (selection + score + commit), host-timed, no synchronisation inside
    m._encode_round(lv, True, 0)
    torch.cuda.synchronize()
    rounds = 20
    t0 = time.perf_counter()
    for r in range(rounds):
        m._encode_round(lv, True, r + 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["encode_round_ms"] = round(dt / rounds * 1e3, 4)
    res["group_encodes_per_s"] = round(N * rounds / dt)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
