#!/usr/bin/env python3
"""Benchmark of the per-datapoint INR training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A *step* is one full training step of PriorBNNmodel.train on one batch of synthetic CIFAR-shaped
signals per GPU: sample posteriors -> upsample net -> A-transform -> batched SIREN forward, MSE,
backward -> KL -> Adam for every per-INR parameter and for the shared mappings
(training_mappings=True, the reference default, main_prior_training.py:114-132).  Workload =
BASELINE.json configs[1]: CIFAR-10 32x32, 4096 INRs per GPU, 3x32 SIREN.  Datapoints shard across
GPUs (weak scaling); the only per-step collective is the gradient sum of the shared mappings.

Prints ONE JSON line on rank 0.  `value` = INRs trained per second for the whole job at the
reference schedule (55 100 Adam steps per INR, main_prior_training.py:106-107,132); the raw
INR-steps/s is reported beside it.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# PMC traffic of the SIREN training kernel per launch (4096 INRs), by pe/dpe storage type (True: bf16)
PMC_FILE = {True: "r05_siren_bf16_pmc.json", False: "r01_siren_bf16_pmc.json"}
STEPS_PER_INR = 200 + 549 * 100   # reference schedule


def _sha16(files):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "recombiner_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def siren_source_sha16():
    """hash of the sources of the headline SIREN kernel (one wave per INR): stamps the committed PMC traffic summary (see roofline)"""
    return _sha16(("siren_mlp_wave.hip", "siren_common.h", "siren_op16.h"))


def siren_census_sha16():
    """hash of every source tools/siren_census.py reads (the three 16-bit SIREN kernel families): stamps its JSON"""
    return _sha16(("siren_mlp_wave.hip", "siren_mlp_bf16.hip", "siren_mlp_wide.hip", "siren_common.h", "siren_op16.h"))


def siren_instance_census(kernel_name, prec, dims_in, n_hidden, hidden, C, n_inrs, px, clock_ghz, n_cu, us):
    """-> the `valu` object of one preset's SIREN kernel: the census row (tools/siren_census.py) of the instance the step selected
    (family read off the profiled kernel name, template arguments from the preset) priced per pipe like the headline kernel's:
    floors = 32-pixel tiles per SIMD x cycles per tile / clock (the clock measured in the headline kernel's launches)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r05_siren_isa_census.json")) as f:
            cen = json.load(f)
    except (OSError, ValueError):
        return None
    if cen.get("source_sha16") != siren_census_sha16():
        return {"note": "stale census: the SIREN kernel sources changed since tools/siren_census.py ran"}
    F, E = dims_in
    T = "DF16b" if prec == 1 else "DF16_"
    fam = "siren_wave_kernel" if "siren_wave" in kernel_name else "siren_wide_kernel" if "siren_wide" in kernel_name else "siren_bf16_kernel"
    args = "I%sLi%dELi%dELi%dELi%dE" % (T, n_hidden, F, E, C) + ("Li%dE" % hidden if fam == "siren_wide_kernel" else "") + "Li2ELb1E"
    row = next((r for r in cen["instances"] if (fam + args) in r["kernel"]), None)
    if row is None:
        return {"note": "no census row for %s%s" % (fam, args)}
    tiles = n_inrs * ((px + 31) // 32)
    per = tiles / (4 * n_cu) / (clock_ghz * 1e9) * 1e3                     # ms per (cycle per tile and SIMD)
    floors = {"matrix_pipe_ms": round(row["mfma_pipe_cycles_per_tile"] * per, 4),
              "transcendental_unit_ms": round(row["transcendental_cycles_per_tile"] * per, 4),
              "lds_ms": round(4 * row["lds_cycles_per_tile_and_wave"] * per, 4),
              "vector_issue_one_stream_ms": round(row["vector_issue_cycles_per_tile"] * per, 4)}
    ms = us * 1e-3
    return {"instance": fam + args, "vgprs": row["vgprs"], "scratch_bytes_per_lane": row["scratch_bytes_per_lane"],
            "scratch_instructions_in_tile_loop": row["scratch_instructions_in_tile_loop"],
            "instructions_per_tile": row["tile_loop_instructions"], "by_class": row["by_class"],
            "tiles_per_launch": tiles, "shader_clock_ghz_assumed": round(clock_ghz, 3), "floors_ms": floors,
            "frac_of_issue_floor": round(floors["vector_issue_one_stream_ms"] / ms, 4),
            "frac_of_largest_pipe_floor": round(max(floors["matrix_pipe_ms"], floors["transcendental_unit_ms"], floors["lds_ms"]) / ms, 4),
            "source": "profiles/r05_siren_isa_census.json (tools/siren_census.py)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed Adam steps (the reference trains 100-200 per train() call)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--inrs", type=int, default=4096, help="INRs per GPU")
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16"],
                    help="bf16 = bf16 MFMA operands / fp32 accumulate / fp32 master weights (BASELINE config[1]); "
                         "fp32 = exact-parity mode (fp32 MFMA, MIOpen upsample net)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (per-kernel table, REC roofline, PSNR@bpp run, presets table)")
    ap.add_argument("--frozen-mappings", action="store_true", help="training_mappings=False")
    ap.add_argument("--lowp-gemm", action="store_true", help="bf16-operand A-transform GEMMs (experimental)")
    ap.add_argument("--no-split-gemm", action="store_true", help="plain fp32 A-transform GEMMs instead of split-bf16 fwd/dgrad")
    ap.add_argument("--split-terms", type=int, default=None, choices=[2, 3],
                    help="A/B switch: 3 = both split-GEMM operands carry a low part, 2 = the mappings enter as bf16")
    ap.add_argument("--split-dgrad-terms", type=int, default=None, choices=[1, 2, 3],
                    help="A/B switch: terms of the A-transform data-gradient GEMM (default: as --split-terms)")
    ap.add_argument("--torch-noise", action="store_true", help="torch.randn + reparam instead of in-kernel Philox noise")
    ap.add_argument("--wgrad-fp32", action="store_true", help="fp32 A weight-gradient GEMMs instead of bf16 high parts")
    ap.add_argument("--pe-fp32", action="store_true", help="store pe / dpe as fp32 instead of bf16 (bf16 mode; bit-identical)")
    ap.add_argument("--stage1-fp32", action="store_true", help="keep the stage-1 upsampling GEMMs in fp32 (bf16 mode)")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="do not load the committed TunableOp GEMM selections")
    return ap.parse_args()


def _cpu_case(cfg, n, threads, seconds_target, max_steps):
    from oracle import ref_cpu as O
    from recombiner_amd import utils
    geo = O.Geometry.from_config(cfg)
    X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, cfg["output_dim"], seed=0)
    p = O.init_prior_params(geo, n, seed=42)
    A = O.make_linear_transform(geo.dims, seed=123)
    up = O.UpsampleNet(geo.data_dim, geo.paddings, geo.layerwise_scale_factors, seed=124)
    s0 = 0.0211547
    lat = geo.latent_grid
    pri = [torch.zeros(geo.d_net), torch.full((geo.d_net,), s0), torch.zeros(*lat, 128), torch.full((*lat, 128), s0)] + [None] * 4
    Xn = X[None].repeat(n, 1, 1)
    O.prior_train(geo, p, Xn, Y, pri, A, up, 1, 2e-4, 1e-8, True, O.Noise())    # warm-up
    t0 = time.perf_counter()
    steps = 0
    while True:
        O.prior_train(geo, p, Xn, Y, pri, A, up, 2, 2e-4, 1e-8, True, O.Noise())
        steps += 2
        el = time.perf_counter() - t0
        if el > seconds_target or steps >= max_steps:
            break
    return n * steps / el, steps


def cpu_baseline(cfg):
    """The CPU oracle (op-for-op restatement of the reference, golden-pinned) timed on the host cores on bounded samples of
    the same workload, training_mappings=True: N = 1024 INRs (amortises the fixed cost of Adam on the shared mappings --
    the fairest CPU figure; this is `value`) and N = 16 (BASELINE configs[0], the reference's own CPU case).
    ~25 s of CPU work together.  Beside them the reference's OWN timing captured in the build container
    (oracle/time_reference.py -> profiles/r02_reference_cpu_timing.json): the provenance link between port and reference."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))      # the GPU box grants 16 host cores per GPU
    torch.set_num_threads(threads)
    big, steps_big = _cpu_case(cfg, 1024, threads, 14.0, 12)
    small, steps_small = _cpu_case(cfg, 16, threads, 6.0, 60)
    ref = None
    try:
        with open(os.path.join(ROOT, "profiles", "r02_reference_cpu_timing.json")) as f:
            ref = json.load(f)
    except (OSError, ValueError):
        ref = None
    return {"value": big / STEPS_PER_INR, "unit": "INRs trained/s", "inr_steps_per_sec": big, "cores": threads, "kind": "port",
            "sample": f"oracle prior_train, CIFAR preset, N=1024 INRs, {steps_big} Adam steps, training_mappings=True, "
                      f"torch CPU fp32, {threads} threads",
            "n16": {"inr_steps_per_sec": small, "value": small / STEPS_PER_INR,
                    "sample": f"same, N=16 INRs (BASELINE configs[0]), {steps_small} Adam steps"},
            "reference_in_build_container": ref}


# ------------------------------------------------------------------------------------------------------------------
# untimed extras of the JSON line: per-kernel table of the step, REC scoring roofline, PSNR@bpp
# ------------------------------------------------------------------------------------------------------------------
def kernel_table(run, steps, n, D):
    """Per-kernel time of `steps` replayed training steps (torch.profiler -> roctracer; the step graph's kernels are traced
    individually) for the eight largest, each priced against the roofline that bounds it with its ALGORITHMIC bytes or
    flops per step.  Profiled steps run a few per cent slower than un-profiled ones."""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run(steps)
        torch.cuda.synchronize()
    rows = {}
    for e in prof.key_averages():
        us = float(getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0))
        if us > 0:
            rows[e.key] = (us / steps, e.count / steps)
    total = sum(v[0] for v in rows.values())
    P, E, Dl = 1024, 16, 512
    per_inr = {   # algorithmic HBM bytes per INR and step
        "siren_bf16_kernel": ("hbm", 2 * P * E * 2 + P * 3 * 4 + 2 * D * 4 + 4),
        "siren_wave_kernel": ("hbm", 2 * P * E * 2 + P * 3 * 4 + 2 * D * 4 + 4),
        # both launches (net parameters + latent grid): loc, log_scale, 4 Adam moments read + written (48 B), the gradient read,
        # the next sample written (fp32; its bf16 copy and the noise, re-drawn since round 4, are not counted)
        "posterior_flat_kernel": ("hbm", 56 * (D + Dl)),
        "reparam_rng_kernel": ("hbm", 16 * (D + Dl)),
        "upconv_bwd3_fused_kernel": ("hbm", 3 * 256 * 64 * 2 + 0 * P),   # dpe 32 KB + h2 32 KB read, dz2 32 KB written
        "upconv_fwd3_lds_kernel": ("hbm", 256 * 64 * 2 + P * E * 2),
        "upconv_fwd2_reg_kernel": ("hbm", 64 * 64 * 2 + 256 * 64 * 2),
        "upconv_dgrad2_reg_kernel": ("hbm", 256 * 64 * 2 + 2 * 64 * 64 * 2),
        "upconv_wgrad_kernel": ("hbm", 64 * 64 * 2 + 256 * 64 * 2),
        "adam_multi_kernel": ("hbm", 0),
    }
    dims_l = [1056, 1056, 1056, 99]
    atrans_flops = 2 * 2 * 2.0 * sum(v * v for v in dims_l) * n          # forward + data gradient, two terms each
    out, gemm_us, gemm_calls = [], 0.0, 0.0
    for name, (us, calls) in rows.items():
        if name.startswith("Cijk_"):
            gemm_us += us
            gemm_calls += calls
    merged = [(k, v) for k, v in rows.items() if not k.startswith("Cijk_")]
    if gemm_us:
        merged.append(("library GEMMs (hipBLASLt: A-transform weight gradient of the wide layers, stage-1 upsample fwd/dgrad/wgrad)",
                       (gemm_us, gemm_calls)))
    merged.sort(key=lambda kv: -kv[1][0])
    gemm_flops = (2.0 * sum(v * v for v in dims_l[:3]) + 3 * 2.0 * 512 * 4096) * n
    for name, (us, calls) in merged[:8]:
        rec = {"kernel": name[:120], "launches_per_step": round(calls, 2), "us_per_step": round(us, 1), "share_of_kernel_time": round(us / total, 4)}
        key = next((k for k in per_inr if k in name), None)
        if name.startswith("library GEMMs"):
            ach = gemm_flops / (us * 1e-6) / 1e12
            rec.update(bound="mfma", achieved=round(ach, 1), peak=2500.0, unit="TFLOP/s", frac=round(ach / 2500.0, 4),
                       alg_flops_per_step=gemm_flops)
        elif "atrans_kernel" in name:
            ach = atrans_flops / (us * 1e-6) / 1e12
            rec.update(bound="mfma", achieved=round(ach, 1), peak=2500.0, unit="TFLOP/s", frac=round(ach / 2500.0, 4),
                       alg_flops_per_step=atrans_flops)
        elif key is not None and per_inr[key][1] > 0:
            b = per_inr[key][1] * n
            ach = b / (us * 1e-6) / 1e9
            rec.update(bound="hbm", achieved=round(ach, 1), peak=8000.0, unit="GB/s", frac=round(ach / 8000.0, 4), alg_bytes_per_step=b)
        out.append(rec)
    return {"source": "torch.profiler (roctracer) over %d replayed steps of the ONE-STREAM form of the step (stream_forks = 0): in the "
                      "shipped three-stream form kernels run beside each other and a kernel's duration is no longer its own" % steps,
            "kernel_us_per_step_total": round(total, 1),
            "kernels_per_step": round(sum(v[1] for v in rows.values()), 1), "top": out}


def rec_roofline(dev):
    """The A* candidate scorer on the CIFAR test batch of main_compression.py (500 rows, K = 65 536, each row's
    largest-KL group): bound = fp64 vector ALU (78.6 TFLOP/s).  `achieved` counts the flops the certified fast scorer
    executes (2 FMA per candidate, element and job + 1 multiply per candidate, element and eight-job batch); SURVEY 8(d)'s
    algorithmic count of the reference arithmetic (14 K g per job) is given beside it."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_rec as BR
    from recombiner_amd import ops
    N, K = 500, 65536
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):        # (the model constructor prints its expected bpp, like the reference's)
        m = BR.build(N, 4.0, dev)
    lv = m._l1
    bits = m._group_kls(lv)
    groups = torch.argmax(bits, dim=1)
    glen = lv.d_glen[groups]
    lens = sorted(set(int(v) for v in glen.cpu().tolist()))
    tables = m._rec_tables(lv, lens, K)
    gum, gmax = m._gumbel(K)
    order = torch.sort(glen, stable=True)[1]
    groups, glen = groups[order], glen[order]
    rows = torch.arange(N, device=dev, dtype=torch.int32)[order].contiguous()
    jobs = ops.RecJobs(rows, lv.d_start[groups].contiguous(), glen.contiguous(), groups.to(torch.int32).contiguous())
    scale, p_scale = ops.softplus_scale(lv.log_scale), ops.softplus_scale(lv.p_log_scale)
    args = (lv.loc, scale, lv.p_loc, p_scale, tables, gum, jobs)
    i_e = ops.rec_score(*args, ops.REC_EXACT)[0]
    i_f, _, unc, _ = ops.rec_score(*args, ops.REC_FAST, gumbel_absmax=gmax)
    ms_f = BR.timed(lambda: ops.rec_score(*args, ops.REC_FAST, gumbel_absmax=gmax), 20)
    ms_e = BR.timed(lambda: ops.rec_score(*args, ops.REC_EXACT), 5)
    sum_g = int(glen.sum())
    executed = (4.0 + 1.0 / 8) * K * sum_g
    ach = executed / (ms_f * 1e-3) / 1e12
    return {"kernel": "A* candidate scoring, certified fast form (rcb_rec_score_argmax, RCB_REC_FAST: prep + score + merge + arbiter)",
            "bound": "fp64 vector ALU", "achieved": round(ach, 2), "peak": 78.6, "unit": "TFLOP/s", "frac": round(ach / 78.6, 4),
            "traffic": None, "avg_call_ms": round(ms_f, 4), "executed_flops_per_call": executed,
            "reference_arithmetic_flops_per_call": 14.0 * K * sum_g,
            "reference_arithmetic_tflops_equivalent": round(14.0 * K * sum_g / (ms_f * 1e-3) / 1e12, 1),
            "exact_scorer_ms": round(ms_e, 4), "indices_identical_to_exact_scorer": bool(torch.equal(i_e, i_f)),
            "jobs_decided_by_exact_arbiter": int(unc.sum()), "jobs": N, "candidates": K, "mean_group_len": round(sum_g / N, 2),
            "group_encodes_per_sec": round(N / (ms_f * 1e-3))}


def psnr_at_bpp(dev, precision):
    """The PSNR@bpp half of the metric, measured: the 32 smooth test images of tests/golden/psnr_smooth_cifar.npz compressed
    from the REFERENCE-written prior checkpoint beside it with the reference run's schedule (400 optimisation epochs, every
    group A*-encoded, 6 fine-tune epochs per round), production path (graph replay, device noise) in the bench's precision
    mode.  The reference's own result for these images at this rate is in the fixture (reference CPU run)."""
    import numpy as np
    from recombiner_amd import bitstream, drivers, utils
    g = os.path.join(ROOT, "tests", "golden")
    d = np.load(os.path.join(g, "psnr_smooth_cifar.npz"), allow_pickle=False)
    cfg = json.loads(str(d["cfg"]))
    ck = drivers.load_checkpoint(os.path.join(g, "PRIOR_ref_smooth_cifar.pkl.gz"))
    n = int(d["n_test"])
    X, _ = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], 1, 3, seed=0)
    Xd = X.to(dev)[None].expand(n, -1, -1)
    Y = torch.from_numpy(d["Y_test"]).to(dev)
    t0 = time.perf_counter()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        dist, model = drivers.compress(cfg, "cifar", ck, Xd, Y, device=dev, n_epochs=int(d["n_opt"]), lr=float(d["lr"]),
                                       precision=precision, finetune_epochs=int(d["n_ft"]))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    blob = bitstream.encode(model)
    bpp = bitstream.payload_bits(blob) / (n * 1024)
    ref = np.asarray(d["psnr"], dtype=np.float64)
    return {"images": n, "bpp": round(float(bpp), 4), "psnr_db_mean": round(float(np.mean(dist)), 3),
            "psnr_db_min": round(float(np.min(dist)), 3), "psnr_db_max": round(float(np.max(dist)), 3),
            "reference_psnr_db_mean": round(float(ref.mean()), 3), "reference_bpp": round(float(d["bpp"]), 4),
            "delta_db": round(float(np.mean(dist) - ref.mean()), 3), "seconds": round(el, 2),
            "adam_steps": int(d["n_opt"]) + int(d["n_groups"]) * int(d["n_ft"]), "groups": int(d["n_groups"]),
            "what": "32 smooth synthetic 32x32 images, prior checkpoint written by the reference (tests/golden), full compression "
                    "(optimise, A* encode all groups, fine-tune between rounds), precision mode of this bench line",
            "product_trained_prior": rd_trained(dev, precision)}


def rd_trained(dev, precision, runs_per_rate=3):
    """The other half of PSNR@bpp: the prior (and the mappings) TRAINED BY THE PRODUCT in this bench's precision mode -- the
    path the throughput number times -- carried through to rate-distortion points: tests/golden/rd_trained_cifar.npz holds, for
    two rate targets, the reference's own EM loop (training_mappings=True) on 64 smooth images and its compression of 16
    others, four repetitions each; here the product does the same from scratch (drivers.rd_point) a few times per rate.
    Single runs scatter by ~0.6 dB / ~3 % bpp on either side (every A* index is a random draw): means are compared, the
    PSNR difference at matched rate through the slope between the reference's two points."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_util import smooth_images
    from recombiner_amd import drivers, utils
    from golden_util import load_rd_fixture
    n64 = all(os.path.exists(os.path.join(ROOT, "tests", "golden", "rd_trained_cifar_n64_r%d.npz" % r)) for r in (0, 1))
    d = load_rd_fixture(64 if n64 else 16)          # round 4: the reference's runs on 64 held-out images when the fixture is there
    cfg = json.loads(str(d["cfg"]))
    Ytr = smooth_images(int(d["n_train"]), cfg["pixel_sizes"], int(d["train_seed"]))
    Yte = smooth_images(int(d["n_test"]), cfg["pixel_sizes"], int(d["test_seed"]))
    X, _ = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], 1, 3, seed=0)
    sched = dict(n_em_iter=int(d["n_iter"]), first_epochs=int(d["first_epochs"]), epochs=int(d["epochs"]), lr=float(d["lr"]),
                 n_opt=int(d["n_opt"]), finetune_epochs=int(d["n_ft"]))
    ref = [(float(np.mean(d[f"r{ri}_bpp"])), float(np.mean(d[f"r{ri}_psnr"]))) for ri in range(len(d["max_bitrate"]))]
    slope = (ref[0][1] - ref[1][1]) / (ref[0][0] - ref[1][0])
    out = []
    t0 = time.perf_counter()
    for ri, rate in enumerate(d["max_bitrate"]):
        runs = [drivers.rd_point(cfg, "cifar", X, Ytr, Yte, float(rate), device=dev, seed=42 + s, precision=precision, **sched)
                for s in range(runs_per_rate)]
        bpp = float(np.mean([r["bpp"] for r in runs]))
        psnr = float(np.mean([r["psnr"].mean() for r in runs]))
        out.append({"max_bitrate": float(rate), "bpp": round(bpp, 4), "psnr_db_mean": round(psnr, 3), "runs": runs_per_rate,
                    "reference_bpp": round(ref[ri][0], 4), "reference_psnr_db_mean": round(ref[ri][1], 3),
                    "reference_runs": int(np.asarray(d[f"r{ri}_bpp"]).size),
                    "delta_db_at_matched_rate": round(psnr - ref[ri][1] - slope * (bpp - ref[ri][0]), 3),
                    "final_kl_bits_per_inr": round(float(np.mean([r["trajectory"][-1, 0] for r in runs])), 1),
                    "bit_budget": [float(v) for v in d[f"r{ri}_budget"]]})
    return {"points": out, "reference_slope_db_per_bpp": round(slope, 3), "seconds": round(time.perf_counter() - t0, 1),
            "what": "64 smooth synthetic training images, " + str(int(d["n_test"])) + " test images; EM loop %d iterations (%d + %d x %d Adam steps, lr %g, "
                    "training_mappings=True), then optimise %d epochs + A* encode every group with %d fine-tune epochs per round"
                    % (sched["n_em_iter"], sched["first_epochs"], sched["n_em_iter"] - 1, sched["epochs"], sched["lr"], sched["n_opt"],
                       sched["finetune_epochs"])}


def presets_table(dev, clock_ghz=None):
    """BASELINE configs[2..4] (and the reference presets they vary) plus the test-time path on this box, untimed by the
    driver's step clock but inside its run.
    prior_training: the training step of each preset in its 16-bit mode (production path: device noise, graph replay) at the
      SMALL size the reference's drivers touch per call (2 photos / 8 clips / 4 clips) and at a GPU-FILLING shard (24 Kodak
      photos, 1024 audio clips = the per-GPU shard of configs[3], 32 video clips), each with its dominant kernel from a
      profile of a few steps and, where that kernel is the SIREN one, its recomputed roofline fractions.
    test_time: TestBNNmodel.train (S = 5) and one A* encode round of the first level -- what main_compression.py:87-162
      runs -- for the CIFAR batch of 500, for ONE Kodak photo / audio clip / video clip (the reference's unit of work, built
      from a briefly trained prior through drivers.build_checkpoint / build_test_model) and for a batch of 8 of each."""
    import contextlib
    import io
    import warnings
    import numpy as np
    from torch.profiler import ProfilerActivity, profile
    from recombiner_amd import config, drivers, utils
    from recombiner_amd import prior_model as PM

    def prior_setup(name, n_data, width, prec):
        cfg = dict(config.configs[name])
        cfg["hidden_dims"] = [width] * len(cfg["hidden_dims"])
        per = int(np.prod(cfg["patch_nums"])) if cfg["patch"] else 1
        n = n_data * per
        X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, cfg["output_dim"], seed=0)
        m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                             cfg["upsample_factors"], cfg["latent_dim"], cfg["patch"], cfg["patch_nums"],
                             cfg["hierarchical_patch_nums"], random_seed=42, device=dev)
        m.precision = prec
        torch.manual_seed(1)
        lt = PM.LinearTransform(m.dims).to(dev)
        up = PM.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"]).to(dev)
        s0, D = 0.0211547, m._d_net
        lat = list(m.lpe_loc.shape[1:])
        pri = [torch.zeros(D, device=dev), torch.full((D,), s0, device=dev), torch.zeros(lat, device=dev), torch.full(lat, s0, device=dev)]
        pri += ([torch.zeros(D, device=dev), torch.full((D,), s0, device=dev)] * 2) if cfg["patch"] else [None] * 4
        return cfg, n, m, lt, up, pri, X.to(dev), Y.to(dev)

    # (the small cases first: measured right after a 40 GB case the 192-INR Kodak step replayed at twice its usual time with
    # the same kernel times -- launch gaps inside the replayed graph -- while tools/bench_presets.py in a fresh process gives
    # 0.71 ms; the order below keeps every small case ahead of the large ones)
    runs = [("kodak-w48 (configs[2]), 2 photos", "kodak", 2, 48, 1, 50), ("audio (configs[3]), 8 clips", "audio", 8, 32, 1, 50),
            ("video-w64-f16 (configs[4]), 4 clips", "video", 4, 64, 2, 50), ("kodak, 2 photos", "kodak", 2, 32, 1, 50),
            ("video, 4 clips", "video", 4, 32, 1, 50),
            ("kodak-w48 (configs[2]), 24 photos", "kodak", 24, 48, 1, 20),
            ("video-w64-f16 (configs[4]), 32 clips", "video", 32, 64, 2, 10),
            ("audio (configs[3]), 1024 clips = one rank's shard of the 8192", "audio", 1024, 32, 1, 8)]
    out = []
    for label, name, n_data, width, prec, steps in runs:
        cfg, n, m, lt, up, pri, X, Yd = prior_setup(name, n_data, width, prec)
        Xd = X[None].expand(n, -1, -1)

        def run(k):
            return m.train(k, 2e-4, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            run(6)
            # best of three timed calls: the first replays after another workload have shown launch gaps inside the replayed graph
            # (a 192-INR step at twice its usual time with unchanged kernel times) that a second call no longer has
            calls = []
            for _rep in range(3 if steps >= 20 else 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                _, _, elbo = run(steps)
                e1.record()
                torch.cuda.synchronize()
                calls.append(e0.elapsed_time(e1) / steps)
            ms = sorted(calls)[len(calls) // 2]            # the MEDIAN call is what the record reports; all calls are kept beside it
            nprof = max(2, min(10, steps // 2))
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                run(nprof)
                torch.cuda.synchronize()
        rows = sorted(((float(getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0)) / nprof, e.key)
                       for e in prof.key_averages()), reverse=True)
        total = sum(r[0] for r in rows) or 1.0
        top_us, top = rows[0]
        px, C, E, D = int(np.prod(cfg["pixel_sizes"])), cfg["output_dim"], 16, m._d_net
        rec = {"preset": label, "datapoints": n_data, "inrs": n, "pixels_per_inr": px, "hidden": width,
               "operands": "bf16" if prec == 1 else "f16", "ms_per_step": round(ms, 4),
               "timing": "median of %d timed call(s) of %d steps" % (len(calls), steps), "ms_per_step_calls": [round(c, 4) for c in calls],
               "inr_steps_per_sec": round(n / (ms * 1e-3)),
               "pixel_steps_per_sec": round(n * px / (ms * 1e-3)), "graph_replay": m._ws is not None and m._ws["graphs"] is not None,
               "finite": bool(np.isfinite(elbo).all()), "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
               "dominant_kernel": top[:80], "dominant_kernel_us": round(top_us, 1), "dominant_kernel_share": round(top_us / total, 3)}
        siren = [r for r in rows if "siren" in r[1]]
        if siren:      # the SIREN kernel of this preset (dominant or not): recomputed fractions of the two chip rooflines
            s_us = sum(r[0] for r in siren)
            bts = (2 * px * E * 2 + px * C * 4 + 2 * D * 4 + 4) * n       # pe + dpe (bf16), target, wvec + dwvec, sse
            dims = m.dims
            fl = 6.0 * px * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)) * n
            rec.update(siren_us=round(s_us, 1), siren_share=round(s_us / total, 3), siren_alg_bytes=bts,
                       siren_hbm_frac=round(bts / (s_us * 1e-6) / 8e12, 4), siren_alg_flops=fl,
                       siren_mfma_frac=round(fl / (s_us * 1e-6) / 2.5e15, 4))
            lb = max(siren, key=lambda r: r[0])                     # the loss / backward instance (the step has no other SIREN launch)
            ck = clock_ghz if (clock_ghz and clock_ghz == clock_ghz) else 2.1
            rec["valu"] = siren_instance_census(lb[1], prec, (cfg["fourier_dim"], E), len(cfg["hidden_dims"]), width, C, n, px, ck,
                                                torch.cuda.get_device_properties(dev).multi_processor_count, lb[0])
        out.append(rec)
        del m, lt, up, Xd, Yd, X, pri
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()

    # ---- test-time path ------------------------------------------------------------------------------------------------
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_rec as BR

    def time_test_model(tm, Xd, Yd, what, steps=60):
        tm.precision = 1
        tm.train(Xd, Yd, 24, torch.optim.Adam(tm.parameters(), lr=2e-4), False, sample_size=5)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        tm.train(Xd, Yd, steps, torch.optim.Adam(tm.parameters(), lr=2e-4), False, sample_size=5)
        e1.record()
        torch.cuda.synchronize()
        ms_tt = e0.elapsed_time(e1) / steps
        lv = tm._levels[-1]                       # the level the reference encodes first (level 3 when patched, else level 1)
        for glen in np.unique((lv.end - lv.start)):
            tm._table(lv, int(glen), 65536)
        tm._encode_round(lv, True, 0)
        ms_round = BR.timed(lambda: tm._encode_round(lv, True, 1), 6)
        n = Yd.shape[0]
        return {"what": what, "inrs": n, "samples": 5, "ms_per_step": round(ms_tt, 4), "inr_sample_steps_per_sec": round(5 * n / (ms_tt * 1e-3)),
                "encode_round_ms": round(ms_round, 4), "encode_round_rows": int(lv.loc.shape[0]), "encode_round_groups_in_level": int(lv.n_groups)}

    tt = []
    with contextlib.redirect_stdout(io.StringIO()):
        tm = BR.build(500, 4.0, dev)
    cfgc = config.configs["cifar"]
    X, Y = utils.synthetic_inputs(cfgc["pixel_sizes"], cfgc["fourier_dim"], 500, 3, seed=0)
    tt.append(time_test_model(tm, X.to(dev)[None].expand(500, -1, -1), Y.to(dev),
                              "TestBNNmodel.train + encode round, CIFAR batch of 500 (data/load_data.py:92-94), bf16 mode, graph replay", 100))
    del tm
    for name, width, prec in (("kodak", 48, 1), ("audio", 32, 1), ("video", 64, 1)):
        # a prior to compress from: a few training steps on two datapoints, then the checkpoint objects of main_prior_training.py
        cfg, n, m, lt, up, pri, X, Yd = prior_setup(name, 2, width, 1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.train(12, 1e-3, X[None].expand(n, -1, -1), Yd, *pri, lt, up, 1e-6, training_mappings=True)
        ck = drivers.build_checkpoint(m, lt, up, *pri, 1e-6)
        per = int(np.prod(cfg["patch_nums"]))
        del m
        for n_data in (1, 8):
            Xn, Yn = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n_data * per, cfg["output_dim"], seed=3)
            with contextlib.redirect_stdout(io.StringIO()):
                tm = drivers.build_test_model(cfg, name, ck, n_data * per, dev, 42)
            tt.append(time_test_model(tm, Xn.to(dev)[None].expand(n_data * per, -1, -1), Yn.to(dev),
                                      f"{name} (width {width}): {n_data} datapoint(s) = {n_data * per} INRs "
                                      f"(main_compression.py:87-162 runs one per process), bf16 mode", 40))
            del tm
            torch.cuda.empty_cache()
    return {"prior_training": out, "test_time": tt}


def comm_report(m, lt, up, dev):
    """Sharded step (every rank calls this): bytes of the two per-step all-reduces (gradients of the A matrices, of the conv
    weights), their duration on their own, the duration of each captured segment of the step, and the slack = compute
    that runs while a bucket travels (A bucket: upsampling-net backward + posterior update; conv bucket: posterior update)
    minus that bucket's all-reduce.  Also checks that the shared mappings are bit-identical on every rank after training."""
    import torch.distributed as dist
    w = m._ws
    buckets_obj = w["flat"] if w is not None else None          # dist.GradBuckets
    flat = buckets_obj.flat if buckets_obj is not None else None
    n_a = sum(q.numel() for q in lt.A)
    chk = torch.stack([torch.cat([q.detach().double().reshape(-1) for q in list(lt.parameters()) + list(up.parameters())]).sum(),
                       torch.cat([(q.detach().double() ** 2).reshape(-1) for q in list(lt.parameters()) + list(up.parameters())]).sum()])
    hi, lo = chk.clone(), chk.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    rep = {"mappings_identical_across_ranks": bool(torch.equal(hi, lo)), "backend": dist.get_backend(),
           "step_form_choice": os.environ.get("RCB_STEP_FORM_CHOICE", "model default (captured on one rank, four segments on more)")}
    if flat is not None and w["graphs"] is not None and w["graphs"][0] == "one":
        # RCCL: both all-reduces are captured INSIDE the one step graph (issued from the forked streams, overlapped by the
        # graph's own dependencies): no host work between segments; what can be timed from outside is the collective alone
        def timed1(fn, reps=10):
            evs = []
            for _ in range(reps):
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            t = torch.tensor([sorted(a_.elapsed_time(b_) for a_, b_ in evs)[reps // 2]], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        scratch = flat.clone()
        ar = [timed1(lambda b=b: dist.all_reduce(b, group=m.dp_group)) for b in (scratch[:n_a], scratch[n_a:])]
        rep.update({"step_form": "one captured graph per step, both all-reduces inside it (RCCL)",
                    "allreduce_bytes_per_step": int(flat.numel() * 4), "bucket_bytes": [int(n_a * 4), int((flat.numel() - n_a) * 4)],
                    "allreduce_ms_alone": [round(x, 4) for x in ar]})
        return rep
    if flat is None or w["graphs"] is None or w["graphs"][0] != "segments":
        rep["note"] = "the step did not run as captured segments (frozen mappings or eager stepping)"
        return rep

    def timed(fn, reps=10):
        evs = []
        for _ in range(reps):
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        t = torch.tensor([sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    buckets = [flat[:n_a], flat[n_a:]]
    # the collectives are timed on a scratch copy (the live buckets hold this step's reduced gradients), and everything the
    # segment replays below advance -- parameters, Adam moments, step / noise counters, the carried-over sample -- is put
    # back afterwards: the replays apply Adam steps from LOCAL gradients (no collective between them), so without the
    # restore the ranks' mappings would drift apart right after having been certified identical
    scratch = flat.clone()
    ar = [timed(lambda b=b: dist.all_reduce(b, group=m.dp_group)) for b in (scratch[:n_a], scratch[n_a:])]
    del scratch
    live = [q.data for q in list(m.parameters()) + list(lt.parameters()) + list(up.parameters())]
    live += [w[k] for k in ("state_flat", "step_t", "rng_ctr", "mse_buf", "kl_buf")] + [flat]
    live += [getattr(t, "buf", t) for k in ("smp_net", "smp_lpe") if k in w for t in w[k] if t is not None]   # (ops.Planes -> its buffer)
    saved = [t.clone() for t in live]
    graphs = w["graphs"][1]
    seg = []
    for _ in range(3):                                    # whole steps, segment by segment (no collectives: timing only)
        row = []
        for g in graphs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            row.append((e0, e1))
        torch.cuda.synchronize()
        seg.append([a.elapsed_time(b) for a, b in row])
    seg = [sorted(col)[1] for col in zip(*seg)]
    for t, c in zip(live, saved):
        t.copy_(c)
    torch.cuda.synchronize()
    rep.update({"allreduce_bytes_per_step": int(flat.numel() * 4), "bucket_bytes": [int(b.numel() * 4) for b in buckets],
                "allreduce_ms_alone": [round(x, 4) for x in ar],
                "segment_ms": {"1a sample..A-transform backward": round(seg[0], 4), "1b upsampling-net backward": round(seg[1], 4),
                               "2 posterior update": round(seg[2], 4), "3 Adam on the mappings": round(seg[3], 4)},
                "overlap_window_ms": [round(seg[1] + seg[2], 4), round(seg[2], 4)],
                "slack_ms": [round(seg[1] + seg[2] - ar[0], 4), round(seg[2] - ar[1], 4)]})
    return rep


def segment_host_cost(dev, cfg, n, tm, single_graph_ms, captured=False):
    """What the SHARDED form of the step costs on top of the one-rank step, measured without a second GPU: the same 4096-INR
    step with the two all-reduces of a ONE-rank RCCL communicator (PriorBNNmodel.force_segments), (captured=True) inside the
    one step graph -- the production form on RCCL -- or (False) as four captured segments around host-enqueued collectives,
    against the collective-free single-graph step of the main measurement."""
    import torch.distributed as dist
    from recombiner_amd import utils
    from recombiner_amd import prior_model as PM
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(29687 + (1 if captured else 0))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, cfg["output_dim"], seed=0)
        m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"], cfg["pixel_sizes"],
                             cfg["upsample_factors"], cfg["latent_dim"], cfg["patch"], cfg["patch_nums"],
                             cfg["hierarchical_patch_nums"], random_seed=42, device=dev)
        m.precision, m.dp_group, m.force_segments, m.capture_collectives = 1, dist.group.WORLD, True, captured
        torch.manual_seed(123)
        lt = PM.LinearTransform(m.dims).to(dev)
        torch.manual_seed(124)
        up = PM.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"]).to(dev)
        D, s0 = m._d_net, 0.0211547
        pri = [torch.zeros(D, device=dev), torch.full((D,), s0, device=dev), torch.zeros(2, 2, 128, device=dev),
               torch.full((2, 2, 128), s0, device=dev)] + [None] * 4
        Xd, Yd = X.to(dev)[None].expand(n, -1, -1), Y.to(dev)

        def run(k):
            return m.train(k, 2e-4, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=tm)
        run(6)
        run(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(100)
        torch.cuda.synchronize()
        seg_ms = (time.perf_counter() - t0) / 100 * 1e3
        kind = m._ws["graphs"][0] if (m._ws is not None and m._ws["graphs"] is not None) else "eager"
        return {"what": "4096 INRs, one rank, RCCL communicator of size 1: two all-reduces per step, " +
                        ("captured inside the step graph" if captured else "host-enqueued between four captured segments") +
                        ", against the collective-free single-graph step", "step_form": kind, "segmented_ms_per_step": round(seg_ms, 4),
                "single_graph_ms_per_step": round(single_graph_ms, 4), "host_cost_us_per_step": round((seg_ms - single_graph_ms) * 1e3, 1),
                "note": "no N > 1 run exists on this pool: the scaling bench of the driver is the only place RCCL moves bytes between GPUs"}
    finally:
        if own:
            dist.destroy_process_group()


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through torch.distributed.run as a CHILD
    process and exit with its code.  Nothing in this parent has touched the GPU yet (device_count() does not initialise
    it), and the parent is not replaced (no exec)."""
    import subprocess
    ndev = torch.cuda.device_count()
    rehearsal = os.environ.get("RCB_DIST_BACKEND", "nccl") != "nccl"
    if ndev < a.gpus and not rehearsal:
        sys.exit(f"bench.py: --gpus {a.gpus} requested but only {ndev} GPU(s) are visible: one rank per GPU over RCCL "
                 f"needs {a.gpus} devices (RCB_DIST_BACKEND=gloo rehearses the sharded code path on fewer; never for numbers)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # --standalone: torchrun picks AND holds the rendezvous port itself (no bind / close / reuse race with other jobs on the box)
    run = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(a.gpus)]
    probe_capture_form(a, env, run, rehearsal)
    sys.exit(subprocess.run(run + [os.path.abspath(__file__)] + sys.argv[1:], env=env).returncode)


def probe_capture_form(a, env, run, rehearsal):
    """Which form of the sharded step the ranks will run (recorded in comm.step_form_choice).  The form with both all-reduces
    captured inside the step graph has only ever executed on one-rank communicators; a hang of RCCL under capture on N ranks
    would take the measured run with it.  So the launcher first runs tools/rccl_capture_probe.py on the same N ranks as a
    child with a hard timeout, and only a clean RCCL_CAPTURE_OK from every rank selects the captured form
    (RCB_CAPTURE_COLLECTIVES=1); a failure, a timeout or a gloo rehearsal select the four-segment form (=0).  An explicit
    RCB_CAPTURE_COLLECTIVES in the environment is left alone."""
    import subprocess
    if "RCB_CAPTURE_COLLECTIVES" in os.environ:
        env["RCB_STEP_FORM_CHOICE"] = "RCB_CAPTURE_COLLECTIVES=%s given by the caller" % os.environ["RCB_CAPTURE_COLLECTIVES"]
        return
    if rehearsal:
        env["RCB_CAPTURE_COLLECTIVES"], env["RCB_STEP_FORM_CHOICE"] = "0", "four segments (gloo rehearsal: no probe)"
        return
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "rccl_capture_probe.py")
    timeout_s = float(os.environ.get("RCB_PROBE_TIMEOUT_S", "180"))
    try:
        # (own session: on a timeout the whole process group of the probe -- torchrun and its ranks -- is ended, by its id)
        p = subprocess.Popen(run + [probe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=timeout_s)
            ok = p.returncode == 0 and out.count("RCCL_CAPTURE_OK") >= a.gpus and "RCCL_CAPTURE_NOT_OK" not in out
            why = "probe ok on %d ranks" % a.gpus if ok else "probe failed (exit code %s)" % p.returncode
        except subprocess.TimeoutExpired:
            import signal
            os.killpg(p.pid, signal.SIGKILL)
            p.wait()
            ok, why = False, "probe timed out after %.0f s" % timeout_s
    except Exception as e:                       # (no torchrun, no probe file ...)
        ok, why = False, "probe could not run: %r" % (e,)
    env["RCB_CAPTURE_COLLECTIVES"] = "1" if ok else "0"
    env["RCB_STEP_FORM_CHOICE"] = ("collectives captured inside the step graph: " if ok else "four segments around host-enqueued collectives: ") + why
    print("bench.py: sharded step form -- " + env["RCB_STEP_FORM_CHOICE"], file=sys.stderr, flush=True)


def probe_capture_form_in_rank(ws):
    """The same probe for ranks that an EXTERNAL launcher started (torchrun ... bench.py --gpus N: bench.py's own launcher, and
    with it probe_capture_form, never ran).  Before this process touches the GPU, every rank starts tools/rccl_capture_probe.py
    as a child on its own rendezvous (MASTER_PORT + 1, the elastic agent's variables removed so that the child of rank 0 hosts
    the store) with a hard timeout; -> True / False for THIS rank, or None when nothing was probed (an explicit
    RCB_CAPTURE_COLLECTIVES, a choice already made by bench.py's launcher, a gloo rehearsal).  main() then agrees on the minimum
    over the ranks through the real communicator.  Every failure mode -- no probe file, a busy port, a hang -- ends in the
    four-segment form."""
    import subprocess
    if "RCB_CAPTURE_COLLECTIVES" in os.environ or "RCB_STEP_FORM_CHOICE" in os.environ:
        return None
    if os.environ.get("RCB_DIST_BACKEND", "nccl") != "nccl":
        return None
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "rccl_capture_probe.py")
    timeout_s = float(os.environ.get("RCB_PROBE_TIMEOUT_S", "180"))
    try:
        env = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC", "TORCH_NCCL_ASYNC"))}
        env["MASTER_ADDR"] = os.environ.get("MASTER_ADDR", "127.0.0.1")
        env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
        env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = str(ws), os.environ.get("RANK", "0"), os.environ.get("LOCAL_RANK", "0")
        p = subprocess.Popen([sys.executable, probe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                             start_new_session=True)
        try:
            out, _ = p.communicate(timeout=timeout_s)
            return bool(p.returncode == 0 and "RCCL_CAPTURE_OK" in out and "RCCL_CAPTURE_NOT_OK" not in out)
        except subprocess.TimeoutExpired:
            import signal
            os.killpg(p.pid, signal.SIGKILL)
            p.wait()
            return False
    except Exception:                            # noqa: BLE001 -- anything at all: the safe form
        return False


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started {ws} rank(s)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("RCB_DIST_BACKEND", "nccl")
    if ws > 1 and backend == "nccl" and ndev < ws:
        sys.exit(f"bench.py: {ws} RCCL ranks need {ws} GPUs, {ndev} visible")
    dev = torch.device("cuda", local % max(ndev, 1))
    # (before anything touches the GPU: the probe runs as a child of this rank)
    probed = probe_capture_form_in_rank(ws) if (ws > 1 or os.environ.get("RCB_RANK_PROBE_FORCE")) else None
    torch.cuda.set_device(dev)
    if ws > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL over xGMI on ROCm.  RCB_DIST_BACKEND=gloo exists only to rehearse the N>1 code path on a
        # box with fewer GPUs than ranks (ranks then share a device); it is never used for reported numbers.
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        # n_gpus of the JSON line = the ranks the communicator itself reports, proven by a collective
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)
        ws_seen = int(one.item())
        if ws_seen != a.gpus or torch.distributed.get_world_size() != a.gpus:
            sys.exit(f"bench.py: communicator has {ws_seen} ranks, --gpus {a.gpus}")
        if probed is not None:                  # every rank probed (or none did): the captured form only if ALL of them passed
            agree = torch.tensor([1.0 if probed else 0.0], device=dev)
            torch.distributed.all_reduce(agree, op=torch.distributed.ReduceOp.MIN)
            probed = bool(agree.item() > 0.5)
    if probed is not None:
        os.environ["RCB_CAPTURE_COLLECTIVES"] = "1" if probed else "0"
        os.environ["RCB_STEP_FORM_CHOICE"] = (("collectives captured inside the step graph: " if probed else
                                               "four segments around host-enqueued collectives: ")
                                              + ("in-rank probe ok on %d ranks" % ws if probed else "in-rank probe failed or timed out"))
        if rank == 0:
            print("bench.py: sharded step form -- " + os.environ["RCB_STEP_FORM_CHOICE"], file=sys.stderr, flush=True)

    from recombiner_amd import config, ops, tuning, utils
    tuned = False if a.no_tuned_gemms else tuning.enable_tuned_gemms()
    from recombiner_amd import prior_model as PM
    cfg = config.configs["cifar"]
    n = a.inrs
    X, Y = utils.synthetic_inputs(cfg["pixel_sizes"], cfg["fourier_dim"], n, cfg["output_dim"], seed=rank)
    m = PM.PriorBNNmodel(cfg["input_dim"], cfg["hidden_dims"], cfg["output_dim"], n, cfg["data_dim"],
                         cfg["pixel_sizes"], cfg["upsample_factors"], cfg["latent_dim"], cfg["patch"],
                         cfg["patch_nums"], cfg["hierarchical_patch_nums"], random_seed=42 + rank, device=dev)
    m.precision = 1 if a.precision == "bf16" else 0
    m.lowp_gemm = a.lowp_gemm
    m.stage1_bf16 = not a.stage1_fp32
    m.pe_bf16 = not a.pe_fp32
    m.split_gemm = not a.no_split_gemm
    if a.split_terms is not None:
        m.split_terms = a.split_terms
    if a.split_dgrad_terms is not None:
        m.split_dgrad_terms = a.split_dgrad_terms
    m.wgrad_bf16 = not a.wgrad_fp32
    m.fused_noise = not a.torch_noise
    torch.manual_seed(123)
    lt = PM.LinearTransform(m.dims).to(dev)
    torch.manual_seed(124)
    up = PM.Upsample(cfg["data_dim"], cfg["paddings"], cfg["layerwise_scale_factors"]).to(dev)
    if ws > 1:
        # the shared mappings' gradients are summed over all ranks every step (the only per-step collective); the mappings
        # themselves start from rank 0's values
        m.dp_group = torch.distributed.group.WORLD
        for prm in list(lt.parameters()) + list(up.parameters()):
            torch.distributed.broadcast(prm.data, 0)
    torch.manual_seed(1000 + rank)
    s0 = 0.0211547
    D = m._d_net
    pri = [torch.zeros(D, device=dev), torch.full((D,), s0, device=dev), torch.zeros(2, 2, 128, device=dev),
           torch.full((2, 2, 128), s0, device=dev), None, None, None, None]
    Xd = X.to(dev)[None].expand(n, -1, -1)     # one coordinate grid shared by every INR (stride-0 view)
    Yd = Y.to(dev)
    tm = not a.frozen_mappings

    def run(k):
        return m.train(k, 2e-4, Xd, Yd, *pri, lt, up, 1e-8, training_mappings=tm)

    def fence():
        if ws > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # set-up, outside both the warm-up and the timed region: the first train() call with more than three steps warms up
    # eagerly and captures the step as HIP graph(s); later calls of any length replay them.  Without this a run with
    # --warmup < 4 would capture inside the timed region.
    run(4)
    if a.warmup > 0:
        run(a.warmup)
    fence()
    t0 = time.perf_counter()
    run(a.steps)
    fence()
    el = time.perf_counter() - t0
    if ws > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        el = float(tt.item())

    # ---- roofline of the hot-path kernel (fused SIREN fwd + MSE + bwd), measured live ----------
    roof = None
    # The kernel is timed with events on its launch stream in the duty cycle it has inside the step: two training steps run
    # between consecutive timed launches (on every rank: the sharded step contains collectives).  Ten back-to-back launches
    # of this VALU-heavy kernel alone pull the clock down within a few launches (276 -> 335 us in one rocprofv3 trace),
    # which is not the state the step runs it in.
    reps = 10
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    if rank == 0:
        meta = m._meta(Xd, 16)
        pe = torch.randn(n, 1024, 16, device=dev) * 0.1
        pe16 = m.precision != 0 and m.pe_bf16          # the storage type the training step uses for pe / dpe
        if pe16:
            pe = pe.bfloat16()
        # launched as the step launches it: rows on 128-byte lines, and (16-bit modes) the bf16 copy of the gradient that
        # the A transform's weight-gradient GEMM reads -- 2 D extra bytes per INR that the algorithmic figure does NOT count
        ld = (D + 31) // 32 * 32
        wv = torch.empty(n, ld, device=dev)[:, :D]
        wv.copy_((torch.rand(n, D, device=dev) * 2 - 1) * 0.02)
        want16 = bool(m.precision != 0 and m.split_gemm and not m.lowp_gemm and m.wgrad_bf16)
        xf16 = ops.xf_bf16(Xd, m.precision) if m.precision in (1, 2) else None

        probe = torch.zeros(256, 4, device=dev, dtype=torch.int64)     # per-workgroup clock stamps of the LAST launch

        def launch():
            return ops.siren_loss_bwd(Xd, pe, wv, Yd, 1.0 / 3072, meta, want_bf16=want16, xf16=xf16,
                                      clock_probe=probe if m.precision == 1 else None)
        for _ in range(2):
            launch()
    clocks = []
    for e0, e1 in evs:
        run(2)
        if rank == 0:
            # an untimed launch first: the stream is busy while the host submits e0 / kernel / e1, so the interval holds
            # the kernel and not the host's launch latency
            launch()
            e0.record()
            launch()
            e1.record()
            if m.precision == 1:
                e1.synchronize()
                clocks.append(ops.shader_clock_ghz(probe))
    if rank == 0:
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / reps
        siren_clock_ghz = (sum(clocks) / len(clocks)) if clocks else float("nan")
        dims = m.dims
        flops = 6.0 * 1024 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)) * n   # fwd + 2x bwd
        # algorithmic HBM bytes per INR: pe read + dpe written (P*16 elements each, 2 B in bf16 storage, else 4 B),
        # target P*C*4, wvec read + dwvec written, sse
        alg_bytes = (2 * 1024 * 16 * (2 if pe16 else 4) + 1024 * 3 * 4 + 2 * D * 4 + 4) * n
        t = ms * 1e-3
        share = round(ms / (el / a.steps * 1e3), 4)
        if m.precision == 0:
            # fp32 MFMA path: 0.51 ms of matrix work at peak vs 0.09 ms of HBM traffic -> matrix-bound
            ach = flops / t / 1e12
            roof = {"kernel": "fused SIREN fwd+MSE+bwd, fp32 MFMA (rcb_siren_loss_bwd)", "bound": "mfma",
                    "achieved": round(ach, 3), "peak": 157.3, "unit": "TFLOP/s", "frac": round(ach / 157.3, 4),
                    "traffic": None, "avg_launch_ms": round(ms, 4), "alg_flops_per_launch": flops,
                    "alg_bytes_per_launch": alg_bytes, "share_of_step": share}
        else:
            # 16-bit operand path: 0.03 ms of matrix work at peak vs 0.09 ms of HBM traffic -> HBM-bound
            # PMC counters cannot be collected inside this run: the figure comes from the committed rocprofv3 passes over THIS
            # kernel.  It is tied to the kernel's source: the summary records the hash of the SIREN kernel files it was
            # measured on, and a mismatch (kernel edited since) reports no traffic instead of a stale one.
            traffic, traffic_note = None, None
            try:
                with open(os.path.join(ROOT, "profiles", PMC_FILE[pe16])) as f:
                    pmc = json.load(f)
                if n == 4096:
                    if pmc.get("kernel_source_sha16") == siren_source_sha16():
                        traffic = pmc["hbm_bytes_per_launch"]
                    else:
                        traffic_note = "stale: the SIREN kernel sources changed since the counters were collected"
            except (OSError, KeyError, ValueError):
                traffic = None
            ach = alg_bytes / t / 1e9
            # What actually limits this kernel is neither of the two: it is vector-ISSUE bound (one sine and one cosine per
            # hidden unit and pixel, packed conversions, the address / loss arithmetic).  `valu` prices the tile loop's
            # instruction census (tools/siren_census.py, from the compiler's assembly of these very sources) at the SIMD's issue
            # costs and at the shader clock MEASURED inside the timed launches (rcb_siren_desc.clock_probe: every one of the
            # first 256 workgroups reads s_memtime and the 100 MHz s_memrealtime on its own CU at its start and end): floor =
            # tiles per SIMD x issue cycles per tile / clock.  `bound` keeps the contract's vocabulary
            # (the higher of the HBM and MFMA fractions); `limiter` names the real one.
            valu = None
            try:
                with open(os.path.join(ROOT, "profiles", "r05_siren_isa_census.json")) as f:
                    cen = json.load(f)
                if cen.get("source_sha16") == siren_census_sha16():
                    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
                    n_simd = 4 * n_cu
                    tiles = n * (1024 // 32)
                    cyc = cen["vector_issue_cycles_per_tile"]
                    per = tiles / n_simd / (siren_clock_ghz * 1e9) * 1e3          # ms per (cycle per tile and SIMD)
                    fl = cen["floors_per_tile"]
                    # every SIMD holds two tiles (two waves); per tile and SIMD: the matrix pipe and the transcendental unit
                    # serve both waves, the CU's one LDS serves the four SIMDs, one wave's vector issue stream is the upper
                    # end of the vector-issue cost (two waves interleave)
                    floors = {"matrix_pipe_ms": round(fl["matrix_pipe_per_simd"] * per, 4),
                              "transcendental_unit_ms": round(fl["transcendental_unit_per_simd"] * per, 4),
                              "lds_ms": round(fl["lds_per_cu_for_one_tile_on_each_simd"] * per, 4),
                              "vector_issue_one_stream_ms": round(cyc * per, 4)}
                    floor_ms = cyc * per
                    valu = {"instructions_per_tile": cen["instructions"], "by_class": cen["by_class"],
                            "vector_issue_cycles_per_tile": cyc, "transcendental_cycles_per_tile": cen["transcendental_cycles_per_tile"],
                            "mfma_pipe_cycles_per_tile": cen["mfma_pipe_cycles_per_tile"],
                            "lds_cycles_per_tile_and_wave": cen["lds_cycles_per_tile_and_wave"],
                            "scratch_instructions_in_tile_loop": cen["scratch_instructions_in_tile_loop"],
                            "tiles_per_launch": tiles, "simds": n_simd, "shader_clock_ghz_measured": round(siren_clock_ghz, 3),
                            "floors_at_measured_clock": floors,
                            "issue_floor_ms_at_measured_clock": round(floor_ms, 4),
                            "issue_floor_ms_at_2.4ghz": round(tiles / n_simd * cyc / 2.4e9 * 1e3, 4),
                            "frac_of_issue_floor": round(floor_ms / ms, 4),
                            "frac_of_largest_pipe_floor": round(max(floors["matrix_pipe_ms"], floors["transcendental_unit_ms"], floors["lds_ms"]) / ms, 4),
                            "source": "profiles/r05_siren_isa_census.json (tools/siren_census.py)"}
                else:
                    valu = {"note": "stale census: the SIREN kernel sources changed since tools/siren_census.py ran"}
            except (OSError, KeyError, ValueError):
                valu = None
            busy = (valu or {}).get("frac_of_issue_floor")
            limiter = ("vector issue" if (busy is not None and busy >= 0.8) else
                       "no single pipe: two waves per SIMD share a matrix pipe, a transcendental unit and (with the other three SIMDs) "
                       "the LDS, each 30-50 % busy, and the phases of the two waves overlap only partly (profiles/r05_siren_sq_pmc.json)"
                       if busy is not None else "unknown (no census)")
            roof = {"kernel": "fused SIREN fwd+MSE+bwd, bf16 MFMA (rcb_siren_loss_bwd)", "bound": "hbm", "limiter": limiter,
                    "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                    "traffic": traffic, "traffic_source": "profiles/%s (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)" % PMC_FILE[pe16],
                    "traffic_note": traffic_note,
                    "avg_launch_ms": round(ms, 4), "alg_bytes_per_launch": alg_bytes, "alg_flops_per_launch": flops,
                    "mfma_tflops": round(flops / t / 1e12, 1), "mfma_frac": round(flops / t / 2.5e15, 4), "valu": valu,
                    "share_of_step": share}
    # ---- sharded runs: what the per-step collectives move, and how much compute they have to hide behind -----------------
    comm = None
    if ws > 1:
        comm = comm_report(m, lt, up, dev)
    cpu = None
    extras = {}
    if rank == 0 and ws == 1 and not a.no_extras:
        def one_stream_table():
            # per-kernel durations are taken from the one-stream form of the same step (a second capture, dropped afterwards)
            keep = os.environ.get("RCB_FORK")
            os.environ["RCB_FORK"] = "0"
            try:
                run(6)
                tab = kernel_table(run, 10, n, D)
            finally:
                if keep is None:
                    del os.environ["RCB_FORK"]
                else:
                    os.environ["RCB_FORK"] = keep
                run(6)
            # the same sum for the SHIPPED (three-stream) form: kernels stretch while they run beside each other, so this total
            # against the step time says how much of the co-running is time-slicing rather than overlap
            shipped = kernel_table(run, 10, n, D)
            tab["shipped_form"] = {"kernel_us_per_step_total": shipped["kernel_us_per_step_total"],
                                   "kernels_per_step": shipped["kernels_per_step"],
                                   "sum_over_step_time": round(shipped["kernel_us_per_step_total"] / (el / a.steps * 1e6), 3),
                                   "one_stream_sum_over_step_time": round(tab["kernel_us_per_step_total"] / (el / a.steps * 1e6), 3)}
            return tab
        for key, fn in (("kernels", one_stream_table), ("rec", lambda: rec_roofline(dev)),
                        ("psnr_bpp", lambda: psnr_at_bpp(dev, m.precision)), ("presets", lambda: presets_table(dev, siren_clock_ghz)),
                        ("sharded_step_cost_captured", lambda: segment_host_cost(dev, cfg, n, tm, el / a.steps * 1e3, True)),
                        ("sharded_step_cost_segments", lambda: segment_host_cost(dev, cfg, n, tm, el / a.steps * 1e3, False))):
            try:
                extras[key] = fn()
            except Exception as exc:            # extras never take the bench line down; the failure is visible in it
                extras[key] = {"error": repr(exc)[:300]}
    if rank == 0 and ws == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(cfg)
    if rank == 0:
        inr_steps = n * ws * a.steps / el
        out = {"metric": "INRs trained/sec (whole node) + PSNR@bpp vs reference, CIFAR-10", "value": inr_steps / STEPS_PER_INR,
               "unit": "INRs trained/s", "n_gpus": ws, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if m.precision == 0 else "bf16", "data": "synthetic",
               "inr_steps_per_sec": inr_steps, "steps_per_inr_reference_schedule": STEPS_PER_INR,
               "config": {"workload": f"CIFAR-10 32x32, {n} INRs/GPU, 3x32 SIREN (in 32 = 16 Fourier + 16 upsampled pe), "
                                      f"S=1, training_mappings={tm}, Adam lr 2e-4", "inrs_per_gpu": n,
                          "parallelism": f"datapoint-sharded x{ws}", "tuned_library_gemms": bool(tuned)},
               "roofline": roof, "cpu_baseline": cpu}
        # the WHOLE step against the two chip rooflines, with SURVEY section 8(d)'s algorithmic figures per INR-step (CIFAR
        # preset, S = 1): unavoidable HBM traffic 0.34 MB (posterior parameters + Adam moments read and written, noise, targets,
        # pe in and its gradient out), arithmetic 19.46 (MLP) + 20.13 (A transform) + 191.9 (upsampling net as the reference
        # evaluates it) MFLOP.  The `roofline` object above describes the largest kernel (a fifth of the step); this one says
        # how far the step as a sequence of kernels sits from a fused ideal.
        if m.precision != 0 and cfg["pixel_sizes"] == [32, 32]:
            sb, sf, ts = 0.34e6 * n, (19.46 + 20.13 + 191.9) * 1e6 * n, el / a.steps
            out["step_roofline"] = {"alg_bytes_per_step": sb, "alg_flops_per_step": sf, "hbm_frac": round(sb / ts / 8e12, 4),
                                    "mfma_frac": round(sf / ts / 2.5e15, 4), "ms_per_step": round(ts * 1e3, 4),
                                    "source": "SURVEY.md section 8(d) per-INR-step figures x INRs per GPU"}
        if comm is not None:
            out["comm"] = comm
        out.update(extras)
        print(json.dumps(out))
    if ws > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
