"""Copies the round-5 rocprofv3 summaries from gpurun_out/prof_r05 (tools/collect_r05.sh) into profiles/ and writes the
HBM-traffic records bench.py reads:  python3 tools/summarize_r05.py   (build container, after the gpurun call)

  profiles/r05_<label>_kernel_stats.csv     rocprofv3 --kernel-trace --stats, one per workload
  profiles/r05_siren_bf16_pmc.json          FETCH_SIZE x 2 + WRITE_SIZE of the SIREN kernel (MI355X_MICROARCH.md, HBM section),
                                            stamped with the hash of the kernel's sources (bench.siren_source_sha16)
  profiles/r05_atrans_pmc.json              the same for the A-transform kernel + its SQ counters
  profiles/r05_presets_step_times.log       tools/bench_presets.py
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "gpurun_out", "prof_r05")
SIREN = "siren_wave_kernel"          # the default family of the headline instance (siren_mlp_wave.hip)
DST = os.path.join(ROOT, "profiles")


def counters(label, match):
    """per-dispatch counter values of the kernels whose name contains `match`, in dispatch order: {counter: [values]}"""
    out = collections.defaultdict(dict)
    files = sorted(glob.glob(os.path.join(SRC, label, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:           # gpurun merges into an existing directory: only the newest run of a label counts
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                d = out[r["Counter_Name"]]
                d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return {c: [v for _, v in sorted(d.items())] for c, d in out.items()}


def stats_row(label, match):
    for f in glob.glob(os.path.join(SRC, label + "_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if match in r["Name"]:
                return dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3)
    return None


def mean(v):
    return sum(v) / len(v)


def main():
    import bench
    for f in glob.glob(os.path.join(SRC, "*_kernel_stats.csv")):
        label = os.path.basename(f)[:-len("_kernel_stats.csv")]
        name = "r05_bench_bf16_kernel_stats.csv" if label == "bench" else "r05_%s_kernel_stats.csv" % label
        shutil.copy(f, os.path.join(DST, name))
        print("copied", name)
    if os.path.exists(os.path.join(SRC, "presets.log")):
        shutil.copy(os.path.join(SRC, "presets.log"), os.path.join(DST, "r05_presets_step_times.log"))

    # ---- SIREN: run_siren.py launches the loss kernel 3 + reps times with dpe, then as often without
    fetch = counters("pmc_siren_fetch", SIREN).get("FETCH_SIZE", [])
    write = counters("pmc_siren_write", SIREN).get("WRITE_SIZE", [])
    if fetch and write:
        # the launches without dpe (second loop) and the forward-only ones (third) write far less: keep those within 20 % of
        # the largest write
        with_dpe = [i for i, w in enumerate(write) if w > 0.8 * max(write)]
        fk, wk = mean([fetch[i] for i in with_dpe]), mean([write[i] for i in with_dpe])
        alg = bench.siren_alg_bytes(4096) if hasattr(bench, "siren_alg_bytes") else 425836544
        rec = {
            "kernel": "siren_wave_kernel<__bf16,3,16,16,3,MODE_LOSS,dpe> (rcb_siren_loss_bwd, one wave per row), 4096 INRs x 1024 px, pe / dpe stored as bf16, "
                      "launched as in the training step: rows on 128-byte lines, bf16 copy of dwvec written by the epilogue "
                      "(4096 x 3267 x 2 B = 26.8 MB that the algorithmic figure does not count)",
            "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 tools/run_siren.py bf16 4096 3 pe16 32 step (separate passes, tools/collect_r05.sh); launches that write dpe",
            "kernel_source_sha16": bench.siren_source_sha16(),
            "launches_counted": len(with_dpe),
            "FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1),
            "fetch_correction": "x2: gfx950 FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane streaming loads (MI355X_MICROARCH.md, HBM)",
            "fetch_bytes_corrected": int(fk * 1024 * 2), "write_bytes": int(wk * 1024),
            "hbm_bytes_per_launch": int(fk * 1024 * 2 + wk * 1024),
            "algorithmic_bytes_per_launch": alg,
            "avg_launch_us_in_bench_step": (stats_row("bench", SIREN) or {}).get("avg_us"),
            "avg_launch_us_back_to_back": (stats_row("siren", SIREN) or {}).get("avg_us"),
        }
        json.dump(rec, open(os.path.join(DST, "r05_siren_bf16_pmc.json"), "w"), indent=1)
        print("r05_siren_bf16_pmc.json", rec["hbm_bytes_per_launch"], "B per launch,", rec["kernel_source_sha16"])

    # ---- A transform: every launch of atrans_kernel is the same size (forward and data gradient of 4096 rows)
    fa = counters("pmc_atrans_fetch", "atrans_kernel").get("FETCH_SIZE", [])
    wa = counters("pmc_atrans_write", "atrans_kernel").get("WRITE_SIZE", [])
    sq = {c: mean(v) for c, v in counters("pmc_atrans_sq", "atrans_kernel").items()}
    if fa and wa:
        D = 3 * 1056 + 99
        rec = {
            "kernel": "atrans_kernel<2> (rcb_atrans_apply): [4096 x 3267] fp32 rows times the packed bf16 images of A (1056^2 x 3 + 99^2), two split terms",
            "command": "rocprofv3 --kernel-trace --pmc ... -- python3 tools/run_atrans.py 4096 2 4 (separate passes, tools/collect_r05.sh)",
            "FETCH_SIZE_KB": round(mean(fa), 1), "WRITE_SIZE_KB": round(mean(wa), 1),
            "fetch_bytes": int(mean(fa) * 2048), "write_bytes": int(mean(wa) * 1024),
            "hbm_bytes_per_launch": int(mean(fa) * 2048 + mean(wa) * 1024),
            "fetch_note": "x2 applied: the known-bytes probe of round 4 (tools/native/glds_fetch_probe.cpp, r04_glds_fetch_probe.json) "
                          "reads exactly half of 1 GiB for ordinary 16-byte loads AND for global_load_lds_dwordx4 -- LDS-DMA requests are "
                          "counted like streaming loads, so the kernel's read traffic is 2 x FETCH_SIZE: ONE ratio",
            "traffic_over_algorithmic": round((mean(fa) * 2048 + mean(wa) * 1024) / (4096 * (3 * 1056 + 99) * 4 * 2 + 2 * (3 * 1056 * 1056 + 128 * 128) * 2), 3),
            "algorithmic_bytes_per_launch": 4096 * D * 4 * 2 + 2 * (3 * 1056 * 1056 + 128 * 128) * 2,
            "algorithmic_flops_per_launch": 2 * 2 * 4096 * (3 * 1056 * 1056 + 99 * 99),
            "avg_launch_us_in_bench_step": (stats_row("bench", "atrans_kernel") or {}).get("avg_us"),
            "avg_launch_us_alone": (stats_row("atrans", "atrans_kernel") or {}).get("avg_us"),
            "SQ_mean_per_launch": sq,
        }
        json.dump(rec, open(os.path.join(DST, "r05_atrans_pmc.json"), "w"), indent=1)
        print("r05_atrans_pmc.json", rec["FETCH_SIZE_KB"], rec["WRITE_SIZE_KB"])


def siren_sq():
    """SQ counters of the two SIREN families (tools/ab_siren_wave.py under two PMC passes): per-launch means"""
    rec = {}
    for label in ("pmc_siren_sq_a", "pmc_siren_sq_b"):
        for fam, match in (("wave_per_row", "siren_wave_kernel"), ("workgroup_per_row", "siren_bf16_kernel")):
            for c, v in counters(label, match).items():
                rec.setdefault(fam, {})[c] = mean(v)
    if rec:
        for fam, d in rec.items():
            if "SQ_WAVE_CYCLES" in d:
                for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
                    if k in d:
                        d[k + "_over_WAVE_CYCLES"] = round(d[k] / d["SQ_WAVE_CYCLES"], 4)
            if "SQ_LDS_IDX_ACTIVE" in d and "SQ_BUSY_CYCLES" in d:
                d["lds_busy_frac"] = round(d["SQ_LDS_IDX_ACTIVE"] / (d["SQ_BUSY_CYCLES"] / 32 * 256), 4)     # busy cycles: sum over 32 SEs
                d["lds_conflict_frac"] = round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], 4)
                d["mfma_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["SQ_BUSY_CYCLES"] / 32 * 256 * 4), 4)
        out = {"what": "rocprofv3 --pmc (two passes) over tools/ab_siren_wave.py 4096 1 10 0,1: the two width-32 bf16 SIREN loss / backward "
                       "families on the same inputs, means over their launches; *_over_WAVE_CYCLES = share of the waves' lifetime",
               "families": rec}
        json.dump(out, open(os.path.join(DST, "r05_siren_sq_pmc.json"), "w"), indent=1)
        print("r05_siren_sq_pmc.json", {f: {k: v for k, v in d.items() if k.endswith(("_CYCLES", "frac"))} for f, d in rec.items()})


if __name__ == "__main__":
    main()
    siren_sq()
