#!/bin/bash
# SQ counters of the two width-32 SIREN loss / backward families, one PMC pass each group:  bash tools/pmc_siren.sh   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_siren
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 tools/ab_siren_wave.py 4096 1 10 0,1 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES --output-format csv -d $OUT/b -- python3 tools/ab_siren_wave.py 4096 1 10 0,1 > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for grp in "ab":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_siren/{grp}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "siren" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        print(k, len(next(iter(c.values()))), " ".join(f"{n}={v:.4g}" for n, v in sorted(m.items())))
PY
