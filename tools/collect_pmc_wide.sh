#!/bin/bash
# HBM traffic counters of the wide SIREN kernel (separate --pmc passes, as for the width-32 kernel):  bash tools/collect_pmc_wide.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_wide
rm -rf $OUT && mkdir -p $OUT
for W in 64 48; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$W -- python3 tools/run_siren.py bf16 4096 3 pe16 $W > $OUT/fetch_$W.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$W -- python3 tools/run_siren.py bf16 4096 3 pe16 $W > $OUT/write_$W.log 2>&1
done
python3 - <<'PY'
import csv, glob
for W in (64, 48):
    for kind, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        vals = []
        for f in glob.glob("gpurun_out/pmc_wide/%s_%d/*/*counter_collection.csv" % (kind, W)):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == name and "siren_wide" in r["Kernel_Name"]:
                    vals.append((r["Kernel_Name"][-40:], float(r["Counter_Value"])))
        print(W, name, sorted(set(round(v) for _, v in vals)))
PY
