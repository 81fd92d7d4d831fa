#!/bin/bash
# same-box A/B of two builds of the library by per-kernel rocprofv3 averages:  bash tools/ab_kernels.sh <prev.so> [pattern]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PAT=${2:-upconv}
rm -rf gpurun_out/ab_new gpurun_out/ab_prev
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_new -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/ab_new.json 2> gpurun_out/ab_new.err
export RCB_LIB=$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_prev -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/ab_prev.json 2> gpurun_out/ab_prev.err
unset RCB_LIB
python3 - "$PAT" <<'PY'
import csv, glob, sys
pat = sys.argv[1]
def load(d):
    f = glob.glob("gpurun_out/%s/*/*kernel_stats.csv" % d)[0]
    return {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
new, prev = load("ab_new"), load("ab_prev")
for k in sorted(new, key=lambda k: -new[k]):
    if pat in k or "siren" in k:
        print("%-70s new %8.1f us   prev %8.1f us" % (k[:70], new[k], prev.get(k, float("nan"))))
PY
grep -o "ms_per_step\": [0-9.]*" gpurun_out/ab_new.json | sed "s/^/new  /"; grep -o "ms_per_step\": [0-9.]*" gpurun_out/ab_prev.json | sed "s/^/prev /"
