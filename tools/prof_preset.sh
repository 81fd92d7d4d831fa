#!/bin/bash
# rocprofv3 kernel stats of one preset's eager training step:  bash tools/prof_preset.sh <tag> <preset> [datapoints] [width] [precision]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 tools/prof_preset.py "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/run/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot / 1e6)
    for r in rows[:22]:
        print(f'{r["Name"][:100]:100s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"])/1e3:8.1f} pct {float(r["Percentage"]):5.1f}')
PY
