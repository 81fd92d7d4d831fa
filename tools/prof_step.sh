#!/bin/bash
# per-kernel rocprofv3 averages of the bench step:  bash tools/prof_step.sh <tag> [dir with bench.py]   (on the GPU box)
TAG=${1:-step}
DIR=${2:-.}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $DIR
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/bench/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot / 1e6)
    for r in rows[:28]:
        print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"])/1e3:8.1f} pct {float(r["Percentage"]):5.1f}')
PY
