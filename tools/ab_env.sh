#!/bin/bash
# same-box A/B of environment switches of the step:  bash tools/ab_env.sh "VAR=0" ["VAR2=0" ...]  -> ms/step with the defaults and
# with each assignment, two rounds each (boxes drift: read differences well above the round-to-round spread only)
cd $GRAFT_REPO_ROOT
run() { env "$@" python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep -o "ms_per_step\": [0-9.]*" | cut -d" " -f2; }
for round in 1 2; do
  echo "round $round  default           $(run RCB_NOP=1)"
  for kv in "$@"; do echo "round $round  $kv    $(run $kv)"; done
done
