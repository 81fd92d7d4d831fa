#!/bin/bash
# same-box A/B of the bench step with the two width-32 SIREN families (RCB_SIREN_WAVE=0 / 1), two rounds
for r in 1 2; do
  for w in 0 1; do
    RCB_SIREN_WAVE=$w python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null > gpurun_out/ab_bench_w$w.json
    python - "$w" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_bench_w%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("wave", sys.argv[1], "ms_per_step", d["ms_per_step"], "roofline", {k: r[k] for k in r if k in ("achieved", "frac", "kernel", "kernel_ms", "avg_launch_ms")})
PY
  done
done
