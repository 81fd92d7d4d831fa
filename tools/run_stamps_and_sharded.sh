#!/bin/bash
# GPU-box helper: phase stamps of the SIREN kernel + the one-rank sharded-step forms
cd $GRAFT_REPO_ROOT
python tools/siren_stamps.py 2>&1 | grep -v amdgpu > gpurun_out/r04_siren_stamps.log
cat gpurun_out/r04_siren_stamps.log
python - <<'PY' 2>&1 | grep -v amdgpu
import json, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from recombiner_amd import config
dev = torch.device("cuda", 0)
cfg = config.configs["cifar"]
for cap in (True, False):
    print(json.dumps(bench.segment_host_cost(dev, cfg, 4096, True, 0.0, cap)), flush=True)
PY
