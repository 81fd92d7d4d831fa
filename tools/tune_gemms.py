"""One-off: tune the library GEMMs of the bench workload on this GPU and write the TunableOp result file.
   python tools/tune_gemms.py <out.csv>"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recombiner_amd import tuning
out = os.path.abspath(sys.argv[1])
tuning.enable_tuned_gemms(out, tune=True)
sys.argv = ["bench.py", "--steps", "12", "--warmup", "4", "--no-cpu-baseline", "--no-tuned-gemms"]
import bench
bench.main()
print("TunableOp writes", out, "at interpreter exit")
