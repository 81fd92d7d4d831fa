"""Timeline of ONE replayed training step from a rocprofv3 kernel trace (which kernel ran when, beside what):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
    python3 tools/step_timeline.py gpurun_out/tl [step index from the end, default 3]
A step is delimited by consecutive step_begin_kernel launches; times in us relative to the step's first kernel."""
import csv
import glob
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "step_begin_kernel" in r["Kernel_Name"]]
i0, i1 = starts[-back - 1], starts[-back]
t0 = int(rows[i0]["Start_Timestamp"])
span = (max(int(r["End_Timestamp"]) for r in rows[i0:i1]) - t0) / 1e3
periods = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 for a, b in zip(starts[-12:-1], starts[-11:])]
print("step of %d kernels, %.1f us from first start to last end; step period (start to start, last 11 steps) median %.1f us -> %.1f us between replays" % (
    i1 - i0, span, sorted(periods)[len(periods) // 2], sorted(periods)[len(periods) // 2] - span))
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%8.1f %8.1f %7.1f  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), name[:70]))
