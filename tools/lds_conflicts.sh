#!/bin/bash
# LDS bank conflicts of every kernel of the bench step (PMC pass of its own):  bash tools/lds_conflicts.sh   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_lds
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/run -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_lds/run/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f'{"kernel":70s} {"n":>4s} {"lds_active":>12s} {"conflict":>12s} {"confl/act":>9s} {"lds/busy":>9s} {"mfma/busy":>9s}')
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if m.get("SQ_LDS_IDX_ACTIVE", 0) < 1e4:
        continue
    busy = m["SQ_BUSY_CYCLES"] / 32 * 256          # busy cycles are summed over 32 shader engines; LDS counters over 256 CUs
    print(f'{k:70s} {len(c["SQ_BUSY_CYCLES"]):4d} {m["SQ_LDS_IDX_ACTIVE"]:12.3g} {m["SQ_LDS_BANK_CONFLICT"]:12.3g} '
          f'{m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]:9.2f} {m["SQ_LDS_IDX_ACTIVE"] / busy:9.2f} {m["SQ_VALU_MFMA_BUSY_CYCLES"] / (busy * 4):9.2f}')
PY
