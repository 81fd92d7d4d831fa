#!/bin/bash
# rocprofv3 evidence for the round (run on the GPU box; summaries are copied to profiles/ afterwards):
#   bash tools/collect_profiles.sh <tag>
# kernel stats of the bench step, of the A* scorer, and HBM traffic of the SIREN kernel (PMC in passes of their own).
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rec -- python3 tools/bench_rec.py > $OUT/rec.json 2> $OUT/rec.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/siren -- python3 tools/run_siren.py bf16 4096 10 pe16 > $OUT/siren.log 2> $OUT/siren.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/run_siren.py bf16 4096 3 pe16 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/run_siren.py bf16 4096 3 pe16 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq -- python3 tools/run_siren.py bf16 4096 3 pe16 > $OUT/pmc_sq.log 2>&1
ls -R $OUT | head -60
