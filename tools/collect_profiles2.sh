#!/bin/bash
# rocprofv3 kernel stats of the other workloads (test-time step + encode rounds, the other presets, the width variants):
#   bash tools/collect_profiles2.sh <tag>      -> gpurun_out/prof2_<tag>/<label>_kernel_stats.csv
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof2_$TAG
rm -rf $OUT && mkdir -p $OUT
run() {
  label=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$label -- "$@" > $OUT/$label.log 2> $OUT/$label.err
  cp $OUT/$label/*/*kernel_stats.csv $OUT/${label}_kernel_stats.csv 2>/dev/null
  echo "$label: $(tail -n 2 $OUT/$label.log | tr '\n' ' ' | cut -c1-300)"
}
run testtime_compress python3 tools/bench_compress.py bf16
run kodak python3 tools/prof_preset.py kodak 2 32 1
run audio python3 tools/prof_preset.py audio 8 32 1
run video python3 tools/prof_preset.py video 4 32 1
run protein python3 tools/prof_preset.py protein 4096 32 1
run kodak_w48 python3 tools/prof_preset.py kodak 2 48 1
run video_w64_f16 python3 tools/prof_preset.py video 4 64 2
run cifar_w64 python3 tools/prof_preset.py cifar 4096 64 1
run siren_wide_w64 python3 tools/run_siren.py bf16 4096 10 pe16 64
run siren_wide_w48 python3 tools/run_siren.py bf16 4096 10 pe16 48
run siren_wide_w64_f16 python3 tools/run_siren.py f16 4096 10 pe16 64
python3 tools/bench_presets.py > $OUT/presets.log 2>&1
cat $OUT/presets.log | grep "ms/step"
