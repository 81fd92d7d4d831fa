#!/bin/bash
# rocprofv3 evidence of round 5 (run on the GPU box; tools/summarize_r05.py then writes what is kept under profiles/):
#   bash tools/collect_r05.sh && python3 python3 tools/summarize_r05.py
# Kernel statistics of the bench step and of every preset, the A-transform kernels alone, and HBM traffic (PMC, passes of
# their own, never combined with other trace domains) of the SIREN kernel and the A-transform kernel.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r05
rm -rf $OUT && mkdir -p $OUT
run() {
  label=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$label -- "$@" > $OUT/$label.log 2> $OUT/$label.err
  cp $OUT/$label/*/*kernel_stats.csv $OUT/${label}_kernel_stats.csv 2>/dev/null
  echo "$label: $(tail -n 1 $OUT/$label.log | cut -c1-200)"
}
pmc() {
  label=$1; shift; counters=$1; shift
  rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $OUT/$label -- "$@" > $OUT/$label.log 2>&1
  echo "$label: done"
}
run bench python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras
run atrans python3 tools/run_atrans.py 4096 2 10
run siren python3 tools/run_siren.py bf16 4096 10 pe16 32 step
run testtime_compress python3 tools/bench_compress.py bf16
run kodak python3 tools/prof_preset.py kodak 2 32 1
run audio python3 tools/prof_preset.py audio 8 32 1
run video python3 tools/prof_preset.py video 4 32 1
run kodak_w48 python3 tools/prof_preset.py kodak 2 48 1
run video_w64_f16 python3 tools/prof_preset.py video 4 64 2
# the presets at a rank's shard (BASELINE configs[2..4]) and the test-time step of a batch of 8 photos
run audio_1024clips python3 tools/prof_preset.py audio 1024 32 1
run kodak_w48_24photos python3 tools/prof_preset.py kodak 24 48 1
run video_w64_f16_32clips python3 tools/prof_preset.py video 32 64 2
run testtime_kodak_w48_8photos python3 tools/prof_testtime.py kodak 8 48 20
# the stand-alone benchmarks of the round's new phase-conv / stage-1 kernels at the shard shapes (their printed TB/s are from
# these durations)
run kernels_stage1 python3 tools/bench_stage1.py
run kernels_wgrad1d python3 tools/bench_wgrad1d.py
run kernels_phaseconv python3 tools/bench_phaseconv.py
pmc pmc_siren_fetch FETCH_SIZE python3 tools/run_siren.py bf16 4096 3 pe16 32 step
pmc pmc_siren_write WRITE_SIZE python3 tools/run_siren.py bf16 4096 3 pe16 32 step
pmc pmc_atrans_fetch FETCH_SIZE python3 tools/run_atrans.py 4096 2 4
pmc pmc_atrans_write WRITE_SIZE python3 tools/run_atrans.py 4096 2 4
pmc pmc_atrans_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" python3 tools/run_atrans.py 4096 2 4
# SQ counters of the two width-32 SIREN families (same inputs, one process): waits, issue, LDS, matrix pipe
pmc pmc_siren_sq_a "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" python3 tools/ab_siren_wave.py 4096 1 10 0,1
pmc pmc_siren_sq_b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES" python3 tools/ab_siren_wave.py 4096 1 10 0,1
python3 tools/bench_presets.py > $OUT/presets.log 2>&1
grep "ms/step" $OUT/presets.log
