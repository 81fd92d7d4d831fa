#!/bin/bash
# kernel time of rcb_upconv_weff_grad under rocprofv3 for the library selected by RCB_LIB:  bash tools/weff_grad_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_wg_$1
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/weff_grad_time.py > $OUT/log 2>&1
grep -h "weff_grad" $OUT/*/*kernel_stats.csv | cut -d, -f2-5
