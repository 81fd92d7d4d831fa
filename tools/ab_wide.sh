#!/bin/bash
# wide SIREN kernels alone, two builds on one box: bash tools/ab_wide.sh <other.so>
cd $GRAFT_REPO_ROOT
for rep in 1; do
for w in 48 64; do
  for p in bf16 f16; do
    a=$(python3 tools/run_siren.py $p 4096 10 pe16 $w | grep -m1 loss_bwd)
    b=$(RCB_LIB=$GRAFT_REPO_ROOT/$1 python3 tools/run_siren.py $p 4096 10 pe16 $w | grep -m1 loss_bwd)
    echo "W=$w $p  new: $a   |   other: $b"
  done
done
done
