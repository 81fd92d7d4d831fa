#!/bin/bash
# same-box comparison of library builds by the SIREN kernel's rocprofv3 average inside the bench step:
#   bash tools/ab_siren.sh label1=lib1.so label2=lib2.so ...      (a label without "=" uses the in-tree library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@" "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  rm -rf gpurun_out/abs_$label
  if [ "$lib" != "$spec" ]; then export RCB_LIB=$GRAFT_REPO_ROOT/$lib; else unset RCB_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abs_$label -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/abs_$label.json 2> gpurun_out/abs_$label.err
  f=$(ls gpurun_out/abs_$label/*/*kernel_stats.csv | head -1)
  us=$(grep -i siren $f | head -1 | awk -F, '{print $(NF-4)}')
  ms=$(grep -o 'ms_per_step": [0-9.]*' gpurun_out/abs_$label.json | cut -d' ' -f2)
  echo "$label: siren avg ns $us   step ms $ms"
done
