#!/bin/bash
# register spills of every instance of the wide SIREN kernel (build-container check, no GPU):  bash tools/wide_spills.sh
cd "$(dirname "$0")/../recombiner_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c siren_mlp_wide.hip -o /tmp/wide.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
name=None
for line in sys.stdin:
    if 'error' in line or 'warning' in line: print(line)
    m=re.search(r'Function Name: (\S+)',line)
    if m: name=m.group(1); continue
    m=re.search(r'(VGPRs Spill): (\d+)',line)
    if m and 'ELi0ELb0' not in name: print(name[-60:], 'spill', m.group(2))
"
