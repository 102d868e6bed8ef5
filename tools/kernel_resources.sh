#!/bin/bash
# VGPRs / spills / LDS of the kernels of one HIP source whose mangled name matches a regex:  tools/kernel_resources.sh hn_gemm.hip 'conv3x3_direct'
set -eu
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/multitask_hydranet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 ${HN_TUNING:+-DHN_TUNING} -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys
pat = re.compile(sys.argv[1])
name = None
rows = {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m: name = m.group(1); rows[name] = {}; continue
    m = re.search(r'remark:\s+(VGPRs|AGPRs|VGPRs Spill|SGPRs|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\d+)', line)
    if m and name: rows[name][m.group(1)] = int(m.group(2))
for n, r in rows.items():
    if pat.search(n): print(n, r)
" "${2:-.}"
