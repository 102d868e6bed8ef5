#!/bin/bash
# every kernel of the library whose register allocation spills (VGPRs Spill > 0), per source file -- run after touching shared kernel code:
# an unrolled loop added to one template path cost the 128-cout direct conv 22 spilled VGPRs (-0.6 % of the step) in round 5
R=$(cd "$(dirname "$0")/.." && pwd)
for f in hn_gemm hn_norm hn_fused hn_stencil hn_loss hn_post; do
  $R/tools/kernel_resources.sh $f.hip '.' 2>/dev/null | grep -v "Spill': 0}" | sed "s/^/$f: /"
done
echo "(expected: gemm_tn_group_kernel<128,128,2,2,64,-2> with 1 spilled VGPR)"
