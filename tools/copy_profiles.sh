#!/bin/bash
# gpurun_out/prof (tools/run_profiles.sh) + gpurun_out/segtrace_out.txt (tools/seg_trace.sh) -> profiles/${ROUND}_*
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/gpurun_out/prof
ROUND=${ROUND:-r05}
cp $P/bench_n1.json $R/profiles/${ROUND}_bench_n1.json
cp $P/kernel_stats.csv $R/profiles/${ROUND}_kernel_stats_hipgraph_b16_512x1024.csv
cp $P/dominant_dispatches.csv $R/profiles/${ROUND}_dominant_dispatches.csv
cp $P/dominant_pmc.json $R/profiles/${ROUND}_dominant_pmc.json
cp $P/pmc_FETCH_SIZE.csv $R/profiles/${ROUND}_dominant_pmc_FETCH_SIZE.csv
cp $P/pmc_WRITE_SIZE.csv $R/profiles/${ROUND}_dominant_pmc_WRITE_SIZE.csv
cp $P/pmc_MFMA.csv $R/profiles/${ROUND}_dominant_pmc_MFMA.csv
cp $P/roofline_table.md $R/profiles/${ROUND}_roofline_table.md
[ -f $R/gpurun_out/segtrace_out.txt ] && cp $R/gpurun_out/segtrace_out.txt $R/profiles/${ROUND}_seg_decoder_dispatches_phase_vs_fullres.txt
ls -la $R/profiles | grep ${ROUND}
