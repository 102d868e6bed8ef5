#!/bin/bash
# gpurun_out/prof (tools/run_profiles.sh) + gpurun_out/segtrace_out.txt (tools/seg_trace.sh) -> profiles/r03_*
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/gpurun_out/prof
cp $P/bench_n1.json $R/profiles/r03_bench_n1.json
cp $P/kernel_stats.csv $R/profiles/r03_kernel_stats_hipgraph_b16_512x1024.csv
cp $P/dominant_dispatches.csv $R/profiles/r03_dominant_dispatches.csv
cp $P/dominant_pmc.json $R/profiles/r03_dominant_pmc.json
cp $P/pmc_FETCH_SIZE.csv $R/profiles/r03_dominant_pmc_FETCH_SIZE.csv
cp $P/pmc_WRITE_SIZE.csv $R/profiles/r03_dominant_pmc_WRITE_SIZE.csv
cp $P/pmc_MFMA.csv $R/profiles/r03_dominant_pmc_MFMA.csv
cp $P/roofline_table.md $R/profiles/r03_roofline_table.md
[ -f $R/gpurun_out/segtrace_out.txt ] && cp $R/gpurun_out/segtrace_out.txt $R/profiles/r03_seg_decoder_dispatches_phase_vs_fullres.txt
ls -la $R/profiles | grep r03
