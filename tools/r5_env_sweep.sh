#!/bin/bash
# HIP runtime (clr) environment knobs vs the captured step: img/s, ms/step, loss (must stay 16596.0898)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5h
rm -rf $O; mkdir -p $O
cd $R
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['loss'])"; }
for e in "X=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" \
         "DEBUG_CLR_SKIP_RELEASE_SCOPE=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_KERNARG_COPY_OPT=0" "GPU_MAX_HW_QUEUES=1" "DEBUG_HIP_DYNAMIC_QUEUES=0" "X=1"; do
  run "$e" >> $O/sweep.log 2>&1
done
cat $O/sweep.log
