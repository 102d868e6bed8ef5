#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5e
rm -rf $O; mkdir -p $O
cd $R
export HN_TUNING=ab
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "bifpn_node or epilogue_reductions" > $O/tests_node.log 2>&1; echo "rc $?" >> $O/tests_node.log
for v in 1 0 1 0; do HN_SEPNODE=$v python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 40 --warmup 10 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('sepnode(split) $v', j['value'], j['ms_per_step'])"; done > $O/ab.log 2>&1
for v in 1 0; do HN_SEPNODE=$v python bench.py --no-cpu-baseline --infer --res 1152x1920 --batch 32 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('infer sepnode(split) $v', j['value'], j['ms_per_step'])"; done >> $O/ab.log 2>&1
HN_SEPNODE=1 bash tools/step_timeline.sh > /dev/null 2>&1; cp gpurun_out/trace/step.csv $O/step512_node.csv
grep -n "passed\|failed\|^FAILED" $O/tests_node.log | tail -5; cat $O/ab.log
