#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5c
rm -rf $O; mkdir -p $O
cd $R
python tools/scratch/xblock_hw60_diag.py > $O/hw60.log 2>&1
python tools/scratch/xblock_hw60_diag.py 376 8 8 8 > $O/hw64.log 2>&1
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "bifpn_node or epilogue_reductions" > $O/tests_node.log 2>&1; echo "rc $?" >> $O/tests_node.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize2_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x > $O/tests_model.log 2>&1; echo "rc $?" >> $O/tests_model.log
export HN_TUNING=ab
for v in 1 0 1 0; do HN_SEPNODE=$v python bench.py --no-cpu-baseline --no-extras --no-roofline 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('sepnode $v', j['value'], j['ms_per_step'])"; done > $O/ab.log 2>&1
for v in 1 0; do HN_SEPNODE=$v python bench.py --no-cpu-baseline --infer --res 1152x1920 --batch 32 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('infer sepnode $v', j['value'], j['ms_per_step'])"; done >> $O/ab.log 2>&1
tail -30 $O/hw60.log; grep -n "passed\|failed\|^FAILED" $O/tests_node.log $O/tests_model.log | tail; cat $O/ab.log
