#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "tail_fold or se_excite" > $O/tests_kernel.log 2>&1; tail -3 $O/tests_kernel.log
export HN_TUNING=ab
ARGS="--no-cpu-baseline --no-extras --no-roofline --steps 60 --warmup 10"
one() { python3 bench.py $ARGS $2 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms')"; }
for r in 1 2 3; do
  one both
  HN_TAIL_FOLD=0 one no_tail_fold
done 2>&1 | tee $O/ab2.log
