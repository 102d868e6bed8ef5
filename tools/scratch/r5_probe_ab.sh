#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5m; mkdir -p $O
export HN_TUNING=ab
ARGS="--no-cpu-baseline --no-extras --no-roofline --steps 60 --warmup 10"
one() { python3 bench.py $ARGS $2 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms')"; }
for r in 1 2 3; do
  one shipped
  HN_PROBE_A=1 one no_bn1_apply_launch
  HN_PROBE_B=1 one no_bn2_reduce_launch
done 2>&1 | tee $O/probe_ab.log
