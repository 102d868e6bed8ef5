#!/bin/bash
# round 5: tail fold of the reduce passes + hn_se_gate_apply row-block policy -- parity tests, microbench, same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "tail_fold or se_excite" > $O/tests_kernel.log 2>&1; tail -3 $O/tests_kernel.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -q > $O/tests_model.log 2>&1; tail -3 $O/tests_model.log
(cd tools && timeout 600 python3 bench_gate_apply.py > $O/bench_gate_apply.txt 2>&1); cut -c1-400 $O/bench_gate_apply.txt | tail -12
export HN_TUNING=ab
ARGS="--no-cpu-baseline --no-extras --no-roofline --steps 60 --warmup 10"
one() { python3 bench.py $ARGS $2 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms')"; }
for r in 1 2 3; do
  one both
  HN_TAIL_FOLD=0 one no_tail_fold
  HN_SE_GATE_APPLY=0 one no_gate_apply
  HN_TAIL_FOLD=0 HN_SE_GATE_APPLY=0 one neither
done 2>&1 | tee $O/ab.log
for r in 1 2; do
  one infer_gate_apply "--infer --batch 32 --res 1152x1920"
  HN_SE_GATE_APPLY=0 one infer_two_launch "--infer --batch 32 --res 1152x1920"
  one r640_both "--res 640x640"
  HN_TAIL_FOLD=0 HN_SE_GATE_APPLY=0 one r640_neither "--res 640x640"
done 2>&1 | tee -a $O/ab.log
