#!/bin/bash
# round 5: hn_se_gate_apply -- parity tests, then same-box A/B of the training step and the inference step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "se_excite or se_gate" > $O/tests_kernel.log 2>&1; tail -3 $O/tests_kernel.log
timeout 1200 python -m pytest tests/test_model_gpu.py -q > $O/tests_model.log 2>&1; tail -3 $O/tests_model.log
export HN_TUNING=ab
ARGS="--no-cpu-baseline --no-extras --no-roofline --steps 60 --warmup 10"
one() { python3 bench.py $ARGS $2 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms')"; }
for r in 1 2 3; do
  one gate_apply
  HN_SE_GATE_APPLY=0 one two_launch
  HN_LIB_AB=$R/multitask_hydranet_amd/libhydranet_hip_cw32.so one cw32
  HN_LIB_AB=$R/multitask_hydranet_amd/libhydranet_hip_cw64.so one cw64
done 2>&1 | tee $O/ab.log
for r in 1 2; do
  one infer_gate_apply "--infer --batch 32 --res 1152x1920"
  HN_SE_GATE_APPLY=0 one infer_two_launch "--infer --batch 32 --res 1152x1920"
  one r640_gate_apply "--res 640x640"
  HN_SE_GATE_APPLY=0 one r640_two_launch "--res 640x640"
done 2>&1 | tee -a $O/ab.log
