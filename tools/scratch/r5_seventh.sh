#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5g
rm -rf $O; mkdir -p $O
cd $R
export HN_TUNING=ab
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "packed_dots or grouped_conv or epilogue_reductions or batchnorm3 or deferred" > $O/tests_k.log 2>&1; echo "rc $?" >> $O/tests_k.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize2_gpu.py -q -m gpu > $O/tests_model.log 2>&1; echo "rc $?" >> $O/tests_model.log
for v in 1024 0 1024 0; do HN_GCONV_DOT_MAX_HW=$v python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 40 --warmup 10 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('gconv_dot max_hw $v', j['value'], j['ms_per_step'])"; done > $O/ab.log 2>&1
for v in 1024 0; do HN_GCONV_DOT_MAX_HW=$v python bench.py --no-cpu-baseline --no-extras --no-roofline --res 640x640 --steps 40 --warmup 10 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('640: gconv_dot max_hw $v', j['value'], j['ms_per_step'])"; done >> $O/ab.log 2>&1
bash tools/step_timeline.sh > /dev/null 2>&1; cp gpurun_out/trace/step.csv $O/step512.csv
grep -n "passed\|failed\|^FAILED" $O/tests_k.log $O/tests_model.log | tail; cat $O/ab.log
python3 - <<'PY'
import csv, collections, os
rows=list(csv.DictReader(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r5g/step512.csv")))
d=collections.OrderedDict()
for r in rows:
    if "gconv_dot" in r["kernel"] or "conv3x3_direct_kernel<64" in r["kernel"]:
        d.setdefault((r["kernel"][:40], r["grid_x"]),[]).append(float(r["dur_us"]))
for k,v in d.items(): print(k, len(v), round(sum(v)/len(v),1))
PY
