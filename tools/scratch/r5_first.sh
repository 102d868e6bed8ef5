#!/bin/bash
# round-5 first GPU call: new data-parallel / cache tests, baseline bench at both resolutions, per-kernel stats of the 640x640 step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5a
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_post_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
python bench.py --no-cpu-baseline --no-extras > $O/b512.log 2>&1
python bench.py --no-cpu-baseline --no-extras --res 640x640 > $O/b640.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $R
for res in 640x640 512x1024; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$res -- python3 bench.py --no-cpu-baseline --no-optimizer --no-extras --no-roofline --res $res --steps 8 --warmup 2 > $O/kt$res.log 2>&1
  f=$(find $O/kt$res -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$res.csv
  rm -rf $O/kt$res
done
tail -3 $O/tests.log; tail -1 $O/b512.log | cut -c1-300; tail -1 $O/b640.log | cut -c1-300
