#!/bin/bash
# whole -m gpu suite + a default bench line (what the driver runs at round end)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5full
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
python tools/scratch/ddp_trainer_diag.py cap 2>&1 | grep "^step" > $O/diag_cap.log
python bench.py > $O/bench.log 2>&1
tail -4 $O/tests.log; cat $O/diag_cap.log | cut -c1-120; tail -1 $O/bench.log | cut -c1-400
