#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5f
rm -rf $O; mkdir -p $O
cd $R
python tools/overlap_probe.py 40 > $O/overlap_probe.txt 2>&1
python tools/overlap_probe.py 120 >> $O/overlap_probe.txt 2>&1
timeout 2700 python -m pytest tests/ -q -m gpu > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
grep -v "^\[\|Warning" $O/overlap_probe.txt | tail -16; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/tests.log | tail -12
