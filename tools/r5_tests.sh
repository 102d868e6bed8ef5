#!/bin/bash
# whole -m gpu suite (no -x: every failure in one call)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5tests
rm -rf $O; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests/ -q -m gpu "$@" > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
grep -n "passed\|failed\|^FAILED\|^ERROR" $O/tests.log | tail -30
