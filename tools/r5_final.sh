#!/bin/bash
# round-end check on one box: the whole -m gpu suite, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5final
rm -rf $O; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests/ -q -m gpu > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
python bench.py > $O/bench.log 2>&1; echo "rc $?" >> $O/bench.log
grep -n "passed\|failed\|^FAILED\|^ERROR" $O/tests.log | tail -5; tail -4 $O/smoke.log; tail -2 $O/bench.log | cut -c1-400
