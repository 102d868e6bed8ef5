#!/bin/bash
# same-box sweep of the heuristic knobs (bench.py HN_KNOBS -> hn_debug_knob): prints img/s per setting; baseline first and last
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export HN_TUNING=1    # hn_debug_knob lives in the tuning build only
run() { echo -n "$1: "; HN_KNOBS="$1" python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; }
echo -n "default: "; python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
for s in "$@"; do run "$s"; done
echo -n "default: "; python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
