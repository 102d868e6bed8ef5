#!/bin/bash
# A/B of environment knobs on one box: tools/ab_env.sh "HN_X=0" "HN_X=1" ... (each run: bench.py --no-extras --no-roofline, value + ms/step)
cd "$(dirname "$0")/.."
export HN_TUNING=1    # HN_KNOBS / HN_TN / HN_DIRECT_PIPE need the tuning build of the library
for rep in 1 2; do
for kv in "$@"; do
    out=$(env $kv python bench.py --no-extras --no-roofline --steps 30 2>/dev/null | tail -1)
    echo "$kv rep$rep $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ms_per_step_median"])')"
done
done
