#!/bin/bash
export HN_TUNING=${HN_TUNING:-ab}    # product library; the package reads HN_LIB_AB / policy switches only under HN_TUNING=1|ab (_lib.policy)
# Same-box A/B of the whole training step: working-tree library (A) against every alternate in-tree build named on the command line
# (multitask_hydranet_amd/libhydranet_hip_<name>.so, e.g. "B" from tools/ab_head.sh), interleaved, REPS rounds (default 2).
#   tools/ab_run.sh B [C ...]        prints: variant img/s ms_per_step      (BENCH_ARGS="--infer --batch 32 --res 1152x1920": another workload)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ARGS="${BENCH_ARGS:-} --no-cpu-baseline --no-extras --no-roofline --steps ${STEPS:-60} --warmup 10"
one() {
  python3 bench.py $ARGS 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms', d.get('ms_per_step_median'))"
}
for r in $(seq ${REPS:-2}); do
  one A
  for v in "$@"; do HN_LIB_AB=$R/multitask_hydranet_amd/libhydranet_hip_$v.so one $v; done
done
