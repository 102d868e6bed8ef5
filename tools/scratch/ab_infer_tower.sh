cd $GRAFT_REPO_ROOT
export HN_TUNING=${HN_TUNING:-ab}    # policy switches are read from the environment only under HN_TUNING=1|ab (_lib.policy)
ARGS="--infer --batch 32 --res 1152x1920 --no-cpu-baseline --no-extras --no-roofline --steps 20 --warmup 5"
one() { python3 bench.py $ARGS 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3))"; }
for r in 1 2 3; do one epilogue; HN_TOWER_BN_IN_GEMM=0 one pass; done
