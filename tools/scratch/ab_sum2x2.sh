cd $GRAFT_REPO_ROOT
export HN_TUNING=${HN_TUNING:-ab}    # policy switches are read from the environment only under HN_TUNING=1|ab (_lib.policy)
ARGS="--no-cpu-baseline --no-extras --no-roofline --steps 60 --warmup 10"
one() { python3 bench.py $ARGS 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3))"; }
for r in 1 2 3; do one off; HN_FUSE_SUM2X2=1 one quads4; HN_FUSE_SUM2X2=1 HN_LIB_AB=$GRAFT_REPO_ROOT/multitask_hydranet_amd/libhydranet_hip_W3.so one quads3; HN_FUSE_SUM2X2=1 HN_LIB_AB=$GRAFT_REPO_ROOT/multitask_hydranet_amd/libhydranet_hip_NQ.so one generic; done
