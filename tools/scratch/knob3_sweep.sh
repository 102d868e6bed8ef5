cd $GRAFT_REPO_ROOT
for k in "" "3=128" "3=256" "3=1024" "3=128,2=1536" ; do
  for r in 1 2; do
  HN_TUNING=1 HN_KNOBS="$k" python3 bench.py --no-cpu-baseline --no-extras --no-roofline --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('knobs [$k]', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3))"
  done
done
