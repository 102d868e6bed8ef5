#!/bin/bash
# Round-end measurement on the GPU box: default bench line, rocprofv3 kernel stats of the same command, PMC passes on the dominant launch.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python bench.py > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > $O/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-optimizer --no-extras > $O/bench_rocprof.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
t=$(find $O/kt -name "*kernel_trace.csv" | head -1); head -1 "$t" > $O/dominant_dispatches.csv; grep "conv3x3_direct_kernel<128" "$t" | tail -600 >> $O/dominant_dispatches.csv
rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --dominant-only --steps 10 > $O/pmc_$c.log 2>&1
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); head -1 "$f" > $O/pmc_$c.csv; grep "conv3x3_direct" "$f" >> $O/pmc_$c.csv
  rm -rf $O/pmc_$c
done

# derive the per-launch HBM traffic of the dominant launch (FETCH_SIZE is reported in KB at half the bytes on gfx950: x2; WRITE_SIZE in KB)
python3 - <<'PY'
import csv, json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof/"
def mean(col, path):
    rows = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name") == col]
    rows = rows[4:] if len(rows) > 8 else rows              # skip the skip-operand conv and the warm-up launches
    return sum(float(r["Counter_Value"]) for r in rows) / max(len(rows), 1), len(rows)
try:
    f, nf = mean("FETCH_SIZE", O + "pmc_FETCH_SIZE.csv")
    w, nw = mean("WRITE_SIZE", O + "pmc_WRITE_SIZE.csv")
    json.dump({"fetch_size_kb": f, "write_size_kb": w, "launches": [nf, nw], "hbm_bytes_per_launch": (2 * f + w) * 1024,
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); separate --pmc passes"},
              open(O + "dominant_pmc.json", "w"), indent=1)
except Exception as e:
    print("pmc summary failed:", e)
PY
