#!/bin/bash
# Round-end measurement on the GPU box: default bench line, rocprofv3 kernel stats of the captured step, the priced dominant launch's
# dispatch rows, PMC passes (HBM traffic + MFMA counters) on the dominant launch, the counter-backed roofline table of the whole step.
# -> gpurun_out/prof/*; tools/copy_profiles.sh copies the summaries to profiles/${ROUND}_*
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r05}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-optimizer --no-extras --no-roofline > $O/bench_rocprof.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
t=$(find $O/kt -name "*kernel_trace.csv" | head -1)
# the PRICED launch only: decoder.3's phase-form conv = the 6th conv3x3_direct_kernel<128,...> dispatch of every step (steps are delimited by
# the batched weight pack that opens each forward); the eager warm-up steps and the capture run are dropped (their launches are not replays)
python3 - "$t" > $O/dominant_dispatches.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
packs = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("pack_w_batched_kernel")] + [len(rows)]
w = csv.writer(sys.stdout)
w.writerow(["step", "Kernel_Name", "Grid_Size_X", "Workgroup_Size_X", "VGPR_Count", "Start_Timestamp", "End_Timestamp", "duration_us"])
for s in range(3, len(packs) - 1):
    d = [r for r in rows[packs[s]:packs[s + 1]] if "conv3x3_direct_kernel<128" in r["Kernel_Name"]]
    if len(d) >= 6:
        r = d[5]
        w.writerow([s, r["Kernel_Name"], r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["Start_Timestamp"], r["End_Timestamp"],
                    "%.2f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)])
PY
rm -rf $O/kt
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$i -- python3 bench.py --dominant-only --steps 10 > $O/pmc_$i.log 2>&1
  f=$(find $O/pmc_$i -name "*counter_collection.csv" | head -1); head -1 "$f" > $O/pmc_$i.csv; grep "conv3x3_direct" "$f" >> $O/pmc_$i.csv
  rm -rf $O/pmc_$i
done
mv $O/pmc_1.csv $O/pmc_FETCH_SIZE.csv; mv $O/pmc_2.csv $O/pmc_WRITE_SIZE.csv; mv $O/pmc_3.csv $O/pmc_MFMA.csv

# per-launch HBM traffic and MFMA counters of the dominant launch (FETCH_SIZE is reported in KB at half the bytes on gfx950: x2; WRITE_SIZE in KB)
python3 - <<'PY'
import csv, json, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof/"
def vals(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: v[4:] if len(v) > 8 else v for k, v in d.items()}      # skip the skip-operand conv and the warm-up launches
mean = lambda v: sum(v) / max(len(v), 1)
try:
    f, w, m = vals(O + "pmc_FETCH_SIZE.csv")["FETCH_SIZE"], vals(O + "pmc_WRITE_SIZE.csv")["WRITE_SIZE"], vals(O + "pmc_MFMA.csv")
    gui, busy, mops = mean(m["GRBM_GUI_ACTIVE"]), mean(m["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(m["SQ_INSTS_VALU_MFMA_MOPS_BF16"])
    json.dump({"workload": {"batch": 16, "res": "512x1024", "form": "phase"},
               "fetch_size_kb": mean(f), "write_size_kb": mean(w), "launches": [len(f), len(w)], "hbm_bytes_per_launch": (2 * mean(f) + mean(w)) * 1024,
               "mfma": {"SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui, "SQ_BUSY_CYCLES": mean(m["SQ_BUSY_CYCLES"]),
                        "SQ_INSTS_VALU_MFMA_MOPS_BF16": mops, "executed_flop_from_mops": mops * 512,
                        "mfma_busy_frac": busy / (gui / 8 * 256 * 4)},
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); separate --pmc passes; MFMA busy = "
                       "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs); bench.py --dominant-only --steps 10"},
              open(O + "dominant_pmc.json", "w"), indent=1)
except Exception as e:
    print("pmc summary failed:", repr(e))
PY
# the default bench line LAST, so that its roofline.traffic is this round's counter result (bench.py reads the newest profiles/rNN_dominant_pmc.json)
[ -s $O/dominant_pmc.json ] && cp $O/dominant_pmc.json $R/profiles/${ROUND}_dominant_pmc.json
python bench.py > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > $O/bench_n1.json
bash tools/roofline_table.sh > $O/roofline_table.log 2>&1
cp $R/gpurun_out/roofline/roofline_table.md $O/roofline_table.md
cat $O/dominant_pmc.json; head -3 $O/dominant_dispatches.csv; wc -l $O/dominant_dispatches.csv
