cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/tr_probe
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_probe -- python3 tools/graph_share_probe.py plain > $R/gpurun_out/tr_probe.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(R+"/gpurun_out/tr_probe/**/*kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
# last replay = after the last pack_w_batched_kernel
idx=[i for i,n in enumerate(names) if "pack_w_batched" in n]
seq=names[idx[-1]:]
out=[f"launches in last replay: {len(seq)}"]
cnt=collections.Counter(n.split("(")[0][:60] for n in seq if "rocclr" in n or "at::native" in n)
for k,v in cnt.most_common(): out.append(f"{v:4d} {k}")
# context of copies
for i,n in enumerate(seq):
    if "copyBuffer" in n or "fillBuffer" in n:
        out.append("  " + seq[i-1].split("(")[0][:50] + "  ->  " + n[:30] + "  ->  " + (seq[i+1].split("(")[0][:50] if i+1 < len(seq) else "END"))
open(R+"/gpurun_out/probe_copies.txt","w").write("\n".join(out)+"\n")
PY
rm -rf gpurun_out/tr_probe
