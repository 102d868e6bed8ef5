cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for s in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_$s -- python3 tools/trace_section.py $s > $R/gpurun_out/tr_$s.log 2>&1
  python tools/trace_section.py --parse gpurun_out/tr_$s 6 > gpurun_out/tr_$s.txt 2>&1
  rm -rf gpurun_out/tr_$s
done
