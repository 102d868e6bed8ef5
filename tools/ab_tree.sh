#!/bin/bash
# Build the WORKING TREE's kernels into multitask_hydranet_amd/libhydranet_hip_$1.so (a named variant for tools/ab_run.sh); extra hipcc
# flags (e.g. -DSOME_EXPERIMENT=2) may follow the name.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
T=$(mktemp -d)
objs=""
for f in hn_gemm hn_norm hn_fused hn_stencil hn_loss hn_post hn_xstage; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c $R/multitask_hydranet_amd/csrc/$f.hip -o $T/$f.o &
  objs="$objs $T/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/multitask_hydranet_amd/libhydranet_hip_$NAME.so $objs
rm -rf $T
ls -la $R/multitask_hydranet_amd/libhydranet_hip_$NAME.so
