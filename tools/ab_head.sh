#!/bin/bash
# Build variant B = the library as of git revision $1 (default HEAD) into multitask_hydranet_amd/libhydranet_hip_B.so (same ABI assumed),
# so that one gpurun call can time the working tree (A) against it on the SAME box:  tools/ab_build.sh run
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-HEAD}
T=$(mktemp -d)
for f in hn_common.h hn_gemm.hip hn_norm.hip hn_fused.hip hn_stencil.hip hn_loss.hip hn_post.hip hn_xstage.hip; do git -C $R show $REV:multitask_hydranet_amd/csrc/$f > $T/$f; done
objs=""
for f in hn_gemm hn_norm hn_fused hn_stencil hn_loss hn_post hn_xstage; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c $T/$f.hip -o $T/$f.o &
  objs="$objs $T/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/multitask_hydranet_amd/libhydranet_hip_B.so $objs
rm -rf $T
ls -la $R/multitask_hydranet_amd/libhydranet_hip_B.so
