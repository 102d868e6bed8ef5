#!/bin/bash
export HN_BENCH_LIVE_TRAFFIC=0
export HN_TUNING=${HN_TUNING:-ab}    # product library; the package reads HN_LIB_AB / policy switches only under HN_TUNING=1|ab (_lib.policy)
# Same-box A/B of two builds: `tools/ab_build.sh save` copies the current in-tree library to libhydranet_hip_B.so (variant B); change /
# revert the sources, rebuild (variant A), then on the GPU box run bench.py with and without HN_LIB_AB=<repo>/multitask_hydranet_amd/libhydranet_hip_B.so.
# (touch the .so after editing sources back, or bench.py's build() recompiles on the box.)
R=$(cd "$(dirname "$0")/.." && pwd)
case "$1" in
  save) cp $R/multitask_hydranet_amd/libhydranet_hip.so $R/multitask_hydranet_amd/libhydranet_hip_B.so ;;
  run) for i in 1 2; do python $R/bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c80-175; HN_LIB_AB=$R/multitask_hydranet_amd/libhydranet_hip_B.so python $R/bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c80-175; done ;;
  *) echo "usage: ab_build.sh save|run" ;;
esac
