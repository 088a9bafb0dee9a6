#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the GPU BVH build (tools/build_time.py under MI355PT_BVH_BUILDER=gpu)
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out/bvh_prof
export MI355PT_BVH_BUILDER=gpu
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/bvh_prof -o bvh -- python3 $ROOT/tools/build_time.py
cd $ROOT
f=$(find gpurun_out/bvh_prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/bvh_prof/kernel_stats.csv
head -20 gpurun_out/bvh_prof/kernel_stats.csv
