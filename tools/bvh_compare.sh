#!/bin/bash
# GPU box: host-built vs GPU-built BVH -- build time, tree size and render throughput on the headline scene and a 1 M-triangle mesh
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for b in host gpu; do
  echo "== builder=$b scene 3 1080p" 
  MI355PT_BVH_BUILDER=$b timeout -k 10 300 python3 tools/perf_probe.py --scene 3 --reps 3 --tag bvh_$b
  echo "== builder=$b scene 17 1080p"
  MI355PT_BVH_BUILDER=$b timeout -k 10 300 python3 tools/perf_probe.py --scene 17 --strategy nee --reps 2 --tag bvh_$b
  echo "== builder=$b big meshes"
  MI355PT_BVH_BUILDER=$b timeout -k 10 300 python3 tools/build_time.py
done
