#!/bin/bash
# GPU box: SAH constants of the BVH builders against render throughput.  The overrides exist only in a -DMI355PT_TUNING build:
#   tools/build_variant.sh tuning -DMI355PT_TUNING   (here), then   MI355PT_LIB=build_variants/libmi355pt_tuning.so tools/sah_sweep.sh
cd "$(dirname "$0")/.."
for scene in ${1:-3}; do
for leaf in ${LEAFS:-2 3 4}; do
  for ct in ${CTS:-1.0 1.5 2.0 3.0}; do
    echo -n "scene=$scene leaf=$leaf cost_tri=$ct "
    MI355PT_BVH_LEAF=$leaf MI355PT_BVH_COST_TRI=$ct timeout -k 10 200 python3 tools/perf_probe.py --scene $scene --slice 1024 --reps 1 --no-stats | grep -o '"Msamples_s": [0-9.]*'
  done
done
done
