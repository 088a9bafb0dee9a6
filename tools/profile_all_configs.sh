#!/bin/bash
# GPU box, from the repo root: the committed counter evidence of a build — tools/profile_bench.sh (kernel trace, FETCH / WRITE / TCC, SQ, TA,
# TCP passes, each its own run) for every BASELINE config on one GPU.  Results land in gpurun_out/prof_<tag>/; the summaries to commit:
#   cp gpurun_out/prof_c2/pmc_valu.json profiles/pmc_valu.json; cp gpurun_out/prof_c3/pmc_valu.json profiles/pmc_valu_c3.json; ... (likewise pmc_traffic)
# usage: tools/profile_all_configs.sh [c2 c3 c4 c5]
cd "$(dirname "$0")/.."
CONFIGS=${@:-c2 c3 c4 c5}
for c in $CONFIGS; do
  case $c in
    c2) STEPS=4 tools/profile_bench.sh c2 || exit 1;;
    c3) STEPS=2 PASS_TIMEOUT=300 tools/profile_bench.sh c3 --scene 10 --spp 4096 || exit 1;;
    c4) STEPS=2 PASS_TIMEOUT=400 tools/profile_bench.sh c4 --scene 8 --width 4096 --height 4096 --spp 1024 || exit 1;;
    c5) STEPS=2 PASS_TIMEOUT=300 tools/profile_bench.sh c5 --scene 17 --strategy nee --spp 16384 || exit 1;;
  esac
  echo "== $c done"
done
