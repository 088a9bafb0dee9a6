#!/bin/bash
# usage (GPU box): tools/ab.sh "<scene list>" lib1.so lib2.so ...   — perf_probe each variant in build_variants/
R=${GRAFT_REPO_ROOT:-$PWD}
SCENES=$1; shift
for s in $SCENES; do for v in "$@"; do
  MI355PT_LIB=$R/build_variants/$v timeout -k 10 120 python3 $R/tools/perf_probe.py --scene $s --tag $v --reps 2 --no-stats ${AB_ARGS:-} | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['tag'], 'scene', d['scene'], d['Msamples_s'])" || exit 1
done; done
