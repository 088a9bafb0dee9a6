#!/bin/bash
# usage (GPU box): tools/icache_pmc.sh [scene] [strategy]   -- instruction-cache counters of the production kernel
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH --output-format csv -d $R/gpurun_out/icache -- python3 $R/tools/perf_probe.py --reps 1 --scene ${1:-3} --strategy ${2:-mis} --slice 1024 --no-stats > $R/gpurun_out/icache.log 2>&1 || { tail -5 $R/gpurun_out/icache.log; exit 1; }
python3 - $R/gpurun_out/icache <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pt_kernel<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: sum(v) / len(v) for k, v in agg.items()})
PY
