#!/bin/bash
# usage (GPU box): tools/ray_bench.sh — kernel times of probe_intersect on coherent / incoherent rays, lock-step (mode 1) vs
# dynamic fetch (mode 2); MI355PT_LIB selects a variant library
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/rb
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rb -- python3 $R/tools/ray_bench.py > $R/gpurun_out/rb.log 2>&1 || { tail -5 $R/gpurun_out/rb.log; exit 1; }
grep -E "mode|identical" $R/gpurun_out/rb.log
python3 - $R/gpurun_out/rb <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "probe_intersect" in r["Kernel_Name"]]
    for r in rows[1:]:
        ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        n = int(r.get("Grid_Size", 0))
        print(r["Kernel_Name"][4:34], "ms", round(ms, 3))
PY
