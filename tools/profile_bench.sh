#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [bench.py args...]
# 1. rocprofv3 --kernel-trace --stats around bench.py (per-kernel time)
# 2. separate --pmc passes (never combined with tracing): FETCH_SIZE, then WRITE_SIZE + L2 hit/miss
# Writes gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_traffic.json}; copy what should be judged into profiles/.
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 1 --no-cpu-baseline $@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
python3 - "$OUT" "$ARGS" <<'PY'
import csv, glob, sys, collections, json, shutil
out_dir, args = sys.argv[1], sys.argv[2]
res = {"command": "python3 bench.py " + args, "kernel": "pt_kernel<false,false,FEAT> (the timed launches: grid >= 1024 workgroups)"}
# ---- kernel stats
for f in glob.glob(out_dir + "/trace/*/*_kernel_stats.csv"):
    shutil.copy(f, out_dir + "/kernel_stats.csv")
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("void pt::pt_kernel<false, false") or "pt_kernel<false, false" in r["Name"]:
            res.setdefault("kernel_stats", []).append({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")})
# per-launch durations of the timed launches (skip the warmup launch and the 4-sample STATS launch, which is another kernel)
for f in glob.glob(out_dir + "/trace/*/*_kernel_trace.csv"):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "pt_kernel<false, false" in r["Kernel_Name"]]
    if d:
        res["launches_ns"] = d
        res["avg_launch_ms_excluding_warmup"] = sum(d[1:]) / max(len(d) - 1, 1) / 1e6
# ---- counters: per launch averages over the timed launches
cnt = collections.defaultdict(list)
for f in sorted(glob.glob(out_dir + "/pmc_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "pt_kernel<false, false" in r["Kernel_Name"]:
            cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in cnt.items():
    res[k + "_per_launch"] = sum(v[1:]) / max(len(v) - 1, 1)
if "FETCH_SIZE_per_launch" in res and "WRITE_SIZE_per_launch" in res:
    # MI355X_MICROARCH.md (HBM): counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> double it
    res["hbm_bytes_per_launch"] = (2.0 * res["FETCH_SIZE_per_launch"] + res["WRITE_SIZE_per_launch"]) * 1024.0
    res["correction"] = "hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024  [gfx950: FETCH_SIZE reports half of the fetched bytes]"
if "TCC_HIT_sum_per_launch" in res:
    h, m = res["TCC_HIT_sum_per_launch"], res["TCC_MISS_sum_per_launch"]
    res["l2_hit_rate"] = h / max(h + m, 1.0)
json.dump(res, open(out_dir + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "launches_ns"}))
PY
tail -1 $OUT/trace.log
