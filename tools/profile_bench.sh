#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [bench.py args...]
# 1. rocprofv3 --kernel-trace --stats around bench.py (per-kernel time)
# 2. separate --pmc passes (never combined with tracing): FETCH_SIZE; WRITE_SIZE + L2 hit/miss; SQ VALU / wait counters
# Writes gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_traffic.json,pmc_valu.json}; copy what should be judged into profiles/.
# Every pass is checked: a failing pass stops the script with the tail of its log.
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${STEPS:-4} --warmup 1 --no-cpu-baseline $@"
pass() {   # pass <dir> <rocprofv3 options...>
  local d=$1; shift
  timeout -k 10 ${PASS_TIMEOUT:-400} rocprofv3 "$@" --output-format csv -d $OUT/$d -- python3 $R/bench.py $ARGS > $OUT/$d.log 2>&1 || { echo "pass $d failed"; tail -5 $OUT/$d.log; exit 1; }
}
pass trace --kernel-trace --stats
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass pmc_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
pass pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
pass pmc_grbm --pmc GRBM_GUI_ACTIVE
# the vector-memory path (TA = address unit, TCP = L1): one pass per group, each fits one pass on gfx950
pass pmc_ta --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
pass pmc_tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
python3 - "$OUT" "$ARGS" "$R" <<'PY'
import csv, glob, sys, collections, json, shutil, re
out_dir, args, root = sys.argv[1], sys.argv[2], sys.argv[3]
IS_PT = lambda name: "pt_kernel<false" in name          # the production variants (the instrumented one is pt_kernel<true, ...>)
line = None
for l in open(out_dir + "/trace.log"):
    if l.startswith("{") and '"metric"' in l:
        line = json.loads(l)
res = {"command": "python3 bench.py " + args, "kernel": "pt_kernel<false,FEAT,MODE> (the timed launches)"}
if line:
    m = re.match(r"scene(\d+) (\d+)x(\d+) (\w+)\+(\w+), (\d+)-spp job, (\d+) sample", line["config"]["workload"])
    res["workload"] = {"scene": int(m.group(1)), "width": int(m.group(2)), "height": int(m.group(3)), "spp": int(m.group(6)),
                       "spp_per_step": int(m.group(7)), "strategy": m.group(4), "sampler": m.group(5), "n_gpus": line["n_gpus"]}
    res["samples_per_launch"] = line["config"]["samples_per_step"] // line["n_gpus"]
    res["bench_line"] = line
sys.path.insert(0, root)
import importlib
res["library"] = importlib.import_module("toy-cpu-pathtracing_amd").Product().version()
for f in glob.glob(out_dir + "/trace/*/*_kernel_stats.csv"):
    shutil.copy(f, out_dir + "/kernel_stats.csv")
    for r in csv.DictReader(open(f)):
        if IS_PT(r["Name"]):
            res.setdefault("kernel_stats", []).append({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")})
for f in glob.glob(out_dir + "/trace/*/*_kernel_trace.csv"):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if IS_PT(r["Kernel_Name"])]
    if d:
        res["launches_ns"] = d
        res["avg_launch_ms_excluding_warmup"] = sum(d[1:]) / max(len(d) - 1, 1) / 1e6
def counters(pattern):
    cnt = collections.defaultdict(list)
    for f in sorted(glob.glob(out_dir + "/" + pattern + "/*/*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if IS_PT(r["Kernel_Name"]):
                cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k + "_per_launch": sum(v[1:]) / max(len(v) - 1, 1) for k, v in cnt.items()}   # skip the warm-up launch
traffic = dict(res); traffic.pop("bench_line", None)
traffic.update(counters("pmc_fetch")); traffic.update(counters("pmc_write"))
if "FETCH_SIZE_per_launch" in traffic and "WRITE_SIZE_per_launch" in traffic:
    # MI355X_MICROARCH.md (HBM): counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> double it
    traffic["hbm_bytes_per_launch"] = (2.0 * traffic["FETCH_SIZE_per_launch"] + traffic["WRITE_SIZE_per_launch"]) * 1024.0
    traffic["correction"] = "hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024  [gfx950: FETCH_SIZE reports half of the fetched bytes]"
if "TCC_HIT_sum_per_launch" in traffic:
    h, m = traffic["TCC_HIT_sum_per_launch"], traffic["TCC_MISS_sum_per_launch"]
    traffic["l2_hit_rate"] = h / max(h + m, 1.0)
json.dump(traffic, open(out_dir + "/pmc_traffic.json", "w"), indent=1)
valu = {k: res[k] for k in ("command", "kernel", "workload", "samples_per_launch", "library") if k in res}
valu.update(counters("pmc_sq1")); valu.update(counters("pmc_sq2")); valu.update(counters("pmc_grbm")); valu.update(counters("pmc_ta")); valu.update(counters("pmc_tcp"))
if "SQ_INSTS_VALU_per_launch" in valu and "samples_per_launch" in valu:
    ms = res.get("avg_launch_ms_excluding_warmup", 0.0)
    iv = valu["SQ_INSTS_VALU_per_launch"]
    valu["wave_instr_per_sample"] = iv / valu["samples_per_launch"]
    valu["lane_use"] = valu["SQ_THREAD_CYCLES_VALU_per_launch"] / (64.0 * iv)
    valu["unprofiled_launch_ms"] = ms
    if ms:
        valu["valu_Ginstr_per_s"] = iv / (ms * 1e-3) / 1e9
        valu["issue_frac_of_1228.8G"] = valu["valu_Ginstr_per_s"] / 1228.8
    if ms and "GRBM_GUI_ACTIVE_per_launch" in valu:     # summed over the 8 XCDs: the clock the GPU held during the launch
        valu["gpu_clock_ghz"] = valu["GRBM_GUI_ACTIVE_per_launch"] / 8.0 / (ms * 1e-3) / 1e9
    if ms and "TCP_TOTAL_CACHE_ACCESSES_sum_per_launch" in valu:
        cyc = ms * 1e-3 * valu.get("gpu_clock_ghz", 2.4) * 1e9
        valu["l1_tag_lookups_per_clk_per_cu"] = valu["TCP_TOTAL_CACHE_ACCESSES_sum_per_launch"] / 256.0 / cyc
        if "TA_TA_BUSY_sum_per_launch" in valu: valu["ta_busy"] = valu["TA_TA_BUSY_sum_per_launch"] / 256.0 / cyc
        if "SQ_INSTS_VMEM_RD_per_launch" in valu: valu["l1_lookups_per_wave_load"] = valu["TCP_TOTAL_CACHE_ACCESSES_sum_per_launch"] / valu["SQ_INSTS_VMEM_RD_per_launch"]
    wc = valu.get("SQ_WAVE_CYCLES_per_launch")
    if wc:   # SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md, cycle constants)
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if k + "_per_launch" in valu:
                valu["share_of_wave_cycles_" + k] = valu[k + "_per_launch"] / wc
json.dump(valu, open(out_dir + "/pmc_valu.json", "w"), indent=1)
print(json.dumps({k: v for k, v in valu.items() if not k.endswith("_per_launch")}))
print(json.dumps({k: v for k, v in traffic.items() if k in ("hbm_bytes_per_launch", "l2_hit_rate", "avg_launch_ms_excluding_warmup", "kernel_stats")}))
PY
tail -1 $OUT/trace.log
