#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <tag> [perf_probe args...]
# Runs separate rocprofv3 --pmc passes (never combined with tracing) around tools/perf_probe.py and
# prints per-launch averages for the production pt_kernel<false,...> launches.  TA / TCP groups are split so that each fits one pass.
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
MAXL=${PMC_LINES:-99}
while read -r line; do
  [ -z "$line" ] && continue
  [ $i -ge $MAXL ] && break
  # every pass is checked: a counter group that does not fit one pass makes rocprofv3 abort the (GPU-initialised) process
  timeout -k 10 200 rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -- python3 $R/tools/perf_probe.py --reps 2 "$@" > $OUT/p$i.log 2>&1 \
    || { echo "pass $i ($line) failed:"; tail -5 $OUT/p$i.log; exit 1; }
  i=$((i+1))
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM_RD
TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
LIST
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pt_kernel<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
        out["_launches"] = len(v)
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
print(json.dumps(out))
PY
