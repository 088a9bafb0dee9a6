set -e
mkdir -p gpurun_out
: > gpurun_out/r2_pp_div3.jsonl
for s in "3 mis" "10 mis" "8 mis" "17 nee" "19 mis" "12 mis" "14 mis" "15 mis"; do
  set -- $s
  timeout -k 10 150 python tools/perf_probe.py --scene $1 --strategy $2 --slice 64 --reps 2 >> gpurun_out/r2_pp_div3.jsonl
done
