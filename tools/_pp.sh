set -e
: > gpurun_out/r2_pp4.jsonl
for s in "3 mis" "0 mis" "8 mis" "17 nee"; do set -- $s
  timeout -k 10 150 python tools/perf_probe.py --scene $1 --strategy $2 --slice 1024 --reps 1 >> gpurun_out/r2_pp4.jsonl
done
