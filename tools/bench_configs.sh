#!/bin/bash
# GPU box: bench.py on the BASELINE configs (C2 default with CPU baseline, then C3, C4, C5, C1 on one GPU)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err || exit 1
cut -c1-200 gpurun_out/bench_c2.json
: > gpurun_out/bench_other.jsonl
timeout -k 10 300 python3 bench.py --scene 10 --spp 4096 --steps 2 --no-cpu-baseline >> gpurun_out/bench_other.jsonl || exit 1
timeout -k 10 300 python3 bench.py --scene 8 --width 4096 --height 4096 --spp 1024 --steps 2 --no-cpu-baseline >> gpurun_out/bench_other.jsonl || exit 1
timeout -k 10 300 python3 bench.py --scene 17 --strategy nee --spp 16384 --steps 2 --no-cpu-baseline >> gpurun_out/bench_other.jsonl || exit 1
timeout -k 10 300 python3 bench.py --scene 3 --width 256 --height 256 --spp 16 --strategy pt --sampler random --steps 16 --no-cpu-baseline >> gpurun_out/bench_other.jsonl || exit 1
cut -c1-200 gpurun_out/bench_other.jsonl
