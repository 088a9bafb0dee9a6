#!/bin/bash
# GPU box: per-rank throughput of one shard of N (emulated on one GPU) at the bench's launch shape (whole 1024-spp frame per launch)
cd "$(dirname "$0")/.."
for shards in 1 2 4 8; do
  echo -n "shards=$shards "
  timeout -k 10 200 python3 tools/perf_probe.py --shards $shards --slice ${1:-1024} --reps 2 --no-stats | grep -o '"Msamples_s": [0-9.]*'
done
