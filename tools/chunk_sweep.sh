#!/bin/bash
# GPU box: per-rank throughput of one shard of N for different sample-chunk counts (MI355PT_CHUNKS tuning override)
cd "$(dirname "$0")/.."
for shards in 8 4 2; do
  for c in auto 1 2 4 8 16; do
    if [ $c = auto ]; then unset MI355PT_CHUNKS; else export MI355PT_CHUNKS=$c; fi
    echo -n "shards=$shards chunks=$c "
    timeout -k 10 120 python3 tools/perf_probe.py --shards $shards --slice 64 --reps 4 --no-stats | grep -o '"Msamples_s": [0-9.]*'
  done
done
