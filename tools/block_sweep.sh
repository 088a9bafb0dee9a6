#!/bin/bash
# GPU box: work-item block size (MI355PT_BLOCK = log2 of the block side) against samples per launch
cd "$(dirname "$0")/.."
for b in 0 1 2; do
  echo "== parity with MI355PT_BLOCK=$b"
  MI355PT_BLOCK=$b timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "image_parity or config1 or edge_case or full_size or shards_tile or consistency or other_scenes" 2>&1 | tail -2 || exit 1
done
for slice in 64 256 1024; do
  for b in auto 3 2 1 0; do
    if [ $b = auto ]; then unset MI355PT_BLOCK; else export MI355PT_BLOCK=$b; fi
    echo -n "slice=$slice block=$b "
    timeout -k 10 200 python3 tools/perf_probe.py --slice $slice --reps 3 --no-stats ${1:-} | grep -o '"Msamples_s": [0-9.]*'
  done
done
