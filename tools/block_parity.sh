#!/bin/bash
cd "$(dirname "$0")/.."
for b in 0 1; do
  echo "== parity with MI355PT_BLOCK=$b"
  MI355PT_BLOCK=$b timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "image_parity or config1 or edge_case or full_size or shards_tile or consistency or other_scenes" 2>&1 | grep -v "^$" | tail -25
done
