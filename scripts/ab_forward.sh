#!/bin/bash
# interleaved A/B of forward-kernel variants (VIT_DEBUG_FLAGS = 4096 * OPT) on the same box
cd "$(dirname "$0")/.."
for r in 1 2; do
 for o in 0 1 2 4; do
  VIT_DEBUG_FLAGS=$((4096*o)) python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batch ${PB:-128} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('round', $r, 'OPT', $o, 'fwd_ms', round(d['kernels_ms']['forward'],3))"
 done
done
