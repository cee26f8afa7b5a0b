#!/bin/bash
# emission prefetch depth A/B for banded_floor_forward_kernel: VIT_DEBUG_FLAGS 16384 -> 2 frames, 0 -> 4, 32768 -> 8
cd "$(dirname "$0")/.."
for f in 16384 0 32768; do
  for b in 1 128 512; do
    VIT_DEBUG_FLAGS=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags', $f, 'B', $b, 'Mframes/s', round(d['value'],1), 'fwd_ms', round(d['kernels_ms']['forward'],2), 'bt_ms', round(d['kernels_ms']['backtrace'],2))"
  done
done
