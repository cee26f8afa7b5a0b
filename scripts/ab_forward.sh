#!/bin/bash
# A/B of forward-kernel variants selected by VIT_DEBUG_FLAGS (timing experiments)
cd "$(dirname "$0")/.."
for f in ${FLAGS:-0 4096}; do
  for b in ${BATCHES:-1 128 512}; do
    VIT_DEBUG_FLAGS=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags', $f, 'B', $b, 'Mframes/s', round(d['value'],1), 'fwd_ms', round(d['kernels_ms']['forward'],2), 'bt_ms', round(d['kernels_ms']['backtrace'],2))"
  done
done
