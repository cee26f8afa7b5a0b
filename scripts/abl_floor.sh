#!/bin/bash
# timing-only ablations of banded_floor_forward_kernel (results are WRONG under these flags)
cd "$(dirname "$0")/.."
run() { VIT_DEBUG_FLAGS=$1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $2 --states $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$4', 'B', $2, 'S', $3, 'fwd_ms', round(d['kernels_ms']['forward'],2))"; }
for b in 1 128; do
  run 0 $b 361 full
  run 65536 $b 361 no_dpp_reduce
  run 131072 $b 361 window_8_of_32
  run 262144 $b 361 no_global
  run 524288 $b 361 no_barrier
  run 0 $b 250 full_4waves
  run 0 $b 190 full_3waves
done
