#!/bin/bash
cd "$(dirname "$0")/.."
for b in 128 1024; do
 for cw in "0 0" "14 256" "28 256" "28 128" "32 64" "16 128"; do
  set -- $cw
  if [ "$1" = "0" ]; then OPT=""; else OPT="--option bt_chunks=$1 --option bt_warm=$2"; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --serial $OPT --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B', $b, 'chunks/warm', '$cw', 'bt_ms', round(d['kernels_ms']['backtrace'],3), 'fwd_ms', round(d['kernels_ms']['forward'],2))"
 done
done
