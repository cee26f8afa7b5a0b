#!/bin/bash
# throughput vs songs per GPU (forward + back-trace), banded and dense kernels
cd "$(dirname "$0")/.."
for b in 1 32 128 256 512 704 1024 2048; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --serial --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('banded B', $b, 'Mframes/s', round(d['value'],1), 'fwd_ms', round(d['kernels_ms']['forward'],2), 'bt_ms', round(d['kernels_ms']['backtrace'],2))"
done
for b in 128 1024; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial --batch $b --algo dense 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense B', $b, 'Mframes/s', round(d['value'],1), 'fwd_ms', round(d['kernels_ms']['forward'],2), 'bt_ms', round(d['kernels_ms']['backtrace'],2))"
done
