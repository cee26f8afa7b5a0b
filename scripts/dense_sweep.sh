#!/bin/bash
cd "$(dirname "$0")/.."
for b in 128 512 1024; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial --batch $b --algo dense --frames ${FR:-6000} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense S=361 B', $b, 'Mframes/s', round(d['value'],1), 'fwd_ms', round(d['kernels_ms']['forward'],2), 'bt_ms', round(d['kernels_ms']['backtrace'],2))"
done
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial --batch 256 --states 722 --transition dense --emissions dense --f16 --frames ${FR:-6000} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense S=722 f16 B 256', 'Mframes/s', round(d['value'],1), 'fwd_ms', round(d['kernels_ms']['forward'],2), 'bt_ms', round(d['kernels_ms']['backtrace'],2))"
