#!/bin/bash
# dense forward kernel: songs per workgroup (option dense_songs) vs batch, T shortened
cd "$(dirname "$0")/.."
for b in 128 1024; do
 for ns in 1 2 4 8; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial --option dense_songs=$ns --batch $b --algo dense --frames ${FR:-3000} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense S=361 B', $b, 'NS', $ns, 'Mframes/s', round(d['value'],1), 'fwd_ms', round(d['kernels_ms']['forward'],2))"
 done
done
