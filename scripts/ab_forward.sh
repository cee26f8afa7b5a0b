#!/bin/bash
# A/B of banded forward forms (bench.py --option forward_form=N: 1 one target per lane, 2 two targets per lane, 3 scan,
# 4 one song per wavefront, 5 never the wave form; all decode the same bits)
cd "$(dirname "$0")/.."
for f in ${FORMS:-0 1 2 3 4}; do
  for b in ${BATCHES:-1 128 512}; do
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --serial --option forward_form=$f --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forward_form', $f, 'B', $b, 'Mframes/s', round(d['value'],1), 'fwd_ms', round(d['kernels_ms']['forward'],2), 'bt_ms', round(d['kernels_ms']['backtrace'],2))"
  done
done
