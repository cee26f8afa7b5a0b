#!/bin/bash
# LDS window-shift A/B (option win_shift overrides the plan's choice; every value is functionally correct)
cd "$(dirname "$0")/.."
for sh in 0 1 2 3 default; do
  for b in 1 128; do
    if [ $sh = default ]; then OPT=""; else OPT="--option win_shift=$sh"; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --serial --algo group $OPT --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('shift', '$sh', 'B', $b, 'Mframes/s', round(d['value'],1), 'fwd_ms', round(d['kernels_ms']['forward'],2))"
  done
done
