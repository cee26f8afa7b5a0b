for f in 0 1 2 4 8 3 7 15; do
  VIT_DEBUG_FLAGS=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags', $f, 'fwd_ms', round(d['kernels_ms']['forward'],2))"
done
