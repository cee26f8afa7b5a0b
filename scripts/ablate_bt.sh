for f in 0 64 128 192; do
  VIT_DEBUG_FLAGS=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags', $f, 'bt_ms', round(d['kernels_ms']['backtrace'],2), 'fwd_ms', round(d['kernels_ms']['forward'],2))"
done
