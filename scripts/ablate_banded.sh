# timing ablations of the scan-form kernel (results WRONG): needs the hooks build, `make -C viterbi_spl_amd/csrc TIMING=1`
for f in 0 1 2 4 8 3 7 15; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --serial --option timing=$f --option forward_form=3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags', $f, 'fwd_ms', round(d['kernels_ms']['forward'],2))"
done
