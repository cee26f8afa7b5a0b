#!/bin/bash
# HBM traffic counters for the bench kernels: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (TCC slots: FETCH_SIZE costs 3, WRITE_SIZE 2), no tracing domains combined.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r01}
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>> $OUT/err.log
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "vit::" in k:
            rows[k.split("(")[0].split("vit::")[-1][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in rows.items():
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, {c: f"{x:.4g} (n={len(d[c])})" for c, x in res[k].items()})
json.dump(res, open(out + "/traffic_raw.json", "w"), indent=1)
PY
find $OUT -name "*counter_collection.csv" -delete
