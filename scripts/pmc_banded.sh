#!/bin/bash
# SQ counters for the forward kernel (separate --pmc passes; no tracing domains combined).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_${1:-r01}
rm -rf $OUT && mkdir -p $OUT
run() { rocprofv3 --pmc "$@" --output-format csv -d $OUT/$1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>> $OUT/err.log; }
run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
PMC_OUT=$OUT python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
import os
for f in glob.glob(os.environ.get("PMC_OUT", "gpurun_out/pmc_r01") + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "vit::" in k:
            rows[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in rows.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
