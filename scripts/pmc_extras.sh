#!/bin/bash
# SQ counters (two passes) + FETCH_SIZE / WRITE_SIZE (one pass each) of scripts/prof_extras.py (the kernels outside the headline).
# usage: pmc_extras.sh TAG   -> gpurun_out/pmc_extras_TAG/summary.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r02}
OUT=gpurun_out/pmc_extras_$TAG
rm -rf $OUT && mkdir -p $OUT
run() { d=$1; shift; rocprofv3 "$@" --output-format csv -d $OUT/$d -- python3 scripts/prof_extras.py > /dev/null 2>> $OUT/err.log; }
run sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
python3 - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
print("# scripts/pmc_extras.sh: rocprofv3 --pmc passes (no tracing domains) over scripts/prof_extras.py; per-launch means")
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "vit::" in k:
            rows[k.split("(")[0].split("vit::")[-1][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(rows.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} n={len(v)} mean={sum(v)/len(v):.5g}")
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fs, ws = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        print(f"   HBM bytes per launch (2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024): {2 * fs * 1024 + ws * 1024:.4g}")
    if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d:
        bc, ia = sum(d["SQ_LDS_BANK_CONFLICT"]) / len(d["SQ_LDS_BANK_CONFLICT"]), sum(d["SQ_LDS_IDX_ACTIVE"]) / len(d["SQ_LDS_IDX_ACTIVE"])
        print(f"   LDS bank-conflict cycles / LDS active cycles: {bc / ia if ia else 0:.4f}")
PY
find $OUT -name "*counter_collection.csv" -delete
cat $OUT/summary.txt
