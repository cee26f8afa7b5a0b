#!/bin/bash
# SQ counters (two passes) + FETCH_SIZE / WRITE_SIZE (one pass each) + kernel stats of scripts/prof_target.py "$@".
# usage: pmc_target.sh TAG B algo [steps T opt=val ...]   -> gpurun_out/pmc_TAG/summary.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
run() { d=$1; shift; rocprofv3 "$@" --output-format csv -d $OUT/$d -- python3 scripts/prof_target.py $ARGS > /dev/null 2>> $OUT/err.log; }
ARGS="$*"
run stats --kernel-trace --stats
run sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run grbm --pmc GRBM_GUI_ACTIVE
python3 - "$OUT" "$ARGS" > $OUT/summary.txt <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
print("# scripts/pmc_target.sh", sys.argv[2])
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "vit::" in k:
            rows[k.split("(")[0].split("vit::")[-1][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in rows.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} n={len(v)} mean={sum(v)/len(v):.5g}")
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    print("# kernel stats (vit:: kernels; rocprofv3 --kernel-trace --stats)")
    for k, line in enumerate(open(f)):
        if k == 0 or "vit::" in line:
            print(line.rstrip())
PY
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/summary.txt
