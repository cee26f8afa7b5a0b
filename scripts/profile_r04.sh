#!/bin/bash
# Round-4 profiles (run on the GPU box from the repo root): kernel stats of every kernel family (>= 10 launches each) and PMC passes of
# the emission builder, the lane back-trace and the B = 1024 full-history decode.  Outputs under gpurun_out/prof_r04/; the summaries
# (r04_kernel_stats.csv, r04_pmc_*.txt) are what gets copied to profiles/.
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r04
[ -z "$ONLY_PMC" ] && rm -rf $OUT; mkdir -p $OUT
cd /tmp
if [ -z "$ONLY_PMC" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/scripts/prof_r04.py > $OUT/stats.log 2> $OUT/stats.err
# per (kernel, grid) rows from the trace: the same kernel at 1024 and 2048 songs must not share a line
python3 - "$OUT/stats" > $OUT/r04_kernel_stats.csv <<'PY'
import csv, glob, collections, sys
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vit::" in r["Kernel_Name"]:
            rows[(r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print('"Name","GridSize","WorkgroupSize","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs"')
for (k, g, w), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f'"{k}",{g},{w},{len(v)},{sum(v)},{sum(v) / len(v):.1f},{min(v)},{max(v)}')
PY
find $OUT -name "*kernel_trace.csv" -delete
echo "kernel stats done"; cat $OUT/r04_kernel_stats.csv | cut -c1-200
fi
pmc() {   # pmc NAME PART...: SQ counters (two passes) + FETCH_SIZE / WRITE_SIZE (one pass each), no tracing domain
    name=$1; shift
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/$name/sq1 -- python3 $ROOT/scripts/prof_r04.py "$@" > /dev/null 2>> $OUT/err.log
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$name/sq2 -- python3 $ROOT/scripts/prof_r04.py "$@" > /dev/null 2>> $OUT/err.log
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$name/fetch -- python3 $ROOT/scripts/prof_r04.py "$@" > /dev/null 2>> $OUT/err.log
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$name/write -- python3 $ROOT/scripts/prof_r04.py "$@" > /dev/null 2>> $OUT/err.log
    python3 - "$OUT/$name" "$*" > $OUT/r04_pmc_$name.txt <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
print("# scripts/profile_r04.sh: rocprofv3 --pmc passes (SQ counters in two passes, FETCH_SIZE, WRITE_SIZE; no tracing domains) over scripts/prof_r04.py", sys.argv[2], "; per-launch means")
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "vit::" in k:
            rows[k.split("(")[0].split("vit::")[-1][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(rows.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} n={len(v)} mean={sum(v)/len(v):.6g}")
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fs, ws = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        print(f"   HBM bytes per launch (2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024): {2 * fs * 1024 + ws * 1024:.5g}   (read {2 * fs * 1024:.5g}, written {ws * 1024:.5g})")
PY
    find $OUT/$name -name "*counter_collection.csv" -delete
    echo "pmc $name done"
}
if [ -n "$ONLY_PMC" ]; then
    for part in $ONLY_PMC; do pmc $part $part; done     # e.g. ONLY_PMC="configs4" scripts/profile_r04.sh
else
    for part in ${PMC_PARTS-obs lane b1024 configs4}; do pmc $part $part; done      # PMC_PARTS="lane b1024": only those after the kernel stats
fi
ls -la $OUT
