#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command (headline only: no sweep, no nested PMC passes); writes
# profiles/<tag>_kernel_stats.csv (header + this library's kernels) and profiles/<tag>_bench_under_rocprof.json.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-traffic > $OUT/bench.json 2> $OUT/bench.err
STATS=$(find $OUT -name "*kernel_stats.csv" | head -1)
(head -1 "$STATS"; grep "vit::" "$STATS") > $OUT/${TAG}_kernel_stats.csv
tail -1 $OUT/bench.json > $OUT/${TAG}_bench_under_rocprof.json
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/${TAG}_kernel_stats.csv
python3 - "$OUT/${TAG}_bench_under_rocprof.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("bench.py under rocprof:", round(d["value"], 1), "Mframes/s; forward avg (HIP events)", round(d["roofline"]["avg_launch_ms"], 3), "ms; back-trace", round(d["kernels_ms"]["backtrace"], 3), "ms")
PY
