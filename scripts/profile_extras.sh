#!/bin/bash
# rocprofv3 kernel-trace + stats of scripts/prof_extras.py; writes profiles/<tag>_kernel_stats_extras.csv (header + this library's kernels)
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r02}
OUT=gpurun_out/prof_extras_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/prof_extras.py > $OUT/run.log 2> $OUT/run.err
STATS=$(find $OUT -name "*kernel_stats.csv" | head -1)
(head -1 "$STATS"; grep "vit::" "$STATS") > $OUT/${TAG}_kernel_stats_extras.csv
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/${TAG}_kernel_stats_extras.csv
