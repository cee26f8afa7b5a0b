#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command; summaries are copied to profiles/ by hand.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*stats*.csv" | head
cat $OUT/bench.json
