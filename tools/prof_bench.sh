#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command (copied into profiles/ by hand afterwards)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_bench
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
grep "^{\"metric" $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
