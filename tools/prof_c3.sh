#!/bin/bash
# rocprofv3 kernel stats of BASELINE config 3 (CSR n = 102400, k = 50, fp64, one vector): kernel durations vs wall time
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_c3${1:+_$1}
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_configs.py c3 > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
grep "^C3" $OUT/run.log
head -14 $OUT/kernel_stats.csv | cut -c1-200
