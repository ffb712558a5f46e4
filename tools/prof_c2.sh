#!/bin/bash
# rocprofv3 kernel stats of the C2-shaped run of tools/bench_configs.py (n = 45730, d = 9 ARD, k = 30, 8 probes)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_c2
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_configs.py c2 > $OUT/run.log 2>&1
grep "^C2" $OUT/run.log
python3 $R/tools/trace_summary.py $OUT mfx:: | cut -c1-170 | head -16
