#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  OUT=$R/gpurun_out/prof_mv_stats_$v
  rm -rf $OUT && mkdir -p $OUT
  MFX_RBF_PACK=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_matvec_one.py 64 10 > $OUT/log 2>&1
  echo "== pack=$v"; cat $OUT/log | grep -v amdgpu | tail -1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-200
done
