#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_c5
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_configs.py ${CFG:-c5} > $OUT/log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:16]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:8.2f} ms avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
