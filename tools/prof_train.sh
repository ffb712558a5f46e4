#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_train
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 experiments/applications/gaussian_process/train/optim_logml_adjoints_adaptive.py --name prof --seed 1 --dataset protein --rank_precon 15 --num_partitions 10 --num_matvecs 10 --num_samples 10 --num_epochs 20 --cg_tol 1.0 > $OUT/log 2>&1
grep "seconds per epoch" $OUT/log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:14]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:8.2f} ms avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
