#!/bin/bash
# PMC passes over the gradient GEMM alone (separate runs per counter group)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_grad
rm -rf $OUT && mkdir -p $OUT
python3 $R/tools/bench_grad.py 131072 2560 3
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_grad.py 131072 2560 1 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; continue; }
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  echo "== $grp"
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if "grad_h" in row["Kernel_Name"]:
        a = acc[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, (v, c) in acc.items():
    print(f"  {k}: total {v:.4g} over {c} rows")
PY
done
