#!/bin/bash
# PMC passes over the pipelined Gram matvec alone (C4 shape, 64 probes)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_matvec
rm -rf $OUT && mkdir -p $OUT
python3 $R/tools/bench_matvec_one.py 64 10
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_matvec_one.py 64 4 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; continue; }
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  echo "== $grp"
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if "apply_h3" in row["Kernel_Name"] and "false, false>" not in row["Kernel_Name"]:  # not the empty range-guard launch
        a = acc[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, (v, c) in acc.items():
    print(f"  {k}: per launch {v / c:.4g} ({c} launches)")
PY
done
