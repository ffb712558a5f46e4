#!/bin/bash
# PMC passes over the producer / consumer Gram matvec alone (C4 shape, 64 vectors); MFX_LIBRARY_PATH selects the build.
# usage: tools/prof_pc.sh [outdir-name] [kernel-name-substring]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAME=${1:-prof_pc}
FILTER=${2:-pc_apply}
OUT=$R/gpurun_out/$NAME
rm -rf $OUT && mkdir -p $OUT
python3 $R/tools/bench_matvec_one.py 64 10
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_matvec_one.py 64 4 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; continue; }
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  echo "== $grp"
  python3 - "$f" "$FILTER" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in row["Kernel_Name"] and "false, false>" not in row["Kernel_Name"]:  # not the empty range-guard launch
        a = acc[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, (v, c) in acc.items():
    print(f"  {k}: per launch {v / c:.4g} ({c} launches)")
PY
  rm -rf $OUT/p$i
done
