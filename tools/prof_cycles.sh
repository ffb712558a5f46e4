#!/bin/bash
# one PMC pass over the Gram matvec (C4 shape, 64 vectors): shader cycles, clock and matrix-pipe share of the kernel whose name
# contains $1 (default pc_apply).  MFX_* environment selects the kernel / the build, P the number of vectors (default 64).   usage: [P=32] tools/prof_cycles.sh [filter] [tag]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
FILTER=${1:-pc_apply}
OUT=/tmp/prof_cycles_$$
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/bench_matvec_one.py ${P:-64} 6 > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
python3 - "$OUT" "$FILTER" "${2:-}" <<'PY'
import csv, sys, glob, collections
out, filt, tag = sys.argv[1:4]
cc = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(cc)):
    if filt in row["Kernel_Name"]:
        acc[(row["Dispatch_Id"], row["Counter_Name"])].append(float(row["Counter_Value"]))
dur = {}
for row in csv.DictReader(open(kt)):
    if filt in row["Kernel_Name"]:
        dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
ids = sorted({k[0] for k in acc}, key=int)
ids = [i for i in ids if dur.get(i, 0) > 0.5][1:]   # skip the empty range-guard launches and the first (cold) launch
def avg(name): return sum(sum(acc[(i, name)]) for i in ids) / len(ids)
ms = sum(dur[i] for i in ids) / len(ids)
cyc = avg("GRBM_GUI_ACTIVE") / 8
print(f"{tag or filt}: {ms:.3f} ms, {cyc/1e6:.2f} Mcycles, {cyc/ms*1e-6:.2f} GHz, matrix pipe {avg('SQ_VALU_MFMA_BUSY_CYCLES')/1024/cyc*100:.1f} %, "
      f"VALU insts {avg('SQ_INSTS_VALU')/1e9:.3f} G ({avg('SQ_ACTIVE_INST_VALU')*4/avg('SQ_INSTS_VALU'):.2f} cycles each), "
      f"wait-inst {avg('SQ_WAIT_INST_ANY')/avg('SQ_WAVE_CYCLES')*100:.0f} % of wave cycles  [{len(ids)} launches]")
PY
rm -rf $OUT
