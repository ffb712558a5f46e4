#!/bin/bash
# The C4 accuracy gate's margin, measured on 16 probe sets instead of four: SLQ log-det value and gradient of the shipped default mode
# (f16x3) against the fp64 HIP path at n = 131072, d = 8, k = 40, 64 +-1 probes, probe keys 0 .. 15.  The fp64 reference of a key is
# taken from profiles/ when it is there (32 s each otherwise) and kept under gpurun_out/acc16/ (copy the new ones to
# profiles/r05b_accuracy_16_seeds/).   usage: [TAG=name] tools/run_accuracy_16seeds.sh ["0 1 2 ..."] [mode]   (TAG: a name for the table
# of a build / environment variant; MFX_* variables pass through to the library)
set -e
SEEDS=${1:-"0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15"}
MODE=${2:-f16x3}
OUT=gpurun_out/acc16${TAG:+_$TAG}
mkdir -p $OUT
for s in $SEEDS; do
  D=$OUT/seed$s
  mkdir -p $D
  REF=profiles/r05b_accuracy_16_seeds/f64_refs/seed${s}_f64.json
  [ -f $REF ] || REF=profiles/r02a_accuracy/other_probe_sets/seed${s}_f64.json
  [ "$s" = "0" ] && [ ! -f $REF ] && REF=profiles/r02a_accuracy/n131072_0_f64.json
  if [ -f $REF ]; then
    cp $REF $D/n131072_0_f64.json
  else
    timeout -k 10 600 python tools/accuracy_gate.py --mode f64 --seed $s --out $D/n131072_0_f64.json > /dev/null
  fi
  timeout -k 10 300 python tools/accuracy_gate.py --mode $MODE --seed $s --tag seed$s --out $D/n131072_1_${MODE}.json > /dev/null
  python tools/accuracy_gate.py --table $D | grep -v '^f64\|^ *n ' | tee -a $OUT/table_${MODE}.log
done
echo "== ${TAG:-default} $MODE"
python3 - $OUT/table_${MODE}.log <<'PY'
import sys
rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
mx = [float(r[-2]) for r in rows]
print(f"{len(mx)} probe sets: worst gradient component max {max(mx):.2e}, median {sorted(mx)[len(mx)//2]:.2e}, above 8e-5: {sum(m > 8e-5 for m in mx)}, above 1e-4: {sum(m > 1e-4 for m in mx)}")
PY
