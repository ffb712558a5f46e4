#!/bin/bash
# fp32 arithmetic modes against the fp64 HIP path at n = 65536 and the C4 size n = 131072 (p probes each) -> table.
# usage: tools/run_accuracy_gate.sh [p] [sizes...]      output: gpurun_out/acc/*.json + gpurun_out/accuracy_gate.log
# The fp64 reference of a size is taken from profiles/r02a_accuracy/ when it is there (32 s at n = 131072 otherwise).
set -e
P=${1:-64}
shift || true
SIZES=${@:-"65536 131072"}
OUT=gpurun_out/acc
mkdir -p $OUT
for n in $SIZES; do
  echo "== n=$n p=$P" | tee -a gpurun_out/accuracy_gate_progress.log
  if [ -f profiles/r02a_accuracy/n${n}_0_f64.json ] && [ "$P" = "64" ]; then
    cp profiles/r02a_accuracy/n${n}_0_f64.json $OUT/
  else
    timeout -k 10 600 python tools/accuracy_gate.py --mode f64 --n $n --p $P --out $OUT/n${n}_0_f64.json
  fi
  for mode in fp32 f16x3-matvec f16x3; do
    timeout -k 10 300 python tools/accuracy_gate.py --mode $mode --n $n --p $P --out $OUT/n${n}_1_${mode}.json
  done
done
python tools/accuracy_gate.py --table $OUT | tee gpurun_out/accuracy_gate.log
