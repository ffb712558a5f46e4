#!/bin/bash
# How much of the C4 accuracy table is rounding noise?  The same comparison with other +-1 probe sets (seed s: its own fp64 reference,
# taken from profiles/r02a_accuracy/other_probe_sets/ when it is there: 32 s each otherwise).
# usage: tools/run_accuracy_seeds.sh "1 2 3"     output: gpurun_out/acc_seed<s>/ + one table per seed on stdout
set -e
for s in ${1:-"1 2 3"}; do
  OUT=gpurun_out/acc_seed$s
  mkdir -p $OUT
  if [ -f profiles/r02a_accuracy/other_probe_sets/seed${s}_f64.json ]; then
    cp profiles/r02a_accuracy/other_probe_sets/seed${s}_f64.json $OUT/n131072_0_f64.json
  else
    timeout -k 10 600 python tools/accuracy_gate.py --mode f64 --seed $s --out $OUT/n131072_0_f64.json > /dev/null
  fi
  for mode in f16x3-matvec f16x3; do
    timeout -k 10 300 python tools/accuracy_gate.py --mode $mode --seed $s --tag seed$s --out $OUT/n131072_1_${mode}.json > /dev/null
  done
  python tools/accuracy_gate.py --table $OUT
done
