#!/bin/bash
# matvec error per input kind and column-split count, then the end-to-end gradient error with forced splits
set -e
for mode in f16x3-matvec fp32; do
  timeout -k 10 300 python tools/diag_matvec_bias.py --mode $mode
done
for s in 2 4 16; do
  MFX_RBF_SPLIT=$s timeout -k 10 300 python tools/diag_matvec_bias.py --mode f16x3-matvec
done
mkdir -p gpurun_out/acc2
cp profiles/r02a_accuracy/n131072_0_f64.json gpurun_out/acc2/
for s in 4 16; do
  MFX_RBF_SPLIT=$s timeout -k 10 300 python tools/accuracy_gate.py --mode f16x3-matvec --tag split$s --n 131072 --p 64 --out gpurun_out/acc2/n131072_3_split$s.json
done
python tools/accuracy_gate.py --table gpurun_out/acc2
