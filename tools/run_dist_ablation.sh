#!/bin/bash
# accuracy / speed ablation of the f16x3 distance variant (MFX_RBF_DIST=1) of the pipelined Gram kernel
set -e
for n in 16384 65536; do
  echo "== n=$n"
  MFX_RBF_DIST=0 MFX_RBF_MODE=1 python tools/accuracy_check.py --n $n 2>&1 | grep -v amdgpu.ids | sed "s/^/dist0 mode1 /"
  MFX_RBF_DIST=1 MFX_RBF_MODE=1 python tools/accuracy_check.py --n $n --skip64 2>&1 | grep -v amdgpu.ids | sed "s/^/dist1 mode1 /"
  MFX_RBF_DIST=0 MFX_RBF_MODE=2 python tools/accuracy_check.py --n $n --skip64 2>&1 | grep -v amdgpu.ids | sed "s/^/dist0 mode2 /"
  MFX_RBF_DIST=1 MFX_RBF_MODE=2 python tools/accuracy_check.py --n $n --skip64 2>&1 | grep -v amdgpu.ids | sed "s/^/dist1 mode2 /"
done
echo "== speed"
MFX_RBF_DIST=1 python tools/bench_matvec.py 2>&1 | grep -v amdgpu.ids
