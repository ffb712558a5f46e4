#!/bin/bash
# launch-bound configs with the fused cooperative Arnoldi kernels (default) and without (MFX_FUSED=0), alternating
for i in 1 2; do
  for f in 1 0; do  # (historical: MFX_FUSED=1 selected the cooperative kernels of commit 0a8fa0a, since removed)
    echo "== MFX_FUSED=$f"
    MFX_FUSED=$f timeout -k 10 300 python tools/bench_configs.py c1 c3 c5 2>&1 | grep -v amdgpu.ids
  done
done
