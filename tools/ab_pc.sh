#!/bin/bash
# same-box alternating A/B of the Gram matvec: producer / consumer kernel (MFX_RBF_PC=1, default) vs the same-program kernel
# it replaces (MFX_RBF_PC=0), C4 shape, 64 vectors.   usage: tools/ab_pc.sh [rounds] [reps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${1:-3}
REPS=${2:-20}
for i in $(seq $ROUNDS); do
  MFX_RBF_PC=0 python3 $R/tools/bench_matvec_one.py 64 $REPS | sed 's/^/h3 (MFX_RBF_PC=0)  /'
  MFX_RBF_PC=1 python3 $R/tools/bench_matvec_one.py 64 $REPS | sed 's/^/pc (default)       /'
done
