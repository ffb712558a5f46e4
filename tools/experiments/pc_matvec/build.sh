#!/bin/bash
# tools/experiments/pc_matvec/build.sh <name> "<flags>"  ->  tools/ab/pc_model_<name>.bin  (the producer / consumer kernel alone, with
# cycle stamps; flags e.g. -DMFX_PC_DIAG=1 .. 7, -DMFX_PC_PRIO=1|2, -DMFX_PC_SWAP=1: see the macros at the top of mfx_rbf_pc.hip)
set -e
cd "$(dirname "$0")/../../.."
mkdir -p tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I experiments-lanczos-adjoints_amd/csrc -I tools/experiments/pc_matvec \
  -DMFX_PC_STAMP=1 $2 tools/experiments/pc_matvec/pc_model.hip -o tools/ab/pc_model_$1.bin 2>&1 | grep -E "error" || true
echo built tools/ab/pc_model_$1.bin
