#!/bin/bash
# tools/pc_model_build.sh <name> "<flags>"  ->  tools/ab/pc_model_<name>.bin
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I experiments-lanczos-adjoints_amd/csrc -DMFX_PC_STAMP=1 $2 tools/pc_model.hip -o tools/ab/pc_model_$1.bin 2>&1 | grep -E "error" || true
echo built tools/ab/pc_model_$1.bin
