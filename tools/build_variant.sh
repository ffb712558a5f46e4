#!/bin/bash
# build an alternative libmfx.so for same-call A/B runs (MFX_LIBRARY_PATH): tools/build_variant.sh <name> <file.hip> "<extra flags>"
# only <file.hip> is recompiled with the extra flags; the other objects are the in-tree ones.  Output: tools/ab/libmfx_<name>.so
set -e
cd "$(dirname "$0")/../experiments-lanczos-adjoints_amd/csrc"
NAME=$1; SRC=$2; FLAGS=$3
mkdir -p ../../tools/ab
OBJ=${TMPDIR:-/tmp}/mfx_${NAME}_${SRC%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function $FLAGS -c $SRC -o $OBJ
OTHERS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $OBJ -ldl -o ../../tools/ab/libmfx_${NAME}.so
echo built tools/ab/libmfx_${NAME}.so
