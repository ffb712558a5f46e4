#!/bin/bash
# build an alternative libmfx.so for same-call A/B runs (MFX_LIBRARY_PATH): tools/build_variant.sh <name> <file.hip>[,<file2.hip>...] "<extra flags>"
# only the named sources are recompiled with the extra flags; the other objects are the in-tree ones.  Output: tools/ab/libmfx_<name>.so
set -e
cd "$(dirname "$0")/../experiments-lanczos-adjoints_amd/csrc"
NAME=$1; SRCS=${2//,/ }; FLAGS=$3
mkdir -p ../../tools/ab
OBJS=""; SKIP=""
for SRC in $SRCS; do
  OBJ=${TMPDIR:-/tmp}/mfx_${NAME}_${SRC%.hip}.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function $FLAGS -c $SRC -o $OBJ &
  OBJS="$OBJS $OBJ"; SKIP="$SKIP ${SRC%.hip}.o"
done
wait
OTHERS=$(for o in *.o; do case " $SKIP " in *" $o "*) ;; *) echo $o;; esac; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $OBJS -ldl -o ../../tools/ab/libmfx_${NAME}.so
echo built tools/ab/libmfx_${NAME}.so
