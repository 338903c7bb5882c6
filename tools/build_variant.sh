#!/bin/bash
# Builds an A/B variant of libvstar_hip.so: tools/build_variant.sh NAME "<extra hipcc flags, e.g. -DGEMM_EXP=1>" [file ...]
# Only the listed kernel files (default: gemm256) are recompiled with the flags; output vstar_amd/csrc/build/ab/lib_NAME.so
# (in-tree so that it travels to the GPU box; git-ignored).  Use with VSTAR_LIB=... (tools/ab_gemm.sh).
set -e
NAME=$1; EXTRA=$2; shift; shift
FILES=${@:-gemm256}
cd "$(dirname "$0")/../vstar_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p build/ab/$NAME
OBJS=$(ls build/*.o)
for f in $FILES; do
  [ "$f" = gemm4w ] && EXTRA="$EXTRA -mllvm -amdgpu-spill-vgpr-to-agpr=0"      # see build.sh
  $HIPCC $FLAGS $EXTRA -c $f.hip -o build/ab/$NAME/$f.o &
  if [ -f build/f16_$f.o ]; then $HIPCC $FLAGS $EXTRA -DVSTAR_LP_F16 -c $f.hip -o build/ab/$NAME/f16_$f.o & fi
  OBJS=$(echo "$OBJS" | grep -v "build/$f.o" | grep -v "build/f16_$f.o")
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS build/ab/$NAME/*.o -o build/ab/lib_$NAME.so
echo "built $(realpath build/ab/lib_$NAME.so)"
