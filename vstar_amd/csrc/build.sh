#!/bin/bash
# Builds libvstar_hip.so (gfx950 only) in-tree next to the Python package.
set -e
cd "$(dirname "$0")"
OUT=../libvstar_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p build
pids=()
for f in gemm gemm256 norm attention elementwise heads preprocess engine; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.hpp -nt build/$f.o ] || [ kernels.hpp -nt build/$f.o ] || [ gemm_epilogue.hpp -nt build/$f.o ] || [ ../../include/vstar_hip.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $(realpath $OUT)"
