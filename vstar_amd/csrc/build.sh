#!/bin/bash
# Builds libvstar_hip.so (gfx950 only) in-tree next to the Python package.
set -e
cd "$(dirname "$0")"
OUT=../libvstar_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p build
rm -f build/gemm256a.o build/f16_gemm256a.o     # (round-3 experiment, moved to tools/experiments/gemm256a)
pids=()
HDRS="common.hpp kernels.hpp gemm_epilogue.hpp gemm256_direct_epilogue.hpp mx.hpp gemm4w_loop.inc engine_base.hpp llm_cached.hpp ../../include/vstar_hip.h ../../include/vstar_vqa.h"
stale() {  # stale <object> <source>
  [ ! -f "$1" ] && return 0
  [ "$2" -nt "$1" ] && return 0
  for h in $HDRS; do [ -f "$h" ] && [ "$h" -nt "$1" ] && return 0; done
  return 1
}
# bf16 instantiation: every kernel file + the VSM engine
# gemm4w keeps its 256 accumulators in AGPRs BEHIND the compiler's back (they are clobbers of the K-loop asm statement, read back by
# v_accvgpr_read statements in the epilogue): the compiler must never use AGPRs as VGPR spill space there — it did (a2..a9, round 6)
extra() { [ "$1" = gemm4w ] && echo "-mllvm -amdgpu-spill-vgpr-to-agpr=0"; }
for f in gemm gemm256 gemm4w norm attention elementwise decode quant heads preprocess engine comm; do
  if stale build/$f.o $f.hip; then $HIPCC $FLAGS $(extra $f) -c $f.hip -o build/$f.o & pids+=($!); fi
done
# fp16 instantiation (-DVSTAR_LP_F16): the dtype-generic kernel files + the VQA-LLM engine
for f in gemm gemm256 gemm4w norm attention elementwise decode vqa_engine; do
  [ -f $f.hip ] || continue
  if stale build/f16_$f.o $f.hip; then $HIPCC $FLAGS $(extra $f) -DVSTAR_LP_F16 -c $f.hip -o build/f16_$f.o & pids+=($!); fi
done
# the hash of the kernel sources THIS binary is built from (vstar_amd/provenance.py::kernel_source_hash, same algorithm), compiled
# into the library and exported as vstar_build_source_hash(): evidence files are stamped with the LOADED library's value, so a
# .hip edited between building and profiling / summarising shows up as a mismatch instead of a matching stamp.
SRC_HASH=$(python3 - <<'PY'
import glob, hashlib, os
h = hashlib.sha256()
for path in sorted(glob.glob("*.hip") + glob.glob("*.hpp") + ["build.sh"]):
    h.update(os.path.basename(path).encode())
    h.update(open(path, "rb").read())
print(h.hexdigest()[:16])
PY
)
cat > build/src_hash.cpp.new <<EOF2
extern "C" const char* vstar_build_source_hash(void) { return "$SRC_HASH"; }
EOF2
if ! cmp -s build/src_hash.cpp.new build/src_hash.cpp; then mv build/src_hash.cpp.new build/src_hash.cpp; else rm build/src_hash.cpp.new; fi
if [ ! -f build/src_hash.o ] || [ build/src_hash.cpp -nt build/src_hash.o ]; then g++ -O2 -fPIC -c build/src_hash.cpp -o build/src_hash.o; fi
fail=0
for p in "${pids[@]}"; do wait $p || fail=1; done
[ $fail -eq 0 ] || { echo "build failed"; exit 1; }
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $(realpath $OUT)"
