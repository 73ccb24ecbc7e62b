#!/bin/bash
# Builds the REFERENCE's own GPU test programs -- tests/test_gpu.cu (small_test, options_test, inf_test, grad_check
# through compute_rnnt_loss with RNNT_GPU) and tests/test_time.cu (its timing harness) -- UNMODIFIED, from the sources
# where they lie under $REF, against THIS repo's include/rnnt.h and libwarprnnt.so.  Test infrastructure
# (tests/test_reference_sources.py, -m gpu); nothing is copied into the repository, only the two binaries land in the
# git-ignored oracle/_ref/ (they travel to the GPU box; cross-compiled here: host code only, no kernels).
# The .cu files name nine CUDA runtime entry points; they are mapped to their HIP twins on the COMMAND LINE of this
# build (the reference's `cudaStream_t` is CUDA's `CUstream_st*`, which is exactly the `CUstream` of rnnt.h, so the
# stream variable keeps that type and is cast where HIP wants a hipStream_t).
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
LIB=$ROOT/warp-transducer_amd/lib
[ -f "$REF/tests/test_gpu.cu" ] || { echo "reference checkout absent: keeping prebuilt oracle/_ref/ref_test_* (if any)"; exit 0; }
[ -f "$LIB/libwarprnnt.so" ] || { echo "libwarprnnt.so not built yet"; exit 0; }
OUT=$HERE/_ref
mkdir -p "$OUT"
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
MAP=(-DcudaStream_t=CUstream
     '-DcudaStreamCreate(p)=hipStreamCreate(reinterpret_cast<hipStream_t*>(p))'
     '-DcudaMemcpyAsync(d,s,n,k,st)=hipMemcpyAsync(d,s,n,k,reinterpret_cast<hipStream_t>(st))'
     -DcudaMalloc=hipMalloc -DcudaMemcpy=hipMemcpy -DcudaFree=hipFree
     -DcudaMemcpyHostToDevice=hipMemcpyHostToDevice -DcudaMemcpyDeviceToHost=hipMemcpyDeviceToHost)
g++ -O1 -std=c++14 -I"$ROOT/include" -I"$REF/tests" -c "$REF/tests/random.cpp" -o "$TMP/random.o"
for t in test_gpu test_time; do
  if [ "$OUT/ref_$t" -nt "$REF/tests/$t.cu" ] && [ "$OUT/ref_$t" -nt "$ROOT/include/rnnt.h" ] && [ "$OUT/ref_$t" -nt "$0" ]; then
    echo "up to date: $OUT/ref_$t"; continue
  fi
  hipcc --offload-arch=gfx950 -O1 -std=c++14 -w -x hip -include hip/hip_runtime.h "${MAP[@]}" \
        -I"$ROOT/include" -I"$REF/tests" -c "$REF/tests/$t.cu" -o "$TMP/$t.o"
  hipcc --offload-arch=gfx950 "$TMP/$t.o" "$TMP/random.o" -o "$OUT/ref_$t" -L"$LIB" -lwarprnnt -Wl,-rpath,"$LIB"
  echo "built $OUT/ref_$t from $REF/tests/$t.cu"
done
