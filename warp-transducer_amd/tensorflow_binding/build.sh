#!/bin/sh
# Builds warprnnt_tensorflow/kernels.so against the installed tensorflow-rocm and ../lib/libwarprnnt.so.
set -e
here=$(cd "$(dirname "$0")" && pwd)
python - <<'PY' >/dev/null || { echo "tensorflow is not importable: nothing built" >&2; exit 1; }
import tensorflow
PY
TF_CFLAGS=$(python -c 'import tensorflow as tf; print(" ".join(tf.sysconfig.get_compile_flags()))')
TF_LFLAGS=$(python -c 'import tensorflow as tf; print(" ".join(tf.sysconfig.get_link_flags()))')
make -C "$here/.." lib/libwarprnnt.so
hipcc -std=c++17 -O2 -shared -fPIC -DTENSORFLOW_USE_ROCM=1 "$here/warprnnt_op.cc" \
      -o "$here/warprnnt_tensorflow/kernels.so" -I"$here/../../include" $TF_CFLAGS $TF_LFLAGS \
      -L"$here/../lib" -lwarprnnt -Wl,-rpath,"$here/../lib"
echo "built $here/warprnnt_tensorflow/kernels.so"
