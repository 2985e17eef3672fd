#!/bin/bash
# Build an A/B variant of libcoregex_hip.so: scripts/build_variant.sh NAME "EXTRA FLAGS" file.hip [file.hip ...]
# Only the named sources are recompiled with the extra flags (into coregex_amd/csrc/build_var/NAME/); everything else links
# from the product's objects.  Output: coregex_amd/variants/libcoregex_hip_NAME.so — selected with CXG_LIB_PATH (coregex_amd/_lib.py).
# Variants are experiments (ablations give WRONG rows): never loaded by the product, tests or bench.py's default run.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2; shift 2
SRC=$ROOT/coregex_amd/csrc
make -s -j8 -C $SRC
TORCH_LIB=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
mkdir -p $SRC/build_var/$NAME $ROOT/coregex_amd/variants
OBJS=""
for o in $(cd $SRC/build && find . -name '*.o' | sed 's|^\./||'); do
  s=${o%.o}
  use=$SRC/build/$o
  for f in "$@"; do
    if [ "$f" = "$s" ]; then
      mkdir -p $(dirname $SRC/build_var/$NAME/$o)
      /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-function $EXTRA -x hip -c $SRC/$s -o $SRC/build_var/$NAME/$o
      use=$SRC/build_var/$NAME/$o
    fi
  done
  OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $OBJS -o $ROOT/coregex_amd/variants/libcoregex_hip_$NAME.so -L$TORCH_LIB -Wl,-rpath,$TORCH_LIB
echo built coregex_amd/variants/libcoregex_hip_$NAME.so
