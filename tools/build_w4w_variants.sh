#!/bin/bash
# Same-box A/B builds of conv_wino4w.hip: tools/build_w4w_variants.sh "MACRO=v[,MACRO=v...]" ...
#   -> poco_amd/lib/exp/libpoco_hip_w4w_<macros>.so (select with POCO_HIP_LIB)
cd $(dirname $0)/..
mkdir -p poco_amd/lib/exp
OBJS=$(ls poco_amd/lib/obj/*.o | grep -v "/conv_wino4w.o")
for v in "$@"; do
  DEFS=""; for m in ${v//,/ }; do DEFS="$DEFS -D$m"; done
  TAG=${v//,/_}
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DNDEBUG $DEFS -x hip -c poco_amd/csrc/conv_wino4w.hip -o /tmp/w4w_$TAG.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/w4w_$TAG.o -o poco_amd/lib/exp/libpoco_hip_w4w_$TAG.so && echo built $TAG ) &
done
wait
