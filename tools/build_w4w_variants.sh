#!/bin/bash
# Same-box A/B builds of conv_wino4w.hip: tools/build_w4w_variants.sh "<DMAW> <LAYOUT>" ... -> poco_amd/lib/exp/libpoco_hip_w4w_<DMAW>_<LAYOUT>.so
cd $(dirname $0)/..
mkdir -p poco_amd/lib/exp
OBJS=$(ls poco_amd/lib/obj/*.o | grep -v "/conv_wino4w.o")
for v in "$@"; do
  set -- $v
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DNDEBUG -DW4W_DMAW=$1 -DW4W_LAYOUT=$2 ${3:+-D$3} -x hip -c poco_amd/csrc/conv_wino4w.hip -o /tmp/w4w_$1_$2$3.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/w4w_$1_$2$3.o -o poco_amd/lib/exp/libpoco_hip_w4w_$1_$2$3.so && echo built $1 $2 $3 ) &
done
wait
