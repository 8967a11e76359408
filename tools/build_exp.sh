#!/bin/bash
# Build timing-experiment variants of the library: tools/build_exp.sh conv_wino.hip WINO_EXP 1 2 4 ...
# -> poco_amd/lib/exp/libpoco_hip_<MACRO>_<v>.so (select with POCO_HIP_LIB=...)
SRC=$1; MACRO=$2; shift 2
cd $(dirname $0)/..
mkdir -p poco_amd/lib/exp
OBJS=$(ls poco_amd/lib/obj/*.o | grep -v "/${SRC%.hip}.o")
for v in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DNDEBUG -D$MACRO=$v -x hip -c poco_amd/csrc/$SRC -o /tmp/exp_$v.o &&
  hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/exp_$v.o -o poco_amd/lib/exp/libpoco_hip_${MACRO}_$v.so && echo built $v &
done
wait
