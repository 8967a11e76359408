#!/bin/bash
# Whole-library build with extra compiler flags (timing experiments on one box, see tools/ab_lib.sh):
#   tools/build_flags.sh <name> <flags...>   ->   poco_amd/lib/exp/libpoco_hip_G_<name>.so
N=$1; shift
D=/tmp/objs_$N; mkdir -p $D poco_amd/lib/exp
for s in poco_amd/csrc/*.hip; do
  b=$(basename $s .hip)
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DNDEBUG "$@" -x hip -c $s -o $D/$b.o > $D/$b.log 2>&1 &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 $D/*.o -o poco_amd/lib/exp/libpoco_hip_G_$N.so && echo built $N
