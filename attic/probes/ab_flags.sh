#!/bin/bash
# A/B of whole-library builds (tools/build_flags.sh) on ONE box: bash tools/ab_flags.sh <tag> base a b ...
# per library: uses-weighted sum of all tuned entries at B = 64 and 32, then the three bench lines
TAG=$1; shift
mkdir -p gpurun_out/r3x
OUT=gpurun_out/r3x/abflags_$TAG.txt
: > $OUT
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset POCO_HIP_LIB; else export POCO_HIP_LIB=$PWD/poco_amd/lib/exp/libpoco_hip_${TAG}_$v.so; fi
  L="$v: t64 $(python tools/s2_table_time.py 64 k 2>/dev/null | tail -1 | sed 's/sum over uses: //') t32 $(python tools/s2_table_time.py 32 k 2>/dev/null | tail -1 | sed 's/sum over uses: //')"
  L="$L | w48 $(python bench.py --no-side --no-cpu-baseline --no-stream 2>/dev/null | val)"
  L="$L | r50 $(python bench.py --variant resnet50-cliff --no-side --no-cpu-baseline --no-stream 2>/dev/null | val)"
  L="$L | pare $(python bench.py --variant hrnet_w32-pare --batch 32 --no-side --no-cpu-baseline --no-stream 2>/dev/null | val)"
  echo "$L" >> $OUT
done
done
