# A/B of alternative library builds on ONE box: bash tools/ab_lib.sh <tag> v1 v2 ...  (poco_amd/lib/exp/libpoco_hip_<tag>_<v>.so; "base" = the in-tree build)
TAG=$1; shift
mkdir -p gpurun_out/r3x
OUT=gpurun_out/r3x/ab_$TAG.txt
: > $OUT
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset POCO_HIP_LIB; else export POCO_HIP_LIB=$PWD/poco_amd/lib/exp/libpoco_hip_${TAG}_$v.so; fi
  L="$v:"
  for shp in "64 56 56 48 48 3 1 1 3 2 4 8 1 8" "64 28 28 96 96 3 1 1 3 2 4 16 1 8" "64 14 14 192 192 3 1 1 3 2 4 16 2 8" "32 56 56 32 32 3 1 1 2 2 4 8 1 8" "32 28 28 64 64 3 1 1 2 2 4 16 1 8"; do
    L="$L $(python tools/conv_time.py $shp 2>&1 | tail -1)"
  done
  if [ "$BENCH" = 1 ]; then L="$L | $(python bench.py --no-side --no-cpu-baseline --no-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"; fi
  echo "$L" >> $OUT
done
done
