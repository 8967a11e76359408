# usage: [MACRO=..] [TOOL=..] [PAT=..] bash tools/exp_run.sh v1 v2 ...   (0 = the regular build)
for v in "$@"; do
  if [ $v = 0 ]; then unset POCO_HIP_LIB; else export POCO_HIP_LIB=$PWD/poco_amd/lib/exp/libpoco_hip_${MACRO:-WINO_EXP}_$v.so; fi
  echo "EXP $v"; python ${TOOL:-tools/wino_slope.py} ${TOOLARGS} 2>&1 | grep "${PAT:-, 4)}"
done
