cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
for L in w4w_9_1 w4w_8_1 "w4w_8_1W4W_HOLD=2" w4w_6_1 w4w_9_1; do
  echo "== $L" >> gpurun_out/r6b/ab.log
  POCO_HIP_LIB=poco_amd/lib/exp/libpoco_hip_$L.so timeout 300 python tools/w4w_ab.py 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6b/ab.log
done
cat gpurun_out/r6b/ab.log
