cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6q
for L in libpoco_hip "exp/libpoco_hip_w4w_W4W_HOLD=2" "exp/libpoco_hip_w4w_W4W_HOLD=3" "exp/libpoco_hip_w4w_W4W_HOLD=4" libpoco_hip; do
  echo "== $L" >> gpurun_out/r6q/ab.log
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/w4w_ab.py 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6q/ab.log
done
cat gpurun_out/r6q/ab.log
