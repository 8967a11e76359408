cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6p
for L in libpoco_hip "exp/libpoco_hip_w4w_W4W_HOLD=1" "exp/libpoco_hip_w4w_W4W_BIAS11=1" "exp/libpoco_hip_w4w_W4W_HOLD=1_W4W_BIAS11=1" libpoco_hip "exp/libpoco_hip_w4w_W4W_HOLD=1_W4W_BIAS11=1"; do
  echo "== $L" >> gpurun_out/r6p/ab.log
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/w4w_ab.py 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6p/ab.log
done
cat gpurun_out/r6p/ab.log
POCO_HIP_LIB="poco_amd/lib/exp/libpoco_hip_w4w_W4W_HOLD=1_W4W_BIAS11=1.so" timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "whole_position" 2>&1 | tail -3
for L in libpoco_hip "exp/libpoco_hip_w4w_W4W_HOLD=1_W4W_BIAS11=1"; do
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/fwd_time.py hrnet_w48_cls-cliff 64 2>&1 | grep -v amdgpu.ids
done
