cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6v
timeout 1500 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "whole_position" > gpurun_out/r6v/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r6v/t1.log
tail -3 gpurun_out/r6v/t1.log
for L in libpoco_hip "exp/libpoco_hip_w4w_W4W_RESEARLY=0" libpoco_hip "exp/libpoco_hip_w4w_W4W_RESEARLY=0"; do
  echo "== $L" >> gpurun_out/r6v/ab.log
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/w4w_ab.py 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6v/ab.log
done
cat gpurun_out/r6v/ab.log
for L in libpoco_hip "exp/libpoco_hip_w4w_W4W_RESEARLY=0" libpoco_hip "exp/libpoco_hip_w4w_W4W_RESEARLY=0"; do
POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/fwd_time.py hrnet_w48_cls-cliff 64 2>&1 | grep -v amdgpu.ids
done
