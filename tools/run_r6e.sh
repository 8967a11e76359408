cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
timeout 1200 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "whole_position" > gpurun_out/r6e/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r6e/t1.log
tail -5 gpurun_out/r6e/t1.log
for L in libpoco_hip "exp/libpoco_hip_w4w_W4W_RD=3" "exp/libpoco_hip_w4w_W4W_DMAWAVES=0" libpoco_hip; do
  echo "== $L" >> gpurun_out/r6e/ab.log
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/w4w_ab.py 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6e/ab.log
done
cat gpurun_out/r6e/ab.log
POCO_HIP_LIB="poco_amd/lib/exp/libpoco_hip_w4w_W4W_TRACE=1.so" timeout 300 python tools/w4w_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6e/trace.log
cat gpurun_out/r6e/trace.log
for v in "hrnet_w48_cls-cliff 64" "resnet50-cliff 64" "hrnet_w32-pare 32"; do
  timeout 300 python tools/fwd_time.py $v 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6e/fwd.log
done
cat gpurun_out/r6e/fwd.log
