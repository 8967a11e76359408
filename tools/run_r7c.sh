cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7c
timeout 1500 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "whole_position" > gpurun_out/r7c/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r7c/t1.log
tail -3 gpurun_out/r7c/t1.log
for L in libpoco_hip exp/libpoco_hip_prev "exp/libpoco_hip_w4w_W4W_PINOFF=1" libpoco_hip exp/libpoco_hip_prev; do
  echo "== $L" >> gpurun_out/r7c/ab.log
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/w4w_ab.py 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r7c/ab.log
done
cat gpurun_out/r7c/ab.log
POCO_HIP_LIB="poco_amd/lib/exp/libpoco_hip_w4w_W4W_TRACE=1.so" timeout 300 python tools/w4w_trace.py 2>&1 | grep -v amdgpu.ids | grep "MFMA wave [03]" 
for L in libpoco_hip exp/libpoco_hip_prev libpoco_hip exp/libpoco_hip_prev; do
POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/fwd_time.py hrnet_w48_cls-cliff 64 2>&1 | grep -v amdgpu.ids
done
