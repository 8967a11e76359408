cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6l
timeout 1200 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "whole_position" > gpurun_out/r6l/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r6l/t1.log
tail -8 gpurun_out/r6l/t1.log
for L in libpoco_hip libpoco_hip; do
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/w4w_ab.py 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6l/ab.log
done
cat gpurun_out/r6l/ab.log
for v in "resnet50-cliff 64" "hrnet_w32-pare 32"; do
  timeout 300 python tools/fwd_time.py $v 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6l/fwd.log
done
cat gpurun_out/r6l/fwd.log
