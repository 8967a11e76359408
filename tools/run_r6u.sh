cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6u
timeout 1500 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "whole_position" > gpurun_out/r6u/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r6u/t1.log
tail -4 gpurun_out/r6u/t1.log
for L in libpoco_hip "exp/libpoco_hip_w4w_W4W_PEEL=0" "exp/libpoco_hip_w4w_W4W_ATPK=0" "exp/libpoco_hip_w4w_W4W_PEEL=0_W4W_ATPK=0" libpoco_hip "exp/libpoco_hip_w4w_W4W_PEEL=0_W4W_ATPK=0"; do
  echo "== $L" >> gpurun_out/r6u/ab.log
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/w4w_ab.py 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6u/ab.log
done
cat gpurun_out/r6u/ab.log
for v in "hrnet_w48_cls-cliff 64" "hrnet_w32-pare 32" "resnet50-cliff 64"; do timeout 300 python tools/fwd_time.py $v 2>&1 | grep -v amdgpu.ids; done
