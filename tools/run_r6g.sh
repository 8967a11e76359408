cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
for v in "hrnet_w48_cls-cliff 64" "resnet50-cliff 64" "hrnet_w32-pare 32" "hrnet_w48_cls-cliff 128" "hrnet_w48_cls-cliff 16"; do
  timeout 300 python tools/fwd_time.py $v 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6g/fwd.log
done
cat gpurun_out/r6g/fwd.log
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r6g/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6g/tests.log
tail -15 gpurun_out/r6g/tests.log
