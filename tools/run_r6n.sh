cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6n
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "tuned_table or bench_batch or golden or timed_out" > gpurun_out/r6n/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r6n/t1.log
tail -6 gpurun_out/r6n/t1.log
for v in "hrnet_w48_cls-cliff 64" "hrnet_w48_cls-cliff 128" "resnet50-cliff 64" "hrnet_w32-pare 32"; do
  timeout 300 python tools/fwd_time.py $v 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6n/fwd.log
done
cat gpurun_out/r6n/fwd.log
