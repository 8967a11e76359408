set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "timed_out or fused_regressor or golden_b2" > gpurun_out/r6a/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r6a/t1.log
POCO_HIP_LIB=poco_amd/lib/exp/libpoco_hip_W4W_TRACE_1.so timeout 300 python tools/w4w_trace.py > gpurun_out/r6a/trace.log 2>&1
timeout 600 python tools/w4w_solo.py 64 > gpurun_out/r6a/solo.log 2>&1
timeout 300 python tools/fwd_time.py hrnet_w48_cls-cliff 64 > gpurun_out/r6a/fwd.log 2>&1
timeout 300 python tools/fwd_time.py resnet50-cliff 64 >> gpurun_out/r6a/fwd.log 2>&1
timeout 300 python tools/fwd_time.py hrnet_w32-pare 32 >> gpurun_out/r6a/fwd.log 2>&1
tail -5 gpurun_out/r6a/t1.log; cat gpurun_out/r6a/trace.log gpurun_out/r6a/fwd.log; tail -30 gpurun_out/r6a/solo.log
