cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7b
POCO_HIP_LIB="poco_amd/lib/exp/libpoco_hip_w4w_W4W_TRACE=1.so" timeout 300 python tools/w4w_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r7b/trace.log
cat gpurun_out/r7b/trace.log
