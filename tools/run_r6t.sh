cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6t
timeout 900 python tools/tail_cfg.py hrnet_w48_cls-cliff 64 3 "2,4,2,2,2,1,11" 8 1 2>&1 | grep -v amdgpu.ids > gpurun_out/r6t/tail7.log
cat gpurun_out/r6t/tail7.log
