cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
timeout 600 python tools/ab_shape_cfg.py hrnet_w48_cls-cliff 64 7x7x384x384 "2,4,2,2,2,1,11" "1,3,2,1,8,8,13" 2 "w4_min_plane=7" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6i/p7.log
timeout 600 python tools/ab_shape_cfg.py hrnet_w48_cls-cliff 64 7x7x384x384 "2,4,2,2,2,1,11" "1,2,2,1,8,8,13" 1 "w4_min_plane=7" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6i/p7.log
cat gpurun_out/r6i/p7.log
timeout 1500 python tools/w4w_tune.py hrnet_w48_cls-cliff 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r6i/tune_w48_64.log
tail -30 gpurun_out/r6i/tune_w48_64.log
