cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6o
timeout 1500 python tools/w4w_tune.py hrnet_w32-pare 32 --write 2>&1 | grep -v amdgpu.ids > gpurun_out/r6o/tune_pare_32.log
grep -v "^    " gpurun_out/r6o/tune_pare_32.log | tail -20
timeout 1500 python tools/w4w_tune.py resnet50-cliff 64 --write 2>&1 | grep -v amdgpu.ids > gpurun_out/r6o/tune_r50_64.log
grep -v "^    " gpurun_out/r6o/tune_r50_64.log | tail -12
timeout 1500 python tools/w4w_tune.py hrnet_w48_cls-cliff 16 --write 2>&1 | grep -v amdgpu.ids > gpurun_out/r6o/tune_w48_16.log
grep -v "^    " gpurun_out/r6o/tune_w48_16.log | tail -12
cp poco_amd/tuned/gfx950.json gpurun_out/r6o/gfx950.json
