cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6w
timeout 1500 python tools/w4w_tune.py hrnet_w32-pare 32 --write --two-pass 2>&1 | grep -v amdgpu.ids > gpurun_out/r6w/tune_pare_32.log
grep -v "^    " gpurun_out/r6w/tune_pare_32.log | tail -3
timeout 1500 python tools/w4w_tune.py hrnet_w48_cls-cliff 128 --write 2>&1 | grep -v amdgpu.ids > gpurun_out/r6w/tune_w48_128.log
grep -v "^    " gpurun_out/r6w/tune_w48_128.log | tail -3
timeout 1500 python tools/w4w_tune.py hrnet_w48_cls-cliff 32 --write 2>&1 | grep -v amdgpu.ids > gpurun_out/r6w/tune_w48_32.log
grep -v "^    " gpurun_out/r6w/tune_w48_32.log | tail -3
timeout 1500 python tools/w4w_tune.py resnet50-cliff 64 --write 2>&1 | grep -v amdgpu.ids > gpurun_out/r6w/tune_r50_64.log
grep -v "^    " gpurun_out/r6w/tune_r50_64.log | tail -3
cp poco_amd/tuned/gfx950.json gpurun_out/r6w/gfx950.json
python tools/oplist.py hrnet_w48_cls-cliff 1 > gpurun_out/r6w/oplist_w48_b1.txt 2>&1
