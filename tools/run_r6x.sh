cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6x
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "split_k_small_batch" 2>&1 | tail -3
for b in 1 4; do
timeout 1200 python tools/splitk_tune.py hrnet_w48_cls-cliff $b --write 2>&1 | grep -v amdgpu.ids > gpurun_out/r6x/sk_w48_$b.log
grep -v "^    " gpurun_out/r6x/sk_w48_$b.log | tail -12
done
cp poco_amd/tuned/gfx950.json gpurun_out/r6x/gfx950.json
