cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/retune
python tools/tune_context_wide.py hrnet_w48_cls-cliff 64 24 8 15 gpurun_out/retune/t1.json > gpurun_out/retune/w48_64.log 2>&1
python tools/tune_context_wide.py hrnet_w48_cls-cliff 128 16 6 10 gpurun_out/retune/t2.json > gpurun_out/retune/w48_128.log 2>&1
python tools/tune_context_wide.py hrnet_w32-pare 32 16 6 15 gpurun_out/retune/t3.json > gpurun_out/retune/pare_32.log 2>&1
python tools/tune_context_wide.py resnet50-cliff 64 12 6 15 gpurun_out/retune/t4.json > gpurun_out/retune/r50_64.log 2>&1
cp poco_amd/tuned/gfx950.json gpurun_out/retune/gfx950.json
tail -3 gpurun_out/retune/*.log
python bench.py --no-stream --no-cpu-baseline 2>/dev/null | cut -c1-200
