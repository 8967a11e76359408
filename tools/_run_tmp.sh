cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7h
for vb in "hrnet_w48_cls-cliff 64 200" "hrnet_w48_cls-cliff 128 60" "hrnet_w48_cls-cliff 1 300" "hrnet_w32-pare 32 200" "resnet50-cliff 64 200" "hrnet_w48_cls-cliff 16 150"; do
timeout 900 python tools/stress.py $vb 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/r7h/stress.log
done
cat gpurun_out/r7h/stress.log
