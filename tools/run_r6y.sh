cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6y
for vb in "hrnet_w32-pare 1" "resnet50-cliff 1" "hrnet_w48_cls-cliff 16" "hrnet_w32-pare 4" "resnet50-cliff 4"; do
set -- $vb
timeout 1200 python tools/splitk_tune.py $1 $2 --write 2>&1 | grep -v amdgpu.ids > gpurun_out/r6y/sk_$1_$2.log
tail -2 gpurun_out/r6y/sk_$1_$2.log
done
cp poco_amd/tuned/gfx950.json gpurun_out/r6y/gfx950.json
