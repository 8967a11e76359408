cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7g
for vb in "hrnet_w48_cls-cliff 128" "hrnet_w48_cls-cliff 32" "hrnet_w48_cls-cliff 64" "hrnet_w32-pare 32"; do
set -- $vb
timeout 1500 python tools/w4w_tune.py $1 $2 --write 2>&1 | grep -v amdgpu.ids > gpurun_out/r7g/tune_$1_$2.log
tail -2 gpurun_out/r7g/tune_$1_$2.log
done
cp poco_amd/tuned/gfx950.json gpurun_out/r7g/gfx950.json
