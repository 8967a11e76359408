cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6z
python tools/oplist.py hrnet_w48_cls-cliff 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r6z/oplist_w48_b16.txt
grep "stage4.1.branches.[0-3].0.conv1" gpurun_out/r6z/oplist_w48_b16.txt
for c in "1,1,16,1,1,1,5" "1,1,16,1,3,1,5" "1,1,16,1,2,1,5" "1,1,8,1,1,1,5"; do
timeout 300 python tools/ab_shape_cfg.py hrnet_w48_cls-cliff 16 7x7x384x384 "1,1,4,2,8,4,4" "$c" 1 2>&1 | grep -v amdgpu.ids | tail -2
done
for c in "1,1,16,1,1,1,5" "1,1,16,1,3,1,5" "1,1,8,1,1,1,5"; do
timeout 300 python tools/ab_shape_cfg.py hrnet_w48_cls-cliff 16 14x14x192x192 "1,2,4,2,14,1,4" "$c" 1 2>&1 | grep -v amdgpu.ids | tail -2
done
