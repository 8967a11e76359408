cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6k
for L in libpoco_hip "exp/libpoco_hip_w4w_W4W_YIELD=1" "exp/libpoco_hip_w4w_W4W_YIELD=2" "exp/libpoco_hip_w4w_W4W_YIELD=4" libpoco_hip; do
  echo "== $L" >> gpurun_out/r6k/ab.log
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/w4w_ab.py 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6k/ab.log
done
cat gpurun_out/r6k/ab.log
timeout 600 python tools/ab_shape_cfg.py hrnet_w48_cls-cliff 128 7x7x384x384 "2,4,2,2,2,1,11" "1,3,2,1,8,8,13" 2 "w4_min_plane=7" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6k/p7.log
cat gpurun_out/r6k/p7.log
