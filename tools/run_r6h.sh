cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
for L in libpoco_hip "exp/libpoco_hip_w4w_W4W_NT2LAYOUT=1" libpoco_hip "exp/libpoco_hip_w4w_W4W_NT2LAYOUT=1"; do
  echo "== $L" >> gpurun_out/r6h/ab.log
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/w4w_ab.py 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6h/ab.log
done
cat gpurun_out/r6h/ab.log
for L in libpoco_hip "exp/libpoco_hip_w4w_W4W_NT2LAYOUT=1"; do
  echo "== $L" >> gpurun_out/r6h/fwd.log
  for v in "resnet50-cliff 64" "hrnet_w32-pare 32"; do
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/fwd_time.py $v 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6h/fwd.log
  done
done
cat gpurun_out/r6h/fwd.log
# 7x7 planes on ALG 13 (rectangular items of 8 whole images) against the table's ALG 11 entry, inside the forward
timeout 600 python tools/ab_shape_cfg.py hrnet_w48_cls-cliff 64 7x7x384x384 "2,4,2,2,0,0,11" "1,3,2,1,8,8,13" 2 "w4_min_plane=7" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6h/p7.log
cat gpurun_out/r6h/p7.log
