cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6s
timeout 1500 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "whole_position or tuned_table" > gpurun_out/r6s/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r6s/t1.log
tail -4 gpurun_out/r6s/t1.log
for L in libpoco_hip "exp/libpoco_hip_w4w_W4W_ASMMAX=0" libpoco_hip "exp/libpoco_hip_w4w_W4W_ASMMAX=0"; do
  echo "== $L" >> gpurun_out/r6s/ab.log
  POCO_HIP_LIB=poco_amd/lib/$L.so timeout 300 python tools/w4w_ab.py 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6s/ab.log
done
cat gpurun_out/r6s/ab.log
python - <<'P' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0,'.')
import numpy as np, torch
from poco_amd import ops
x=torch.randn(64,7,7,384,device='cuda:0'); w=(np.random.default_rng(0).standard_normal((384,384,3,3))/60).astype(np.float32)
for cfg in [(1,3,2,1,8,8,13),(2,4,2,2,2,1,11)]:
    print(cfg, min(ops.bench_conv2d(x,w,1,cfg=cfg,iters=40)[0]*1e3 for _ in range(3)))
P
for v in "hrnet_w48_cls-cliff 64" "hrnet_w48_cls-cliff 128"; do timeout 300 python tools/fwd_time.py $v 2>&1 | grep -v amdgpu.ids; done
timeout 600 python tools/conv_traffic.py 64 7 7 384 384 "1,3,2,1,8,8,13;2,4,2,2,2,1,11" 2>&1 | grep -v amdgpu.ids | tail -5; timeout 600 python tools/conv_traffic.py 64 14 14 192 192 "2,3,2,1,16,2,13" 2>&1 | grep -v amdgpu.ids | tail -3
