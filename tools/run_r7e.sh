cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7e
timeout 1500 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "mosaic or whole_position" > gpurun_out/r7e/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r7e/t1.log
tail -5 gpurun_out/r7e/t1.log
python - <<'P' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0,'.')
import numpy as np, torch
from poco_amd import ops
x=torch.randn(64,14,14,192,device='cuda:0'); w=(np.random.default_rng(0).standard_normal((192,192,3,3))/40).astype(np.float32)
for cfg in [(2,3,2,1,16,2,13),(2,3,2,1,16,0,13),(1,3,2,1,16,0,13),(2,3,2,1,32,0,13),(2,3,2,1,4,0,13)]:
    print(cfg, round(min(ops.bench_conv2d(x,w,1,cfg=cfg,iters=40)[0]*1e3 for _ in range(3)),1))
P
for c in "2,3,2,1,16,0,13" "1,3,2,1,16,0,13" "2,3,2,1,32,0,13"; do
timeout 300 python tools/ab_shape_cfg.py hrnet_w48_cls-cliff 64 14x14x192x192 "2,3,2,1,16,2,13" "$c" 2 2>&1 | grep -v amdgpu.ids | tail -4
done
