cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7a
ALL=1 python tools/oplist.py hrnet_w48_cls-cliff 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r7a/oplist_w48_b64.txt
