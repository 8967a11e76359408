mkdir -p gpurun_out/r3d
O=gpurun_out/r3d
run() {
python -X faulthandler -c "
import sys; sys.path.insert(0,'.')
import torch
from poco_amd import synth
from tests import util
m=util.make_engine('hrnet_w48_cls-cliff',max_batch=4)
b=util.cuda_batch(synth.synth_batch(4,5),torch.device('cuda:0'))
o=m(b); torch.cuda.synchronize()
out=m._alloc_outputs(4,False)
m.graph_forward(b,out); torch.cuda.synchronize(); print('graph1 ok', flush=True)
" 2>&1 | grep -c "graph1 ok"
}
for mm in 1 2 3 8; do echo "mode2 maxmod=$mm: $(POCO_XDEP_MODE=2 POCO_XDEP_MAXMOD=$mm run)" >> $O/dbg_bisect.log; done
echo "mode2 sumjoin: $(POCO_XDEP_MODE=2 POCO_XDEP_SUMJOIN=1 run)" >> $O/dbg_bisect.log
echo "mode3 sumjoin: $(POCO_XDEP_MODE=3 POCO_XDEP_SUMJOIN=1 run)" >> $O/dbg_bisect.log
