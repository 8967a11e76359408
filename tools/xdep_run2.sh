O=gpurun_out/r3e; mkdir -p $O
python -m pytest tests/test_model_gpu.py tests/test_demo_gpu.py tests/test_multirank_gpu.py -q -x -m gpu 2>&1 | tail -4 > $O/parity.log
python tools/stress.py > $O/stress.log 2>&1
python tools/stress.py hrnet_w32-pare 32 100 >> $O/stress.log 2>&1
python -c "
import sys; sys.path.insert(0,'.')
from tests import util
for var,B in (('hrnet_w48_cls-cliff',64),('hrnet_w48_cls-cliff',128),('hrnet_w32-pare',32)):
    m=util.make_engine(var,max_batch=B); print(var,B,'workspace MB',m.workspace_bytes()/1e6, 'ops', len(m.ops()))
" > $O/ws.txt 2>&1
python -m poco_amd.tune --variant hrnet_w48_cls-cliff --batch 64 --in-context --out $O/t.json > $O/tune.log 2>&1
python -m poco_amd.tune --variant hrnet_w32-pare --batch 32 --in-context --out $O/t.json >> $O/tune.log 2>&1
cp poco_amd/tuned/gfx950.json $O/gfx950.json
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('small_batch'))"; }
for rep in 1 2; do
  echo "w48: $(python bench.py --no-cpu-baseline --no-stream 2>/dev/null | val)" >> $O/after.txt
  echo "pare: $(python bench.py --variant hrnet_w32-pare --batch 32 --no-cpu-baseline --no-stream 2>/dev/null | val)" >> $O/after.txt
done
