import json, sys
sys.path.insert(0, '/root/repo')
from pathlib import Path
from poco_amd import tune
v, B = sys.argv[1], int(sys.argv[2])
res = tune.tune_in_context(v, B, top_shapes=int(sys.argv[3]), top_cands=int(sys.argv[4]), iters=int(sys.argv[5]))
full = json.loads(tune.TABLE.read_text())
full.update({k: x for k, x in res.items() if x["cfg"][0] > 0})
Path(sys.argv[6]).write_text(json.dumps(full, indent=0, sort_keys=True))
tune.TABLE.write_text(json.dumps(full, indent=0, sort_keys=True))
