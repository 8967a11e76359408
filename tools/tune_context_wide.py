"""Wider in-context tuning pass than `python -m poco_amd.tune --in-context`:
    tools/tune_context_wide.py <variant> <batch> <top_shapes> <top_cands> <iters> <out.json>
Re-picks the configuration of the <top_shapes> most expensive conv shapes inside the whole 4-lane forward, trying the
<top_cands> best solo candidates each, and writes the merged table to <out.json> (and to poco_amd/tuned/gfx950.json)."""
import json, sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from pathlib import Path
from poco_amd import tune
v, B = sys.argv[1], int(sys.argv[2])
res = tune.tune_in_context(v, B, top_shapes=int(sys.argv[3]), top_cands=int(sys.argv[4]), iters=int(sys.argv[5]))
full = json.loads(tune.TABLE.read_text())
full.update({k: x for k, x in res.items() if x["cfg"][0] > 0})
Path(sys.argv[6]).write_text(json.dumps(full, indent=0, sort_keys=True))
tune.TABLE.write_text(json.dumps(full, indent=0, sort_keys=True))
