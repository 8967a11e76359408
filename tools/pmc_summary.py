"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel family.
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv ...> [out.json]"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        m = re.search(r"(conv_\w+_kernel|conv_wino_kernel|\w+_kernel)", name)
        fam = m.group(1) if m else name[:40]
        if "conv_" in fam:
            fam = "conv(all MFMA variants)"
        c = r["Counter_Name"]
        agg[fam][c] += float(r["Counter_Value"])
        calls[fam][c] += 1
out = {}
for fam, cs in agg.items():
    d = {k: v for k, v in cs.items()}
    d["dispatches"] = max(calls[fam].values())
    out[fam] = d
json.dump(out, open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout, indent=1)
