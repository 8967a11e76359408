"""Aggregate rocprofv3 --pmc counter_collection CSVs PER KERNEL SYMBOL (template arguments kept, e.g. `conv_wino4p_kernel<3>`).
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv ...> <out.json> [forwards per PMC pass]

Per symbol: the raw counter sums over all its dispatches, `dispatches`, and the derived ratios the design is steered by
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) / (GRBM_GUI_ACTIVE / 8 XCDs)   (fraction of the kernel's own
               active time in which the MFMA pipe of a SIMD was busy, averaged over SIMDs)
  mfma_gflop = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 / 1e9 (executed, not algorithmic)
  hbm_bytes  = 2 x FETCH_SIZE KB (gfx950 correction for 16 B/lane streams, MI355X_MICROARCH.md) + WRITE_SIZE KB
`_meta` records the kernel-source digest the profile was taken from (bench.py refuses a stale profile for roofline.traffic) and
the number of forwards per pass; `_total` sums every kernel."""
import csv
import glob
import hashlib
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def csrc_digest() -> str:
    h = hashlib.sha256()
    for f in sorted((ROOT / "poco_amd" / "csrc").glob("*")):
        if f.suffix in (".hip", ".h", ".cpp") and f.name != "ops_capi.hip":   # (the stand-alone operators' C wrappers are not on the forward path)
            h.update(f.name.encode())
            # code only: // comments and blank lines do not invalidate a profile
            code = [ln.split("//")[0].rstrip() for ln in f.read_text().splitlines()]
            h.update("\n".join(ln for ln in code if ln.strip()).encode())
    # ... and the tile configurations the launches run with: executed MFMA flops / busy cycles / traffic are properties of
    # (kernel, configuration), and a table-only change must make a committed profile stale too; only the `cfg` lists
    # count, re-measured ms / tflops fields of an unchanged configuration do not
    import json
    table = json.loads((ROOT / "poco_amd" / "tuned" / "gfx950.json").read_text())
    h.update(json.dumps({k: v.get("cfg") for k, v in sorted(table.items())}, sort_keys=True).encode())
    return h.hexdigest()[:16]


def symbol(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"([A-Za-z_]\w*(?:<[^()]*?>)?)\s*\(", name)
    s = m.group(1) if m else name.split("(")[0]
    return s.strip()[:80]


def main():
    root, out = sys.argv[1], sys.argv[2]
    forwards = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            sym = symbol(r.get("Kernel_Name", ""))
            c = r["Counter_Name"]
            agg[sym][c] += float(r["Counter_Value"])
            calls[sym][c] += 1
    res = {}
    tot = defaultdict(float)
    for sym, cs in agg.items():
        d = {k: v for k, v in cs.items()}
        d["dispatches"] = max(calls[sym].values())
        if d.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            d["mfma_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (d["GRBM_GUI_ACTIVE"] / 8.0), 4)
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in d:
            d["mfma_gflop"] = round(d["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / 1e9, 3)
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            d["hbm_bytes"] = round((2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0)
        for k, v in cs.items():
            tot[k] += v
        res[sym] = d
    t = dict(tot)
    if t.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in t:
        t["mfma_busy"] = round(t["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (t["GRBM_GUI_ACTIVE"] / 8.0), 4)
    t["hbm_bytes_per_forward"] = round((2.0 * t.get("FETCH_SIZE", 0.0) + t.get("WRITE_SIZE", 0.0)) * 1024.0 / forwards)
    res = dict(sorted(res.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)))
    res["_total"] = t
    res["_meta"] = {"csrc_sha": csrc_digest(), "forwards": forwards,
                    "note": "counter sums over all dispatches of the profiled command (warm-up + timed forwards)"}
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
