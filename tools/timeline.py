"""Analyse a rocprofv3 --kernel-trace CSV of bench.py: GPU-busy fraction, concurrency, per-stream gaps."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
def _wgs(r):     # workgroups of the launch (the symbol alone does not tell a 128-block 14x14 conv from a 256-block 56x56 one)
    try:
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) * max(1, int(r.get("Grid_Size_Y", 1))) * max(1, int(r.get("Grid_Size_Z", 1)))
        w = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))) * max(1, int(r.get("Workgroup_Size_Y", 1))) * max(1, int(r.get("Workgroup_Size_Z", 1)))
        return g // max(1, w)
    except (TypeError, ValueError):
        return 0


ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48] + f" [{_wgs(r)} wg]",
       r.get("Stream_Id", r.get("Queue_Id", "0"))) for r in rows]
ks.sort()
# last forward = between the last two stem kernels
stems = [i for i, k in enumerate(ks) if "stem_conv" in k[2] or "stem_mfma" in k[2]]
lo, hi = stems[-2], stems[-1]
fw = ks[lo:hi]
t0, t1 = fw[0][0], max(k[1] for k in fw)
print(f"forward {(t1 - t0) / 1e3:.1f} us, {len(fw)} kernels")
ev = []
for s, e, n, q in fw:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = {0: 0, 1: 0, 2: 0, 3: 0, 4: 0}
cur, last = 0, t0
for t, d in ev:
    busy[min(cur, 4)] += t - last
    cur += d; last = t
tot = t1 - t0
print("concurrency histogram (fraction of wall time with k kernels resident):", {k: round(v / tot, 3) for k, v in busy.items()})
per = defaultdict(list)
for s, e, n, q in fw:
    per[q].append((s, e, n))
for q, lst in per.items():
    lst.sort()
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    small = [g for g in gaps if 0 <= g < 20000]
    print(f"queue {q}: {len(lst)} kernels, busy {sum(e - s for s, e, _ in lst) / 1e3:.0f} us, "
          f"median back-to-back gap {sorted(small)[len(small) // 2] / 1e3 if small else -1:.2f} us, sum small gaps {sum(small) / 1e3:.0f} us")
# per-kernel-name average duration in this forward
agg = defaultdict(lambda: [0, 0])
for s, e, n, q in fw:
    a = agg[n.split("(")[0][-40:]]
    a[0] += 1; a[1] += e - s
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {n:42s} n={c:4d} sum {d / 1e3:8.0f} us avg {d / c / 1e3:7.1f} us")

# which kernels run ALONE (concurrency 1): wall time attributed to the single resident kernel, by symbol
solo = defaultdict(float)
active = []
pts = sorted([(s, 0, i) for i, (s, e, n, q) in enumerate(fw)] + [(e, 1, i) for i, (s, e, n, q) in enumerate(fw)])
live, last = set(), t0
for t, kind, i in pts:
    if len(live) == 1:
        solo[fw[next(iter(live))][2][:70]] += t - last
    if kind == 0:
        live.add(i)
    else:
        live.discard(i)
    last = t
print("wall time with exactly one kernel resident, by kernel:")
for n, v in sorted(solo.items(), key=lambda kv: -kv[1])[:22]:
    print(f"  {v / 1e3:8.1f} us  {n}")

# CU demand over time (round 4): a launch with w workgroups wants min(w, 256) CUs (every kernel of the forward that matters keeps
# one workgroup per CU: LDS-bound or persistent); demand < 256 = CUs idle because the resident kernels are too narrow (or absent)
import re as _re
def _wg(n):
    m = _re.search(r"\[(\d+) wg\]", n)
    return min(256, int(m.group(1))) if m else 256
pts = sorted([(s, _wg(n)) for s, e, n, q in fw] + [(e, -_wg(n)) for s, e, n, q in fw])
cur, last = 0, t0
hist = {"0": 0, "1-127": 0, "128-255": 0, "256-383": 0, ">=384": 0}
idle_cu_us = 0.0
for t, dlt in pts:
    dt = t - last
    key = "0" if cur == 0 else "1-127" if cur < 128 else "128-255" if cur < 256 else "256-383" if cur < 384 else ">=384"
    hist[key] += dt
    idle_cu_us += max(0, 256 - cur) * dt / 1e3
    cur += dlt; last = t
print("CU demand of the resident kernels (fraction of wall time):", {k: round(v / tot, 3) for k, v in hist.items()})
print(f"idle CU capacity: {idle_cu_us / 256:.0f} us-equivalents of the whole chip out of {tot / 1e3:.0f} us wall")
cut = defaultdict(float)
for s, e, n, q in fw:
    cut[n[:60]] += (e - s) / 1e3 * _wg(n) / 256
print("CU time by kernel (duration x min(workgroups, 256) / 256, us-equivalents of the whole chip):")
for n, v in sorted(cut.items(), key=lambda kv: -kv[1])[:16]:
    print(f"  {v:8.0f}  {n}")
print(f"  total {sum(cut.values()):.0f}")
