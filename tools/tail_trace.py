"""Kernel-trace view of the END of a forward (regressor tail): python tools/tail_trace.py kernel_trace.csv [n_last]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 45
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]) for r in rows)
stems = [i for i, k in enumerate(ks) if "stem_conv" in k[2] or "stem_mfma" in k[2]]
fw = ks[stems[-2]:stems[-1]]
t0, t1 = fw[0][0], max(k[1] for k in fw)
print(f"forward {(t1 - t0) / 1e3:.1f} us, {len(fw)} kernels; last {n_last}:")
for s, e, n in fw[-n_last:]:
    print(f"  start {(s - t0) / 1e3:9.1f}  end {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:7.1f} us  {n}")
