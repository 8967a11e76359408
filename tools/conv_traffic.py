"""HBM traffic per launch of one conv shape under several tile configurations (rocprofv3 PMC passes FETCH_SIZE / WRITE_SIZE, separate
passes, no tracing domains mixed in; gfx950 correction 2 x FETCH_SIZE as tools/pmc_summary.py).
  python tools/conv_traffic.py B H W Cin Cout "cfg7;cfg7;..."          (on the GPU box; prints MB per launch and the algorithmic MB)
  python tools/conv_traffic.py --child B H W Cin Cout cfg7 iters          (what rocprofv3 runs)"""
import csv, glob, os, subprocess, sys, tempfile
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R))


def child(a):
    import numpy as np, torch
    from poco_amd import ops
    B, H, W, Cin, Cout = map(int, a[:5])
    cfg = tuple(int(x) for x in a[5].split(","))
    x = torch.randn(B, H, W, Cin, device="cuda:0")
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    ops.bench_conv2d(x, w, 1, cfg=cfg, iters=int(a[6]))


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2:])
    B, H, W, Cin, Cout = map(int, sys.argv[1:6])
    alg_mb = B * H * W * (Cin + Cout) * 4 / 1e6
    print(f"{B}x{H}x{W} {Cin}->{Cout}: algorithmic {alg_mb:.1f} MB per launch (input + output once)")
    for cfg in sys.argv[6].split(";"):
        tot = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(dir="/tmp")
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                            str(Path(__file__).resolve()), "--child", *map(str, (B, H, W, Cin, Cout)), cfg, "6"],
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if "conv_wino4" in r.get("Kernel_Name", ""):
                        t = tot.setdefault(ctr, [0.0, 0])
                        t[0] += float(r["Counter_Value"]); t[1] += 1
        f, wv = tot.get("FETCH_SIZE", [0, 1]), tot.get("WRITE_SIZE", [0, 1])
        rd, wr = 2.0 * f[0] * 1024 / max(f[1], 1) / 1e6, wv[0] * 1024 / max(wv[1], 1) / 1e6
        print(f"  cfg ({cfg}): read {rd:.1f} MB + write {wr:.1f} MB = {rd + wr:.1f} MB per launch = {(rd + wr) / alg_mb:.2f} x algorithmic ({f[1]} launches counted)")


if __name__ == "__main__":
    main()
