"""bench.py - crops/sec of the POCO hot path on MI355X (contract: see DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--variant hrnet_w48_cls-cliff] [--batch 64]

A "step" is one pass of  model(batch)  (backbone -> head -> SMPL-LBS -> confidence MLP) over one
batch of synthetic 224x224 crops that is already resident in HBM, with seeded random-init weights
of the named architecture (the reference's checkpoints and the SMPL model are license-gated).
N>1: one process per GPU (torch.distributed / RCCL), `--batch` crops per GPU (weak scaling), and a
RCCL all-gather of the packed SMPL parameters [pose 216 | betas 10 | cam 3 | var 24 | conf 1] per
step.  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` on its own STARTS the N ranks (it re-executes itself under
`python -m torch.distributed.run --nproc-per-node N`, rendezvous on 127.0.0.1); launched by torchrun
(WORLD_SIZE set) it is one of the ranks.  With the nccl backend it refuses to run with fewer visible
GPUs than ranks.  The line carries `dist` = what the process group itself reports (backend, world size).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL; must be set before HIP initialises

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32 dense peak
FLOW_LAYERS = {"hrnet_w32-pare": 3, "hrnet_w48_cls-cliff": 1, "resnet50-cliff": 1}


def load_spec(variant):
    return [(n, tuple(s)) for n, s in json.loads((ROOT / "tests" / "golden" / f"spec_{variant}.json").read_text())]


def build_model(variant, max_batch, device):
    from poco_amd import synth
    from poco_amd.model import POCO
    m = POCO(backbone=variant, num_flow_layers=FLOW_LAYERS[variant], max_batch=max_batch, smpl=synth.synth_smpl(7),
             device=device)
    w = synth.synth_state_dict(load_spec(variant), 0)
    m.load_state_dict({k: v for k, v in w.items() if v.dtype != np.int64}, strict=True)
    return m.finalize()


def pmc_traffic(variant, B):
    """HBM bytes per forward from the committed rocprofv3 --pmc passes of this same command
    (profiles/r01_pmc_*_summary.json: FETCH_SIZE x2 [gfx950 correction for 16 B/lane streams] + WRITE_SIZE,
    KB -> bytes, / 5 forwards).  bench.py cannot run the counters itself; null if no profile matches."""
    f = ROOT / "profiles" / "r01_pmc_w48cliff_b64_summary.json"
    if variant != "hrnet_w48_cls-cliff" or B != 64 or not f.exists():
        return None
    d = json.loads(f.read_text())
    conv = d.get("conv(all MFMA variants)", {})
    if "FETCH_SIZE" not in conv:
        return None
    tot = sum((2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) for v in d.values())
    return round(tot * 1024.0 / 5.0)


def cpu_baseline(variant, seconds_budget=25.0):
    """The oracle (CPU restatement of the reference path, oracle/poco_ref.py) timed on this host's cores
    on a bounded sample of the same workload.  Reported beside the GPU number; never the product path."""
    from oracle import poco_ref
    from poco_amd import synth
    cores = os.cpu_count() or 1
    w = synth.synth_state_dict(load_spec(variant), 0)
    sd = poco_ref.to_torch({k: v for k, v in w.items() if v.dtype != np.int64})
    smpl = poco_ref.to_torch(synth.synth_smpl(7))
    Bs = 32
    batch = poco_ref.to_torch(synth.synth_batch(Bs, 1234))
    t_start = time.time()
    best = None
    # torch's CPU conv does not scale to every core of a 2-socket host: probe a few thread counts
    for th in sorted({min(cores, c) for c in (16, 32, 64, 128)}):
        if best is not None and time.time() - t_start > 0.5 * seconds_budget:
            break
        torch.set_num_threads(th)
        poco_ref.poco_forward(variant, sd, smpl, batch)      # warm-up at this thread count
        t0 = time.time()
        poco_ref.poco_forward(variant, sd, smpl, batch)
        dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, th)
    threads = best[1]
    torch.set_num_threads(threads)
    times = [best[0]]
    while len(times) < 4 and (time.time() - t_start) < seconds_budget:
        t0 = time.time()
        poco_ref.poco_forward(variant, sd, smpl, batch)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return {"value": round(Bs / med, 2), "unit": "crops/s", "cores": threads, "kind": "port",
            "sample": f"oracle/poco_ref.py (torch CPU fp32, {threads} threads of {cores} logical cpus) on {Bs} crops of "
                      f"{variant}, median of {len(times)} passes at the best of 16/32/64/128 threads"}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this same script, one per GPU."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and ndev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible and the nccl (RCCL) backend needs one "
                         f"per rank (use --backend gloo to exercise the multi-rank path on fewer devices)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--variant", default="hrnet_w48_cls-cliff", choices=list(FLOW_LAYERS))
    ap.add_argument("--batch", type=int, default=64, help="crops per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--lanes", type=int, default=4, help="HIP streams for independent branches (1 = single stream)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay of the forward")
    ap.add_argument("--no-dominant", action="store_true",
                    help="skip the per-op HIP-event pass behind roofline.dominant (use under rocprofv3 so that the "
                         "kernel statistics contain only the warm-up + timed forwards)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only to exercise the multi-rank path on a "
                         "box with fewer GPUs than ranks: ranks then share devices round-robin)")
    ap.add_argument("--check-gather", action="store_true",
                    help="N>1: after the timed region rank 0 recomputes every rank's batch and checks the gathered rows "
                         "bitwise against them (and every rank checks its own rows)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > 1 and world > ndev:
        raise SystemExit(f"rank {rank}: {world} ranks but only {ndev} GPU(s) visible; RCCL needs one GPU per rank")
    device = torch.device(f"cuda:{local_rank % max(ndev, 1)}")
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "collective": "RCCL all_gather_into_tensor over xGMI" if args.backend == "nccl"
                     else "gloo all_gather staged through the host (smoke path)",
                     "devices_visible": ndev}
    else:
        dist_info = {"backend": None, "world_size": 1, "collective": None, "devices_visible": ndev}

    from poco_amd import synth
    B = args.batch
    model = build_model(args.variant, B, device)
    model.set_num_lanes(args.lanes)
    batch = {k: torch.from_numpy(v).to(device) for k, v in synth.synth_batch(B, 1234 + rank).items()}
    out = model._alloc_outputs(B, want_segm=False)
    from poco_amd import dist as pdist
    gathered = torch.empty(world * B, pdist.REC, device=device) if world > 1 else None
    flops_per_crop = sum(f for _, f, _ in model.ops())

    def step():
        if args.no_graph:
            model(batch, out=out)
        else:
            model.graph_forward(batch, out)
        if world > 1 and not args.no_gather:
            rec = pdist.pack_records(out, head=args.variant)
            if args.backend == "nccl":
                dist.all_gather_into_tensor(gathered, rec)          # RCCL over xGMI, device buffers
            else:                                                   # gloo smoke path: staged through the host
                parts = [torch.empty(B, pdist.REC) for _ in range(world)]
                dist.all_gather(parts, rec.cpu())
                gathered.copy_(torch.cat(parts, 0))

    for _ in range(args.warmup):
        step()
    stream = torch.cuda.current_stream()       # the stream the kernels are launched on
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record(stream)
    for i in range(args.steps):
        step()
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    ev_ms = float(np.mean(step_ms))
    ev_med = float(np.median(step_ms))

    gather_check = None
    if world > 1 and args.check_gather and not args.no_gather:
        mine = pdist.pack_records(out, head=args.variant)
        ok = bool(torch.equal(gathered[rank * B:(rank + 1) * B], mine))
        if rank == 0:
            for r in range(1, world):       # same device type, deterministic kernels: bitwise equality is expected
                br = {k: torch.from_numpy(v).to(device) for k, v in synth.synth_batch(B, 1234 + r).items()}
                o = model(br)
                ok = ok and bool(torch.equal(gathered[r * B:(r + 1) * B], pdist.pack_records(o, head=args.variant)))
        flag = torch.tensor([1.0 if ok else 0.0], device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_check = bool(flag.item() == 1.0)

    dominant = None
    if rank == 0 and world == 1 and not args.no_dominant:
        # per-kernel view (HIP events around every op, launched back to back on ONE stream so kernels do not
        # overlap): group the conv ops by the kernel symbol they launch, pick the symbol with the most time
        from collections import defaultdict
        model.set_num_lanes(1)
        prof = model.profile_ops(batch, iters=5)
        model.set_num_lanes(args.lanes)
        agg = defaultdict(lambda: [0, 0.0, 0.0])
        for i, (nm, fl, ty, ms) in enumerate(prof):
            d = model.conv_desc(i)
            if d is None:
                continue
            a = agg[model.kernel_symbol(d, model.conv_cfg(i, B))]
            a[0] += 1; a[1] += ms; a[2] += fl * B
        sym, (n, ms, fl) = max(agg.items(), key=lambda kv: kv[1][1])
        dominant = {"kernel": sym, "launches_per_step": n, "avg_us": round(ms / n * 1e3, 2),
                    "share_of_kernel_time": round(ms / sum(p[3] for p in prof), 3),
                    "gflop_per_launch": round(fl / n / 1e9, 3), "tflops": round(fl / (ms * 1e-3) / 1e12, 2)}

    if rank == 0:
        value = world * B * args.steps / elapsed
        achieved = flops_per_crop * B / (ev_ms * 1e-3) / 1e12
        line = {
            "metric": "person-crops/sec (224x224) POCO-CLIFF bs=64" if args.variant.endswith("cliff")
                      else "person-crops/sec (224x224) POCO-PARE",
            "value": round(value, 2), "unit": "crops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "step_ms_events": {"mean": round(ev_ms, 4), "median": round(ev_med, 4), "min": round(min(step_ms), 4),
                               "max": round(max(step_ms), 4),
                               "crops_per_s_at_median": round(B / (ev_med * 1e-3), 1),
                               "note": "per-step HIP events on rank 0's launch stream (SURVEY 8(d): median of the steps)"},
            "higher_is_better": True, "scaling": "weak", "dist": dist_info,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.variant} forward (backbone+head+SMPL-LBS+confidence MLP), "
                                   f"{B} crops/GPU of 224x224, fp32 MFMA", "variant": args.variant,
                       "crops_per_gpu": B, "global_batch": world * B,
                       "parallelism": f"dp{world} (crop sharding" + (f", {'RCCL' if args.backend == 'nccl' else 'gloo (smoke path)'} all-gather of 254-float SMPL records)"
                                                                          if world > 1 and not args.no_gather else ")")},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": pmc_traffic(args.variant, B),
                         "note": f"whole forward: algorithmic {flops_per_crop/1e9:.3f} GFLOP/crop x {B} crops / mean "
                                 f"HIP-event forward time {ev_ms:.3f} ms on the launch stream (MFMA conv kernels are >96% "
                                 "of the kernel time, profiles/); `dominant` = the kernel symbol with the most time, "
                                 "algorithmic flops of its launches / their HIP-event time on one stream"},
        }
        if gather_check is not None:
            line["dist"]["gather_check"] = ("ok: all gathered rows bitwise equal to the per-rank forwards" if gather_check
                                            else "FAILED")
        if dominant is not None:
            dominant["frac"] = round(dominant["tflops"] / PEAK_F32_MFMA_TFLOPS, 4)
            line["roofline"]["dominant"] = dominant
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.variant)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if gather_check is False:
        raise SystemExit("gathered records differ from the per-rank outputs")


if __name__ == "__main__":
    main()
