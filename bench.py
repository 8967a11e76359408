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


def build_model(variant, max_batch, device, options=None):
    from poco_amd import synth
    from poco_amd.model import POCO
    m = POCO(backbone=variant, num_flow_layers=FLOW_LAYERS[variant], max_batch=max_batch, smpl=synth.synth_smpl(7),
             device=device, keep_state_dict=False, engine_options=options)
    w = synth.synth_state_dict(load_spec(variant), 0)
    m.load_state_dict({k: v for k, v in w.items() if v.dtype != np.int64}, strict=True)
    return m.finalize()


def csrc_digest() -> str:
    """sha256 over the kernel sources (what a committed rocprofv3 profile must have been taken from)."""
    import hashlib
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


def pmc_traffic(variant, B):
    """HBM bytes per forward from the committed rocprofv3 --pmc passes of this same command
    (profiles/r*_pmc_*_summary.json written by tools/refresh_profiles.sh + tools/pmc_summary.py: FETCH_SIZE x2
    [gfx950 correction for 16 B/lane streams, MI355X_MICROARCH.md] + WRITE_SIZE, KB -> bytes, / forwards profiled).
    bench.py cannot run the counters itself.  Returns (bytes or None, note): None when no profile matches this
    workload OR when the profile was taken from different kernel sources than the ones in the tree (stale)."""
    tag = {"hrnet_w48_cls-cliff": "w48cliff", "resnet50-cliff": "resnet50cliff", "hrnet_w32-pare": "w32pare"}[variant]
    cands = sorted((ROOT / "profiles").glob(f"r*_pmc_{tag}_b{B}_summary.json"))
    if not cands:
        return None, "no committed PMC profile for this workload"
    d = json.loads(cands[-1].read_text())
    meta = d.get("_meta", {})
    if meta.get("csrc_sha") != csrc_digest():
        return None, (f"{cands[-1].name} is stale: taken from kernel sources {meta.get('csrc_sha', 'unrecorded')}, tree has "
                      f"{csrc_digest()} (re-run tools/refresh_profiles.sh on the GPU box)")
    fw = float(meta.get("forwards", 5))
    tot = sum((2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) for k, v in d.items() if not k.startswith("_"))
    return round(tot * 1024.0 / fw), f"{cands[-1].name} (same kernel sources, {int(fw)} forwards per PMC pass)"


def pmc_executed(variant, B):
    """What the committed PMC pass of this workload says the kernels EXECUTE per forward: MFMA flops (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512,
    MI355X_MICROARCH.md) and MFMA-pipe busy cycles per SIMD, whole forward and per kernel symbol.  Both are properties of the
    launches (kernel + tile configuration), not of their timing, so `roofline.frac` = executed flops / THIS run's time / peak;
    it is <= 1 by construction.  Uses the newest committed profile of the workload and says whether it was taken from the kernel
    sources in the tree (`fresh`); a stale profile still prices the bulk of the forward correctly but is labelled."""
    tag = {"hrnet_w48_cls-cliff": "w48cliff", "resnet50-cliff": "resnet50cliff", "hrnet_w32-pare": "w32pare"}[variant]
    cands = sorted((ROOT / "profiles").glob(f"r*_pmc_{tag}_b{B}_summary.json"))
    if not cands:
        return None
    d = json.loads(cands[-1].read_text())
    meta = d.get("_meta", {})
    fw = float(meta.get("forwards", 5))
    per = {k: v for k, v in d.items() if not k.startswith("_")}
    return {"file": cands[-1].name, "fresh": meta.get("csrc_sha") == csrc_digest(), "forwards": fw,
            "gflop_per_forward": sum(v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) for v in per.values()) * 512.0 / 1e9 / fw,
            "busy_per_simd": sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in per.values()) / fw / 1024.0,
            "per_symbol": {k: (v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0 / 1e9 / max(v.get("dispatches", 1), 1))
                           for k, v in per.items()}}


# ------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (torch-CPU restatement of the reference path) on this host's cores
# ------------------------------------------------------------------------------------------------------------
def _oracle_setup(variant, B):
    from oracle import poco_ref
    from poco_amd import synth
    w = synth.synth_state_dict(load_spec(variant), 0)
    sd = poco_ref.to_torch({k: v for k, v in w.items() if v.dtype != np.int64})
    smpl = poco_ref.to_torch(synth.synth_smpl(7))
    batch = poco_ref.to_torch(synth.synth_batch(B, 1234))
    return lambda: poco_ref.poco_forward(variant, sd, smpl, batch)


def cpu_worker(argv):
    """One instance of the multi-instance leg: pinned to its own cores, warm-up, handshake, timed passes."""
    variant, idx, threads, crops, passes, first_cpu = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), int(argv[4]), int(argv[5])
    try:
        os.sched_setaffinity(0, set(range(first_cpu, first_cpu + threads)))
    except OSError:
        pass
    torch.set_num_threads(threads)
    fwd = _oracle_setup(variant, crops)
    fwd()
    print("READY", flush=True)
    sys.stdin.readline()
    t0 = time.time()
    for _ in range(passes):
        fwd()
    print(f"DONE {time.time() - t0:.4f}", flush=True)


def cpu_baseline(variant, B=64, passes=3, inst_threads=32):
    """the CPU restatement of the reference path on the host cores, B crops per pass, one warm-up + `passes`
    timed passes (median) for one instance on 8 threads (comparable with the survey's provisional numbers), 16 and 32 threads
    (where torch's CPU convolutions scale best), all physical cores (bounded: they collapse on very wide pools), and - what
    the node's CPUs can do together on this workload - one pinned `inst_threads`-thread instance per `inst_threads` physical
    cores, each working on its own share of B crops at the same time.  `value` is the best of these."""
    import subprocess
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        phys = os.cpu_count() or 1
    logical = os.cpu_count() or 1
    runs = []
    fwd = _oracle_setup(variant, B)
    for th in sorted({min(8, phys), min(16, phys), min(32, phys), phys}):
        torch.set_num_threads(th)
        t0 = time.time()
        fwd()
        warm = time.time() - t0
        npass = passes if warm < 6.0 else 0                  # bound the leg: a pool that wide is slow, its warm-up pass is the sample
        ts = [warm] if npass == 0 else []
        for _ in range(npass):
            t0 = time.time()
            fwd()
            ts.append(time.time() - t0)
        runs.append({"instances": 1, "threads_per_instance": th, "crops_per_pass": B, "passes": max(npass, 1),
                     "crops_per_s": round(B / float(np.median(ts)), 2)})
    legs = []
    for T in sorted({inst_threads, 16}, reverse=True):       # 4 x 32 and 8 x 16 on a 128-core host
        P = max(1, phys // T)
        if P < 2:
            continue
        per = max(4, B // P)                     # the B crops are shared out: P instances x B/P crops at the same time
        procs = [subprocess.Popen([sys.executable, str(Path(__file__).resolve()), "--cpu-worker", variant, str(i), str(T),
                                   str(per), "2", str(i * T)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, text=True, env=dict(os.environ, OMP_NUM_THREADS=str(T)))
                 for i in range(P)]
        ok = all(p.stdout.readline().strip() == "READY" for p in procs)
        if ok:
            t0 = time.time()
            for p in procs:
                p.stdin.write("GO\n"); p.stdin.flush()
            done = [p.stdout.readline() for p in procs]
            wall = time.time() - t0
            ok = all(d.startswith("DONE") for d in done)
        for p in procs:
            try:
                p.wait(timeout=60)
            except subprocess.TimeoutExpired:
                p.kill()
        if ok:
            legs.append(f"{P} pinned instances x {T} threads")
            runs.append({"instances": P, "threads_per_instance": T, "crops_per_pass": per * P, "passes": 2,
                         "pinned": f"instance i on cpus [{T}i, {T}i+{T})",
                         "crops_per_s": round(2 * per * P / wall, 2)})
    best = max(runs, key=lambda r: r["crops_per_s"])
    return {"value": best["crops_per_s"], "unit": "crops/s", "cores": best["instances"] * best["threads_per_instance"],
            "kind": "port", "host": {"physical_cores": phys, "logical_cpus": logical}, "runs": runs,
            "sample": f"oracle/poco_ref.py (torch CPU fp32) on {B} crops of {variant} per pass, 1 warm-up + up to {passes} timed "
                      f"passes (median); `value` = the best of: 1 instance x 8 / 16 / 32 / {phys} threads, "
                      f"{' and '.join(legs) if legs else 'no multi-instance leg'} sharing the {B} crops (2 passes)"}


STAGE_THREADS = int(os.environ.get("POCO_STAGE_THREADS", "4"))     # host threads copying decoded frames into the pinned ring


def streaming_leg(variant, device, batch=128, people=4, batches=20):
    """BASELINE.json config #5 shape on this GPU (was tools/bench_video.py): synthetic 1080p uint8 frames cross PCIe
    once each (pinned ring filled by 4 staging threads, copy stream), `people` boxes per frame are cropped + normalised on the GPU into the
    resident batch, hipGraph forward at bs=`batch`, 253-float records come back.  Detector / tracker are out of scope
    (boxes are synthetic); friends.mp4 is not in the tree."""
    from poco_amd.stream import CropStream
    H, W = 1080, 1920
    m = build_model(variant, batch, device)
    fpb = batch // people
    cs = CropStream(m, (H, W), batch, ring=2 * fpb)
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(4)]
    boxes = np.stack([np.array([rng.uniform(0.2, 0.8) * W, rng.uniform(0.3, 0.7) * H, s, s], np.float32)
                      for s in rng.uniform(150, 600, people)])

    def one(i, buf):
        slots = cs.upload_many([frames[(i * fpb + k) % len(frames)] for k in range(fpb)], threads=STAGE_THREADS)
        return cs.run([(k, boxes) for k in slots], buf)

    for i in range(3):
        one(i, i & 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = None
    for i in range(batches):
        h, _ = one(i, i & 1)
        ev = torch.cuda.Event()
        ev.record()
        if done is not None:
            done[0].synchronize()                     # consume the previous batch's records while this one runs
            cs.check()                                # (poco_status: a timed-out in-kernel wait would make them invalid)
            _ = float(done[1][0, 0])
        done = (ev, h)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del cs, m
    return {"workload": f"{variant} streaming: {H}x{W} uint8 frames over PCIe, {people} people/frame, GPU crop+normalise, "
                        f"hipGraph forward bs={batch}, 254-float records back (BASELINE config #5 shape, 1 GPU, synthetic frames)",
            "frames_per_s": round(batches * fpb / dt, 1), "crops_per_s": round(batches * batch / dt, 1),
            "ms_per_batch": round(dt / batches * 1e3, 2), "pcie_in_MB_per_batch": round(fpb * H * W * 3 / 1e6, 1)}


def streaming_cpu_reference(variant, people=4, frames=3, threads=(16, 32)):
    """BASELINE config #5's "wall-clock FPS vs CPU reference": the reference's per-frame pipeline on this host's cores - per detection
    cv2.getAffineTransform + cv2.warpAffine + ToTensor + Normalize (pocolib/core/tester.py:182-203 via vibe_image_utils.py:58-107,
    restated in oracle/crop_np.py), then ONE forward per frame on that frame's detections (tester.py:205-213), restated in
    oracle/poco_ref.py - on the same synthetic 1080p frames / boxes as the GPU leg.  Bounded sample: 1 warm-up frame + `frames` timed
    frames per thread count; the best thread count is reported.  The oracle is the checker / baseline here, never the product path."""
    from oracle import crop_np, poco_ref
    from poco_amd import synth
    from poco_amd.tester import calculate_bbox_info, calculate_focal_length
    H, W = 1080, 1920
    w = synth.synth_state_dict(load_spec(variant), 0)
    sd = poco_ref.to_torch({k: v for k, v in w.items() if v.dtype != np.int64})
    smpl = poco_ref.to_torch(synth.synth_smpl(7))
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(2)]
    boxes = np.stack([np.array([rng.uniform(0.2, 0.8) * W, rng.uniform(0.3, 0.7) * H, s, s], np.float32)
                      for s in rng.uniform(150, 600, people)])

    def one_frame(img):
        crops = crop_np.crop_normalize_np(img, boxes)                             # [people, 3, 224, 224] float32
        b = {"img": crops}
        if variant.endswith("cliff"):
            shp = np.tile(np.array([[H, W]], np.float32), (people, 1))
            fl = np.full(people, calculate_focal_length(H, W), np.float32)
            b.update(bbox_info=np.stack([calculate_bbox_info(boxes[i, :2], boxes[i, 2] / 200.0, (H, W)) for i in range(people)]), focal_length=fl,
                     scale=(boxes[:, 2] / 200.0).astype(np.float32), center=boxes[:, :2].astype(np.float32), orig_shape=shp)
        return poco_ref.poco_forward(variant, sd, smpl, poco_ref.to_torch(b))

    runs = []
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        phys = os.cpu_count() or 1
    for th in sorted({min(t, phys) for t in threads}):
        torch.set_num_threads(th)
        one_frame(imgs[0])
        t0 = time.time()
        for i in range(frames):
            one_frame(imgs[i & 1])
        dt = time.time() - t0
        runs.append({"threads": th, "frames_per_s": round(frames / dt, 3), "crops_per_s": round(frames * people / dt, 2)})
    best = max(runs, key=lambda r: r["frames_per_s"])
    return {"frames_per_s": best["frames_per_s"], "crops_per_s": best["crops_per_s"], "threads": best["threads"], "kind": "port",
            "runs": runs, "sample": f"oracle/crop_np.py (OpenCV 4.5.5 fixed-point warpAffine restated) + oracle/poco_ref.py, {people} "
                                    f"detections per 1080p frame, one forward per frame as tester.py:182-213, 1 warm-up + {frames} timed frames"}


HBM_PEAK_TBS = 8.0      # MI355X_MICROARCH.md: HBM3E ~8 TB/s (about 6.3 achievable with a streaming kernel)


def _time_launches(fn, iters=50, warm=5):
    """mean microseconds of `fn()` (which only enqueues kernels on torch's current stream) over `iters` back-to-back calls."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream()
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def side_kernels(device):
    """the HBM/latency-bound side kernels are reported in GB/s against their ALGORITHMIC bytes (the minimal HBM
    traffic of the operation: every input and parameter read once, every output written once), at the BASELINE sizes
    (VERDICT r2 next #4): part-attention pool (config #2: B = 32, 56x56, C = 128 and 64), SMPL-LBS (B = 64), the GPU crop
    (config #5: 128 crops of one 1080p frame) and the RealNVP flow (config #3: N = B*24 = 1536 and 3072 rows, 2 and 6 coupling
    layers, log_prob and forward_p, with a per-row context and with the reference's per-crop context repeated 24x).  Each
    entry: mean microseconds per call over back-to-back launches between two HIP events on the launch stream."""
    import ctypes as C
    from poco_amd import synth
    from poco_amd._lib import check, current_stream, fptr, lib
    from poco_amd.model import POCO
    from poco_amd.tester import crop_normalize
    L = lib()
    out = {}

    def entry(us, nbytes, note):
        gbs = nbytes / (us * 1e-6) / 1e9
        return {"us": round(us, 2), "algorithmic_MB": round(nbytes / 1e6, 3), "GB_per_s": round(gbs, 1),
                "frac_of_hbm_peak": round(gbs / (HBM_PEAK_TBS * 1e3), 4), "note": note}

    # -- part-attention pool (keypoint_attention.py:34-48 as pare_head.py:794-796 calls it) ----------------------------
    B, H, W = 32, 56, 56
    g = torch.Generator(device=device).manual_seed(1)
    heat = torch.randn((B, H, 2, W, 16), device=device, generator=g)
    L.poco_bench_part_attention.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_int, C.POINTER(C.c_float), C.c_void_p]
    for Cc in (128, 64):
        feat = torch.randn((B, H, Cc // 16, W, 16), device=device, generator=g)
        dst = torch.empty((B, Cc, 24), device=device)
        ms = C.c_float(0)
        check(L.poco_bench_part_attention(fptr(heat), 32, fptr(feat), Cc, B, H, W, fptr(dst), 50, C.byref(ms), current_stream()),
              "poco_bench_part_attention")
        out[f"part_attention_B{B}_C{Cc}"] = entry(ms.value * 1e3, B * (H * W * (32 + Cc) + Cc * 24) * 4,
                                                  f"{B} crops x 56x56 x ({Cc} features + 32 heat-map channels) read once, [B,{Cc},24] written")
    del heat

    # -- engine-bound operators: SMPL-LBS and the flow (resnet50-cliff shell, 3 flow blocks = 6 coupling layers) ------
    def shell(nfl):
        m = POCO(backbone="resnet50-cliff", num_flow_layers=nfl, max_batch=64, smpl=synth.synth_smpl(7), device=device,
                 keep_state_dict=False, engine_options={"flow_ctx_rows": 3072})      # per-row contexts at N = 3072 below
        spec = [(n, shp) for n, shp, _ in m.expected_tensors() if not n.startswith("smpl.")]
        w = synth.synth_state_dict(spec, 0)
        m.load_state_dict({k: v for k, v in w.items() if v.dtype != np.int64}, strict=True)
        return m.finalize()

    m6 = shell(3)
    Bl = 64
    r = np.random.default_rng(5)
    betas = torch.from_numpy(r.standard_normal((Bl, 10)).astype(np.float32)).to(device)
    R = torch.linalg.qr(torch.randn(Bl * 24, 3, 3, generator=torch.Generator().manual_seed(2)))[0].reshape(Bl, 24, 3, 3).to(device)
    verts = torch.empty(Bl, 6890, 3, device=device)
    j49 = torch.empty(Bl, 49, 3, device=device)
    L.poco_smpl_lbs.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5

    def lbs():
        check(L.poco_smpl_lbs(m6._h, Bl, betas.data_ptr(), R.data_ptr(), verts.data_ptr(), j49.data_ptr(), current_stream()), "poco_smpl_lbs")

    V = 6890
    model_bytes = (207 * V * 3 + 10 * V * 3 + V * 3 + V * 24) * 4
    out[f"smpl_lbs_B{Bl}"] = entry(_time_launches(lbs), model_bytes + Bl * (V * 3 + 49 * 3 + 216 + 10) * 4,
                                   f"body model ({model_bytes/1e6:.1f} MB: posedirs, shapedirs, template, skinning weights) read once + "
                                   f"{Bl} x (6890x3 vertices + 49 joints) written")

    # -- GPU crop + normalise (vibe_image_utils.py:94-107,233-266) ---------------------------------------------------
    Hf, Wf, Nc = 1080, 1920, 128
    frame = torch.from_numpy(r.integers(0, 256, (Hf, Wf, 3), dtype=np.uint8)).to(device)
    side = r.uniform(150, 600, Nc)
    boxes = torch.from_numpy(np.stack([r.uniform(0.2, 0.8, Nc) * Wf, r.uniform(0.3, 0.7, Nc) * Hf, side, side], 1).astype(np.float32)).to(device)
    crops = torch.empty(Nc, 3, 224, 224, device=device)
    L.poco_crop_normalize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]

    def crop():
        check(L.poco_crop_normalize(frame.data_ptr(), Hf, Wf, boxes.data_ptr(), Nc, 1.0, 224, crops.data_ptr(), current_stream()), "poco_crop_normalize")

    out[f"crop_normalize_{Nc}x1080p"] = entry(_time_launches(crop), Hf * Wf * 3 + Nc * 3 * 224 * 224 * 4,
                                               f"one 1080p uint8 frame read once + {Nc} x [3,224,224] fp32 crops written "
                                               "(byte-exact cv2 fixed-point warpAffine)")
    del frame, crops

    # -- RealNVP (real_nvp.py:25-65; nf_head.py:93-110: N = B*24 rows, context of a crop repeated for its 24 joints) --
    L.poco_realnvp_rep.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    m2 = shell(1)
    for tag, m, layers in (("L2", m2, 2), ("L6", m6, 6)):
        wbytes = layers * 2 * (521 * 64 + 64 + 64 * 64 + 64 + 9 * 64 + 9) * 4
        for N in (1536, 3072):
            x = torch.from_numpy(np.abs(r.standard_normal((N, 9))).astype(np.float32)).to(device)
            crop_ctx = torch.from_numpy(r.standard_normal((N // 24, 512)).astype(np.float32)).to(device)
            row_ctx = crop_ctx.repeat_interleave(24, 0).contiguous()
            for rep, cx, rtag in ((24, crop_ctx, "ctx_per_crop"), (1, row_ctx, "ctx_per_row")):
                for fwd, nm, ow in ((0, "log_prob", 1), (1, "forward_p", 9)):
                    o = torch.empty(N * ow, device=device)

                    def flow(m=m, x=x, cx=cx, o=o, fwd=fwd, N=N, rep=rep):
                        check(L.poco_realnvp_rep(m._h, N, x.data_ptr(), cx.data_ptr(), rep, o.data_ptr(), fwd, current_stream()), "poco_realnvp")

                    nb = N * (9 + ow) * 4 + (N // rep) * 512 * 4 + wbytes
                    gf = 2.0 * (N // rep) * 512 * layers * 128 + 2.0 * N * layers * 2 * (16 * 64 + 64 * 64 + 64 * 16)
                    e = entry(_time_launches(flow), nb,
                              f"{N} rows x 9 + {N // rep} context rows x 512 read, {layers} coupling layers' weights ({wbytes/1e3:.0f} KB) "
                              f"read once, {ow} float(s) per row written; 2 launches (context GEMM + MFMA coupling kernel)")
                    e["gflop"] = round(gf / 1e9, 3)
                    e["tflops_f32_mfma"] = round(gf / (e["us"] * 1e-6) / 1e12, 2)
                    out[f"realnvp_{nm}_{tag}_N{N}_{rtag}"] = e
    del m2, m6
    return out


def small_batch_leg(model, variant, device, sizes=(1, 4, 16), steps=30):
    """the latency regime.  hipGraph replay of the forward at B = 1 / 4 / 16 on the SAME engine (the tuned
    table has entries for these batch sizes; a photo folder with one person per image used to run here - the folder mode now
    batches across images, poco_amd/tester.py:iter_frame_results)."""
    from poco_amd import synth
    out = {}
    for b in sizes:
        if b > model.max_batch:
            continue
        batch = {k: torch.from_numpy(v).to(device) for k, v in synth.synth_batch(b, 77).items()}
        o = model._alloc_outputs(b, want_segm=False)
        for _ in range(5):
            model.graph_forward(batch, o)
        us = _time_launches(lambda: model.graph_forward(batch, o), iters=steps, warm=3)
        out[f"B{b}"] = {"ms_per_forward": round(us / 1e3, 4), "crops_per_s": round(b / (us * 1e-6), 1)}
    return out


def roofline_block(variant, B, flops_per_crop, ev_ms):
    """`roofline` of a forward that took ev_ms (mean HIP-event time on the launch stream).  frac = EXECUTED fp32-MFMA flops per forward
    (PMC: SQ_INSTS_VALU_MFMA_MOPS_F32 of the committed rocprofv3 pass of this workload - Winograd kernels execute 2.25x / 4x fewer
    MFMAs than the direct-convolution count, padding lanes and masked tiles are included) / time / 157.3 TFLOP/s: a bound, <= 1.
    algorithmic_frac = direct-convolution flop count of the model / time / peak: what the work is worth, may exceed 1."""
    alg = flops_per_crop * B / (ev_ms * 1e-3) / 1e12
    ex = pmc_executed(variant, B)
    traffic, traffic_note = pmc_traffic(variant, B)
    r = {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
         "algorithmic": round(alg, 2), "algorithmic_frac": round(alg / PEAK_F32_MFMA_TFLOPS, 4),
         "traffic": traffic, "traffic_source": traffic_note}
    if ex:
        ach = ex["gflop_per_forward"] / (ev_ms * 1e-3) / 1e3
        r.update({"achieved": round(ach, 2), "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                  "frac_kind": "EXECUTED fp32-MFMA flops (PMC) / this run's forward time / peak; `algorithmic_frac` prices the same time at "
                               "the direct-convolution flop count (Winograd F(2x2) / F(4x4) execute 2.25x / 4x fewer MFMAs)",
                  "executed_gflop_per_forward": round(ex["gflop_per_forward"], 1),
                  "executed_source": f"{ex['file']}: SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 / {int(ex['forwards'])} forwards"
                                     + ("" if ex["fresh"] else " [STALE: taken from other kernel sources than the tree's - re-run tools/refresh_profiles.sh]"),
                  "mfma_pipe_busy": {"frac_at_2.4GHz": round(ex["busy_per_simd"] / (ev_ms * 1e-3 * 2.4e9), 4),
                                     "busy_cycles_per_simd_per_forward": round(ex["busy_per_simd"]),
                                     "source": "SQ_VALU_MFMA_BUSY_CYCLES of the same PMC pass (single lane) / 1024 SIMDs / (this run's mean forward time x 2.4 GHz)"}})
    else:
        r.update({"achieved": None, "frac": None, "frac_kind": "absent: no committed PMC profile for this workload (profiles/r*_pmc_*_summary.json)"})
    r["note"] = (f"whole forward: {B} crops / mean HIP-event forward time {ev_ms:.3f} ms on the launch stream; algorithmic "
                 f"{flops_per_crop/1e9:.3f} GFLOP/crop (SURVEY 8(d)); MFMA conv kernels are >96% of the kernel time (profiles/)")
    return r, ex


def dominant_kernel(model, batch, B, lanes, ex):
    """per-kernel view (HIP events around every op, launched back to back on ONE stream so kernels do not overlap): group the conv
    ops by the kernel symbol they launch, pick the symbol with the most time; both fractions for it."""
    from collections import defaultdict
    model.set_num_lanes(1)
    prof = model.profile_ops(batch, iters=5)
    model.set_num_lanes(lanes)
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for i, (nm, fl, ty, ms) in enumerate(prof):
        d = model.conv_desc(i)
        if d is None:
            continue
        a = agg[model.kernel_symbol(d, model.conv_cfg(i, B))]
        a[0] += 1; a[1] += ms; a[2] += fl * B
    sym, (n, ms, fl) = max(agg.items(), key=lambda kv: kv[1][1])
    dom = {"kernel": sym, "launches_per_step": n, "avg_us": round(ms / n * 1e3, 2),
           "share_of_kernel_time": round(ms / sum(p[3] for p in prof), 3),
           "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3), "algorithmic_tflops": round(fl / (ms * 1e-3) / 1e12, 2),
           "algorithmic_frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
    if ex:
        hit = [v for k, v in ex["per_symbol"].items() if k.startswith(sym)]
        if hit:
            dom["executed_gflop_per_launch"] = round(hit[0], 3)
            dom["frac"] = round(hit[0] / (ms / n * 1e-3) / 1e3 / PEAK_F32_MFMA_TFLOPS, 4)
    return dom


def variant_leg(variant, B, device, steps=30, warmup=10):
    """One more BASELINE config in the same driver line: hipGraph replay of `variant` at B crops on this GPU,
    the same timed-region rules as the headline (barrier-free single GPU: synchronize, W warm-ups, K steps between HIP events)."""
    from poco_amd import synth
    m = build_model(variant, B, device)
    batch = {k: torch.from_numpy(v).to(device) for k, v in synth.synth_batch(B, 1234).items()}
    # every tensor POCO.forward returns is written inside the timed region - for PARE that includes pred_segm_mask [B,25,56,56]
    # (poco.py:99-129 via pare_head.py:669-752; VERDICT r5 weak #8: the leg used to leave it out without saying so)
    segm = variant.endswith("-pare")
    out = m._alloc_outputs(B, want_segm=segm)
    for _ in range(warmup):
        m.graph_forward(batch, out)
    st = torch.cuda.current_stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record(st)
    for i in range(steps):
        m.graph_forward(batch, out)
        ev[i + 1].record(st)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    m.check_status()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    fpc = sum(f for _, f, _ in m.ops())
    roof, ex = roofline_block(variant, B, fpc, float(np.mean(ms)))
    roof["dominant"] = dominant_kernel(m, batch, B, 4, ex)
    res = {"workload": f"{variant} forward, {B} crops of 224x224, fp32 MFMA, hipGraph replay, all reference outputs written"
                       + (" incl. pred_segm_mask" if segm else ""), "value": round(B * steps / wall, 2),
           "unit": "crops/s", "steps": steps, "warmup": warmup, "ms_per_step": round(wall / steps * 1e3, 4),
           "step_ms_events": {"mean": round(float(np.mean(ms)), 4), "median": round(float(np.median(ms)), 4)},
           "roofline": roof}
    del m
    return res


def summary_of(line) -> dict:
    """Every BASELINE config's number in < 600 characters, as the LAST key of the JSON line (a record that keeps
    only the tail of the line lost the ResNet-50 / small-batch numbers behind the verbose baseline legs)."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    r = line.get("roofline", {})
    s = {"cps": line.get("value"), "ms": line.get("ms_per_step"), "n_gpus": line.get("n_gpus"), "frac": r.get("frac"),
         "alg_frac": r.get("algorithmic_frac"), "dom": [g(r, "dominant", "kernel"), g(r, "dominant", "avg_us"), g(r, "dominant", "frac")]}
    for key, tag in (("resnet50-cliff_b64", "r50_b64"), ("hrnet_w32-pare_b32", "pare_b32")):
        v = g(line, "variants", key)
        if v:
            s[tag] = [v.get("value"), g(v, "roofline", "frac"), g(v, "roofline", "algorithmic_frac")]
    sb = line.get("small_batch") or {}
    if sb:
        s["ms_b1_4_16"] = [g(sb, f"B{b}", "ms_per_forward") for b in (1, 4, 16)]
        s["cps_b16"] = g(sb, "B16", "crops_per_s")
    st = line.get("streaming_cfg5") or {}
    if st:
        s["stream"] = {"cps": st.get("crops_per_s"), "fps": st.get("frames_per_s"), "cpu_fps": st.get("cpu_reference_fps")}
    cb = line.get("cpu_baseline") or {}
    if cb:
        s["cpu"] = [cb.get("value"), cb.get("cores")]
    return {k: v for k, v in s.items() if v is not None}


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
    ap.add_argument("--no-stream", action="store_true", help="skip the config-#5 streaming leg (`streaming_cfg5`)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--split-f16", action="store_true",
                    help="EXPERIMENT (never the headline): every plain 1x1 conv on the split-fp16 GEMM (fp16 hi + lo, 3 MFMAs per "
                         "product, csrc/exp/gemm1x1h.hip); the line is labelled and its dtype says so.  Needs the experiment build: "
                         "python -m poco_amd.build --experiments; POCO_HIP_LIB=poco_amd/lib/exp/libpoco_hip_experiments.so")
    ap.add_argument("--no-variants", action="store_true", help="skip the `variants` block (resnet50-cliff bs=64, hrnet_w32-pare bs=32)")
    ap.add_argument("--no-side", action="store_true", help="skip the `side_kernels` block (HBM/latency-bound kernels in GB/s)")
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
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2:])
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
    model = build_model(args.variant, B, device, options={"split_f16": 1} if args.split_f16 else None)
    model.set_num_lanes(args.lanes)
    batch = {k: torch.from_numpy(v).to(device) for k, v in synth.synth_batch(B, 1234 + rank).items()}
    out = model._alloc_outputs(B, want_segm=False)
    from poco_amd import dist as pdist
    gathered = torch.empty(world * B, pdist.REC, device=device) if world > 1 else None
    flops_per_crop = sum(f for _, f, _ in model.ops())
    checked = []            # the RCCL buffers are validated once, before the first collective

    def step():
        # N > 1: the step is the graph replay + ONE collective on the engine's own record buffer (poco_outputs_t.record is written
        # by a kernel inside the graph) - no eager packing launches between them
        if args.no_graph:
            model(batch, out=out)
        else:
            model.graph_forward(batch, out)
        if world > 1 and not args.no_gather:
            rec = out["record"]
            if args.backend == "nccl":
                if not checked:
                    pdist.check_collective_buffers(rec, gathered, world, "nccl", device)
                    checked.append(True)
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
    model.check_status()       # poco_status: a timed-out in-kernel wait in any of the timed forwards would void the number (raises)
    step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    ev_ms = float(np.mean(step_ms))
    ev_med = float(np.median(step_ms))

    gather_check = None
    if world > 1 and args.check_gather and not args.no_gather:
        mine = pdist.pack_records(out, head=args.variant)       # the host-side reference packing: must be what the engine wrote
        ok = bool(torch.equal(gathered[rank * B:(rank + 1) * B, :253], mine[:, :253])) and \
            bool((gathered[rank * B:(rank + 1) * B, 253] - mine[:, 253]).abs().max() <= 1e-6)
        if rank == 0:
            for r in range(1, world):       # same device type, deterministic kernels: bitwise equality is expected
                br = {k: torch.from_numpy(v).to(device) for k, v in synth.synth_batch(B, 1234 + r).items()}
                o = model(br)
                ok = ok and bool(torch.equal(gathered[r * B:(r + 1) * B], o["record"]))
        flag = torch.tensor([1.0 if ok else 0.0], device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_check = bool(flag.item() == 1.0)

    if rank == 0:
        value = world * B * args.steps / elapsed
        roof, ex = roofline_block(args.variant, B, flops_per_crop, ev_ms)
        if world == 1 and not args.no_dominant:
            roof["dominant"] = dominant_kernel(model, batch, B, args.lanes, ex)
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
            "roofline": roof,
        }
        if args.split_f16:
            line["metric"] += " [EXPERIMENT: 1x1 convs in split fp16]"
            line["dtype"] = "f32 except the plain 1x1 convs: fp16 hi + lo split, 3 v_mfma_f32_16x16x32_f16 per product, fp32 accumulation (experiment)"
            line["experiment"] = ("not the headline; narrower arithmetic than the reference's fp32 in the 1x1 convs (22 mantissa "
                                  "bits per operand), gated by the stress fixtures at 1e-3 (tests/test_model_gpu.py::test_split_f16_experiment_passes_the_gate); "
                                  "`roofline` still prices the run against the fp32-MFMA peak")
        if gather_check is not None:
            line["dist"]["gather_check"] = ("ok: all gathered rows bitwise equal to the per-rank forwards" if gather_check
                                            else "FAILED")
        if world == 1 and not args.no_side:
            line["small_batch"] = small_batch_leg(model, args.variant, device)
            line["side_kernels"] = side_kernels(device)
        if world == 1 and not args.no_variants and args.variant == "hrnet_w48_cls-cliff" and B == 64:
            # BASELINE configs #3 (ResNet-50 wording) and #2 in the same driver line
            line["variants"] = {"resnet50-cliff_b64": variant_leg("resnet50-cliff", 64, device),
                                "hrnet_w32-pare_b32": variant_leg("hrnet_w32-pare", 32, device)}
        if world == 1 and not args.no_stream:
            line["streaming_cfg5"] = streaming_leg(args.variant, device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.variant, B)
            if "streaming_cfg5" in line:
                try:
                    line["streaming_cfg5"]["cpu_reference"] = streaming_cpu_reference(args.variant)
                    line["streaming_cfg5"]["cpu_reference_fps"] = line["streaming_cfg5"]["cpu_reference"]["frames_per_s"]
                    line["streaming_cfg5"]["gpu_over_cpu"] = round(line["streaming_cfg5"]["frames_per_s"] /
                                                                   max(line["streaming_cfg5"]["cpu_reference_fps"], 1e-9), 1)
                except Exception as e:                                     # (the baseline leg must never take the bench line down)
                    line["streaming_cfg5"]["cpu_reference"] = {"error": repr(e)[:200]}
        line["summary"] = summary_of(line)          # LAST key: survives a driver that keeps only the tail of the line
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if gather_check is False:
        raise SystemExit("gathered records differ from the per-rank outputs")


if __name__ == "__main__":
    main()
