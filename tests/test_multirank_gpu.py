"""GPU box (1 device): the N>1 code paths run for real as separate processes - `bench.py --gpus N` and
`demo.py --mode video --gpus N` start their own ranks; with `--backend gloo` the ranks share the one GPU, so everything
except the RCCL transport itself (process group, sharding, packed-record gather, rank-0 merge) is exercised here.
RCCL proper needs >= 2 GPUs: the driver's 8-GPU scaling run (`bench.py --gpus 8`, nccl) is that leg."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from poco_amd import synth
from tests import util

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(cmd, env=None, timeout=900):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable] + cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_gpus2_starts_two_ranks_and_gathers(cuda):
    r = _run(["bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--variant", "resnet50-cliff",
              "--batch", "8", "--check-gather", "--no-stream", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["dist"]["world_size"] == 2 and line["dist"]["backend"] == "gloo"
    assert line["config"]["global_batch"] == 16 and line["scaling"] == "weak"
    assert line["dist"]["gather_check"].startswith("ok"), line["dist"]
    assert sum(ln.startswith("{") for ln in r.stdout.splitlines()) == 1          # ONE json line (rank 0 only)


def test_bench_gpus4_ragged_batch_gathers(cuda):
    """N > 2 ranks and a batch that is no multiple of anything (7 crops per rank)."""
    r = _run(["bench.py", "--gpus", "4", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--variant", "resnet50-cliff",
              "--batch", "7", "--check-gather", "--no-stream", "--no-cpu-baseline", "--no-side"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 4 and line["dist"]["world_size"] == 4 and line["config"]["global_batch"] == 28
    assert line["dist"]["gather_check"].startswith("ok"), line["dist"]
    assert sum(ln.startswith("{") for ln in r.stdout.splitlines()) == 1


def test_bench_config4_shape_eight_ranks_of_64_crops(cuda):
    """VERDICT r4 next #5 / BASELINE config #4 (POCO-CLIFF, 512 crops over 8 ranks): the exact job shape of the driver's scaling run
    - 8 ranks x 64 crops of the headline variant, engine-written records, ONE all_gather_into_tensor per step, every gathered row
    checked against rank 0's own recomputation of the other ranks' batches - rehearsed on the one GPU with the gloo transport
    (eight engines, 8 x 0.9 GB of workspace, share the 288 GB).  Only the RCCL transport itself is left to the 8-GPU node."""
    r = _run(["bench.py", "--gpus", "8", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "64", "--check-gather",
              "--no-stream", "--no-cpu-baseline", "--no-side", "--no-variants", "--no-dominant"], timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 8 and line["dist"]["world_size"] == 8 and line["config"]["global_batch"] == 512
    assert line["config"]["variant"] == "hrnet_w48_cls-cliff" and line["config"]["crops_per_gpu"] == 64
    assert line["dist"]["gather_check"].startswith("ok"), line["dist"]
    assert line["scaling"] == "weak" and line["summary"]["n_gpus"] == 8
    assert sum(ln.startswith("{") for ln in r.stdout.splitlines()) == 1


def test_rccl_buffers_dry_check(cuda):
    """The nccl branch of bench.py hands RCCL exactly these buffers: validated here for an 8-rank job without a second GPU
    (device tensors of this rank's own device, float32, contiguous, recv = world x send), and the check refuses host or
    mis-shaped buffers - so that the first real `--gpus 8` run can only fail on RCCL itself."""
    from poco_amd import dist as pdist
    m = util.make_engine("resnet50-cliff", max_batch=4)
    out = m(util.cuda_batch(synth.synth_batch(4, 1), cuda))
    rec = pdist.pack_records(out, head="resnet50-cliff")
    world = 8
    gathered = torch.empty(world * 4, pdist.REC, device=cuda)
    pdist.check_collective_buffers(rec, gathered, world, "nccl", cuda)
    assert rec.is_cuda and rec.device == gathered.device and rec.shape == (4, pdist.REC)
    with pytest.raises(ValueError, match="RCCL"):
        pdist.check_collective_buffers(rec.cpu(), gathered, world, "nccl", cuda)
    with pytest.raises(ValueError, match="world"):
        pdist.check_collective_buffers(rec, gathered[:-4], world, "nccl", cuda)
    with pytest.raises(TypeError):
        pdist.check_collective_buffers(rec.double(), gathered.double(), world, "nccl", cuda)
    pdist.check_collective_buffers(rec.cpu(), gathered.cpu(), world, "gloo")


def test_bench_rccl_needs_one_gpu_per_rank(cuda):
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs")
    r = _run(["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-stream", "--no-cpu-baseline"])
    assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout)
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())            # no line pretending n_gpus = 2


def test_bench_one_rank_under_a_launcher_equals_plain_run(cuda):
    """`--gpus 1` with and without launcher environment: same code path, same number (within 2 %, one retry)."""
    args = ["bench.py", "--gpus", "1", "--steps", "40", "--warmup", "10", "--variant", "resnet50-cliff", "--no-stream",
            "--no-cpu-baseline", "--no-dominant"]
    for attempt in range(2):
        a = _json_line(_run(args).stdout)
        b = _json_line(_run(args, env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                       "MASTER_PORT": "29555"}).stdout)
        assert a["n_gpus"] == b["n_gpus"] == 1 and a["dist"]["world_size"] == 1
        rel = abs(a["step_ms_events"]["median"] - b["step_ms_events"]["median"]) / a["step_ms_events"]["median"]
        if rel < 0.02:
            break
    assert rel < 0.02, (a["step_ms_events"], b["step_ms_events"])


def test_demo_video_two_ranks_matches_one_process(tmp_path, cuda):
    """demo.py --mode video --gpus 2 (tracks sharded, records all-gathered, rank 0 re-derives the meshes of the other
    rank's tracks with the LBS operator) writes the same poco_results.npz as the single-process run."""
    from PIL import Image
    variant = "resnet50-cliff"
    w = util.synth_weights(variant, profile="stress")
    torch.save({"state_dict": {"model." + k: torch.from_numpy(v) for k, v in w.items()}}, tmp_path / "ckpt.pt")
    np.savez(tmp_path / "smpl.npz", **synth.synth_smpl(7))
    fr = tmp_path / "frames"
    fr.mkdir()
    r = np.random.default_rng(4)
    for i in range(6):
        img = np.clip(128 + 60 * r.standard_normal((6, 8, 3)).repeat(40, 0).repeat(40, 1) + 20 * r.standard_normal((240, 320, 3)), 0, 255)
        Image.fromarray(img.astype(np.uint8)).save(fr / f"{i:06d}.png")
    tracks = {"0": {"bbox": [[160 + 3 * i, 120, 150, 150] for i in range(6)], "frames": list(range(6))},
              "1": {"bbox": [[80, 100 + 2 * i, 90, 120] for i in range(4)], "frames": list(range(2, 6))},
              "2": {"bbox": [[250, 60, 70, 70]], "frames": [3]}}
    (tmp_path / "tracks.json").write_text(json.dumps(tracks))
    base = ["demo.py", "--cfg", "configs/demo_poco_cliff_resnet50.yaml", "--ckpt", str(tmp_path / "ckpt.pt"), "--mode", "video",
            "--vid_file", str(fr), "--batch_size", "4", "--smpl", str(tmp_path / "smpl.npz"), "--no_render", "--smooth",
            "--tracking", str(tmp_path / "tracks.json")]
    one = _run(base + ["--output_folder", str(tmp_path / "o1")])
    assert one.returncode == 0, one.stderr[-3000:]
    two = _run(base + ["--output_folder", str(tmp_path / "o2"), "--gpus", "2", "--dist_backend", "gloo"])
    assert two.returncode == 0, two.stderr[-3000:]
    assert _json_line(two.stdout)["ranks"] == 2
    a = dict(np.load(tmp_path / "o1" / "frames_" / "poco_results.npz"))
    b = dict(np.load(tmp_path / "o2" / "frames_" / "poco_results.npz"))
    assert set(a) == set(b)
    for k in a:
        tol = 2e-2 if k.endswith("smpl_joints2d") else 1e-4            # 2-D joints in full-image pixels (values ~1e3)
        assert a[k].shape == b[k].shape and np.abs(a[k].astype(np.float64) - b[k]).max() <= tol, (k, np.abs(a[k] - b[k]).max())
    assert a["0/verts"].shape == (6, 6890, 3) and np.abs(a["0/pose"] - a["1/pose"][:1]).max() > 1e-2


def test_bench_line_contract(cuda):
    """The ONE JSON line bench.py prints: every field of the driver contract, roofline with a fresh (or explicitly stale) traffic
    source, per-step event statistics, and the CPU baseline + streaming legs on request only."""
    r = _run(["bench.py", "--steps", "5", "--warmup", "2", "--variant", "resnet50-cliff", "--no-stream", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "step_ms_events", "dist"):
        assert k in d, k
    assert d["unit"] == "crops/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["steps"] == 5 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.2 < rf["frac"] < 1.05
    assert (rf["traffic"] is None) == ("stale" in rf["traffic_source"] or "no committed" in rf["traffic_source"])
    assert abs(d["value"] - 64 * 5 / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-3            # value = crops of the job / its wall time
    ev = d["step_ms_events"]
    assert ev["min"] <= ev["median"] <= ev["max"] and abs(ev["mean"] - d["ms_per_step"]) / ev["mean"] < 0.05
    assert "cpu_baseline" not in d and "streaming_cfg5" not in d
