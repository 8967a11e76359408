"""Opt-in validation against the REAL, licence-gated assets.  Nothing here can run in
the build / CI image (no smplx, no SMPL_NEUTRAL.pkl, no data/poco_*.pt); it is the path by which a user who holds the licences
turns the two "parity unpinned" rows (a10 SMPL-LBS, f2 real-asset loaders) into pinned ones on their own machine.

    python tools/validate_assets.py [--smpl-pkl data/smpl/SMPL_NEUTRAL.pkl] [--extra data/J_regressor_extra.npy]
                                    [--smpl-npz data/smpl/SMPL_NEUTRAL.npz]
                                    [--ckpt data/poco_cliff.pt --cfg configs/demo_poco_cliff.yaml] [--device cuda:0]

Every stage whose inputs are missing is reported as SKIPPED (exit code 0); a stage that runs and disagrees is FAILED (exit 1).
  1. SMPL file: convert the licensed pickle with tools/convert_smpl.py (or take --smpl-npz) and check shapes / sparsity /
     kinematic tree against what pocolib/models/head/smpl_head.py:12-34 expects (6890 vertices, 24 joints, 207 pose-blend rows).
  2. LBS parity (needs `smplx` importable + the model directory): smplx.SMPL.forward(pose2rot=False) on random betas / rotations
     against oracle/smpl_np.py (float64) and - with a GPU - against poco_smpl_lbs; also the 21 extra-vertex ids and the 49-joint map
     against smplx's own VertexJointSelector / the reference's constants.  Gate: 1e-5 m (oracle), 1e-4 m (HIP fp32).
  3. Checkpoint: strict load of data/poco_{pare,cliff}.pt through poco_amd.checkpoint (pocolib/models/poco.py:131-154,
     train_utils.py:69-136): reports missing / unexpected / tolerated-unused keys and shape mismatches without needing a GPU;
     with a GPU it finalises the engine and runs one forward on a synthetic crop (finite outputs, orthonormal rotations).
  4. Crop (needs `cv2` importable; licence-free): the reference's per-detection crop - cv2.getAffineTransform +
     cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) on uint8 exactly as pocolib/utils/vibe_image_utils.py:58-107 calls them - against
     the fixed-point restatement oracle/crop_np.py (uint8, byte for byte; the transform matrix bit for bit) and, with a GPU, against
     poco_crop_normalize.  `--crop` runs only this stage.
"""
from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

SKIP, OK, FAIL = "SKIPPED", "OK", "FAILED"


def check_smpl_npz(smpl: dict) -> list:
    """Structural checks of an engine-format body model.  Returns a list of problems (empty = fine)."""
    from poco_amd import synth
    bad = []
    V = smpl["v_template"].shape[0]
    want = {"v_template": (V, 3), "shapedirs": (V, 3, 10), "posedirs": (207, V * 3), "J_regressor": (24, V),
            "J_regressor_extra": (9, V), "lbs_weights": (V, 24), "parents": (24,), "extra_vertex_ids": (21,), "joint_map": (49,)}
    for k, shp in want.items():
        if k not in smpl:
            bad.append(f"missing array {k}")
        elif tuple(smpl[k].shape) != shp:
            bad.append(f"{k}: shape {tuple(smpl[k].shape)}, expected {shp}")
    if bad:
        return bad
    if not np.array_equal(np.asarray(smpl["parents"], np.int64), synth.SMPL_PARENTS.astype(np.int64)):
        bad.append("kinematic tree differs from the SMPL tree (pocolib/utils/kp_utils.py:881-908)")
    if np.abs(smpl["lbs_weights"].sum(1) - 1).max() > 1e-4:
        bad.append("skinning weights do not sum to 1 per vertex")
    if np.abs(smpl["J_regressor"].sum(1) - 1).max() > 1e-3:
        bad.append("J_regressor rows do not sum to 1")
    if int(np.asarray(smpl["extra_vertex_ids"]).max()) >= V or int(np.asarray(smpl["joint_map"]).max()) >= 54:
        bad.append("extra_vertex_ids / joint_map out of range")
    return bad


def stage_smpl_file(args):
    npz = args.smpl_npz
    if args.smpl_pkl and os.path.isfile(args.smpl_pkl) and args.extra and os.path.isfile(args.extra):
        from tools import convert_smpl
        npz = npz or str(Path(args.smpl_pkl).with_suffix(".npz"))
        convert_smpl.main(args.smpl_pkl, args.extra, npz)
    if not npz or not os.path.isfile(npz):
        return SKIP, "no SMPL model (--smpl-pkl + --extra, or --smpl-npz)", None
    smpl = dict(np.load(npz))
    bad = check_smpl_npz(smpl)
    if bad:
        return FAIL, "; ".join(bad), None
    return OK, f"{npz}: {smpl['v_template'].shape[0]} vertices, tree / weights / regressors consistent", smpl


def lbs_against(fn_ref, smpl: dict, n: int = 8, seed: int = 0):
    """max |oracle - ref| over vertices and the 49 joints for random (betas, rotations); fn_ref(betas, rotmat) -> (verts, joints49)."""
    import torch
    from oracle import poco_ref, smpl_np
    r = np.random.default_rng(seed)
    betas = r.standard_normal((n, 10)).astype(np.float32)
    R = poco_ref.rot6d_to_rotmat(torch.from_numpy(r.standard_normal((n * 24, 6)).astype(np.float32))).reshape(n, 24, 3, 3).numpy()
    v64, j64 = smpl_np.smpl_lbs_np(smpl, betas, R)
    v, j = fn_ref(betas, R)
    return float(np.abs(np.asarray(v) - v64).max()), float(np.abs(np.asarray(j) - j64).max()), betas, R, v64, j64


def stage_lbs(args, smpl):
    if smpl is None:
        return SKIP, "no SMPL model"
    try:
        import smplx  # noqa: F401
        import torch
    except ImportError:
        return SKIP, "smplx is not importable (pip install smplx==0.1.28, requirements.txt:7)"
    model_dir = args.smpl_dir or (str(Path(args.smpl_pkl).parent) if args.smpl_pkl else None)
    if not model_dir or not os.path.isdir(model_dir):
        return SKIP, "no smplx model directory (--smpl-dir)"
    from smplx import SMPL as _SMPL
    from smplx.lbs import vertices2joints
    m = _SMPL(model_dir, create_transl=False)
    msgs = []
    # data the product keeps as constants: extra vertex ids (smplx VertexJointSelector) and the 49-joint map
    sel = getattr(getattr(m, "vertex_joint_selector", None), "extra_joints_idxs", None)
    if sel is not None and not np.array_equal(sel.cpu().numpy().astype(np.int64), np.asarray(smpl["extra_vertex_ids"], np.int64)):
        return FAIL, "extra_vertex_ids differ from smplx's VertexJointSelector: " + str(sel.cpu().numpy().tolist())
    Jx = torch.from_numpy(np.asarray(smpl["J_regressor_extra"], np.float32))
    jm = torch.from_numpy(np.asarray(smpl["joint_map"], np.int64))

    def ref(betas, R):                                    # pocolib/models/head/smpl_head.py:22-34 on the real smplx layer
        Rt = torch.from_numpy(R)
        o = m(betas=torch.from_numpy(betas), body_pose=Rt[:, 1:].contiguous(), global_orient=Rt[:, :1].contiguous(), pose2rot=False)
        joints = torch.cat([o.joints, vertices2joints(Jx, o.vertices)], 1)[:, jm]
        return o.vertices.detach().numpy(), joints.detach().numpy()

    dv, dj, betas, R, v64, j64 = lbs_against(ref, smpl)
    msgs.append(f"oracle (float64 restatement) vs smplx.SMPL.forward: vertices {dv:.2e} m, joints49 {dj:.2e} m")
    status = OK if max(dv, dj) < 1e-5 else FAIL
    if torch.cuda.is_available():
        from poco_amd.model import POCO
        eng = POCO(backbone="resnet50-cliff", num_flow_layers=1, max_batch=len(betas), smpl=smpl, device=args.device)
        # LBS only needs the body model; finalize() wants the network too: use the seeded synthetic weights
        from tests import util
        eng.load_state_dict(util.synth_weights("resnet50-cliff"), strict=True)
        eng.finalize()
        v, j = eng.smpl_lbs(torch.from_numpy(betas).to(args.device), torch.from_numpy(R).to(args.device))
        dvh, djh = float(np.abs(v.cpu().numpy() - v64).max()), float(np.abs(j.cpu().numpy() - j64).max())
        msgs.append(f"poco_smpl_lbs (HIP fp32) vs float64: vertices {dvh:.2e} m, joints49 {djh:.2e} m")
        if max(dvh, djh) >= 1e-4:
            status = FAIL
    else:
        msgs.append("no GPU: HIP LBS operator not exercised")
    return status, " | ".join(msgs)


def checkpoint_report(engine_tensors, loaded: dict) -> dict:
    """Strict comparison of a checkpoint's tensors with what the engine declares (names, shapes after dropping size-1 dims)."""
    want = {n: (tuple(s), req) for n, s, req in engine_tensors if not n.startswith("smpl.")}
    squeeze = lambda s: tuple(int(d) for d in s if int(d) != 1)   # noqa: E731
    rep = {"missing": sorted(n for n, (_, req) in want.items() if req and n not in loaded),
           "tolerated_unused_present": sorted(n for n, (_, req) in want.items() if not req and n in loaded),
           "unexpected": sorted(n for n in loaded if n not in want),
           "shape_mismatch": sorted(f"{n}: checkpoint {tuple(loaded[n].shape)} vs engine {want[n][0]}" for n in loaded
                                    if n in want and squeeze(loaded[n].shape) != squeeze(want[n][0]))}
    rep["ok"] = not (rep["missing"] or rep["unexpected"] or rep["shape_mismatch"])
    return rep


def stage_checkpoint(args, smpl):
    if not args.ckpt or not os.path.exists(args.ckpt):
        return SKIP, "no checkpoint (--ckpt data/poco_cliff.pt | data/poco_pare.pt)"
    if not args.cfg or not os.path.isfile(args.cfg):
        return SKIP, "no --cfg yaml for the checkpoint"
    import torch
    from poco_amd.checkpoint import read_checkpoint
    from poco_amd.config import model_kwargs, update_hparams
    from poco_amd.model import POCO
    kw = model_kwargs(update_hparams(args.cfg))
    sd = read_checkpoint(args.ckpt, args.inf_model)
    eng = POCO(**kw, max_batch=2, device=args.device)
    rep = checkpoint_report(eng.expected_tensors(), sd)
    msg = (f"{len(sd)} tensors; missing {len(rep['missing'])}, unexpected {len(rep['unexpected'])}, shape mismatches "
           f"{len(rep['shape_mismatch'])}, tolerated-unused present {len(rep['tolerated_unused_present'])}")
    for k in ("missing", "unexpected", "shape_mismatch"):
        if rep[k]:
            msg += f"\n      {k}: " + ", ".join(rep[k][:6]) + (" ..." if len(rep[k]) > 6 else "")
    if not rep["ok"]:
        return FAIL, msg
    eng.load_state_dict(sd, strict=True)
    if smpl is None or not torch.cuda.is_available():
        return OK, msg + " | strict load OK (no GPU or no SMPL model: forward not run)"
    from poco_amd import synth
    eng.load_smpl(smpl)
    eng.finalize()
    batch = {k: torch.from_numpy(v).to(args.device) for k, v in synth.synth_batch(2, 1).items()}
    out = eng(batch)
    torch.cuda.synchronize()
    finite = all(bool(torch.isfinite(v).all()) for v in out.values() if torch.is_tensor(v))
    Rm = out["pred_pose"].reshape(-1, 3, 3)
    ortho = float((Rm @ Rm.transpose(1, 2) - torch.eye(3, device=Rm.device)).abs().max())
    ok = finite and ortho < 1e-4 and 0 < float(out["var_pose"].min()) and float(out["var_pose"].max()) < 1
    return (OK if ok else FAIL), msg + f" | forward on 2 synthetic crops: finite={finite}, |RR^T-I|={ortho:.1e}"


def crop_against_cv2(cv2, n: int = 64, seed: int = 0, device=None):
    """Random frames / boxes through the real cv2 calls the reference makes vs oracle/crop_np.py (and the HIP kernel on `device`).
    `cv2` is passed in so that the CPU test can hand over a stand-in built from the oracle itself (logic check only)."""
    from oracle import crop_np
    r = np.random.default_rng(seed)
    frame = r.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    boxes = np.stack([r.uniform(-60, 1340, n), r.uniform(-60, 780, n), r.uniform(8, 900, n), r.uniform(8, 900, n)], 1)
    boxes[0] = (112, 112, 224, 224)
    worst = {"matrix_bits": 0, "u8_pixels": 0, "u8_maxdiff": 0, "gpu_values": 0}
    for dt in (np.float32, np.float64):
        for scale in (1.0, 1.1, 1.2):
            bx = boxes.astype(dt).astype(np.float64)                 # numpy 1.18: float32 box * Python float promotes to float64
            crops = []
            for (cx, cy, w, h) in bx:
                src, dst = crop_np.patch_points(cx, cy, w, h, 224, scale)
                M = cv2.getAffineTransform(np.float32(src), np.float32(dst))
                worst["matrix_bits"] += int((np.asarray(M, np.float64) != crop_np.get_affine_transform_cv(src, dst)).sum())
                crops.append(cv2.warpAffine(frame.copy(), M, (224, 224), flags=cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT))
            ref = np.stack(crops)
            mine = crop_np.crop_u8_np(frame, bx, scale)
            d = np.abs(ref.astype(np.int32) - mine.astype(np.int32))
            worst["u8_pixels"] += int((d > 0).sum())
            worst["u8_maxdiff"] = max(worst["u8_maxdiff"], int(d.max()))
            if device is not None:
                import torch
                from poco_amd.tester import crop_normalize
                out = crop_normalize(torch.from_numpy(frame).to(device), torch.from_numpy(boxes.astype(dt)).to(device), scale)
                worst["gpu_values"] += int((out.cpu().numpy() != crop_np.normalize_np(ref)).sum())
    return worst


def stage_crop(args):
    try:
        import cv2
    except Exception as e:                                     # noqa: BLE001
        return SKIP, f"cv2 is not importable ({type(e).__name__}); the reference pins opencv-python==4.5.5.64"
    dev = None
    try:
        import torch
        if torch.cuda.is_available():
            dev = args.device
    except Exception:                                          # noqa: BLE001
        pass
    w = crop_against_cv2(cv2, device=dev)
    msg = (f"cv2 {cv2.__version__}: 384 crops; getAffineTransform entries differing {w['matrix_bits']}, uint8 pixels differing "
           f"{w['u8_pixels']} (max {w['u8_maxdiff']} grey levels)" + (f", HIP kernel values differing {w['gpu_values']}" if dev else " (no GPU: HIP kernel not compared)"))
    ok = w["u8_pixels"] == 0 and w["gpu_values"] == 0
    if not ok and not cv2.__version__.startswith("4.5"):
        msg += " - note: OpenCV >= 4.11 ships a different (floating-point) warpAffine; the reference pins 4.5.5.64"
    return (OK if ok else FAIL), msg


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--smpl-pkl", default="data/smpl/SMPL_NEUTRAL.pkl")
    ap.add_argument("--extra", default="data/J_regressor_extra.npy")
    ap.add_argument("--smpl-npz", default=None)
    ap.add_argument("--smpl-dir", default=None, help="directory smplx.SMPL loads from (default: the folder of --smpl-pkl)")
    ap.add_argument("--ckpt", default="data/poco_cliff.pt")
    ap.add_argument("--cfg", default="configs/demo_poco_cliff.yaml")
    ap.add_argument("--inf_model", default="best")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--crop", action="store_true", help="only stage 4: the crop against the real cv2")
    args = ap.parse_args(argv)
    results = []
    if args.crop:
        st, msg = stage_crop(args)
        print(f"[{st:7s}] 4 crop vs cv2: {msg}")
        return 1 if st == FAIL else 0
    st, msg, smpl = stage_smpl_file(args)
    results.append(("1 SMPL file", st, msg))
    results.append(("2 LBS vs smplx", *stage_lbs(args, smpl)))
    results.append(("3 checkpoint", *stage_checkpoint(args, smpl)))
    results.append(("4 crop vs cv2", *stage_crop(args)))
    for name, st, msg in results:
        print(f"[{st:7s}] {name}: {msg}")
    return 1 if any(st == FAIL for _, st, _ in results) else 0


if __name__ == "__main__":
    sys.exit(main())
