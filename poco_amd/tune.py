"""Measured tile selection for the MFMA conv kernel ("measure, don't guess").

    python -m poco_amd.tune --variant hrnet_w48_cls-cliff --batch 64      # on a GPU box

For every distinct conv shape of a variant at a batch size, time the candidate tile decompositions
(MT,NT,WM,WN,R,NI) with poco_tune_conv and keep the fastest.  Results are merged into
poco_amd/tuned/gfx950.json (committed, so later runs need no re-tuning); POCO.finalize() applies
the table, shapes/batches without an entry fall back to the built-in heuristic.
"""
from __future__ import annotations

import argparse
import ctypes as C
import itertools
import os
import json
import math
import time
from pathlib import Path
from typing import Dict, List, Tuple

TABLE = Path(__file__).resolve().parent / "tuned" / "gfx950.json"


def shape_key(B, H, W, Cin, Cout, ks, stride) -> str:
    return f"{B}x{H}x{W}x{Cin}x{Cout}k{ks}s{stride}"


def load_table() -> Dict[str, List[int]]:
    if TABLE.exists():
        return {k: v["cfg"] for k, v in json.loads(TABLE.read_text()).items()}
    return {}


G1_SCHED_G = (0, 0, 2, 4, 8, 2, 4)   # ALG 6: MFMAs per pinned operand load for NI = 2..6 (gemm1x1.hip g1_sched)


def candidates(B, H, W, Cin, Cout, ks, stride, lds_cap=150 * 1024) -> List[Tuple[int, ...]]:
    pad = (ks - 1) // 2
    Ho = (H + 2 * pad - ks) // stride + 1
    Wo = (W + 2 * pad - ks) // stride + 1
    nT = Cout // 16
    out = set()
    if H == 1 and W == 1 and ks == 1:    # Linear layer: K split over WM waves (ALG 5)
        return [(1, 1, wm, 1, 1, 1, 5) for wm in (1, 2, 4, 8, 16)] + [(4, 1, 1, 1, 1, min(B, 64), 1)]
    if ks == 1 and stride == 1:          # split-K GEMM straight from global memory (small planes / few pixels)
        out.update((1, 1, wm, 1, 1, 1, 5) for wm in (1, 2, 4, 8))
    if ks == 1:                          # register-direct GEMM, no LDS (ALG 6); R = operand prefetch depth; NI = load
        for (MT, NT), (WM, WN), D in itertools.product(                   # schedule: 1 here, 2..6 tried by tune_shape
                ((2, 4), (4, 2), (4, 4), (7, 2), (7, 4), (8, 2)),
                ((1, 1), (1, 2), (2, 1), (1, 4), (2, 2), (4, 1), (1, 8), (2, 4), (4, 2), (8, 1)), (2, 3)):
            if WN > 1 and (WN - 1) * NT >= nT:       # a whole wave column beyond Cout
                continue
            out.add((MT, NT, WM, WN, D, 1, 6))
            if D == 2 and (MT, NT) in ((4, 4), (7, 2), (7, 4), (8, 2)):   # the same through the LDS transposition (ALG 9)
                out.add((MT, NT, WM, WN, 1, 1, 9))
    if ks == 1 and stride == 1 and H * W > 1 and (B * H * W // 16) * nT >= 1024:   # stream-K register-direct GEMM (ALG 14, gemm1x1sk.hip):
        for (MT, NT, D), WM, NI in itertools.product(                                  # WM = waves per block of the persistent grid
                ((7, 4, 2), (7, 2, 2), (7, 2, 3), (4, 4, 2), (4, 4, 3), (4, 2, 3), (2, 4, 3)), (2, 4, 8), (1, 3, 6)):
            if WM <= (4 if MT * NT > 16 else 8):
                out.add((MT, NT, WM, 1, D, NI, 14))
    if ks == 3 and (stride == 2 or os.environ.get("POCO_TUNE_G3_S1")):     # register-direct gather GEMM (ALG 10); R = depth, NI = schedule
        for (MT, NT), (WM, WN), D, NI in itertools.product(
                ((2, 4), (4, 2), (4, 3), (4, 4), (7, 2), (7, 3), (7, 4), (8, 2)),
                ((1, 1), (1, 2), (2, 1), (1, 4), (2, 2), (4, 1), (1, 8), (2, 4), (4, 2), (8, 1)), (2, 3), (1, 3, 6)):
            if WN > 1 and (WN - 1) * NT >= nT:
                continue
            if (nT + NT - 1) // NT * NT - nT >= NT or ((nT + NT - 1) // NT * NT - nT) * 4 > nT:     # > 25 % idle n-tiles
                continue
            if {1: 0, 3: 4, 6: 4}[NI] * (MT + NT) > 4 * MT * NT:
                continue
            out.add((MT, NT, WM, WN, D, NI, 10))
    for MT, NT, WM, WN in itertools.product((4, 7, 13), (1, 2, 3, 4), (1, 2, 4, 8), (1, 2, 3, 4, 6, 8)):
        if WM * WN > 8 or (MT == 13 and NT > (2 if ks == 3 else 3)) or nT % NT:
            continue
        if (nT // NT) % WN and WN > 1 and (nT // NT) > WN:
            continue
        if WN > nT // NT:
            continue
        cap = WM * MT * 16
        for R in range(1, Ho + 1):
            if R * Wo > cap:
                break
            nis = {1, max(1, cap // (R * Wo))} if R == Ho or Ho % R == 0 or R * Wo * 2 > cap else {1}
            for ni in nis:
                if ni > max(1, B * ((Ho + R - 1) // R)):
                    continue
                util = ni * R * Wo / cap
                if util < 0.75:
                    continue
                pr, pw = (R - 1) * stride + ks, (Wo - 1) * stride + ks
                npos = ni * pr * pw
                lds = 4 * ((npos + 15) // 16 * 16) * 16
                if lds <= lds_cap:
                    out.add((MT, NT, WM, WN, R, ni, 0))
                plane = (npos + 63) // 64 * 64
                lds1 = 2 * (4 * plane + ks * ks * WN * NT * 64) * 16
                if lds1 <= 160 * 1024 and (plane // 64 + WM * WN - 1) // (WM * WN) <= 6:
                    out.add((MT, NT, WM, WN, R, ni, 1))
                    out.add((MT, NT, WM, WN, R, ni, 2))
    if ks == 3 and stride == 1:          # Winograd F(2x2,3x3): one 16-tile sub-tile per wave
        TX = (W + 1) // 2
        for NT, WM, WN in itertools.product((1, 2), range(1, 13), (1, 2, 3, 4, 6, 8)):
            if WM * WN > (12 if NT == 1 else 8) or nT % NT or WN > nT // NT:
                continue
            for R in range(2, (H + 1) // 2 * 2 + 1, 2):
                tiles = (R // 2) * TX
                if tiles > WM * 16:
                    break
                for ni in {1, max(1, (WM * 16) // tiles)}:
                    if ni * tiles / (WM * 16) < 0.75 or (ni > 1 and R < H):
                        continue
                    npos = ni * (R + 2) * (2 * TX + 2)
                    plane = (npos + 63) // 64 * 64
                    lds = (4 * plane + 2 * 16 * WN * NT * 64) * 16
                    if lds <= 160 * 1024 and (plane // 64 + WM * WN - 1) // (WM * WN) <= 6:
                        out.add((1, NT, WM, WN, R, ni, 3))
        for NT, WM in itertools.product((1, 2, 3), (1, 2, 3, 4)):
            if nT % NT and NT > 1 and nT > NT:
                pass
            for R in range(2, (H + 1) // 2 * 2 + 1, 2):
                tiles = (R // 2) * TX
                if tiles > WM * 16:
                    break
                for ni in {1, max(1, (WM * 16) // tiles)}:
                    if ni * tiles / (WM * 16) < 0.7 or (ni > 1 and R < H):
                        continue
                    npos = ni * (R + 2) * (2 * TX + 2)
                    plane = (npos + 63) // 64 * 64
                    lds = (8 * plane + 2 * 16 * NT * 64) * 16
                    if lds <= 160 * 1024 and (plane // 64 + WM - 1) // WM <= 6 and (nT % NT == 0 or nT < NT):
                        out.add((1, NT, WM, 2, R, ni, 4))
    if ks == 3 and stride == 1 and H <= 8 and W <= 8 and H * W > 1:     # Winograd F(4x4,3x3) as 36 position GEMMs (ALG 11)
        for (MT, NT), (WM, WN), D in itertools.product(((1, 1), (1, 2), (1, 4), (2, 1), (2, 2), (2, 4), (4, 1), (4, 2), (4, 4), (8, 2)),
                                                       ((1, 1), (2, 1), (1, 2), (4, 1), (2, 2), (1, 4), (8, 1), (4, 2), (2, 4)), (2, 3, 4, 6)):
            if (WN > 1 and (WN - 1) * NT >= nT) or D * (MT + NT) + MT * NT > 56:
                continue
            out.add((MT, NT, WM, WN, D, 1, 11))
    if ks == 3 and stride == 1 and H >= 14 and W >= 14:      # Winograd F(4x4,3x3): ALG 7 (planes >= 28x28) / ALG 8 (>= 14x14); 2 tile groups x 4 position quarters
        TX4, Hc = (W + 3) // 4, (H + 3) // 4 * 4
        for NT in (1, 2, 3):
            ragged = nT % NT and nT > NT       # last n-tile group partly empty: ALG 8 handles it (clamped U fetch, masked stores)
            if ragged and not (NT == 3 and nT % 3 == 2 and nT >= 8):      # worth it only when one of >= 9 slots idles
                continue
            for R in range(4, Hc + 1, 4):
                tps = (R // 4) * TX4
                if tps > 32:
                    break
                nis = {1} if R < Hc else {1, max(1, 32 // tps)}
                for ni in nis:
                    if ni * tps < 20:
                        continue
                    npos = ni * (R + 2) * (4 * TX4 + 2)
                    raw = (npos + npos // 8 + 1 + 63) // 64 * 64
                    if not ragged and H >= 28 and W >= 28 and raw <= 1024 and 2 * max(2 * (raw + 9 * NT * 64), 4096) * 16 <= 160 * 1024:
                        out.add((1, NT, 2, 4, R, ni, 7))
                    # ALG 8 (specialised waves): raw ring 3 deep (<= 1024 slots per slice), U ring 3 deep, V 2 deep, exchange overlay
                    u, v = NT * 576, 2 * 576
                    tot = max(3 * raw + 3 * u + 2 * v, 3 * raw + u + 4096)
                    if raw <= 1024 and tot * 16 <= 160 * 1024:
                        out.add((1, NT, 2, 4, R, ni, 8))
                    # ALG 13 (whole-position MFMA waves, round 5; round 6: U in registers - LDS = raw ring + V only)
                    if raw <= 1024:
                        out.add((1, NT, 2, 1, R, ni, 13))
            # ALG 8 with FLAT items (R = 4, NI = 0; round 4): 32 consecutive tiles of the flattened (image, tile row, tile column)
            # order per item, 6-row strip patch with slots skewed by pos / 16
            fmax = (TX4 - 1 + 32 + TX4 - 1) // TX4
            npos = 6 * (128 + 2 * fmax)
            raw = (npos + npos // 16 + 1 + 63) // 64 * 64
            u, v = NT * 576, 2 * 576
            if raw <= 1024 and max(3 * raw + 3 * u + 2 * v, 3 * raw + u + 4096) * 16 <= 160 * 1024:
                out.add((1, NT, 2, 4, 4, 0, 8))
            if raw <= 1024:
                out.add((1, NT, 2, 1, 4, 0, 13))
            # ... over a MOSAIC of MS x MS images that share their zero borders (R = 4 MS): only where it saves tiles (14 x 14 planes:
            # 15 x 15 tiles per 4 x 4 images instead of 16 x 16)
            for MS in (2, 4, 8):
                TXm, TYm = (MS * (W + 1) - 1 + 3) // 4, (MS * (H + 1) - 1 + 3) // 4
                if -(-B // (MS * MS)) * TXm * TYm >= 0.95 * B * TX4 * (Hc // 4):
                    continue
                fm = (TXm - 1 + 32 + TXm - 1) // TXm
                nposm = 6 * (128 + 2 * fm)
                rawm = (nposm + nposm // 16 + 1 + 63) // 64 * 64
                if rawm <= 1024 and max(3 * rawm + 3 * u + 2 * v, 3 * rawm + u + 4096) * 16 <= 160 * 1024:
                    out.add((1, NT, 2, 4, 4 * MS, 0, 8))
                if rawm <= 1024:
                    out.add((1, NT, 2, 1, 4 * MS, 0, 13))       # (round 6: ALG 13 walks mosaics too)
    return sorted(out)


def _timed(L, B, H, W, Cin, Cout, ks, stride, cands, iters):
    """ms of every candidate of one poco_tune_conv call.  The FIRST configuration of a call measures ~10 % slow (66.5 against 60.4 us for
    the same 14x14 1024->256 kernel in slot 0 and in a later slot, tools/sk_probe.py, round 5: the chip is still clocking up / its
    caches are cold), so slot 0 is a throw-away copy of the first candidate."""
    from ._lib import check
    run = [cands[0]] + list(cands)
    flat = (C.c_int * (7 * len(run)))(*[v for c in run for v in c])
    ms = (C.c_float * len(run))()
    check(L.poco_tune_conv(B, H, W, Cin, Cout, ks, stride, flat, len(run), iters, ms, None), "poco_tune_conv")
    return [ms[i + 1] for i in range(len(cands))]


def solo_times(L, B, H, W, Cin, Cout, ks, stride, iters=8, with_default=False):
    """[(ms, cfg)] of every candidate of one shape, each timed on its own (cfg all-zero = the built-in heuristic)."""
    from ._lib import check
    cands = ([(0, 0, 0, 0, 0, 0, 0)] if with_default else []) + candidates(B, H, W, Cin, Cout, ks, stride)
    ms = _timed(L, B, H, W, Cin, Cout, ks, stride, cands, iters)
    res = [(ms[i], cands[i]) for i in range(len(cands))]
    # ALG 6 load schedules (NI 2..6) only for its best few tilings under hipcc's schedule (NI = 1)
    g6 = [c for _, c in sorted(r for r in res if r[0] > 0 and r[1][6] == 6 and r[1][5] == 1)[:8]]
    cands6 = [c[:5] + (ni, 6) for c in g6 for ni in range(2, 7) if G1_SCHED_G[ni] * (c[0] + c[1]) <= 4 * c[0] * c[1]]
    if cands6:
        ms6 = _timed(L, B, H, W, Cin, Cout, ks, stride, cands6, iters)
        res += [(ms6[i], cands6[i]) for i in range(len(cands6))]
    return res


def tune_shape(L, B, H, W, Cin, Cout, ks, stride, iters=8):
    from ._lib import check
    timed = solo_times(L, B, H, W, Cin, Cout, ks, stride, iters, with_default=True)
    base = timed[0][0]
    cands = [c for _, c in timed]
    res = [r for r in timed if r[0] > 0]
    # re-time the top few with more iterations to reduce noise
    top = sorted(res)[:6]
    cands2 = [c for _, c in top]
    ms2 = _timed(L, B, H, W, Cin, Cout, ks, stride, cands2, iters * 4)
    best_i = min(range(len(cands2)), key=lambda i: ms2[i])
    return cands2[best_i], float(ms2[best_i]), float(base), len(cands)


def tune_model(model, B: int, verbose=True, skip=(), only_ks=0) -> Dict[str, dict]:
    from ._lib import lib
    L = lib()
    L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
    shapes = {}
    for i, (name, flops, ty) in enumerate(model.ops()):
        d = model.conv_desc(i)
        if d is None:
            continue
        H, W, Cin, Cout, ks, stride = d[:6]
        shapes.setdefault((H, W, Cin, Cout, ks, stride), []).append(i)
    out = {}
    t0 = time.time()
    for (H, W, Cin, Cout, ks, stride), idxs in sorted(shapes.items(), key=lambda kv: -len(kv[1])):
        key = shape_key(B, H, W, Cin, Cout, ks, stride)
        if key in skip or (only_ks and (ks != only_ks or H * W == 1)):
            continue
        cfg, ms, base, n = tune_shape(L, B, H, W, Cin, Cout, ks, stride)
        fl = 2.0 * B * ((H + 2 * ((ks - 1) // 2) - ks) // stride + 1) ** 2 * Cout * Cin * ks * ks if H == W else 0
        out[key] = {"cfg": list(cfg), "ms": round(ms, 5), "heuristic_ms": round(base, 5), "uses": len(idxs),
                    "tflops": round(fl / ms / 1e9, 1) if ms > 0 else 0}
        if verbose:
            print(f"{key:34s} x{len(idxs):3d}  {base:.4f} -> {ms:.4f} ms  {out[key]['tflops']:6.1f} TF  cfg={cfg}  ({n} cands)", flush=True)
    if verbose:
        print(f"tuned {len(out)} shapes in {time.time()-t0:.1f}s")
    return out


def apply_table(model, B: int, table=None) -> int:
    """Set the measured configuration of every conv op for batch size B.  Without an entry for exactly B the
    entry of the nearest tuned batch size is used if it is valid for B (tiles are per image / row band, so
    they transfer; the library validates and the built-in heuristic stays in place otherwise)."""
    from ._lib import PocoHipError
    table = load_table() if table is None else table
    by_shape: Dict[str, List[Tuple[int, List[int]]]] = {}
    for k, cfg in table.items():
        b, rest = k.split("x", 1)
        by_shape.setdefault(rest, []).append((int(b), cfg))
    n = 0
    for i, _ in enumerate(model.ops()):
        d = model.conv_desc(i)
        if d is None:
            continue
        rest = shape_key(B, *d[:6]).split("x", 1)[1]
        # nearest tuned batch in log space, ties to the LARGER neighbour (its tiles were chosen with more blocks in flight, the safer
        # direction).  The split-K direct conv (ALG 5 with ks = 3) re-streams all 9 Cin Cout weights per 64-pixel block: its
        # entries were tuned at 1 / 4 crops and only transfer to batch sizes up to the tuned one
        for b, cfg in sorted(by_shape.get(rest, []), key=lambda bc: (abs(math.log(bc[0] / B)), -bc[0])):
            if not cfg or cfg[0] <= 0:
                continue
            if cfg[6] == 5 and d[4] == 3 and B > b:
                continue
            try:
                model.set_conv_cfg(i, B, cfg)
                n += 1
                break
            except PocoHipError:
                continue
    return n


def tune_in_context(variant: str, B: int, top_shapes: int = 12, top_cands: int = 6, iters: int = 15, verbose=True,
                    only=None, hysteresis_ms: float = 0.01) -> Dict[str, dict]:
    """Second tuning pass, measured INSIDE the whole forward (4 lanes): the configuration that wins a solo
    micro-benchmark is not always the one that wins next to the other lanes' kernels (7x7 384->384: 68.6 us solo vs
    75.3 us for the runner-up, yet 17.98 vs 17.75 ms per forward).  For the `top_shapes` most expensive shapes the
    `top_cands` best solo candidates are tried in place (greedy, one pass, eager launches) and the best is kept."""
    import numpy as np
    import torch
    from . import synth
    from ._lib import PocoHipError, lib
    from .model import POCO
    L = lib()
    L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
    fl = {"hrnet_w32-pare": 3, "hrnet_w48_cls-cliff": 1, "resnet50-cliff": 1}[variant]
    import json as _json
    spec = [(n, tuple(sh)) for n, sh in _json.loads((Path(__file__).resolve().parent.parent / "tests" / "golden" /
                                                     f"spec_{variant}.json").read_text())]
    m = POCO(backbone=variant, num_flow_layers=fl, max_batch=B, smpl=synth.synth_smpl(7))
    m.load_state_dict({k: v for k, v in synth.synth_state_dict(spec, 0).items() if v.dtype != np.int64}, strict=True)
    m.finalize()
    batch = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_batch(B, 1).items()}
    out = m._alloc_outputs(B, False)

    def forward_ms():
        for _ in range(3):
            m(batch, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            m(batch, out=out)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    table = json.loads(TABLE.read_text()) if TABLE.exists() else {}
    prof = m.profile_ops(batch, iters=3)
    shapes: Dict[str, List[int]] = {}
    cost: Dict[str, float] = {}
    for i, (nm, f, ty, ms) in enumerate(prof):
        d = m.conv_desc(i)
        if d is None:
            continue
        k = shape_key(B, *d[:6])
        shapes.setdefault(k, []).append(i)
        cost[k] = cost.get(k, 0.0) + ms
    base = forward_ms()
    if verbose:
        print(f"{variant} B={B}: {base:.3f} ms/forward before the in-context pass")
    res = {}
    ranked = [k for k in sorted(cost, key=lambda kk: -cost[kk]) if only is None or only(m.conv_desc(shapes[k][0])[:6])]
    for k in ranked[:top_shapes]:
        idxs = shapes[k]
        H, W, Cin, Cout, ks, stride = m.conv_desc(idxs[0])[:6]
        solo = sorted(r for r in solo_times(L, B, H, W, Cin, Cout, ks, stride) if r[0] > 0)
        if only is not None:       # a restricted pass (e.g. --g3): the best few of EVERY algorithm, so that a new kernel gets its trial
            by_alg: Dict[int, list] = {}
            for r in solo:
                by_alg.setdefault(r[1][6], []).append(r)
            solo = sorted(r for lst in by_alg.values() for r in lst[:3])
        solo = solo[:top_cands]
        cur = tuple(m.conv_cfg(idxs[0], B))
        trial = [cur] + [c for _, c in solo if tuple(c) != cur]
        # ALG 8 on half of the CUs (cfg.MT = 2): always slower alone, but two such launches of different lanes then run out of
        # phase (one's HBM-write-bound store rounds under the other's MFMA-bound K loops)
        trial += [(2,) + tuple(c[1:]) for c in trial[:3] if c[6] == 8 and c[0] == 1]
        best_cfg, best_t = cur, None
        for cfg in trial:
            try:
                for i in idxs:
                    m.set_conv_cfg(i, B, cfg)
            except PocoHipError:
                continue
            t = forward_ms()
            if best_t is None or t < best_t - hysteresis_ms:      # hysteresis against noise
                best_cfg, best_t = tuple(cfg), t
        for i in idxs:
            m.set_conv_cfg(i, B, best_cfg)
        solo_ms = dict((tuple(c), t) for t, c in solo).get(best_cfg, table.get(k, {}).get("ms", 0.0))
        res[k] = {"cfg": list(best_cfg), "ms": round(float(solo_ms), 5), "uses": len(idxs), "in_context": True,
                  "heuristic_ms": table.get(k, {}).get("heuristic_ms", 0.0), "tflops": table.get(k, {}).get("tflops", 0.0)}
        if verbose:
            print(f"  {k:30s} x{len(idxs):3d}  {cur} -> {best_cfg}   forward {best_t:.3f} ms", flush=True)
    if verbose:
        print(f"{variant} B={B}: {forward_ms():.3f} ms/forward after")
    return res


def main():
    import numpy as np
    import torch
    from . import synth
    from .model import POCO
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="hrnet_w48_cls-cliff")
    ap.add_argument("--batch", type=int, nargs="+", default=[64])
    ap.add_argument("--only-missing", action="store_true", help="keep existing table entries, tune new shapes only")
    ap.add_argument("--out", default=str(TABLE), help="where to write the merged table")
    ap.add_argument("--ks", type=int, default=0, help="re-tune only the spatial convs of this kernel size (1 or 3)")
    ap.add_argument("--in-context", action="store_true",
                    help="second pass: re-pick the configuration of the most expensive shapes inside the whole forward")
    ap.add_argument("--small-planes", action="store_true",
                    help="in-context pass over the 3x3 stride-1 shapes on planes <= 8x8 (ALG 11, conv_wino4g.hip, against the tuned entry)")
    ap.add_argument("--g3", action="store_true",
                    help="in-context pass over the 3x3 stride-2 shapes only (ALG 10, gemm3x3.hip, against the tuned LDS-staged entry)")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    fl = {"hrnet_w32-pare": 3, "hrnet_w48_cls-cliff": 1, "resnet50-cliff": 1}[args.variant]
    m = POCO(backbone=args.variant, num_flow_layers=fl, max_batch=1)   # declarations only: shapes, no weights
    full = json.loads(TABLE.read_text()) if TABLE.exists() else {}
    for B in args.batch:
        if args.small_planes:
            res = tune_in_context(args.variant, B, top_shapes=8, top_cands=10, iters=25,
                                  only=lambda d: d[4] == 3 and d[5] == 1 and d[0] <= 8 and d[1] <= 8, hysteresis_ms=0.004)
            full.update({k: v for k, v in res.items() if v["cfg"][0] > 0})
            continue
        if args.g3:
            res = tune_in_context(args.variant, B, top_shapes=40, top_cands=8, iters=25, only=lambda d: d[4] == 3 and d[5] == 2,
                                  hysteresis_ms=0.004)
            full.update({k: v for k, v in res.items() if v["cfg"][0] > 0})
            continue
        if args.in_context:
            full.update({k: v for k, v in tune_in_context(args.variant, B).items() if v["cfg"][0] > 0})
            continue
        full.update({k: v for k, v in tune_model(m, B, skip=set(full) if args.only_missing else (), only_ks=args.ks).items()
                     if v["cfg"][0] > 0})
    out = Path(args.out)
    out.parent.mkdir(exist_ok=True, parents=True)
    out.write_text(json.dumps(full, indent=0, sort_keys=True))
    TABLE.write_text(json.dumps(full, indent=0, sort_keys=True))
    print("wrote", out, len(full), "entries")


if __name__ == "__main__":
    main()
