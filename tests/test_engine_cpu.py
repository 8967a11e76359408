"""CPU (no GPU): the C-ABI library loads, exports every symbol the header declares, and the
engine's tensor declarations equal the reference's state_dict key set (recorded in
tests/golden/spec_*.json by oracle/gen_golden.py).  No compute calls."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

from poco_amd import _lib, synth
from poco_amd.model import POCO, SMPL_KEYS

GOLD = Path(__file__).parent / "golden"
VARIANTS = {"hrnet_w32-pare": 3, "hrnet_w48_cls-cliff": 1, "resnet50-cliff": 1}


def test_library_exports_header_symbols():
    L = _lib.lib()
    syms = _lib.header_symbols()
    assert len(syms) >= 15
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_library_reads_no_environment_variable():
    """the shipped binary has no debug / experiment environment switch compiled in (they silently changed results
    or speed): no POCO_* string in the .so at all (the Python loader's POCO_HIP_LIB override lives in poco_amd/_lib.py), and
    `getenv` appears in csrc/ only behind a compile-time probe macro (POCO_PROBES / W4P_EXP / GH_EXP / G1_DUAL_EXP builds of tools/build_exp.sh)."""
    import re
    blob = _lib.LIB_PATH.read_bytes()
    found = sorted(set(m.decode() for m in re.findall(rb"POCO_[A-Z0-9_]{3,}", blob)))
    assert found == [], found
    for f in sorted((Path(__file__).resolve().parent.parent / "poco_amd" / "csrc").rglob("*")):
        if not f.is_file():
            continue
        depth_stack = []
        for ln in f.read_text().splitlines():
            s = ln.strip()
            if re.match(r"#\s*if", s):
                depth_stack.append(bool(re.search(r"POCO_PROBES|W4P_EXP|GH_EXP|WINO_EXP|G1_DUAL_EXP", s)))
            elif re.match(r"#\s*endif", s) and depth_stack:
                depth_stack.pop()
            code = s.split("//")[0]
            if "getenv(" in code:
                assert any(depth_stack), (f.name, ln)


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_declared_tensors_match_reference_keys(variant):
    m = POCO(backbone=variant, num_flow_layers=VARIANTS[variant], max_batch=2)
    decl = {n: (s, r) for n, s, r in m.expected_tensors()}
    spec = {n: tuple(s) for n, s in json.loads((GOLD / f"spec_{variant}.json").read_text())}
    ref_keys = set(spec)
    eng_keys = {k for k in decl if not k.startswith("smpl.")}
    assert ref_keys - eng_keys == set(), sorted(ref_keys - eng_keys)[:10]      # every checkpoint key is accepted
    assert eng_keys - ref_keys == set(), sorted(eng_keys - ref_keys)[:10]      # nothing invented
    for k, shp in spec.items():
        assert int(np.prod(decl[k][0])) == int(np.prod(shp)), (k, decl[k][0], shp)
    assert {k[5:] for k in decl if k.startswith("smpl.")} == set(SMPL_KEYS)
    # unused-by-forward tensors are tolerated, not required
    for k, (s, req) in decl.items():
        if k.endswith("num_batches_tracked") or k.startswith("flow_head.") or "classifier" in k:
            assert not req, k


def test_strict_loading_errors():
    m = POCO(backbone="resnet50-cliff", num_flow_layers=1, max_batch=1)
    with pytest.raises(_lib.PocoHipError, match="unexpected"):
        m.load_state_dict({"backbone.not_a_layer.weight": np.zeros((1,), np.float32)})
    with pytest.raises(_lib.PocoHipError, match="shape mismatch"):
        m.load_state_dict({"backbone.conv1.weight": np.zeros((64, 3, 3, 3), np.float32)})
    # same element count, different arrangement (a transposed Linear weight): refused dimension by dimension
    with pytest.raises(_lib.PocoHipError, match="shape mismatch"):
        m.load_state_dict({"head.fc2.weight": np.zeros((1, 1024 * 1024), np.float32)})
    with pytest.raises(_lib.PocoHipError, match="shape mismatch"):
        m.load_state_dict({"head.decpose.weight": np.zeros((1024, 144), np.float32)})
    m.load_state_dict({"head.init_pose": np.zeros((144,), np.float32)})          # [1,144] stored flat: size-1 dims are free
    m.load_state_dict({"head.init_pose": np.zeros((1, 144), np.float32)})
    # finalize without weights: strict missing-key report (and no GPU needed to get there)
    with pytest.raises(_lib.PocoHipError, match="missing required tensors"):
        m._finalized = False
        _lib.check(m._L.poco_finalize(m._h), "poco_finalize")


def test_state_dict_round_trip():
    """the mirror class offers state_dict() under the reference's keys."""
    from tests import util
    variant = "resnet50-cliff"
    w = util.synth_weights(variant)
    m = POCO(backbone=variant, num_flow_layers=1, max_batch=1)                # state_dict() works by default, like nn.Module's
    w["backbone.bn1.num_batches_tracked"] = np.array(7, np.int64)             # tolerated-unused entry keeps its dtype
    m.load_state_dict(w, strict=True)
    sd = m.state_dict()
    assert set(sd) == set(w) and all(np.array_equal(sd[k].numpy(), w[k]) and sd[k].numpy().dtype == w[k].dtype for k in w)
    lean = POCO(backbone=variant, num_flow_layers=1, max_batch=1, keep_state_dict=False)      # host copies explicitly dropped, no checkpoint file
    lean.load_state_dict(w, strict=True)
    with pytest.raises(_lib.PocoHipError, match="keep_state_dict"):
        lean.state_dict()
    m2 = POCO(backbone=variant, num_flow_layers=1, max_batch=1)
    assert m2.load_state_dict(sd, strict=True) == []
    assert list(sd)[:2] == [n for n, _, _ in m.expected_tensors() if n in w][:2]


def test_forward_before_finalize_is_an_error():
    m = POCO(backbone="resnet50-cliff", num_flow_layers=1, max_batch=1)
    from poco_amd.model import _Inputs, _Outputs
    rc = m._L.poco_forward(m._h, 1, C.byref(_Inputs()), C.byref(_Outputs()), None)
    assert rc != 0 and b"finalize" in m._L.poco_last_error()


def _header_struct_fields(name):
    """member names of `typedef struct { ... } <name>;` in include/poco_hip.h, in order (comments stripped)"""
    import re
    src = (Path(__file__).resolve().parent.parent / "include" / "poco_hip.h").read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = re.search(r"typedef\s+struct\s*\{([^}]*)\}\s*" + name + r"\s*;", src).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            fields.append((re.sub(r"\s+", " ", decl.rsplit(None, 1)[0].replace("*", " * ")).strip(), decl.rsplit(None, 1)[1].lstrip("*")))
    return fields


def test_binding_field_lists_match_the_header():
    """the INTEGRATION.md ctypes stub listed 14 outputs while poco_outputs_t had 15 - a binding copied from
    it handed poco_forward a struct 8 bytes short.  The header is the single source: model.py's structs AND the stub in
    INTEGRATION.md must list exactly its members, in its order, with the size word first."""
    import re
    from poco_amd import model
    root = Path(__file__).resolve().parent.parent
    doc = (root / "INTEGRATION.md").read_text()
    assert f"ABI = {model.ABI_VERSION}" in doc and "poco_abi_version() == ABI" in doc      # the stub checks the version itself
    header_version = int(re.search(r"#define POCO_ABI_VERSION (\d+)", (root / "include" / "poco_hip.h").read_text()).group(1))
    assert header_version == model.ABI_VERSION
    # the stub's two class statements are executed as written in the document
    stub = re.search(r"^class Inputs\(C\.Structure\):.*?(?=^# struct_size)", doc, flags=re.S | re.M).group(0)
    ns = {"C": C}
    exec(stub, ns)
    for cname, struct, docclass in (("poco_inputs_t", model._Inputs, "Inputs"), ("poco_outputs_t", model._Outputs, "Outputs")):
        hf = _header_struct_fields(cname)
        assert hf[0] == ("uint64_t", "struct_size"), hf[0]
        assert all("float *" in t for t, _ in hf[1:]), hf
        names = [n for _, n in hf]
        for binding in (struct, ns[docclass]):
            assert [n for n, _ in binding._fields_] == names, (binding, "field list != include/poco_hip.h")
            assert binding._fields_[0][1] is C.c_uint64 and all(t is C.c_void_p for _, t in binding._fields_[1:])
            assert C.sizeof(binding) == 8 * len(names)                 # no padding: what the library's sizeof(T) is
        assert struct().struct_size == 8 * len(names)                  # filled in by the constructor


def test_header_states_seven_ints_per_tile_configuration():
    """the header documented poco_tune_conv's `cfgs6` as "6 ints each" and poco_bench_conv2d's `cfg_used6` while the
    library reads / writes CONV_CFG_INTS = 7 per configuration - a C client written from the header under-allocated (OOB read, 4-byte
    overflow).  Every configuration parameter is named cfg*7 now, and the number in the name is the number the library uses."""
    import re
    root = Path(__file__).resolve().parent.parent
    hdr = (root / "include" / "poco_hip.h").read_text()
    n = int(re.search(r"constexpr int CONV_CFG_INTS = (\d+);", (root / "poco_amd" / "csrc" / "common.h").read_text()).group(1))
    assert n == 7
    params = re.findall(r"\bint\s*\*\s*(cfg\w*)", hdr)
    assert len(params) >= 6 and all(p.endswith(str(n)) for p in params), params
    assert "6 ints" not in hdr and "cfgs6" not in hdr and "cfg_used6" not in hdr
    assert "SEVEN ints" in hdr
    capi = (root / "poco_amd" / "csrc" / "ops_capi.hip").read_text()
    assert "cfgs6" not in capi and "cfg_used6" not in capi
    from poco_amd import ops
    assert "* 7" in Path(ops.__file__).read_text() or "7)" in Path(ops.__file__).read_text()


def test_forward_refuses_structs_without_a_valid_size_word():
    """the size word is checked before anything else is read (and before the finalize check, so no GPU is needed)"""
    from poco_amd.model import _Inputs, _Outputs
    m = POCO(backbone="resnet50-cliff", num_flow_layers=1, max_batch=1)
    fwd, err = m._L.poco_forward, m._L.poco_last_error

    class OldOutputs(C.Structure):        # an ABI-3 struct: starts with a pointer
        _fields_ = [(n, C.c_void_p) for n, _ in _Outputs._fields_[1:]]
    old = OldOutputs(pred_pose=0x7F0012345000)
    assert fwd(m._h, 1, C.byref(_Inputs()), C.cast(C.byref(old), C.POINTER(_Outputs)), None) == 1 and b"struct_size" in err()
    for bad in (0, 8, 12, 8192):
        o = _Outputs(); o.struct_size = bad
        assert fwd(m._h, 1, C.byref(_Inputs()), C.byref(o), None) == 1 and b"poco_outputs_t.struct_size" in err(), bad
        i = _Inputs(); i.struct_size = bad
        assert fwd(m._h, 1, C.byref(i), C.byref(_Outputs()), None) == 1 and b"poco_inputs_t.struct_size" in err(), bad

    class Short(C.Structure):             # a binding written from a shorter field list: accepted, the rest reads as NULL
        _fields_ = _Outputs._fields_[:-1]
    s = Short(C.sizeof(Short))
    assert fwd(m._h, 1, C.byref(_Inputs()), C.cast(C.byref(s), C.POINTER(_Outputs)), None) == 3 and b"finalize" in err()

    class Longer(C.Structure):            # a binding from a NEWER header: fine while the unknown members are NULL ...
        _fields_ = _Outputs._fields_ + [("future", C.c_void_p)]
    lg = Longer(C.sizeof(Longer))
    assert fwd(m._h, 1, C.byref(_Inputs()), C.cast(C.byref(lg), C.POINTER(_Outputs)), None) == 3
    lg.future = 0x1000                     # ... and refused as soon as one of them is asked for
    assert fwd(m._h, 1, C.byref(_Inputs()), C.cast(C.byref(lg), C.POINTER(_Outputs)), None) == 1 and b"beyond" in err()


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_tuned_table_entries_are_valid_configurations(variant):
    """poco_amd/tuned/gfx950.json against the library's own validation (no GPU needed: geometry + LDS budget):
    every entry measured for a batch size must be accepted for that batch size, and the nearest-batch fallback
    must leave every conv op with a configuration the library accepts (or the heuristic) at untuned batch sizes."""
    from poco_amd import tune
    table = tune.load_table()
    assert len(table) > 100
    m = POCO(backbone=variant, num_flow_layers=VARIANTS[variant], max_batch=1)      # declarations only
    convs = [i for i, _ in enumerate(m.ops()) if m.conv_desc(i) is not None]
    for B in (64, 32):
        hit = 0
        for i in convs:
            cfg = table.get(tune.shape_key(B, *m.conv_desc(i)[:6]))
            if cfg:
                m.set_conv_cfg(i, B, cfg)            # raises PocoHipError if the entry does not fit the op
                hit += 1
        if (variant, B) in (("hrnet_w48_cls-cliff", 64), ("hrnet_w48_cls-cliff", 32), ("hrnet_w32-pare", 32),
                            ("resnet50-cliff", 64)):
            assert hit == len(convs), (variant, B, hit, len(convs))   # bench batch sizes are fully tuned
    for B in (1, 5, 48, 128):
        n = tune.apply_table(m, B, table)
        assert 0 < n <= len(convs)
        for i in convs:
            cfg = m._L.poco_get_conv_cfg       # the active configuration is always retrievable
            c = (C.c_int * 7)()
            assert cfg(m._h, i, B, c) == 0 and c[6] in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14)
            assert m.kernel_symbol(m.conv_desc(i), tuple(c))       # bench.py names every conv's kernel (roofline.dominant): a new ALG needs its symbol


def test_c_abi_from_plain_c(tmp_path):
    """include/poco_hip.h is C (not C++): compile tests/c_abi/abi_check.c with gcc -std=c99 -Wall -Werror and let it
    drive the host-side part of the ABI through dlopen (declarations, strict loading, error strings)."""
    import os
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "abi_check"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", str(root / "include"),
                    str(root / "tests" / "c_abi" / "abi_check.c"), "-o", str(exe), "-ldl"], check=True)
    # header alone must also pass a strict C syntax check
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", str(root / "include" / "poco_hip.h")],
                   check=True)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([str(exe), str(_lib.LIB_PATH)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.startswith("ok "), (r.returncode, r.stdout, r.stderr)


@pytest.mark.parametrize("xdep", ["1", "0"])
@pytest.mark.parametrize("variant", list(VARIANTS))
def test_schedule_has_no_unsynchronised_cross_lane_read(variant, xdep, monkeypatch):
    """The schedule the engine enqueues (include/poco_hip.h poco_op_sched; csrc/engine.hip Builder::push, enqueue_program), checked
    without a GPU by a vector-clock walk: regions (phases) are separated by joins of all lanes; inside a region an op on lane a may
    read an activation that lane b != a writes in the SAME region only if a has waited (wait_mask, transitively) for an op of b at
    or after the writer.  Covers the one-join-per-module schedule with its open stage boundaries and event dependencies (default)
    and the three-join schedule (option xdep=0).  Also: every cross-lane wait names a lane that has work in the region, and the
    default schedule does contain such waits for the HRNet variants (the transition convs at the stage boundaries)."""
    m = POCO(backbone=variant, num_flow_layers=VARIANTS[variant], max_batch=2, engine_options={"xdep": int(xdep)})      # declarations only
    n = len(m.ops())
    sched = [m.op_sched(i) for i in range(n)]
    names = [o[0] for o in m.ops()]
    waits, cross_reads = 0, 0
    i = 0
    while i < n:
        ph = sched[i][0]
        j = i
        while j < n and sched[j][0] == ph:
            j += 1
        writer = {}                       # act -> list of (lane, op index) inside this region
        last = [-1] * 4                   # last op enqueued on each lane
        vc = [[-1] * 4 for _ in range(4)]   # vc[a][b]: lane a is ordered after op vc[a][b] of lane b
        snap = {}                         # op index -> vector clock of its lane right after it
        for k in range(i, j):
            _, a, mask, rd, wr = sched[k]
            assert 0 <= a < 4 and phase_monotone(sched, k)
            for b in range(4):
                if mask & (1 << b):
                    assert b != a and last[b] >= 0, (names[k], "waits for a lane without work", b)
                    waits += 1
                    vc[a][b] = max(vc[a][b], last[b])
                    for c in range(4):                                  # what lane b had seen when that op was enqueued
                        vc[a][c] = max(vc[a][c], snap[last[b]][c])
            for act, lo, hi in rd:
                for (b, w, wlo, whi) in writer.get(act, []):
                    if b != a and wlo < hi and lo < whi:
                        cross_reads += 1
                        assert vc[a][b] >= w, (variant, names[k], "reads activation", act, (lo, hi), "written by", names[w], "on lane", b,
                                               "without waiting for it")
            for act, lo, hi in wr:
                for (b, w, wlo, whi) in writer.get(act, []):          # two lanes never write overlapping channels unordered
                    if b != a and wlo < hi and lo < whi:
                        assert vc[a][b] >= w, (variant, names[k], "overwrites", act, (lo, hi), "of", names[w], "unordered")
                writer.setdefault(act, []).append((a, k, lo, hi))
            last[a] = k
            vc[a][a] = k
            snap[k] = list(vc[a])
        i = j
    if variant.startswith("hrnet") and xdep == "1":
        assert waits >= 2 and cross_reads >= 2        # transition convs read the K-merged conv's output of another lane
    if xdep == "0":
        assert waits == 0 and cross_reads == 0        # the three-join schedule never reads across lanes inside a region


def phase_monotone(sched, k):
    return k == 0 or sched[k][0] >= sched[k - 1][0]


def test_shipped_library_contains_no_experiments():
    """the split-fp16 GEMM (csrc/exp/gemm1x1h.hip) and the 3-deep rings of ALG 4 are built only by
    `python -m poco_amd.build --experiments`: in the shipped library `split_f16` is an unknown build option and no symbol of
    either is linked in."""
    import subprocess
    from tests import util
    if util.has_experiments():
        pytest.skip("POCO_HIP_LIB points at an experiment build")
    names = subprocess.run(["nm", "-C", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True).stdout
    assert "gemm1x1h" not in names
    blob = _lib.LIB_PATH.read_bytes()
    assert b"gemm1x1h" not in blob and b"split_f16" not in blob
    with pytest.raises(_lib.PocoHipError, match="split_f16"):
        POCO(backbone="resnet50-cliff", num_flow_layers=1, max_batch=1, engine_options={"split_f16": 1})


def test_apply_table_prefers_the_larger_neighbour_and_keeps_split_k_convs_small(monkeypatch):
    """without an entry for B the nearest tuned batch (log space) is used, ties to the LARGER one; an ALG 5 split-K 3x3
    entry (tuned at 1 / 4 crops, re-streams all weights per 64-pixel block) is only transferred to batch sizes up to its own."""
    from poco_amd import tune
    m = POCO(backbone="hrnet_w48_cls-cliff", num_flow_layers=1, max_batch=1)      # declarations only
    convs = [i for i, _ in enumerate(m.ops()) if m.conv_desc(i) is not None]
    i3 = next(i for i in convs if m.conv_desc(i)[4] == 3 and m.conv_desc(i)[5] == 1 and m.conv_desc(i)[0] == 14)     # a 14x14 3x3 stride-1 conv
    H, W, Cin, Cout, ks, st = m.conv_desc(i3)[:6]
    small = [1, 2, 4, 2, 14, 1, 4]
    big = [2, 3, 2, 1, 16, 2, 13]
    table = {tune.shape_key(4, H, W, Cin, Cout, ks, st): small, tune.shape_key(16, H, W, Cin, Cout, ks, st): big}
    def active(B):                                                    # (m.conv_cfg would re-apply the shipped table first)
        c = (C.c_int * 7)()
        assert m._L.poco_get_conv_cfg(m._h, i3, B, c) == 0
        return list(c)
    tune.apply_table(m, 8, table)                                     # 8 is as far from 4 as from 16 in log space
    assert active(8) == big
    splitk = [1, 1, 4, 1, 1, 1, 5]
    table = {tune.shape_key(4, H, W, Cin, Cout, ks, st): splitk}
    before = active(6)
    tune.apply_table(m, 6, table)                                     # 6 > 4: the split-K entry does not travel upwards ...
    assert active(6) == before
    before3 = active(3)
    tune.apply_table(m, 3, table)                                     # ... but downwards it does (if the library accepts it for the op)
    assert active(3) in (splitk, before3)
