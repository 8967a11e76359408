"""Build libpoco_hip.so (hipcc, gfx950) in-tree.

    python -m poco_amd.build [--force]

The shared library is the C-ABI boundary declared in include/poco_hip.h.  It is built into
poco_amd/lib/ so it travels with the repo snapshot to the GPU box (no JIT cache involved).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIBDIR = ROOT / "lib"
OBJDIR = LIBDIR / "obj"
LIB = LIBDIR / "libpoco_hip.so"
ARCH = "gfx950"

FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}",
         "-Wno-unused-result", "-DNDEBUG"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def _sources(experiments: bool = False) -> list[Path]:
    """The shipped library is csrc/*.hip.  csrc/exp/*.hip (labelled experiments that can never be the headline: the split-fp16 1x1
    GEMM) and the `#if POCO_EXPERIMENTS` variants (3-deep rings of ALG 4) are compiled only by `python -m poco_amd.build
    --experiments`, into lib/exp/libpoco_hip_experiments.so (select it with POCO_HIP_LIB)."""
    srcs = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp"))
    if experiments:
        srcs += list((CSRC / "exp").glob("*.hip"))
    return sorted(srcs)


def _digest(src: Path) -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    h.update(src.read_bytes())
    for hdr in sorted(list(CSRC.glob("*.h")) + list((ROOT.parent / "include").glob("*.h"))):
        h.update(hdr.read_bytes())
    return h.hexdigest()


def _compile(src: Path, force: bool, experiments: bool = False) -> tuple[Path, bool]:
    objdir = OBJDIR / "exp" if experiments else OBJDIR
    objdir.mkdir(parents=True, exist_ok=True)
    obj = objdir / (src.stem + ".o")
    stamp = objdir / (src.stem + ".sha")
    dig = _digest(src)
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return obj, False
    cmd = [_hipcc(), *FLAGS, *(["-DPOCO_EXPERIMENTS=1"] if experiments else []), "-x", "hip", "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(dig)
    return obj, True


def build(force: bool = False, verbose: bool = True, experiments: bool = False) -> Path:
    OBJDIR.mkdir(parents=True, exist_ok=True)
    srcs = _sources(experiments)
    LIB = (LIBDIR / "exp" / "libpoco_hip_experiments.so") if experiments else globals()["LIB"]
    LIB.parent.mkdir(parents=True, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, experiments), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(ch for _, ch in results)
    if rebuilt or not LIB.exists():
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[poco_amd.build] linked {LIB} ({LIB.stat().st_size/1e6:.1f} MB)")
    elif verbose:
        print(f"[poco_amd.build] {LIB} up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, experiments="--experiments" in sys.argv)
