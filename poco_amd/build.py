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


def _sources() -> list[Path]:
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _digest(src: Path) -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    h.update(src.read_bytes())
    for hdr in sorted(list(CSRC.glob("*.h")) + list((ROOT.parent / "include").glob("*.h"))):
        h.update(hdr.read_bytes())
    return h.hexdigest()


def _compile(src: Path, force: bool) -> tuple[Path, bool]:
    obj = OBJDIR / (src.stem + ".o")
    stamp = OBJDIR / (src.stem + ".sha")
    dig = _digest(src)
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return obj, False
    cmd = [_hipcc(), *FLAGS, "-x", "hip", "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(dig)
    return obj, True


def build(force: bool = False, verbose: bool = True) -> Path:
    OBJDIR.mkdir(parents=True, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
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
    build(force="--force" in sys.argv)
