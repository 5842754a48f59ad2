"""Build libscanpy_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

`python -m scanpy_amd._build` or `__graft_entry__.build()`.  Objects and the shared library
are written in-tree under scanpy_amd/_lib/ (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "_lib"
LIBNAME = "libscanpy_amd.so"
ARCH = "gfx950"

SOURCES = ["capi.cpp", "hostio.cpp", "knn.hip", "fuzzy.hip", "pca.hip", "gram.hip", "dense.hip", "leiden.hip", "preprocess.hip", "umap.hip"]
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-pass-failed",
            "-ffp-contract=off"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libscanpy_amd.so)")


def _digest(paths: list[Path]) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(CXXFLAGS).encode())
    return h.hexdigest()


def _compile(src: Path, obj: Path, headers: list[Path], verbose: bool) -> None:
    stamp = obj.with_suffix(".sha")
    dig = _digest([src, *headers])
    if obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return
    cmd = [hipcc(), *CXXFLAGS, "-x", "hip", "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(dig)


def build(verbose: bool = True, force: bool = False) -> Path:
    LIBDIR.mkdir(exist_ok=True)
    if force:
        for f in LIBDIR.glob("*"):
            if f.is_file():
                f.unlink()
    headers = sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "scanpy_amd.h"]
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    objs = [LIBDIR / (s.stem + ".o") for s in srcs]
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        list(ex.map(lambda so: _compile(so[0], so[1], headers, verbose), zip(srcs, objs)))
    lib = LIBDIR / LIBNAME
    newest = max(o.stat().st_mtime for o in objs)
    if not lib.exists() or lib.stat().st_mtime < newest:
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(lib)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
