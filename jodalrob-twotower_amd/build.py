"""Builds libtwotower_hip.so (HIP, gfx950 only) in-tree with hipcc.  No GPU needed to compile.

    python jodalrob-twotower_amd/build.py [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
OUT = PKG / "libtwotower_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         f"-I{INCLUDE}", f"-I{CSRC}"] + os.environ.get("TT_EXTRA_HIPCC_FLAGS", "").split()      # (measurement builds: -DTT_TAIL_STAMPS)


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> Path:
    srcs = sorted(CSRC.glob("*.hip"))
    hdrs = sorted(CSRC.glob("*.h")) + sorted(INCLUDE.glob("*.h"))
    objdir = PKG / "build"
    objdir.mkdir(exist_ok=True)

    def compile_one(src: Path):
        obj = objdir / (src.stem + ".o")
        if force or _stale(obj, [src, *hdrs]):
            cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT), *map(str, objs)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
