"""Builds the two native libraries in-tree.

  bigdl_amd/lib/libbigdl_host.so   g++  -O3 -fopenmp   (quantizers, re-layout, CPU matmul)
  bigdl_amd/lib/libbigdl_hip.so    hipcc --offload-arch=gfx950  (all device kernels + decode engine)

Both expose a plain C ABI and are loaded with ctypes, so neither needs the PyTorch
headers: a full rebuild is a few seconds per file and hipcc cross-compiles gfx950
without a GPU present.  Objects are cached by source mtime under build/.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent
PKG = CSRC.parent
LIB = PKG / "lib"
OBJ = PKG.parent / "build" / "obj"

HOST_SOURCES = sorted((CSRC / "host").glob("*.cpp"))
HIP_SOURCES = sorted((CSRC / "hip").glob("*.hip"))

HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fopenmp", "-march=x86-64-v3", "-Wall", "-Wno-unused-function"]
HIP_FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
             "-Wno-unused-variable", "-DNDEBUG"]
# per-file additions.  attn_prefill: MFMA accumulators in ordinary VGPRs -- the online softmax rescales them with VALU
# multiplies, and accumulators kept in AGPRs are copied out and back around every such pass (64 moves per key tile).
HIP_FILE_FLAGS = {"attn_prefill.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _newer(src: Path, dst: Path, extra=()) -> bool:
    if not dst.exists():
        return True
    t = dst.stat().st_mtime
    return any(p.stat().st_mtime > t for p in (src, *extra))


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(map(str, cmd)) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"native build failed: {cmd[0]} {cmd[-1]}")
    return r


def _compile_all(compiler, flags, sources, headers, tag, jobs, file_flags=None):
    objs, todo = [], []
    for s in sources:
        o = OBJ / tag / (s.stem + ".o")
        objs.append(o)
        if _newer(s, o, headers):
            todo.append((s, o))
    if todo:
        (OBJ / tag).mkdir(parents=True, exist_ok=True)
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(lambda so: _run([compiler, *flags, *(file_flags or {}).get(so[0].name, []), "-c", str(so[0]), "-o",
                                         str(so[1])]), todo))
    return objs, bool(todo)


def build_host(jobs=8, force=False) -> Path:
    out = LIB / "libbigdl_host.so"
    headers = list((CSRC / "host").glob("*.h"))
    objs, changed = _compile_all("g++", HOST_FLAGS, HOST_SOURCES, headers, "host", jobs)
    if changed or force or not out.exists():
        LIB.mkdir(exist_ok=True)
        _run(["g++", "-shared", "-fopenmp", "-o", str(out), *map(str, objs)])
    return out


def build_hip(jobs=8, force=False) -> Path | None:
    out = LIB / "libbigdl_hip.so"
    if not HIP_SOURCES:
        return None
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = list((CSRC / "hip").glob("*.h")) + list((CSRC / "hip").glob("*.hpp"))
    objs, changed = _compile_all(hipcc, HIP_FLAGS, HIP_SOURCES, headers, "hip", jobs, HIP_FILE_FLAGS)
    if changed or force or not out.exists():
        LIB.mkdir(exist_ok=True)
        _run([hipcc, "-shared", "--offload-arch=gfx950", "-o", str(out), *map(str, objs)])
    return out


def build_all(jobs=8, force=False):
    return build_host(jobs, force), build_hip(jobs, force)


if __name__ == "__main__":
    h, d = build_all(force="--force" in sys.argv)
    print("built", h, d)
