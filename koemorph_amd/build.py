"""Build libkoemorph_hip.so in-tree with hipcc for gfx950 (MI355X).

The shared library is a plain C-ABI object (include/koemorph.h); nothing in it depends on
torch.  hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.

Every translation unit is compiled to an object of its own (koemorph_amd/lib/obj/, git-ignored), in parallel, and only
when the source, a header it includes or the flags changed; the objects are then linked.  An edit of one kernel file costs
one compile + the link instead of the whole library.
"""
from __future__ import annotations

import concurrent.futures
import glob
import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libkoemorph_hip.so")
SOURCES = ["km_host.cpp", "km_wire.cpp", "km_core.hip", "km_mel.hip", "km_generic.hip", "km_koemorph.hip", "km_kmmf.hip", "km_train.hip", "km_trainp.hip", "km_egemaps.hip", "km_data.hip", "km_api.hip"]
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "koemorph.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


_INC = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)


def _deps(path: str, seen=None) -> set:
    """The file and every project header it includes (transitively; system headers are not tracked)."""
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in _INC.findall(open(path, errors="replace").read()):
        for base in (os.path.dirname(path), CSRC, os.path.join(ROOT, "include")):
            cand = os.path.join(base, inc)
            if os.path.exists(cand):
                _deps(cand, seen)
                break
    return seen


def _flags() -> list:
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall",
             "-Wno-unused-function", "-Wno-unused-const-variable"]
    return flags + os.environ.get("KM_EXTRA_FLAGS", "").split()          # kernel A/B experiments (tools/ab_mel.sh)


def _compile_one(hipcc: str, src: str, flags: list, force: bool, verbose: bool) -> str:
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    obj = os.path.join(OBJ_DIR, f"{os.path.splitext(os.path.basename(src))[0]}.{tag}.o")
    if not force and os.path.exists(obj):
        t = os.path.getmtime(obj)
        if all(os.path.getmtime(d) <= t for d in _deps(src)):
            return obj
    tmp = obj + ".tmp%d" % os.getpid()
    cmd = [hipcc] + flags + ["-c", src, "-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n" + res.stdout + res.stderr)
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)
    os.replace(tmp, obj)
    return obj


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc, flags = _hipcc(), _flags()
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    workers = max(1, min(len(srcs), int(os.environ.get("KM_BUILD_JOBS", "0")) or (os.cpu_count() or 4)))
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        objs = list(pool.map(lambda s: _compile_one(hipcc, s, flags, force, verbose), srcs))
    tmp = LIB_PATH + ".tmp%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
