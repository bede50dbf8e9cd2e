"""Build libkoemorph_hip.so in-tree with hipcc for gfx950 (MI355X).

The shared library is a plain C-ABI object (include/koemorph.h); nothing in it depends on
torch.  hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
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


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function",
           "-Wno-unused-const-variable"]
    cmd += os.environ.get("KM_EXTRA_FLAGS", "").split()          # kernel A/B experiments (tools/ab_mel.sh)
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    tmp = LIB_PATH + ".tmp%d" % os.getpid()
    cmd += ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
