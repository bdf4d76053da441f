"""Builds libngcf_hip.so (hipcc, gfx950) in-tree, next to this file.

The shared library is plain C ABI (include/ngcf_hip.h); it has no torch dependency, so a plain
`hipcc -shared` is the whole build.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
SOURCES = [os.path.join(CSRC, f) for f in ("csr.hip", "spmm.hip", "spmm_swept.hip", "dense.hip", "ops.hip", "backward.hip")]
HEADERS = [os.path.join(os.path.dirname(PKG_DIR), "include", "ngcf_hip.h"), os.path.join(CSRC, "common.h"),
           os.path.join(CSRC, "spmm_device.h")]
LIB = os.path.join(PKG_DIR, "libngcf_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libngcf_hip.so cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip -> libngcf_hip.so for gfx950 (one hipcc call).  Returns the library path."""
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function", "-o", LIB + ".tmp"] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
