"""Builds libngcf_hip.so (hipcc, gfx950) in-tree, next to this file.

The shared library is plain C ABI (include/ngcf_hip.h); it has no torch dependency, so a plain
`hipcc -shared` is the whole build.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
SOURCES = [os.path.join(CSRC, f) for f in ("csr.hip", "spmm.hip", "spmm_swept.hip", "dense.hip", "ops.hip", "backward.hip", "comm.hip", "hostrng.hip")]
HEADERS = [os.path.join(os.path.dirname(PKG_DIR), "include", "ngcf_hip.h"), os.path.join(CSRC, "common.h"),
           os.path.join(CSRC, "spmm_device.h")]
LIB = os.path.join(PKG_DIR, "libngcf_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libngcf_hip.so cannot be built")


FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unused-function", "-ldl"]
FLAGS += os.environ.get("NGCF_EXTRA_HIPCC_FLAGS", "").split()      # lab builds (e.g. -DNGCF_DC=8); part of the source hash
STAMP = LIB + ".srchash"       # content hash of the sources the library was built from (travels with the .so)


def source_hash() -> str:
    """sha256 over the sources, headers and flags: staleness is decided by content, not by mtimes (a snapshot of the
    tree on another machine keeps contents, not necessarily timestamps)."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in SOURCES + HEADERS:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip -> libngcf_hip.so for gfx950 (one hipcc call).  Returns the library path.
    Safe under `torch.distributed.run`: ranks serialise on a file lock, the first one builds into a temp file of its
    own and renames it into place, the others find the library up to date."""
    if not force and not needs_build():
        return LIB
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():        # another process built it while this one waited
                return LIB
            tmp = f"{LIB}.tmp.{os.getpid()}"
            cmd = [_hipcc()] + FLAGS + ["-o", tmp] + SOURCES
            if verbose:
                print(" ".join(cmd), flush=True)
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
            os.replace(tmp, LIB)
            with open(STAMP + ".tmp", "w") as f:
                f.write(source_hash() + "\n")
            os.replace(STAMP + ".tmp", STAMP)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
