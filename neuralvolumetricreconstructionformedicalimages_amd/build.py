"""Build recipe for libnaf_hip.so: plain hipcc, gfx950 only, in-tree output.

The library replaces the JIT `torch.utils.cpp_extension.load` of the reference
(src/encoder/hashencoder/backend.py:6-16); it has no torch/pybind dependency, so the prebuilt `.so`
travels with the repo snapshot and is loaded with ctypes (see _abi.py).
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libnaf_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# Per-source additions.  render_fused.hip (MLP + scatter kernels): the SLP vectoriser pairs neighbouring fp32 multiplies / adds into
# v_pk_*_f32, which cost more issue time beside MFMAs than the two scalar instructions they replace (MI355X_MICROARCH.md, "price of one
# filler beside MFMAs"); same-box A/B at 1 024 rays: 0.272 -> 0.268 ms per step, MLP backward at 65 536 rays 0.687 -> 0.677 ms.
EXTRA_FLAGS = {"render_fused.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libnaf_hip.so cannot be built (ROCm toolchain required)")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(PKG, "..", "include", "naf_hip.h")]


def source_fingerprint():
    """sha256 over the kernel sources and the ABI header: names what a measurement (profiles/pmc_traffic.json) was taken on."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(_deps(), key=os.path.basename):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _deps())


def build_library(force=False, verbose=False, jobs=None):
    """Compile every csrc/*.hip for gfx950 and link lib/libnaf_hip.so.  Returns the library path."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    newest_hdr = max(os.path.getmtime(p) for p in _deps() if not p.endswith(".hip"))

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_hdr):
            return obj
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    with cf.ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
