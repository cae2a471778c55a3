"""Build libovqa_hip.so (HIP kernels + C ABI) for gfx950 with plain hipcc.

In-tree build: objects under ``openvivqa_amd/csrc/_build``, the shared library
next to the sources, so that the ``gpurun`` snapshot carries it to the GPU box.
No torch dependency: the library's ABI is plain C (include/ovqa_hip.h).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libovqa_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-pass-failed",
         "-Wno-unused-result", "-ffp-contract=fast"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build the gfx950 kernels)")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    inc = os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "ovqa_hip.h")
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [inc]


STAMP = os.path.join(CSRC, "_build", "flags.stamp")


def _flag_stamp() -> str:
    """The compile flags this build would use, development extras (OVQA_EXTRA_HIPCC_FLAGS, e.g. -DOVQA_PHASE_PROBE)
    included.  Stored next to the objects: a library left behind by a probe build is NOT taken for the product build
    (mtimes alone cannot tell), and vice versa."""
    return " ".join(FLAGS + os.environ.get("OVQA_EXTRA_HIPCC_FLAGS", "").split())


def _stamp_matches() -> bool:
    try:
        with open(STAMP) as f:
            return f.read().strip() == _flag_stamp()
    except OSError:
        return False


def needs_build() -> bool:
    if not os.path.exists(LIB) or not _stamp_matches():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _sources() + _headers())


def _compile(hipcc, src, obj):
    extra = os.environ.get("OVQA_EXTRA_HIPCC_FLAGS", "").split()  # development builds (e.g. -DOVQA_PHASE_PROBE)
    cmd = [hipcc, *FLAGS, *extra, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return src, r.returncode, r.stdout + r.stderr


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    bdir = os.path.join(CSRC, "_build")
    os.makedirs(bdir, exist_ok=True)
    if not _stamp_matches():  # objects compiled with other flags (a probe build, or none recorded): recompile them all
        force = True
        if os.path.exists(STAMP):
            os.remove(STAMP)
    hdr_t = max(os.path.getmtime(p) for p in _headers())
    jobs = []
    for src in _sources():
        obj = os.path.join(bdir, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append((src, obj))
    with cf.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for src, rc, out in ex.map(lambda j: _compile(hipcc, *j), jobs):
            if verbose and out.strip():
                print(out, file=sys.stderr)
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    objs = [os.path.join(bdir, os.path.basename(s)[:-4] + ".o") for s in _sources()]
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB + ".tmp", *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        f.write(_flag_stamp() + "\n")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB)} bytes)", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
