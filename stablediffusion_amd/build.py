"""Build libsd_engine.so (HIP/C++ for gfx950) and the oracle's compiled helpers, in-tree.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU; the resulting .so is git-ignored but
travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsd_engine.so")
SOURCES = ["igemm.hip", "igemm2.hip", "igemm3.hip", "wsgemm.hip", "pgemm.hip", "ffn.hip", "edge.hip", "norm.hip", "attention.hip", "misc.hip", "probe.hip", "runtime.cpp", "unet.cpp", "vae.cpp", "clip.cpp", "capi.cpp"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast", "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "sd_engine.h"))
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, "-x", "hip", "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(OUT, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
