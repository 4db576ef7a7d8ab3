"""Build libmmpl_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m mmpl_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmmpl_hip.so")
SOURCES = ["device_state.hip", "gemm.hip", "attention.hip", "attn_w64.hip", "elementwise.hip", "vae_kernels.hip", "vae.hip", "t5.hip", "i2v.hip", "probe.hip", "api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed"] + os.environ.get("MMPL_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "mmpl_hip.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    # the compile flags are part of the staleness key: a library built with timing / ablation defines
    # (MMPL_EXTRA_HIPCC_FLAGS, tools/*_sweep.sh) must never survive into a later plain build, whatever the mtimes say
    stamp = os.path.join(objdir, "flags.stamp")
    flags_key = " ".join(FLAGS)
    if not os.path.exists(stamp) or open(stamp).read() != flags_key:
        force = True
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print("[mmpl_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    if jobs and os.path.exists(stamp):
        os.remove(stamp)                   # an interrupted rebuild leaves no stamp -> the next build starts over
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[mmpl_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(flags_key)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
