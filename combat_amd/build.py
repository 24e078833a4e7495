"""Build combat_amd/libcombat_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m combat_amd.build [--force]

The shared library exports exactly the C ABI of include/combat_hip.h; it has no torch or
Python dependency.  Objects are cached under combat_amd/csrc/_obj keyed on source mtimes.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libcombat_hip.so")
SOURCES = ["capi.cpp", "plan.cpp", "comm.cpp", "conv_gemm.hip", "conv3x3.hip", "conv3x3_dma.hip", "conv_gather_dma.hip", "conv_k8.hip", "conv_wgrad.hip", "conv_wgrad3x3.hip", "conv_wgrad3x3_dma.hip", "norm.hip", "elementwise.hip", "trigger.hip", "head.hip", "warp.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result", "-Wno-inline-asm",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get("COMBAT_HIPCC_FLAGS", "").split()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(ROOT, "include", "combat_hip.h"), os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "conv_common.hpp"), os.path.join(CSRC, "conv_dma_epilogue.hpp"), os.path.join(CSRC, "plan.hpp")]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
