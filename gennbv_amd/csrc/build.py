"""Compile libgennbv_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m gennbv_amd.csrc.build [--force]

The shared object lands next to the package (gennbv_amd/libgennbv_hip.so): it is
git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libgennbv_hip.so")
SOURCES = ["voxel.hip", "gae.hip", "envstep.hip", "encoder.hip", "linear.hip", "head.hip", "chamfer.hip", "ppo.hip"]
HEADERS = ["common.h", "conv_split.h", os.path.join("..", "..", "include", "gennbv_hip.h")]
ARCH = "gfx950"


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the gfx950 kernels)")


def flags():
    # -ffp-contract=off: the canonical fp32 order of the voxel / GAE kernels must not be
    # re-fused by the compiler (every FMA in those files is an explicit __fmaf_rn).
    return ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
            "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Every translation unit is compiled (in parallel: they are independent) unless its object is newer than the unit, the
    headers and this script; `force` recompiles all of them."""
    if not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    cc = hipcc()
    hdr_t = max(os.path.getmtime(os.path.join(HERE, h)) for h in HEADERS + [os.path.basename(__file__)])

    def compile_one(s):
        src, o = os.path.join(HERE, s), os.path.join(HERE, os.path.splitext(s)[0] + ".o")
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(src), hdr_t):
            return o
        cmd = [cc] + flags() + ["-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return o
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
