"""Build oracle/_ref/libref_bresenham.so from the REFERENCE's own ray-cast kernel text.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.c header).

The reference's only native code is a CUDA-C string inside
`gennbv/utils.py:43-197` (`bresenham3D_pycuda`).  It is plain C apart from the
CUDA qualifiers, so it compiles unchanged as host C++ behind a macro shim.
This script reads the text *where it lies* under /root/reference, writes the
translation unit to a temporary directory OUTSIDE the repo, and emits only the
shared object into `oracle/_ref/` (git-ignored, but shipped to the GPU box by
gpurun like any other built .so).  No reference source enters the repository.

Exported symbol (ours, a host loop around the reference's __global__ function):

    void ref_ray_casting_3d(const int* source_pts, const int* target_pts,
                            int* trajectory_pts, int* trajectory_lengths,
                            int num_rays, int map_size, int max_pts_per_ray)
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("GENNBV_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref")
OUT_SO = os.path.join(OUT_DIR, "libref_bresenham.so")

_SHIM_HEAD = r"""
#include <algorithm>
#include <cstdlib>
using std::max; using std::min; using std::abs;
#define __device__
#define __global__
#define __forceinline__ inline
#define __restrict__
struct _dim3 { int x, y, z; };
static thread_local _dim3 blockIdx{0,0,0}, blockDim{256,1,1}, threadIdx{0,0,0};
"""

_SHIM_TAIL = r"""
extern "C" __attribute__((visibility("default")))
void ref_ray_casting_3d(const int* source_pts, const int* target_pts, int* trajectory_pts,
                        int* trajectory_lengths, int num_rays, int map_size, int max_pts_per_ray)
{
    const int block = 256;
    const int grid = (num_rays + block - 1) / block;
    blockDim.x = block;
    for (int b = 0; b < grid; ++b)
        for (int t = 0; t < block; ++t) {
            blockIdx.x = b; threadIdx.x = t;
            ray_casting_kernel_3d(source_pts, target_pts, trajectory_pts, trajectory_lengths,
                                  num_rays, map_size, max_pts_per_ray);
        }
}
"""


def reference_kernel_text() -> str:
    path = os.path.join(REF_ROOT, "gennbv", "utils.py")
    src = open(path, "r", encoding="utf-8").read()
    m = re.search(r'kernel_code\s*=\s*"""(.*?)"""', src, flags=re.S)
    if not m:
        raise RuntimeError("kernel_code string not found in reference gennbv/utils.py")
    return m.group(1)


def build(force: bool = False) -> str | None:
    """Returns the .so path, or None when the reference is not present (GPU box)."""
    if not os.path.isdir(os.path.join(REF_ROOT, "gennbv")):
        return OUT_SO if os.path.exists(OUT_SO) else None
    os.makedirs(OUT_DIR, exist_ok=True)
    ref_py = os.path.join(REF_ROOT, "gennbv", "utils.py")
    if (not force and os.path.exists(OUT_SO)
            and os.path.getmtime(OUT_SO) >= max(os.path.getmtime(ref_py), os.path.getmtime(__file__))):
        return OUT_SO
    with tempfile.TemporaryDirectory(prefix="gennbv_ref_") as tmp:
        cpp = os.path.join(tmp, "ref_kernel_host.cpp")
        with open(cpp, "w") as f:
            f.write(_SHIM_HEAD + reference_kernel_text() + _SHIM_TAIL)
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
               "-Wno-unknown-pragmas", "-o", OUT_SO, cpp]
        subprocess.check_call(cmd)
    return OUT_SO


if __name__ == "__main__":
    out = build(force="--force" in sys.argv)
    print(out if out else "reference not present; nothing built")
