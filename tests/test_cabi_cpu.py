"""CPU: the C-ABI library loads and exports every symbol include/gennbv_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "gennbv_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gnbv_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from gennbv_amd import _lib
    from gennbv_amd.csrc import build
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gennbv_hip.h but not exported"
    # the binding table covers the whole header, nothing more, nothing less
    assert sorted(_lib.SIGNATURES) == syms
    bound = _lib.load()
    assert bound.gnbv_abi_version() == 5
    assert bound.gnbv_build_arch() == b"gfx950"
    assert bound.gnbv_voxel_workspace_bytes(256, 64) == 2 * 256 * 8192 * 4


def test_product_path_refuses_cpu_tensors():
    import torch
    from gennbv_amd import _lib, utils
    with pytest.raises(_lib.GennbvHipError):
        utils.grid_occupancy_tri_cls(torch.zeros(2, 4, 4, 4))
    with pytest.raises(_lib.GennbvHipError):
        utils.bresenham3D_pycuda(torch.zeros(1, 3), torch.zeros(4, 3), 16)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "gennbv_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
                assert "liboracle" not in src, fn


@pytest.mark.parametrize("cc,lang", [("gcc", "c"), ("g++", "c++")])
def test_public_header_compiles_as_c_and_cxx(cc, lang, tmp_path):
    """include/gennbv_hip.h is the drop-in boundary: it must parse as plain C and as C++, and a consumer must be able
    to name every declared function and fill the structs."""
    import shutil
    import subprocess
    if shutil.which(cc) is None:
        pytest.skip(cc + " not installed")
    src = tmp_path / ("use." + ("c" if lang == "c" else "cpp"))
    calls = "\n".join(f"    p[{i}] = (void *){s};" for i, s in enumerate(declared_symbols()))
    src.write_text('#include "gennbv_hip.h"\n#include <stddef.h>\nint main(void) {\n    void *p[128];\n' + calls +
                   "\n    GnbvEnvPost e; e.ring_state = NULL; e.ring_len = 100; e.episode_info = NULL; e.episode_state = NULL;\n"
                   "    GnbvEncoderParams q; q.grid_i8 = NULL;\n    return p[0] == NULL && e.ring_len == 0 && q.grid_i8 != NULL;\n}\n")
    subprocess.check_call([cc, "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-but-set-variable", "-x", lang,
                           "-I", os.path.join(ROOT, "include"), str(src)])
