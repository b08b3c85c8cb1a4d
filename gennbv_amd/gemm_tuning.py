"""Library-GEMM algorithm selection for the plain Linear layers (rocBLAS / hipBLASLt via torch).

The five Linear layers of the policy (and their backward) are plain library GEMMs.  The default
heuristics pick poor kernels for these skinny shapes on gfx950 (e.g. 133 us for
[128 x 54000] x [54000 x 256], 21-67 us for the 256-wide layers); PyTorch's TunableOp times the
rocBLAS and hipBLASLt candidates once per shape and keeps the fastest (93 us / 6.7 us).
`profiles/tunableop_gfx950.csv` holds the selections measured on MI355X for the BASELINE shapes;
shapes not in the file are tuned on first use (during warm-up, before any hipGraph capture).
"""
from __future__ import annotations

import os

import torch

_DEFAULT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tunableop_gfx950.csv")
_enabled = False


def enable(results_file: str = _DEFAULT, tune_missing: bool = True) -> bool:
    global _enabled
    if _enabled or not torch.cuda.is_available():
        return _enabled
    try:
        from torch.cuda import tunable
        tunable.enable(True)
        tunable.tuning_enable(bool(tune_missing))
        tunable.set_max_tuning_duration(30)   # ms per candidate
        tunable.set_max_tuning_iterations(20)
        # results are not written at exit (the default file name would litter the cwd)
        tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "gennbv_tunableop_%d.csv" % os.getpid()))
        if results_file and os.path.exists(results_file):
            tunable.read_file(results_file)
        _enabled = True
    except Exception as ex:  # TunableOp missing in this torch build: library defaults stay in place
        print(f"[gennbv_amd] GEMM tuning unavailable: {ex!r}")
        _enabled = False
    return _enabled


def save(results_file: str) -> None:
    """Write the current selections in TunableOp's own CSV format."""
    from torch.cuda import tunable
    with open(results_file, "w") as f:
        for k, v in tunable.get_validators():
            f.write(f"Validator,{k},{v}\n")
        for row in tunable.get_results():
            f.write(",".join(str(x) for x in row) + "\n")
