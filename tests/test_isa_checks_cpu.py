"""Build-time ISA checks (no GPU): properties of the compiled gfx950 code that the hand-counted waits in csrc/conv_split.h rely on."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc needed to produce the assembly")
def test_training_forward_async_requests_are_waited_for_before_any_use(tmp_path):
    # tools/check_async_regs.py: every opaque int8 request of k_conv12_fwd_split<true> is retired by a counted s_waitcnt before any
    # instruction names its destination registers, on every control-flow path (a compiler copy / spill / re-allocation inside
    # the window, or a literal count made too loose by a changed tile split, must fail the build, not a training run)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_regs.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 problem(s)" in r.stdout
