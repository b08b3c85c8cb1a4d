"""Build-time check of the hand-counted vector-memory waits in k_conv12_fwd_split<true> (csrc/conv_split.h).

The training forward's staging waves issue their int8 input requests as `asm volatile("global_load_dwordx4 ... nt")` (ldu4_nt_async) and
wait for them with literal `s_waitcnt vmcnt(N)` statements (wait_vm_keep1): the compiler does not know that the destination registers
are in flight.  This script compiles csrc/encoder.hip to assembly (no GPU needed), walks the control-flow graph of the kernel from
every such request and asserts that on EVERY path the first instruction that names one of the request's destination registers -- as a
source or as a destination: a copy, a spill, a reuse -- comes after an `s_waitcnt vmcnt(N)` that RETIRES the request: N <= the number of
vector-memory instructions issued behind the request on that path (a wave's vector-memory operations are counted in issue order).
A compiler that moves, copies or re-allocates such a register inside the window, or a change of kTilesPerWave / kStageWaves that makes a
literal count too loose, fails the check.

    python tools/check_async_regs.py [--asm /tmp/isa/encoder.s] [--kernel k_conv12_fwd_splitILb1E] [--max-keep 7]
"""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_asm(out: str) -> str:
    src = os.path.join(ROOT, "gennbv_amd", "csrc", "encoder.hip")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
           "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out


def regs_of(text: str) -> set:
    """Vector registers an instruction names (v7, v[4:7])."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def check(asm_path: str, kernel: str, max_keep: int):
    lines = open(asm_path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kernel) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    in_asm = [False] * len(body)
    flag = False
    for i, l in enumerate(body):
        if ";;#ASMSTART" in l:
            flag = True
        in_asm[i] = flag
        if ";;#ASMEND" in l:
            flag = False
    loads = [i for i, l in enumerate(body) if in_asm[i] and re.search(r"global_load_dwordx4\s+v\[\d+:\d+\].*\bnt\b", l)]
    problems, n_paths = [], 0
    for li in loads:
        dst = regs_of(body[li].split(",")[0])
        # the address registers may overlap the destination (v[22:25] <- v[22:23]): only LATER instructions matter
        stack, seen = [(li + 1, False, 0)], set()
        while stack:
            pos, waited, younger = stack.pop()
            while pos < len(body):
                if (pos, waited, younger) in seen:
                    break
                seen.add((pos, waited, younger))
                l = body[pos].split(";")[0].strip() if not body[pos].lstrip().startswith(";;#") else ""
                if not l or l.endswith(":") or l.startswith("."):
                    pos += 1
                    continue
                m = re.match(r"s_waitcnt\s+.*?vmcnt\((\d+)\)", l)
                if m and int(m.group(1)) <= younger:
                    waited = True  # at most N operations stay in flight, and `younger` of them were issued behind the request
                if regs_of(l) & dst and not l.startswith("s_waitcnt"):
                    n_paths += 1
                    if not waited:
                        problems.append(f"line {start + pos + 1}: `{l}` names a register of the request at line {start + li + 1} "
                                        f"(`{body[li].strip()}`) before its wait")
                    break
                if l.startswith("s_endpgm"):
                    break
                if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", l):
                    younger = min(younger + 1, max_keep + 1)
                b = re.match(r"(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)", l)
                if b:
                    tgt = labels[b.group(2)]
                    if b.group(1) == "s_branch":
                        pos = tgt
                        continue
                    stack.append((tgt, waited, younger))
                pos += 1
    return loads, n_paths, problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=None)
    ap.add_argument("--kernel", default="k_conv12_fwd_splitILb1E")
    ap.add_argument("--max-keep", type=int, default=7)
    a = ap.parse_args()
    asm = a.asm or compile_asm("/tmp/isa/encoder_check.s")
    loads, n_paths, problems = check(asm, a.kernel, a.max_keep)
    print(f"{a.kernel}: {len(loads)} opaque requests, {n_paths} first-use sites reached, {len(problems)} problem(s)")
    for p in problems:
        print("  " + p)
    if not loads:
        print("  no opaque request found: the kernel no longer uses ldu4_nt_async?")
        return 2
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
