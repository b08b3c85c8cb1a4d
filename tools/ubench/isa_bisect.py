#!/usr/bin/env python3
"""ISA-level bisection of the 'compiled MFMA loop is slow at >= 2 waves per SIMD' effect (profiles/r01_notes.md).

Compiles isa_bisect_kernel.hip to assembly, writes edited variants, assembles each to a code object:
  k0  as compiled (operands loaded from memory, descending s_waitcnt vmcnt(N) between the MFMAs)      50 / 58 cycles per MFMA at 2 / 4 waves
  k1  in-loop waits deleted, one s_waitcnt vmcnt(0) in front of the loop                               50 / 49
  k2  every in-loop wait turned into vmcnt(0)                                                          50 / 57
  k3  k0 + s_waitcnt vmcnt(0) in front of the loop                                                     50 / 49
  k4  k3 + v_mov_b32 vN, vN for every operand register                                                 33 / 41
  k6  operands loaded straight into AGPRs, MFMAs read a[..]                                            50 / 58
  k7  operands passed through LDS (ds_write + ds_read) after the loads landed                          34 / 34
  k8  k3 + v_mov only for the SrcA registers                                                           34 / 41
  k9  k4 with v_pk_mov_b32 pairs                                                                       34 / 41
Run on the GPU box:  hipcc -O2 -o hsaco_host hsaco_host.cpp && ./hsaco_host k0.hsaco k1.hsaco ...
usage: python isa_bisect.py <outdir>      (needs /opt/rocm: hipcc, clang, ld.lld)
"""
import os, re, subprocess, sys

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
here = os.path.dirname(os.path.abspath(__file__))
LLVM = "/opt/rocm/lib/llvm/bin"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-o", f"{out}/k0.s",
                       f"{here}/isa_bisect_kernel.hip"])
s = open(f"{out}/k0.s").read().split("\n")
li = [i for i, l in enumerate(s) if l.startswith(".LBB0_2:")][0]
le = [i for i, l in enumerate(s) if "s_cbranch_scc1 .LBB0_2" in l][0]
srcs = [re.match(r"\s*v_mfma_f32_16x16x4_f32 v\[\d+:\d+\], v(\d+), v(\d+),", l) for l in s[li:le + 1]]
aops = sorted({int(m.group(1)) for m in srcs if m})
ops = sorted({int(m.group(k)) for m in srcs if m for k in (1, 2)})
amap = {r: i for i, r in enumerate(ops)}


def emit(name, lines):
    open(f"{out}/{name}.s", "w").write("\n".join(lines))
    subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", f"{out}/{name}.s", "-o",
                           f"{out}/{name}.o"])
    subprocess.check_call([f"{LLVM}/ld.lld", "-shared", f"{out}/{name}.o", "-o", f"{out}/{name}.hsaco"])


W0 = ["\ts_waitcnt vmcnt(0)"]
emit("k0", s)
emit("k1", s[:li] + W0 + [l for l in s[li:le + 1] if "s_waitcnt vmcnt" not in l] + s[le + 1:])
emit("k2", s[:li] + [re.sub(r"vmcnt\(\d+\)", "vmcnt(0)", l) for l in s[li:le + 1]] + s[le + 1:])
emit("k3", s[:li] + W0 + s[li:])
emit("k4", s[:li] + W0 + [f"\tv_mov_b32_e32 v{r}, v{r}" for r in ops] + s[li:])
emit("k8", s[:li] + W0 + [f"\tv_mov_b32_e32 v{r}, v{r}" for r in aops] + s[li:])


def to_agpr(l):
    m = re.match(r"(\s*global_load_dword )v(\d+)(,.*)", l)
    if m and int(m.group(2)) in amap:
        return f"{m.group(1)}a{amap[int(m.group(2))]}{m.group(3)}"
    m = re.match(r"(\s*v_mfma_f32_16x16x4_f32 v\[\d+:\d+\], )v(\d+), v(\d+)(, v\[\d+:\d+\].*)", l)
    if m:
        return f"{m.group(1)}a{amap[int(m.group(2))]}, a{amap[int(m.group(3))]}{m.group(4)}"
    return l


nv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", "\n".join(s)).group(1))
acc = int(re.search(r"\.amdhsa_accum_offset (\d+)", "\n".join(s)).group(1))
k6 = [to_agpr(l) for l in s[:le + 1]] + s[le + 1:]
k6 = [l.replace(f".amdhsa_next_free_vgpr {nv}", f".amdhsa_next_free_vgpr {acc + 32}").replace(".agpr_count:     0", ".agpr_count:     32")
      .replace(f".vgpr_count:     {nv}", f".vgpr_count:     {acc + 32}") for l in k6]
emit("k6", k6)
pre = W0 + [f"\tv_lshlrev_b32_e32 v{acc - 2}, 2, v0"]
for r in ops:
    pre += [f"\tds_write_b32 v{acc - 2}, v{r}", "\ts_waitcnt lgkmcnt(0)", f"\tds_read_b32 v{r}, v{acc - 2}", "\ts_waitcnt lgkmcnt(0)"]
k7 = s[:li] + pre + s[li:]
k7 = [l.replace(".amdhsa_group_segment_fixed_size 0", ".amdhsa_group_segment_fixed_size 4096").replace(".group_segment_fixed_size: 0",
      ".group_segment_fixed_size: 4096").replace(f".amdhsa_next_free_vgpr {nv}", f".amdhsa_next_free_vgpr {acc}")
      .replace(f".vgpr_count:     {nv}", f".vgpr_count:     {acc}") for l in k7]
emit("k7", k7)
print("wrote", sorted(f for f in os.listdir(out) if f.endswith(".hsaco")))
