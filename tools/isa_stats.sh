#!/bin/bash
# Per-kernel register / scratch / occupancy summary of a HIP source compiled for gfx950 (no GPU needed).
#   tools/isa_stats.sh gennbv_amd/csrc/encoder.hip [kernel-name-substring]
# Writes the full assembly to /tmp/isa/<name>.s
set -e
src=$1; name=$(basename ${src%.hip}); mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I$(dirname $0)/../include \
    -S --cuda-device-only -o /tmp/isa/$name.s $src 2>&1 | grep -v hip-link || true
python3 - "$name" "${2:-}" <<'PY'
import re, sys
name, filt = sys.argv[1], sys.argv[2]
txt = open(f"/tmp/isa/{name}.s").read()
for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)^; Occupancy: \d+", txt, re.S | re.M):
    k, body = m.group(1), m.group(2)
    if filt and filt not in k: continue
    body = m.group(0)
    g = lambda key: (re.search(key + r"[ :]+(\d+)", body) or [None, "?"])[1]
    loads = len(re.findall(r"global_load", body)); mf = len(re.findall(r"v_mfma", body))
    print(f"{k[:60]:60s} vgpr {g('; NumVgprs'):>4} agpr {g('; NumAgprs'):>4} scratch {g('; ScratchSize'):>5} occ {g('; Occupancy'):>2} lds {g('; LDSByteSize'):>6}  loads {loads} mfma {mf}")
PY
