"""profiles/rNN_voxel_pmc.txt (tools/collect_profiles.sh) -> profiles/rNN_voxel_traffic.json: HBM bytes per
gnbv_update_occ_grid launch from the FETCH_SIZE / WRITE_SIZE passes (KB units; FETCH_SIZE doubled on gfx950 as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes for wide coalesced streaming reads).

    python tools/pmc_to_traffic.py profiles/r01_voxel_pmc.txt [flat|compact] > profiles/r01_voxel_traffic.json
(second argument: the observation rows the microbenchmark wrote -- compact = int8 tri-class rows only, the bench default)
"""
import hashlib, json, os, re, sys

fetch, write, cur, kern = {}, {}, None, None
for line in open(sys.argv[1]):
    if line.startswith("## --pmc"):
        cur = line.split("--pmc", 1)[1].split()
        continue
    m = re.match(r"^(void )?(k_\w+)", line.strip()) if not line.startswith(" ") else None
    if m:
        kern = m.group(2)
        continue
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", line)
    if m and kern:
        (fetch if m.group(1) == "FETCH_SIZE" else write)[kern] = float(m.group(2))
fb = {k: int(v * 1024 * 2) for k, v in fetch.items()}
wb = {k: int(v * 1024) for k, v in write.items()}
out = {"_comment": "HBM traffic of one gnbv_update_occ_grid call of the bench configuration (256 envs x 240x320 x 64^3; binary GT -> "
                   "bit-packed gt/scanned; 1-byte coded probability grid when the kernel list shows k_grid_update_coded): separate "
                   "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, KB units). FETCH_SIZE is doubled per MI355X_MICROARCH.md "
                   "section HBM (gfx950 reports half the bytes of wide coalesced streaming reads); WRITE_SIZE as reported.",
       "source_sha256": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gennbv_amd", "csrc", "voxel.hip"), "rb").read()).hexdigest(),
       "config": {"envs": 256, "height": 240, "width": 320, "grid": 64, "obs": sys.argv[2] if len(sys.argv) > 2 else "compact"},
       "fetch_bytes": fb, "write_bytes": wb, "traffic_bytes_per_launch": sum(fb.values()) + sum(wb.values()),  # (kernels only; the mask fill node is not a kernel-trace row)
       "algorithmic_bytes_per_launch": 256 * (240 * 320 * 8 + 64 ** 3 * 4 * 6 + 200)}
print(json.dumps(out, indent=2))
