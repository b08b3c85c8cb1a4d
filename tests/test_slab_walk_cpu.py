"""CPU: the closed form behind k_ray_slab (gennbv_amd/csrc/voxel.hip, init_ray_slab) against the oracle's sequential walk.

The reference's integer Bresenham (gennbv/utils.py:48-167, restated in oracle/oracle.c orc_bresenham_walk) visits point j of a
ray at a_j = a_0 + s_a j on the dominant axis and b_j = b_0 + s_b floor((2 d_b j + d_a) / (2 d_a)) on a minor axis, with error
term p_j = 2 d_b (j + 1) - d_a - 2 d_a nb_j.  k_ray_slab walks a ray one slab of x-planes [X0, X1) at a time: the first and the
last point inside the slab and the walker's state there follow from that form; here the same arithmetic in Python (small grids,
pure loops) must reproduce the oracle's trajectory -- every in-grid point exactly once, in the slab that owns its x."""
import numpy as np
import pytest

from oracle import oracle as orc


def _slab_points(src, tgt, g, x0, x1):
    """Points of the ray src -> tgt with x in [x0, x1) that lie in the grid: init_ray_slab + RayWalk::step, line by line."""
    out = []
    if max(src[0], tgt[0]) < x0 or min(src[0], tgt[0]) >= x1:
        return out
    d0, d1, d2 = (abs(tgt[i] - src[i]) for i in range(3))
    dm = max(d0, d1, d2)
    ax = dm == d0
    ay = (not ax) and dm == d1
    pa0 = src[0] if ax else (src[1] if ay else src[2])
    pb0 = src[1] if ax else src[0]
    pc0 = src[2] if (ax or ay) else src[1]
    ta = tgt[0] if ax else (tgt[1] if ay else tgt[2])
    tb = tgt[1] if ax else tgt[0]
    tc = tgt[2] if (ax or ay) else tgt[1]
    da, db, dc = dm, (d1 if ax else d0), (d2 if (ax or ay) else d1)
    sa, sb, sc = (1 if pa0 < ta else -1), (1 if pb0 < tb else -1), (1 if pc0 < tc else -1)
    sx = 1 if src[0] < tgt[0] else -1
    klo = max((x0 - src[0]) if sx > 0 else (src[0] - (x1 - 1)), 0)
    khi = min(((x1 - 1) - src[0]) if sx > 0 else (src[0] - x0), d0)
    jlo, jhi = 0, da
    if ax:
        jlo, jhi = klo, khi
    elif d0 > 0:
        if klo >= 1:
            jlo = (2 * da * klo - da + 2 * d0 - 1) // (2 * d0)
        jhi = min(da, (2 * da * (khi + 1) - da + 2 * d0 - 1) // (2 * d0) - 1)
    if jlo > jhi:
        return out
    nb = (2 * db * jlo + da) // (2 * da) if da > 0 else 0
    nc = (2 * dc * jlo + da) // (2 * da) if da > 0 else 0
    pa, pb, pc = pa0 + sa * jlo, pb0 + sb * nb, pc0 + sc * nc
    p1, p2 = 2 * db * (jlo + 1) - da - 2 * da * nb, 2 * dc * (jlo + 1) - da - 2 * da * nc
    for _ in range(jhi - jlo + 1):
        xyz = (pa, pb, pc) if ax else ((pb, pa, pc) if ay else (pb, pc, pa))
        if all(0 <= v < g for v in xyz):
            assert x0 <= xyz[0] < x1
            out.append(xyz)
        ib, ic = p1 >= 0, p2 >= 0
        pb += sb if ib else 0
        pc += sc if ic else 0
        pa += sa
        p1 += (2 * db - 2 * da) if ib else 2 * db
        p2 += (2 * dc - 2 * da) if ic else 2 * dc
    return out


@pytest.mark.parametrize("g,planes", [(5, 1), (16, 3), (16, 16), (20, 2), (33, 32), (33, 7)])
def test_slab_walk_equals_the_oracle_trajectory(g, planes):
    rs = np.random.RandomState(100 * g + planes)
    n = 400
    # sources inside the grid, outside it (the reference emits the in-grid points only), on its faces; targets always inside
    src = np.where(rs.rand(n, 1) < 0.5, rs.randint(0, g, (n, 3)), rs.randint(-g, 2 * g, (n, 3))).astype(np.int32)
    tgt = rs.randint(0, g, (n, 3)).astype(np.int32)
    tgt[:40, 0] = np.clip(src[:40, 0], 0, g - 1)           # x constant along the ray
    tgt[40:60] = np.clip(src[40:60], 0, g - 1)             # single-point / clamped rays
    for s, t in zip(src, tgt):
        traj, lens = orc.bresenham3d(s, t[None, :], g)
        ref = [tuple(int(v) for v in p) for p in traj[0, :lens[0]]]
        got = []
        for x0 in range(0, g, planes):
            got += _slab_points([int(v) for v in s], [int(v) for v in t], g, x0, min(g, x0 + planes))
        assert len(set(got)) == len(got), (s, t)
        assert sorted(got) == sorted(ref), (s, t, planes)


def test_reciprocal_divisions_of_the_slab_setup_are_exact():
    """udiv_rcp / perm_index (voxel.hip): floor(n / d) from a float reciprocal estimate with a two-sided correction, n < 2^22;
    (r P) mod cnt in 32-bit arithmetic from floor((2^32 - 1) / cnt) with one correction."""
    rs = np.random.RandomState(7)
    for d in (1, 2, 3, 5, 16, 20, 33, 127, 128, 400, 1089, 4096, 16129, 16384):
        for nudge in (-1, 0, 1):  # the hardware estimate is good to 1 ulp either way
            rc = np.float32(1.0) / np.float32(d)
            if nudge:
                rc = np.nextafter(rc, np.float32(np.inf if nudge > 0 else -np.inf), dtype=np.float32)
            nn = rs.randint(0, 1 << 22, 20000).astype(np.int64)
            q = (nn.astype(np.float32) * rc).astype(np.int64)
            r = nn - q * d
            q = np.where(r < 0, q - 1, q)
            r = np.where(r < 0, r + d, r)
            q = np.where(r >= d, q + 1, q)
            assert np.array_equal(q, nn // d), (d, nudge)
    for cnt in list(rs.randint(1, 542000, 300)) + [1, 2, 7907, 7919, 76800, 542000]:
        p = 7919 if cnt % 7919 else 7907
        assert cnt * p < 2 ** 32
        r = rs.randint(0, cnt, 100).astype(np.uint64)
        x = r * np.uint64(p)
        est = (x * np.uint64(0xFFFFFFFF // int(cnt))) >> np.uint64(32)
        rem = x - est * np.uint64(cnt)
        rem = np.where(rem >= cnt, rem - np.uint64(cnt), rem)
        assert np.array_equal(rem, (r * np.uint64(p)) % np.uint64(cnt)), cnt
