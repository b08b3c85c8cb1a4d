"""The packed 16-bit Bresenham step of k_ray_list (csrc/voxel.hip::walk_packed), restated with Python integers and compared
with the plain walk of the reference kernel (gennbv/utils.py:48-167) on random in-grid rays.  No GPU needed."""
import random
def s16(x): x &= 0xFFFF; return x - 0x10000 if x & 0x8000 else x
def pk(lo, hi): return (lo & 0xFFFF) | ((hi & 0xFFFF) << 16)
def lo(x): return s16(x); 
def hi(x): return s16(x >> 16)
def pk_ashr15(x): return pk(-1 if lo(x) < 0 else 0, -1 if hi(x) < 0 else 0)
def pk_add(a, b): return pk(lo(a) + lo(b), hi(a) + hi(b))
def pk_mad(a, b, c): return pk(lo(a) * lo(b) + lo(c), hi(a) * hi(b) + hi(c))
def dot2(a, b, c): return (lo(a) * lo(b) + hi(a) * hi(b) + c)
def s32(x): x &= 0xFFFFFFFF; return x - (1 << 32) if x & (1 << 31) else x
def ref_walk(src, tgt, g):
    d = [abs(tgt[i] - src[i]) for i in range(3)]; s = [1 if src[i] < tgt[i] else -1 for i in range(3)]
    dm = max(d)
    if dm == d[0]: ax = (0, 1, 2)
    elif dm == d[1]: ax = (1, 0, 2)
    else: ax = (2, 0, 1)
    p = [src[a] for a in ax]; dd = [d[a] for a in ax]; ss = [s[a] for a in ax]
    p1 = 2 * dd[1] - dd[0]; p2 = 2 * dd[2] - dd[0]
    out = []
    def emit():
        c = [0, 0, 0]
        for k, a in enumerate(ax): c[a] = p[k]
        out.append((c[0] * g + c[1]) * g + c[2])
    emit()
    for i in range(dd[0]):
        if p1 >= 0: p[1] += ss[1]; p1 -= 2 * dd[0]
        if p2 >= 0: p[2] += ss[2]; p2 -= 2 * dd[0]
        p[0] += ss[0]; p1 += 2 * dd[1]; p2 += 2 * dd[2]
        emit()
    return out
def packed_walk(src, tgt, g, parts=1, seg=None):
    gg = g * g
    d = [abs(tgt[i] - src[i]) for i in range(3)]
    dm = max(d)
    ax_ = dm == d[0]; ay = (not ax_) and dm == d[1]
    pa = src[0] if ax_ else (src[1] if ay else src[2]); pb = src[1] if ax_ else src[0]; pc = src[2] if (ax_ or ay) else src[1]
    ta = tgt[0] if ax_ else (tgt[1] if ay else tgt[2]); tb = tgt[1] if ax_ else tgt[0]; tc = tgt[2] if (ax_ or ay) else tgt[1]
    da = dm; db = d[1] if ax_ else d[0]; dc = d[2] if (ax_ or ay) else d[1]
    st_a = gg if ax_ else (g if ay else 1); st_b = g if ax_ else gg; st_c = 1 if (ax_ or ay) else g
    sa = 1 if pa < ta else -1; sb = 1 if pb < tb else -1; sc = 1 if pc < tc else -1
    la, lb, lc = sa * st_a, sb * st_b, sc * st_c
    l = pa * st_a + pb * st_b + pc * st_c
    NEG2DA = pk(-2 * da, -2 * da); DL = pk(2 * db - 2 * da, 2 * dc - 2 * da); LBC = pk(lb, lc)
    Kp = la + lb + lc - (1 << 24)
    out = [l]  # the source voxel, emitted once per workgroup
    if seg is not None:  # (k_ray_list's segments of <= seg steps: segment k starts after k * seg steps)
        parts = max(1, -(-da // seg))
    for part in range(parts):  # (a lane enters the ray after j0 steps through the closed form of the walker's state)
        if parts == 1 and seg is None:
            W = s32(l + (da << 24))
            P = pk(2 * db - da, 2 * dc - da)
        else:
            j0, j1 = ((part * da) // parts, ((part + 1) * da) // parts) if seg is None else (part * seg, min(da, (part + 1) * seg))
            two_da = 2 * max(da, 1)
            nb, nc = (2 * db * j0 + da) // two_da, (2 * dc * j0 + da) // two_da
            W = s32(l + j0 * la + nb * lb + nc * lc + ((j1 - j0) << 24))
            p1, p2 = 2 * db * (j0 + 1) - da - 2 * da * nb, 2 * dc * (j0 + 1) - da - 2 * da * nc
            assert -32768 <= p1 < 32768 and -32768 <= p2 < 32768
            P = pk(p1, p2)
        while W >= (1 << 24):
            m = pk_ashr15(P)
            P = pk_mad(m, NEG2DA, pk_add(P, DL))
            W = s32(dot2(LBC, m, W) + Kp)
            out.append(W & 0xFFFFFF if True else 0)
            assert (W & 0xFFFFFF) < g ** 3, W
    return out
random.seed(1)
for g in (16, 20, 33, 64, 96, 104):  # (the list kernels serve G <= 104; the step counter in W[31:24] is signed: whole rays need da <= 127)
    for it in range(20000 if g <= 64 else 6000):
        src = [random.randrange(g) for _ in range(3)]; tgt = [random.randrange(g) for _ in range(3)]
        if it % 7 == 0: tgt[random.randrange(3)] = src[random.randrange(3)]
        a = ref_walk(src, tgt, g)
        for parts in (1, 2, 3, 4):
            b = packed_walk(src, tgt, g, parts)
            assert a == b, (g, parts, src, tgt, a, b)
        sg = max(16, (g + 2) >> 2)  # (the kernel's segment length)
        assert a == packed_walk(src, tgt, g, seg=sg) and -(-max(abs(tgt[i] - src[i]) for i in range(3)) // sg) <= 4, (g, sg, src, tgt)
print("packed recurrence (whole rays, 2 / 3 / 4 parts per ray, the kernel's fixed-length segments) == reference walk")
