"""The slab test of a ray without an anchor against a cluster's bounding box (rt_device.h: make_raybox / box_open and
the boxes tables_kernel builds), replayed on the CPU in float32 exactly as the kernel evaluates it, against the
reference's own float64 sphere test: a box must never be certified "missed" for a ray that the reference's arithmetic
reports a hit for on ANY sphere inside it.  (The kernel-side use is covered by the GPU parity tests and the fuzz soak;
this pins the inequality and its margins, including rays that graze a box face, start inside it, run parallel to a face,
or start far away.)"""
import numpy as np

F = np.float32


def _fma(a, b, c):
    """float32 fused multiply-add: exact product and sum in float64 (24+24-bit products are exact), one rounding."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def _boxes(c, r2):
    """tables_kernel: bounding box of spheres (centres c (n,8,3) float32, r2 float32), faces rounded outward"""
    c = c.astype(np.float64)
    rad = np.sqrt(r2.astype(np.float64)) * (1.0 + 2.0 ** -20) + 1e-30
    lo = (c - rad[..., None]).min(1)
    hi = (c + rad[..., None]).max(1)
    return (lo - (np.abs(lo) * 2.0 ** -22 + 1e-30)).astype(F), (hi + (np.abs(hi) * 2.0 ** -22 + 1e-30)).astype(F)


def _raybox(o, R, extent2):
    o = o.astype(F); R = R.astype(F)
    m = F(2.0 ** -18) * (np.sqrt(F(extent2)) + (np.abs(o[:, 0]) + np.abs(o[:, 1]) + np.abs(o[:, 2])))
    Rs = np.where(np.abs(R) < F(2.0 ** -40), np.copysign(F(2.0 ** -40), R), R).astype(F)
    inv = (F(1) / Rs).astype(F)                       # v_rcp_f32 is within 1 ulp of this; the margins allow 4
    a = -((o + m[:, None]) * inv)
    b = -((o - m[:, None]) * inv)
    da = np.abs(a) * F(2.0 ** -21); db = np.abs(b) * F(2.0 ** -21)
    clo = np.where(inv > 0, a - da, a + da).astype(F)
    chi = np.where(inv > 0, b + db, b - db).astype(F)
    return inv, clo, chi


def _box_open(lo, hi, inv, clo, chi):
    A = _fma(lo, inv, clo); B = _fma(hi, inv, chi)
    tn = np.minimum(A, B).max(-1)
    tf = np.maximum(A, B).min(-1)
    return ~(tf < np.maximum(tn * F(1.0 - 2.0 ** -18), F(0)))


def _reference_hit(o, R, c, r2):
    """intersections.py:6-38 in float64 on the float32 scene: does sphere (c, r2) report 0 < t for ray (o, R)?"""
    L = o[:, None, :] - c.astype(np.float64)
    a = (R * R).sum(-1)[:, None]
    b = 2.0 * (L * R[:, None, :]).sum(-1)
    cc = (L * L).sum(-1) - r2.astype(np.float64)
    disc = b * b - 4.0 * a * cc
    ok = disc >= 0
    sq = np.sqrt(np.where(ok, disc, 0.0))
    t1 = (-b - sq) / (2.0 * a); t2 = (-b + sq) / (2.0 * a)
    return ok & ((t1 > 0) | (t2 > 0))


def _cases(rng, n, scale, far):
    c0 = rng.uniform(-scale, scale, (n, 1, 3))
    c = (c0 + rng.uniform(-0.12, 0.12, (n, 8, 3)) * scale * np.array([1.0, 1.0, rng.choice([1.0, 0.05])])).astype(F)
    r = (rng.uniform(0.01, 0.06, (n, 8)) * scale).astype(F)
    r2 = (r * r).astype(F)
    lo, hi = _boxes(c, r2)
    kind = rng.integers(0, 7, n)
    # targets: a point on some sphere's surface region, a box corner / face point (grazing), or anywhere nearby
    k = rng.integers(0, 8, n)
    ck = c[np.arange(n), k].astype(np.float64); rk = r[np.arange(n), k].astype(np.float64)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    surf = ck + u * rk[:, None] * rng.choice([0.999999, 1.0, 1.000001, 0.5], n)[:, None]
    face = lo + (hi - lo) * rng.choice([0.0, 1.0, 0.5], (n, 3))
    tgt = np.where((kind < 3)[:, None], surf, face.astype(np.float64))
    dist = rng.choice([0.3, 3.0, 30.0] + ([3000.0] if far else []), n)[:, None] * scale
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    o = tgt - v * dist
    inside = kind == 4
    o[inside] = (lo + (hi - lo) * rng.uniform(0, 1, (n, 3)))[inside]
    d = tgt - o
    # kinds 5, 6: tangent to a sphere at its extreme point along an axis — in the plane of the box face that sphere
    # defines — tilted into the sphere by a hair (these are the rays a box rounded the wrong way would lose)
    tang = kind >= 5
    ax = rng.integers(0, 3, n); sg = rng.choice([-1.0, 1.0], n)
    e = np.zeros((n, 3)); e[np.arange(n), ax] = sg
    touch = ck + e * rk[:, None]
    w = rng.normal(size=(n, 3)); w -= (w * e).sum(-1)[:, None] * e; w /= np.linalg.norm(w, axis=1, keepdims=True)
    w = w - e * rng.choice([0.0, 1e-9, 1e-7, 1e-5], n)[:, None]
    o[tang] = (touch - w * dist)[tang]
    d[tang] = w[tang]
    axis_par = (rng.random(n) < 0.15) & ~tang             # rays parallel to a face: one exact zero component
    d[axis_par, rng.integers(0, 3)] = 0.0
    flip = (rng.random(n) < 0.2) & ~tang                  # pointing away: the box is behind the origin
    d[flip] = -d[flip]
    nrm = np.sqrt((d * d).sum(-1))
    good = nrm > 0
    R = d[good] / nrm[good][:, None]
    R = R / np.sqrt((R * R).sum(-1))[:, None]             # the kernel's direction is normalised twice
    return o[good], R, c[good], r2[good], lo[good], hi[good]


def test_box_never_certifies_a_ray_the_reference_hits():
    rng = np.random.default_rng(20260402)
    total_hits = total_culled = 0
    for scale, far in ((1.0, False), (8.0, True), (0.05, False), (300.0, True)):
        o, R, c, r2, lo, hi = _cases(rng, 120000, scale, far)
        extent2 = F(1.0001 * max((np.abs(c).max() * 1.8 + 1) ** 2, 1.0))
        inv, clo, chi = _raybox(o, R, extent2)
        opened = _box_open(lo, hi, inv, clo, chi)
        hit = _reference_hit(o, R, c, r2).any(-1)
        assert not (hit & ~opened).any(), f"scale {scale}: {(hit & ~opened).sum()} rays culled that the reference hits"
        total_hits += int(hit.sum()); total_culled += int((~opened).sum())
    assert total_hits > 50000 and total_culled > 50000     # the cases exercise both outcomes


def test_the_cases_have_teeth():
    """the same cases against boxes shrunk by a thousandth of their size: the check above must then fail"""
    rng = np.random.default_rng(7)
    o, R, c, r2, lo, hi = _cases(rng, 120000, 8.0, False)
    extent2 = F(1.0001 * (np.abs(c).max() * 1.8 + 1) ** 2)
    inv, clo, chi = _raybox(o, R, extent2)
    sz = hi - lo
    opened = _box_open(lo + sz * F(1e-3), hi - sz * F(1e-3), inv, clo, chi)
    hit = _reference_hit(o, R, c, r2).any(-1)
    assert (hit & ~opened).sum() > 20


def test_box_is_tight_for_a_flat_cluster():
    """a 2 x 4 block of spheres resting on a plane: rays passing one sphere diameter above the layer are certified
    by the box (the bounding sphere of the same block would admit them)"""
    xs, ys = np.meshgrid(np.arange(2) * 0.5, np.arange(4) * 0.5)
    c = np.stack([xs.ravel(), ys.ravel(), np.full(8, 0.2)], -1)[None].astype(F)
    r2 = np.full((1, 8), 0.04, F)
    lo, hi = _boxes(c, r2)
    o = np.array([[-3.0, 0.7, 0.9]]); R = np.array([[1.0, 0.0, 0.0]])
    inv, clo, chi = _raybox(o, R, F(30.0))
    assert not _box_open(lo, hi, inv, clo, chi)[0]
    cs = c[0].mean(0); Rb = np.sqrt(((c[0] - cs) ** 2).sum(-1)).max() + 0.2
    assert abs(0.9 - cs[2]) < Rb                            # the bounding sphere's line test would not certify it
