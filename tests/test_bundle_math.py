"""The geometry of the wave-level bundle pre-cull (rt_device.h: bundle_pass), replayed on the CPU in float32 exactly as
the kernel evaluates it, against brute force in float64: a sphere that ANY ray of the bundle hits must never be culled.
(The kernel-side use of these bounds is covered by the GPU parity tests; this pins the inequalities and their margins.)"""
import numpy as np

F = np.float32


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _hits(o, d, c, r):
    """float64: does ray (o, d) hit sphere (c, r) at some t > 0?  o, d: (n,3); c: (m,3); r: (m,) -> (n, m)"""
    L = o[:, None, :] - c[None, :, :]
    s = (L * d[:, None, :]).sum(-1)
    cc = (L * L).sum(-1) - r[None, :] ** 2
    D = s * s - cc
    root = np.sqrt(np.maximum(D, 0))
    return (D >= 0) & ((-s + root) > 0)


def _cone(dirs):
    """direction_cone(): axis = first direction, sin(g) from cross products; None if not a bundle"""
    X = dirs[0].astype(F)
    d = dirs.astype(F)
    k = np.cross(d, X)
    dt = (d * X).sum(-1)
    s2 = np.where(dt > 0, (k * k).sum(-1), F(1.0)).max().astype(F)
    sing = F(np.sqrt(s2) * F(1 + 2.0 ** -10) + F(2.0 ** -20))
    cosg = F(np.sqrt(max(F(1) - sing * sing, F(0))) * F(1 - 2.0 ** -20))
    return (X, cosg, sing) if sing < 0.5 else None


def _ball(pts):
    C = pts[0].astype(F)
    e = pts.astype(F) - C
    rho = F(np.sqrt((e * e).sum(-1).max()) * F(1 + 2.0 ** -10) + np.abs(C).sum() * F(2.0 ** -18) + F(0.001))
    return C, rho


def _cull_free(c, r2, C, rho, X, cosg, sing):
    c = c.astype(F); r2 = r2.astype(F)
    r = np.sqrt(r2) * F(1 + 2.0 ** -20) + F(2.0 ** -30)
    u = c - C
    D2 = (u * u).sum(-1)
    rD = F(1) / np.sqrt(D2)
    t = D2 * rD - rho
    with np.errstate(divide="ignore", invalid="ignore"):
        sv = (r + rho) / t * F(1 + 2.0 ** -20)
        cb = np.sqrt(np.maximum(F(1) - sv * sv, F(0)))
        lhs = (u * X).sum(-1) * rD
        rhs = cosg * cb - sing * sv
        return (t > 0) & (sv < F(0.98)) & (lhs < rhs - F(2.0 ** -16))


def _cull_anchored(c, r2, A, X, cosg, sing):
    e = (A[None, :] - c).astype(F)                       # the table's {A - c}
    ll = ((A[None, :] - c) ** 2).sum(-1)
    W = (ll - r2) - (ll + r2) * (2.0 ** -19 * 1.0001)    # anchored_tau() without the launch floor (smaller floor = larger tau = harder test)
    tau = np.where(W > 0, np.sqrt(np.maximum(W, 0)) * (1 - 2.0 ** -20), 0.0).astype(F)
    r = np.sqrt(r2.astype(F)) * F(1 + 2.0 ** -20) + F(2.0 ** -30)
    lhs = np.abs((e * X).sum(-1))
    rhs = cosg * tau - sing * r
    return lhs < rhs - F(2.0 ** -16) * (tau + r)


def test_free_bundle_never_culls_a_sphere_some_ray_hits():
    rng = np.random.default_rng(1)
    culled_total = 0
    for trial in range(400):
        n, m = 64, 96
        C0 = rng.uniform(-5, 5, 3)
        spread = 10.0 ** rng.uniform(-3, 0.3)
        pts = C0 + rng.normal(size=(n, 3)) * spread
        axis = _unit(rng.normal(size=3))
        ang = 10.0 ** rng.uniform(-3, -0.4)
        dirs = _unit(axis + rng.normal(size=(n, 3)) * ang)
        cone = _cone(dirs)
        if cone is None:
            continue
        C, rho = _ball(pts)
        c = rng.uniform(-8, 8, (m, 3)).astype(F).astype(np.float64)
        r = rng.uniform(0.05, 1.5, m).astype(F)
        r2 = (r * r).astype(np.float64)                  # the scene stores float32 r*r (intersections.py:21)
        origins = pts + 0.0002 * dirs                    # trace.py:110
        hit_any = _hits(origins, dirs, c, np.sqrt(r2)).any(axis=0)
        cull = _cull_free(c, r2, C, rho, *cone)
        assert not (cull & hit_any).any(), f"trial {trial}: culled a sphere that is hit"
        culled_total += int(cull.sum())
    assert culled_total > 5000                           # and the test has teeth: most spheres are culled


def test_anchored_bundle_never_culls_a_sphere_some_line_hits():
    rng = np.random.default_rng(2)
    culled_total = 0
    for trial in range(400):
        n, m = 64, 96
        A = rng.uniform(-6, 8, 3).astype(F).astype(np.float64)          # the light (float32 in the scene) or the camera
        if trial % 2:                                                    # shadow rays: from a ball of hit points toward A
            C0 = A + _unit(rng.normal(size=3)) * rng.uniform(0.5, 12)
            pts = C0 + rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, -0.3)
            C, rho = _ball(pts)
            U = (C - A.astype(F)).astype(F)
            d2 = (U * U).sum()
            inv = F(1) / np.sqrt(d2)
            X = U * inv
            sing = F(rho * inv * F(1 + 2.0 ** -10) + F(2.0 ** -20))
            cosg = F(np.sqrt(max(F(1) - sing * sing, F(0))) * F(1 - 2.0 ** -20))
            if not sing < 0.5:
                continue
            origins, dirs = pts, _unit(A - pts)
        else:                                                            # primary rays: a narrow cone from the camera
            axis = _unit(rng.normal(size=3))
            dirs = _unit(axis + rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3.5, -1))
            cone = _cone(dirs)
            if cone is None:
                continue
            X, cosg, sing = cone
            origins = np.broadcast_to(A, (n, 3))
        c = rng.uniform(-8, 10, (m, 3)).astype(F).astype(np.float64)
        r = rng.uniform(0.05, 1.5, m).astype(F)
        r2 = (r * r).astype(np.float64)
        hit_any = _hits(origins, dirs, c, np.sqrt(r2)).any(axis=0)
        cull = _cull_anchored(c, r2, A, X, cosg, sing)
        assert not (cull & hit_any).any(), f"trial {trial}"
        culled_total += int(cull.sum())
    assert culled_total > 5000
