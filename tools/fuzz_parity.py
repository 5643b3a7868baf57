#!/usr/bin/env python3
"""Soak test: random scenes / cameras / shader settings, HIP kernel vs the CPU oracle, bit-exact uint8 and float32.
    python tools/fuzz_parity.py [--seconds 240] [--seed 1]
Scenes are biased toward what stresses the conservative float32 cull and the cluster hierarchy: touching and
nested spheres, lights inside/near spheres, cameras inside spheres, huge and tiny radii, far-away geometry."""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import python_ray_tracer_amd as pkg
from python_ray_tracer_amd.scene import Camera
from oracle import oracle as orc

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=240); ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
r = pkg.Renderer(0)
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < a.seconds:
    kind = int(rng.integers(0, 6))
    S = int(rng.choice([0, 1, 3, 8, 17, 40, 64, 97, 130, 200, 256, 260, 400]))
    scale = float(rng.choice([0.05, 1.0, 1.0, 1.0, 30.0]))
    sp = np.zeros((7, S), np.float32)
    sp[0:3] = rng.uniform(-5, 7, (3, S)) * scale
    sp[3] = rng.uniform(0.05, 1.5, S) * scale
    if kind == 1 and S > 2:      # touching / nested spheres
        sp[0:3, 1] = sp[0:3, 0] + np.array([sp[3, 0] + sp[3, 1], 0, 0]); sp[0:3, 2] = sp[0:3, 0]; sp[3, 2] = sp[3, 0] * 0.5
    if kind == 2 and S > 0:      # one huge sphere
        sp[3, 0] = 40.0 * scale
    sp[4:7] = rng.integers(0, 256, (3, S))
    P = int(rng.integers(0, 4)) if kind != 3 else int(rng.integers(1, 8))
    pl = np.zeros((9, P), np.float32)
    if P:
        pl[0:3] = rng.uniform(-3, 3, (3, P)) * scale
        nrm = rng.normal(size=(3, P))
        if kind == 3: nrm = np.eye(3)[:, rng.integers(0, 3, P)] * rng.choice([-1.0, 1.0], P)
        pl[3:6] = nrm / np.linalg.norm(nrm, axis=0, keepdims=True); pl[6:9] = rng.integers(0, 256, (3, P))
    Ln = int(rng.integers(0, 6)) if rng.integers(0, 8) else int(rng.integers(6, 14))     # now and then many lights (the anchored cull table then outgrows its LDS budget)
    li = (rng.uniform(-6, 8, (3, Ln)) * scale).astype(np.float32)
    if kind == 4 and S and Ln: li[:, 0] = sp[0:3, 0]          # a light at a sphere centre
    w, h = int(rng.integers(9, 70)), int(rng.integers(9, 70))
    pos = (rng.uniform(-4, 4, 3) * scale).tolist()
    if kind == 5 and S: pos = (sp[0:3, 0] + 0.3 * sp[3, 0]).astype(float).tolist()   # camera inside a sphere
    cam = Camera((w, h), pos, rng.uniform(-180, 180, 3).tolist(), fov=float(rng.uniform(20, 100)))
    amb, lamb, refl, depth = float(rng.uniform(0, 0.3)), float(rng.uniform(0.1, 1.0)), float(rng.uniform(0, 0.9)), int(rng.integers(0, 9))
    aa = int(rng.choice([0, 0, 1, 2])); spp = int(rng.integers(1, 5)); seed = int(rng.integers(0, 2**31))
    rg = cam.raygen()
    r.set_scene(sp, li, pl); r.set_camera(cam.position, cam.rotation); r.set_raygen(w, h, *rg)
    u8, f32 = r.render(amb, lamb, refl, depth, aa, u8=True, f32=True, spp=spp, seed=seed)
    ref = orc.render(w, h, cam.position, cam.rotation, sp, li, pl, amb, lamb, refl, depth, aa, raygen=rg, want=("u8", "f32"), spp=spp, seed=seed)
    ok = np.array_equal(u8, ref["u8"]) and np.array_equal(f32, ref["f32"], equal_nan=True)
    n += 1
    if not ok:
        bad += 1
        d = (u8 != ref["u8"]).any(axis=0)
        print(f"MISMATCH trial {n}: kind={kind} S={S} P={P} L={Ln} {w}x{h} depth={depth} aa={aa} scale={scale}: {int(d.sum())} px", flush=True)
        np.savez(os.path.join(REPO, "gpurun_out", f"fuzz_fail_{a.seed}_{n}.npz"), spheres=sp, lights=li, planes=pl, pos=np.array(pos),
                 rot=cam.rotation, rg=np.array(rg), w=w, h=h, amb=amb, lamb=lamb, refl=refl, depth=depth, aa=aa, spp=spp, seed=seed)
    if n % 200 == 0:
        print(f"{n} scenes, {bad} mismatches, {time.time()-t0:.0f}s", flush=True)
print(f"DONE {n} scenes, {bad} mismatches")
sys.exit(1 if bad else 0)
