#!/usr/bin/env python3
"""Traversal micro-benchmark (GPU box; run under rocprofv3 --kernel-trace --stats): coherent camera rays vs
incoherent secondary rays through probe_intersect_kernel (lock-step, 64 rays per wave, no refill)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
prod = pkg.Product()
sc = prod.new_scene()
cam = pkg.scenes.load_scene(sc, 0, 1920, 1080)
W, H = 1920, 1080
ys, xs = np.mgrid[0:H, 0:W]
# tile-ordered (8x8) camera rays like the render kernel
tx, ty = xs // 8, ys // 8
order = np.lexsort(((xs % 8).ravel(), (ys % 8).ravel(), tx.ravel(), ty.ravel()))
fx = (xs.ravel()[order] + 0.5); fy = (ys.ravel()[order] + 0.5)
t = np.tan(np.deg2rad(22.5)); aspect = W / H
d_cam = np.stack([(2 * fx / W - 1) * aspect * t, (1 - 2 * fy / H) * t, -np.ones_like(fx)], 1)
d_cam /= np.linalg.norm(d_cam, axis=1, keepdims=True)
f = np.array([0, -0.9, -3.2]); f /= np.linalg.norm(f); s = np.cross(f, [0, 1, 0]); s /= np.linalg.norm(s); u = np.cross(s, f)
d = (d_cam[:, :1] * s + d_cam[:, 1:2] * u - d_cam[:, 2:3] * (-f)).astype(np.float32)
d = (d_cam[:, :1] * s + d_cam[:, 1:2] * u + d_cam[:, 2:3] * (-f)).astype(np.float32)
o = np.zeros_like(d)
os.environ["MI355PT_TRAV"] = "1"
t1, inst, tri, n = sc.probe_intersect(o, d)           # coherent
hit = t1 > 0
p = o[hit] + d[hit] * t1[hit, None]
rng = np.random.default_rng(1)
d2 = rng.normal(size=p.shape).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
flip = np.sum(d2 * n[hit], 1) * np.sum(-d[hit] * n[hit], 1) < 0
d2[flip] *= -1
o2 = (p + n[hit] * np.sign(np.sum(-d[hit] * n[hit], 1))[:, None] * 1e-4).astype(np.float32)
perm = rng.permutation(o2.shape[0])
REP = int(os.environ.get("RAY_REP", "8"))
sets = {"coherent": (np.tile(o, (REP, 1)), np.tile(d, (REP, 1))), "incoherent_tile_order": (np.tile(o2, (REP, 1)), np.tile(d2, (REP, 1))),
        "incoherent_shuffled": (np.tile(o2[perm], (REP, 1)), np.tile(d2[perm], (REP, 1)))}
res = {}
for mode in os.environ.get("RAY_MODES", "1,2").split(","):
    os.environ["MI355PT_TRAV"] = mode[0]
    if len(mode) > 1:
        os.environ["MI355PT_PROBE_LDS"] = mode[2:]       # e.g. "1:13000" = lock-step with 13 KB extra LDS per wave
    for k, (oo, dd) in sets.items():
        tt, ii, tr, _ = sc.probe_intersect(oo, dd)
        res[(mode, k)] = (tt, tr)
        print("mode", mode, k, "rays", oo.shape[0], "hit frac", round(float((tt > 0).mean()), 4), flush=True)
for k in sets:
    if ("1", k) in res and ("2", k) in res:
        print(k, "dyn vs lockstep identical t:", np.array_equal(res[("1", k)][0], res[("2", k)][0]), "tri:", np.array_equal(res[("1", k)][1], res[("2", k)][1]))
