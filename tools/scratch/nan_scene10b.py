#!/usr/bin/env python3
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
import ptoracle
prod, orc = pkg.Product(), ptoracle.Oracle()
W, H, S = 1920, 1080, 4096
sc = prod.new_scene(); cam = pkg.scenes.load_scene(sc, 10, W, H)
osc = orc.new_scene(); ocam = pkg.scenes.load_scene(osc, 10, W, H); orc.set_faithful(osc, False)
for (x, y) in ((1895, 369), (1085, 615)):
    t = (y // 8) * ((W + 7) // 8) + x // 8
    prm = pkg.make_params(S, "mis", "sobol", shard_index=t, shard_count=((W + 7) // 8) * ((H + 7) // 8))
    L, lam, pdf = prod.render_sample_log(sc, cam, prm, 0, S)
    pix = (y & 7) * 8 + (x & 7)
    Lp = L[0, pix]
    bad = np.nonzero(~np.isfinite(Lp).all(axis=1))[0]
    print("pixel", x, y, "gpu non-finite samples", bad.tolist(), Lp[bad].tolist(), lam[0, pix][bad].tolist(), pdf[0, pix][bad].tolist())
    allbad = np.argwhere(~np.isfinite(L[0]).all(axis=2))
    print("  tile: non-finite (pixel, sample)", allbad[:6].tolist())
    for s in bad[:2]:
        xys = np.array([[x, y, s]], np.uint32)
        prm1 = pkg.make_params(S, "mis", "sobol")
        for d in (1, 2, 3, 4, 6, 16):
            prm1.max_depth = d
            Lc, lc, pc = osc.probe_radiance(ocam, prm1, xys)
            prm.max_depth = d
            Lg = prod.render_sample_log(sc, cam, prm, int(s), int(s) + 1)[0][0, pix, 0]
            print("   sample", int(s), "max_depth", d, "gpu", Lg.tolist(), "cpu", Lc[0].tolist())
