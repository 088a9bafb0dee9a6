import sys, importlib, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'oracle'))
import numpy as np
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
import ptoracle
prod, orc = pkg.Product(), ptoracle.Oracle()
pair = {}
for name, be in (("gpu", prod), ("cpu", orc)):
    sc = be.new_scene(); pair[name] = (sc, pkg.scenes.load_scene(sc, 11, 128, 96))
orc.set_faithful(pair["cpu"][0], False)
rng = np.random.default_rng(11); n = 30000
xys = np.stack([rng.integers(0, 128, n), rng.integers(0, 96, n), rng.integers(0, 64, n)], 1).astype(np.uint32)
for strat in ("mis", "nee", "pt"):
    prm = pkg.make_params(64, strat, "sobol")
    Lg, _, _ = pair["gpu"][0].probe_radiance(pair["gpu"][1], prm, xys)
    Lc, _, _ = pair["cpu"][0].probe_radiance(pair["cpu"][1], prm, xys)
    ng, nc = np.isnan(Lg).any(1), np.isnan(Lc).any(1)
    ig, ic = np.isinf(Lg).any(1), np.isinf(Lc).any(1)
    print(strat, "nan gpu", ng.sum(), "cpu", nc.sum(), "both", (ng & nc).sum(), "inf gpu", ig.sum(), "cpu", ic.sum())
    if strat == "mis":
        Lg2, lamg, pdfg = pair["gpu"][0].probe_radiance(pair["gpu"][1], prm, xys[nc][:8])
        Lc2, lamc, pdfc = pair["cpu"][0].probe_radiance(pair["cpu"][1], prm, xys[nc][:8])
        print(xys[nc][:8]); print(Lg2); print(lamg); print(lamc); print(Lc2)
