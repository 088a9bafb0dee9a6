#!/usr/bin/env python3
"""64x48x64-spp frames, GPU vs oracle, for every scene / strategy pair of the sample-for-sample frame test (GPU box only): prints
rmse (tone-mapped), pixels off by more than 0.01, one JSON line per pair.  The limits in tests/test_parity_gpu.py come from this table."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
import ptoracle
prod, orc = pkg.Product(), ptoracle.Oracle()
prod.debug_unlock(True)      # rr_gate_slack is a diagnostic (include/mi355pt_debug.h)
pairs = [(int(a), b) for a, b in (x.split(":") for x in sys.argv[1:])]
for sid, strat in pairs:
    out = {}
    for slack in (0.0, 1e-5):
        pair = {}
        for name, be in (("gpu", prod), ("cpu", orc)):
            sc = be.new_scene()
            if name == "cpu" and os.environ.get("ORACLE_RENDER_LOWERING") == "1":
                orc.set_render_space_lowering(sc, True)      # diagnostic: the oracle intersects pre-transformed triangles like the product
            pair[name] = (sc, pkg.scenes.load_scene(sc, sid, 64, 48, tex_size=128))
        orc.set_faithful(pair["cpu"][0], False)
        prm = pkg.make_params(64, strat, "sobol", rr_gate_slack=slack)
        g = prod.render(pair["gpu"][0], pair["gpu"][1], prm); c = orc.render(pair["cpu"][0], pair["cpu"][1], prm)
        with np.errstate(invalid="ignore"):
            dd = np.nan_to_num(g - c)
        out["slack_%g" % slack] = {"rmse": float(np.sqrt(np.mean(dd ** 2))), "off": int((np.abs(dd).max(axis=2) > 0.01).sum()),
                                   "nan_px_gpu": int(np.isnan(g).any(axis=2).sum()), "nan_px_cpu": int(np.isnan(c).any(axis=2).sum())}
    print(json.dumps({"scene": sid, "strategy": strat, **out}), flush=True)
