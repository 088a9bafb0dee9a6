#!/usr/bin/env python3
"""GPU box: share of samples whose radiance leaves the oracle's (reference's Russian-roulette gate, rr_gate_slack = 0) and the 64-spp frame
metrics for the solid constant-eta plastic scenes — the measurement behind PT_FRAME_INVERSE (pt_path.hpp).  MI355PT_LIB selects the build."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
import ptoracle

prod, orc = pkg.Product(), ptoracle.Oracle()
prod.debug_unlock(True)      # rr_gate_slack is a diagnostic (include/mi355pt_debug.h)
out = {"library": prod.version(), "scenes": {}}
LOWER = os.environ.get("PROBE_RENDER_SPACE_ORACLE") == "1"     # the oracle intersects pre-transformed render-space triangles like the product
out["oracle_render_space_lowering"] = LOWER
for scene_id in (9, 13, 19, 8, 10):
    pair = {}
    for name, be in (("gpu", prod), ("cpu", orc)):
        sc = be.new_scene()
        if name == "cpu" and LOWER:
            orc.set_render_space_lowering(sc, True)
        pair[name] = (sc, pkg.scenes.load_scene(sc, scene_id, 64, 48, tex_size=128))
    orc.set_faithful(pair["cpu"][0], False)
    ys, xs, ss = np.meshgrid(np.arange(48), np.arange(64), np.arange(64), indexing="ij")
    xys = np.stack([xs.ravel(), ys.ravel(), ss.ravel()], 1).astype(np.uint32)
    prm = pkg.make_params(64, "mis", "sobol", rr_gate_slack=0.0)
    Lg, lg, pg = pair["gpu"][0].probe_radiance(pair["gpu"][1], prm, xys)
    Lc, lc, pc = pair["cpu"][0].probe_radiance(pair["cpu"][1], prm, xys)
    share = float((~np.all(np.abs(Lg - Lc) <= 1e-3 * np.abs(Lc) + 1e-4, axis=1)).mean())
    g = prod.render(pair["gpu"][0], pair["gpu"][1], prm); c = orc.render(pair["cpu"][0], pair["cpu"][1], prm)
    out["scenes"][scene_id] = {"flipped_share": share, "frame_rmse": float(np.sqrt(np.mean((g - c) ** 2))),
                               "pixels_off_0.01": int((np.abs(g - c).max(axis=2) > 0.01).sum())}
print(json.dumps(out))
