#!/usr/bin/env python3
"""Share of the 196 608 samples of a 64x48x64-spp frame whose spectral radiance is BIT-EQUAL on GPU and oracle, and the largest relative
difference among the rest (GPU box only).  usage: tools/bit_equal_share.py <scene>:<strategy> ...
LOWERING=auto|no_local_tris|general selects the product's lowering path (mi355pt_scene_debug_set_lowering)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
import ptoracle
prod, orc = pkg.Product(), ptoracle.Oracle()
W, H, S = 64, 48, 64
ys, xs, ss = np.meshgrid(np.arange(H), np.arange(W), np.arange(S), indexing="ij")
xys = np.stack([xs.ravel(), ys.ravel(), ss.ravel()], 1).astype(np.uint32)
for arg in sys.argv[1:]:
    sid, strat = arg.split(":"); sid = int(sid)
    gsc = prod.new_scene(); gsc.debug_set_lowering(os.environ.get("LOWERING", "auto")); gcam = pkg.scenes.load_scene(gsc, sid, W, H, tex_size=128)
    osc = orc.new_scene(); ocam = pkg.scenes.load_scene(osc, sid, W, H, tex_size=128); orc.set_faithful(osc, False)
    prm = pkg.make_params(S, strat, "sobol", max_depth=int(os.environ.get("MAX_DEPTH", "16")))
    Lg, lg, pg = gsc.probe_radiance(gcam, prm, xys)
    Lc, lc, pc = osc.probe_radiance(ocam, prm, xys)
    same = np.all((Lg.view(np.uint32) == Lc.view(np.uint32)) | (np.isnan(Lg) & np.isnan(Lc)), axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.nan_to_num(np.abs(Lg - Lc) / np.maximum(np.abs(Lc), 1e-12))[~same]
    print(json.dumps({"scene": sid, "strategy": strat, "tri_space": prod.scene_info(gsc).split("tri_space=")[-1], "bit_equal_share": round(float(same.mean()), 6),
                      "not_equal": int((~same).sum()), "median_rel_diff_of_those": float(np.median(rel)) if rel.size else 0.0,
                      "over_1e-3": int((rel.max(axis=1) > 1e-3).sum()) if rel.size else 0,
                      "first_not_equal": [(xys[i].tolist(), Lg[i].tolist(), Lc[i].tolist()) for i in np.nonzero(~same)[0][:3]] if os.environ.get("SHOW") else None}), flush=True)
