#!/usr/bin/env python3
"""Where do GPU and oracle paths part?  (GPU box only; oracle = test infrastructure.)

For a scene / strategy renders W x H x spp on the product with the per-sample log (production kernel) and on the oracle in two lowerings:
  reference    the reference's per-primitive ray transform (primitive/impls/triangle_mesh.rs:89-119): ray to local space, hit back to render
  render       pre-transformed render-space triangles (what the product's flat BVH intersects)
plus the same two with the Russian-roulette gate `max(T) >= 1` relaxed to `>= 1 - 1e-5` on BOTH sides (mi355pt_params.rr_gate_slack),
and reports, per arm, the share of samples whose spectral radiance differs from the GPU's by more than 1e-3 relative, and the
FIRST path depth at which they differ (max_depth sweep: a sample first differs at depth d if its radiance agrees for max_depth < d).
usage: tools/divergence_probe.py <scene> <strategy> [--width 64 --height 48 --spp 64 --depths 6]"""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("scene", type=int); ap.add_argument("strategy")
ap.add_argument("--width", type=int, default=64); ap.add_argument("--height", type=int, default=48)
ap.add_argument("--spp", type=int, default=64); ap.add_argument("--depths", type=int, default=6)
a = ap.parse_args()
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
import ptoracle
prod, orc = pkg.Product(), ptoracle.Oracle()
prod.debug_unlock(True)      # rr_gate_slack is a diagnostic (include/mi355pt_debug.h)
W, H, S = a.width, a.height, a.spp
ys, xs, ss = np.meshgrid(np.arange(H), np.arange(W), np.arange(S), indexing="ij")
xys = np.stack([xs.ravel(), ys.ravel(), ss.ravel()], 1).astype(np.uint32)

def oracle_scene(render_lowering):
    sc = orc.new_scene()
    if render_lowering:
        orc.set_render_space_lowering(sc, True)
    cam = pkg.scenes.load_scene(sc, a.scene, W, H, tex_size=128)
    orc.set_faithful(sc, False)
    return sc, cam

gsc = prod.new_scene(); gcam = pkg.scenes.load_scene(gsc, a.scene, W, H, tex_size=128)
out = {"scene": a.scene, "strategy": a.strategy, "samples": int(xys.shape[0])}
for name, rl, slack in (("reference", False, 0.0), ("render", True, 0.0), ("reference_rr_slack_1e-5", False, 1e-5), ("render_rr_slack_1e-5", True, 1e-5)):
    osc, ocam = oracle_scene(rl)
    first = np.full(xys.shape[0], 0, np.int32)        # 0 = never differs
    for d in list(range(1, a.depths + 1)) + [16]:
        prm = pkg.make_params(S, a.strategy, "sobol", max_depth=d, rr_gate_slack=slack)
        Lg, lg, pg = gsc.probe_radiance(gcam, prm, xys)
        Lc, lc, pc = osc.probe_radiance(ocam, prm, xys)
        assert np.array_equal(lg, lc)
        with np.errstate(invalid="ignore"):
            bad = ~np.all((np.abs(Lg - Lc) <= 1e-3 * np.abs(Lc) + 1e-4) | (np.isnan(Lg) & np.isnan(Lc)), axis=1)
        first[(first == 0) & bad] = d
        if d == 16:
            out[name] = {"diverging_share_depth16": round(float(bad.mean()), 6),
                         "first_diverging_depth_histogram": {int(k): int(v) for k, v in zip(*np.unique(first[first > 0], return_counts=True))},
                         "frame_rmse_linear": float(np.sqrt(np.mean((np.nan_to_num(Lg) - np.nan_to_num(Lc)) ** 2)))}
print(json.dumps(out))
