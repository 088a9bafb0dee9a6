#!/usr/bin/env python3
"""How far do the clearcoat coat-weight variants move the picture?  (GPU box; the oracle is test infrastructure.)

SimpleClearcoatPbrMaterial estimates the coat's directional albedo with 64 random GGX samples in EACH of sample(), evaluate() and pdf()
(simple_pbr_clearcoat_material.rs:190-192,318-320,416-418 -> generalized_schlick.rs:893-918).  Variants compared on scene 17 (C5's scene) at the
reference's regression size 200x150, 2048 spp, per strategy:
  oracle shared        one estimate per vertex shared by the three calls (the product's default structure)
  oracle independent   three independent estimates per vertex (the reference's structure)
  oracle lut / gpu lut the estimate's expectation from the 64-entry table (mi355pt_params.albedo_lut)
  gpu shared           the product's default
Metric: RMSE of the tone-mapped frames and the reference's own metric (regression_test.rs:6-40) on the 8-bit frames."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
import ptoracle
prod, orc = pkg.Product(), ptoracle.Oracle(native=True)
W, H, SPP = 200, 150, int(sys.argv[1]) if len(sys.argv) > 1 else 2048

def lin(u):
    s = u.astype(np.float64) / 255.0
    return np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)

def rmse_u8(a, b):
    return float(np.sqrt(np.mean((lin(a) - lin(b)) ** 2)))

for strategy in ("nee", "mis"):
    frames = {}
    for mode in ("shared", "independent", "lut"):
        sc = orc.new_scene(); cam = pkg.scenes.load_scene(sc, 17, W, H); orc.set_faithful(sc, False)
        orc.set_clearcoat_mode(sc, mode, prod)
        frames["oracle_" + mode] = orc.render(sc, cam, pkg.make_params(SPP, strategy, "sobol"))
    gsc = prod.new_scene(); gcam = pkg.scenes.load_scene(gsc, 17, W, H)
    frames["gpu_shared"] = prod.render(gsc, gcam, pkg.make_params(SPP, strategy, "sobol"))
    frames["gpu_lut"] = prod.render(gsc, gcam, pkg.make_params(SPP, strategy, "sobol", albedo_lut=1))
    out = {"scene": 17, "strategy": strategy, "width": W, "height": H, "spp": SPP, "rmse": {}}
    for a, b in (("oracle_independent", "oracle_shared"), ("oracle_lut", "oracle_shared"), ("oracle_lut", "oracle_independent"),
                 ("gpu_shared", "oracle_shared"), ("gpu_lut", "oracle_lut"), ("gpu_lut", "gpu_shared")):
        qa, qb = prod.quantize_u8(frames[a]), prod.quantize_u8(frames[b])
        out["rmse"][f"{a} vs {b}"] = {"tone_mapped": float(np.sqrt(np.mean((frames[a] - frames[b]) ** 2))), "reference_metric_u8": rmse_u8(qa, qb)}
    print(json.dumps(out), flush=True)
