#!/usr/bin/env python3
"""C5's kernel rate on one GPU (scene 17, NEE + ZSobol, 1920x1080, 16384-spp job, launches of 1024 sample indices) with the coat weight from
the 64-sample estimate (the reference's estimator, default) and from the table (mi355pt_params.albedo_lut).  GPU box only."""
import importlib, json, os, sys
import torch  # first: see tests/conftest.py
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
prod = pkg.Product(); sc = prod.new_scene(); cam = pkg.scenes.load_scene(sc, 17, 1920, 1080)
a = torch.zeros((1080, 1920, 3), device="cuda")
for lut in (0, 1, 0, 1):
    prm = pkg.make_params(16384, "nee", "sobol", albedo_lut=lut)
    best = 1e30
    for i in range(3):
        st = pkg.ffi.Stats(); prod.render_accum_device(sc, cam, prm, 1024 * i, 1024 * (i + 1), a.data_ptr(), None, stats=st)
        best = min(best, st.kernel_ms)
    print(json.dumps({"config": "C5 scene17 nee+sobol 1920x1080, 16384-spp job, 1024 sample indices per launch", "albedo_lut": lut,
                      "Msamples_s": round(1920 * 1080 * 1024 / best / 1e3, 1), "library": prod.version()}), flush=True)
