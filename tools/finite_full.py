#!/usr/bin/env python3
"""The linear films of the BASELINE configs at their TRUE size hold no non-finite value (GPU box only; C3 did for one build of round 3: two
samples of 8.5e9, tests/test_parity_gpu.py::test_edge_on_thin_film_sample_stays_finite_like_the_reference)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
prod = pkg.Product()
for sid, W, H, S, strat in ((10, 1920, 1080, 4096, "mis"), (17, 1920, 1080, 16384, "nee"), (3, 1920, 1080, 1024, "mis")):
    sc = prod.new_scene(); cam = pkg.scenes.load_scene(sc, sid, W, H)
    prm = pkg.make_params(S, strat, "sobol")
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    for s0 in range(0, S, 4096):
        prod.render_accum_device(sc, cam, prm, s0, min(S, s0 + 4096), acc.data_ptr())
    torch.cuda.synchronize()
    a = acc.cpu().numpy()
    bad = np.argwhere(~np.isfinite(a).all(axis=2))
    print("scene", sid, "non-finite pixels", len(bad), bad[:6].tolist(), "mean", float(a[np.isfinite(a)].mean()), flush=True)
