#!/usr/bin/env python3
"""Scene set-up time (host sweep-SAH BVH build + lowering + upload) for a large synthetic mesh (GPU box)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
prod = pkg.Product()
SIZES = ((256, 129), (1024, 513)) + (((2048, 1025),) if "big" in sys.argv else ())   # 65 k, 1.05 M, 4.2 M triangles
for n_lon, n_bands in SIZES:
    t0 = time.time()
    m = pkg.assets.load_obj_semantics(pkg.assets.blob_mesh(n_lon, n_bands, seed=3, lobes=(14, 0.30, 8.0, 60, 0.05, 60.0), center=(0.0, 1.2, -1.0), scale=1.0))
    t1 = time.time()
    sc = prod.new_scene(); sc.set_rgb2spec(pkg.scenes.srgb_table())
    g = sc.add_mesh(m)
    d = pkg.ffi.MaterialDesc(); d.type = pkg.ffi.MAT_LAMBERT; d.color = pkg.ffi.Spectrum.constant(0.7); d.normal_tex = pkg.ffi.NONE
    sc.add_instance(g, sc.add_material(d))
    p = pkg.scenes.presets()
    pkg.scenes._room(sc, p)
    cam = pkg.ffi.make_camera((0.0, 3.15221, 6.0), (0.0, -0.9, -3.2), (0.0, 1.0, 0.0), 640, 360)
    t2 = time.time()
    sc.build(cam)
    t3 = time.time()
    img = prod.render(sc, cam, pkg.make_params(16, "mis", "sobol"))
    t4 = time.time()
    _, st = prod.render(sc, cam, pkg.make_params(1024, "mis", "sobol"), want_stats=True)
    print(f"tris={m['idx'].reshape(-1,3).shape[0]} mesh_gen={t1-t0:.2f}s build={t3-t2:.2f}s render16spp={t4-t3:.2f}s Msamples_s_1024spp={640*360*1024/st.kernel_ms/1e3:.0f} info={prod.scene_info(sc)} mean={img.mean():.3f}", flush=True)
