#!/usr/bin/env python3
"""GPU box: how often does the closest-hit merge meet an EXACT tie in t between two triangles?  The cooperative traversal keeps the lower
triangle index (LDS atomicMin on t bits << 32 | triangle), the reference the second child / earlier leaf item (scene/src/bvh.rs:381-388,
413-420): only on such ties can the two disagree about WHICH triangle was hit.  Instrumented kernel (collect_stats = 2), every BASELINE
scene at 1920x1080, 16 sample indices; prints ties per closest-hit ray."""
import ctypes as C
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
prod = pkg.Product()
hip = C.CDLL("libamdhip64.so")
# (scene 17: its instrumented kernel walks closest-hit rays alone (trace_closest_coop), which counts the same way)
for scene_id, strategy, spp in ((3, "mis", 1024), (10, "mis", 4096), (8, "mis", 1024), (17, "nee", 16384)):
    W, H = 1920, 1080
    sc = prod.new_scene()
    cam = pkg.scenes.load_scene(sc, scene_id, W, H)
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), C.c_size_t(W * H * 12)) == 0
    hip.hipMemset(d, 0, C.c_size_t(W * H * 12))
    st = pkg.ffi.Stats()
    prod.render_accum_device(sc, cam, pkg.make_params(spp, strategy, "sobol", collect_stats=2), 0, 16, d.value, None, stats=st)
    s = st.as_dict()
    ties, differ, rays = s["phase_cycles"][9] & 0xffffffff, s["phase_cycles"][9] >> 32, s["closest_rays"]
    print(json.dumps({"scene": scene_id, "strategy": strategy, "closest_rays": rays, "exact_t_ties": ties, "ties_per_ray": ties / max(rays, 1),
                      "ties_between_different_material_or_normal": differ,
                      "closest_hits": s["closest_hits"]}))
    hip.hipFree(d)
