#!/usr/bin/env python3
"""Quick perf probe used while tuning kernels (GPU box only): Msamples/s of the render launch (HIP events inside the
library) and the per-phase wave-cycle shares of the instrumented variant."""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=3)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=1024)
ap.add_argument("--slice", type=int, default=32)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--strategy", default="mis")
ap.add_argument("--sampler", default="sobol")
ap.add_argument("--tag", default="")
ap.add_argument("--max-depth", type=int, default=16)
ap.add_argument("--shards", type=int, default=1, help="render only shard 0 of N (emulates one rank of an N-GPU job)")
ap.add_argument("--tex-size", type=int, default=1024)
ap.add_argument("--lowering", default="auto", help="mi355pt_scene_debug_set_lowering: auto | no_local_tris | general")
ap.add_argument("--no-stats", action="store_true", help="skip the instrumented launch (PC sampling wants only the production kernel)")
a = ap.parse_args()

pkg = importlib.import_module("toy-cpu-pathtracing_amd")
import ctypes as C
prod = pkg.Product()
sc = prod.new_scene()
sc.debug_set_lowering(a.lowering)
cam = pkg.scenes.load_scene(sc, a.scene, a.width, a.height, tex_size=a.tex_size)
hip = C.CDLL("libamdhip64.so")
n = a.width * a.height * 3 * 4
d_acc = C.c_void_p()
assert hip.hipMalloc(C.byref(d_acc), C.c_size_t(n)) == 0
hip.hipMemset(d_acc, 0, C.c_size_t(n))
prm = pkg.make_params(a.spp, a.strategy, a.sampler, max_depth=a.max_depth, shard_index=0, shard_count=a.shards)
ms = []
for i in range(a.reps + 1):
    st = pkg.ffi.Stats()
    k = i % max(a.spp // a.slice, 1)
    prod.render_accum_device(sc, cam, prm, k * a.slice, (k + 1) * a.slice, d_acc.value, None, stats=st)
    ms.append(st.kernel_ms)
best = min(ms[1:])
rate = a.width * a.height * a.slice / a.shards / best / 1e3
if a.no_stats:
    print(json.dumps({"tag": a.tag, "scene": a.scene, "Msamples_s": round(rate, 1), "ms": [round(x, 2) for x in ms]}))
    sys.exit(0)
prm2 = pkg.make_params(a.spp, a.strategy, a.sampler, collect_stats=2, max_depth=a.max_depth, shard_index=0, shard_count=a.shards)
st = pkg.ffi.Stats()
prod.render_accum_device(sc, cam, prm2, 0, a.slice, d_acc.value, None, stats=st)
d = st.as_dict()
ph = d["phase_cycles"]
tot = max(ph[5], 1)
names = ["regen", "closest", "shade", "shadow", "film", "loop", "sh_surface", "sh_bsdf", "sh_nee"]
print(json.dumps({"tag": a.tag, "scene": a.scene, "Msamples_s": round(rate, 1), "ms": [round(x, 2) for x in ms],
                  "phase_share": {k: round(v / tot, 3) for k, v in zip(names, ph[:9]) if k != "loop"},
                  "lane_util": (lambda w: {"closest_nodes": round(d["nodes_closest"] / max(64 * w[0], 1), 3), "closest_tris": round(d["tris_closest"] / max(64 * w[1], 1), 3),
                                           "shadow_nodes": round(d["nodes_shadow"] / max(64 * w[2], 1), 3), "shadow_tris": round(d["tris_shadow"] / max(64 * w[3], 1), 3),
                                           "shade": round(w[5] / max(64 * w[4], 1), 3), "shadow_lanes": round(w[6] / max(64 * w[4], 1), 3),
                                           "wave_steps_per_iter": [round(x / max(w[4], 1), 2) for x in w[:4]]})(d["wave_steps"]),
                  "busy_hist": {k: [round(x / max(sum(h), 1), 3) for x in h] for k, h in zip(("closest", "shadow"), d["busy_hist"])},
                  "divergence": (lambda v: {"classes_per_iter": round(v[1] / max(v[0], 1), 3), "largest_class_share": round(v[3] / max(v[2], 1), 3),
                                            "surface_lanes_per_iter": round(v[2] / max(v[0], 1), 1),
                                            "iters_by_bsdf_classes": [round(x / max(sum(v[4:8]), 1), 3) for x in v[4:8]],
                                            "shade_cycles_per_iter_by_bsdf_classes": [round(c / max(n, 1)) for n, c in zip(v[4:8], v[8:12])]})(d["divergence"]),
                  "per_sample": {k: round(d[k] / max(d["samples"], 1), 2) for k in ("closest_rays", "shadow_rays", "nodes_closest", "tris_closest", "nodes_shadow", "tris_shadow", "bounces")}}))
