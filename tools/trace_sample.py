#!/usr/bin/env python3
"""One sample's path vertices on both sides (GPU box only): the BSDF sample and the light connection of every vertex — wo, wi, f, pdf, the
random numbers, the throughput — printed by the kernel (a variant built with -DPT_TRACE_MORTON=<(morton2(x, y) << log2 spp | s) as u32>u,
tools/build_variant.sh; pt_path.hpp) and by the oracle (PTORACLE_TRACE=x,y,s).  The first number that differs is the root cause: this is how
the six entries of DESIGN.md 2.1's table were found.
usage: PTORACLE_TRACE=x,y,s MI355PT_LIB=build_variants/libmi355pt_<variant>.so tools/trace_sample.py <scene> <strategy> x y s <max_depth>"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
import ptoracle
prod, orc = pkg.Product(), ptoracle.Oracle()
sid, strat, x, y, s, md = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
W, H, S = 64, 48, 64
xys = np.array([[x, y, s]], np.uint32)
gsc = prod.new_scene(); gcam = pkg.scenes.load_scene(gsc, sid, W, H, tex_size=128)
osc = orc.new_scene()
if os.environ.get("ORACLE_RENDER_LOWERING") == "1": orc.set_render_space_lowering(osc, True)
ocam = pkg.scenes.load_scene(osc, sid, W, H, tex_size=128); orc.set_faithful(osc, False)
prm = pkg.make_params(S, strat, "sobol", max_depth=md)
print("gpu", gsc.probe_radiance(gcam, prm, xys)); sys.stdout.flush()
print("cpu", osc.probe_radiance(ocam, prm, xys))
