#!/usr/bin/env python3
"""GPU box: feature mask, kernel set and triangle count of every scene (which specialisation each one runs)."""
import importlib, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
prod = pkg.Product()
names = ["TEX", "DIEL", "CC", "MLIGHT", "ROUGH", "METAL", "DELTA", "ENV"]
for sid in list(range(0, 24)) + [27]:
    sc = prod.new_scene(); cam = pkg.scenes.load_scene(sc, sid, 64, 48, tex_size=64)
    info = prod.scene_info(sc)
    f = int(re.search(r"features=(\d+)", info).group(1))
    print(sid, f, "+".join(n for i, n in enumerate(names) if f >> i & 1) or "-", re.search(r"tris=(\d+)", info).group(1))
