import sys, importlib
sys.path.insert(0, '/root/repo')
import numpy as np
pkg = importlib.import_module("toy-cpu-pathtracing_amd")
prod = pkg.Product()
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for sid in (3, 8):
    sc = prod.new_scene(); cam = pkg.scenes.load_scene(sc, sid, 640, 360, tex_size=256)
    imgs = [prod.render(sc, cam, pkg.make_params(SPP, "mis", "sobol")) for _ in range(3)]
    print(sid, "run-to-run identical:", np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[1], imgs[2]), float(np.abs(imgs[0]-imgs[1]).max()))
