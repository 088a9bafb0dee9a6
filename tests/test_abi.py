"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/mi355pt.h declares.
No compute calls here (there is no GPU in the build container)."""
import ctypes
import os
import re


def test_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(pkg.ffi.LIB_PATH)
    hdr = open(os.path.join(pkg.ffi.ROOT, "include", "mi355pt.h")).read()
    declared = sorted(set(re.findall(r"\b(mi355pt_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in mi355pt.h but not exported"
    assert sorted("mi355pt_" + s for s in pkg.ffi.ABI_SYMBOLS) == declared


def test_struct_layouts_match_header(pkg):
    f = pkg.ffi
    assert ctypes.sizeof(f.Spectrum) == 20
    assert ctypes.sizeof(f.MaterialDesc) == 4 + 20 + 4 + 4 + 4 + 20 + 4 + 4 + 5 * 4 + 20 + 20 + 8
    assert ctypes.sizeof(f.Camera) == 48
    assert ctypes.sizeof(f.Params) == 36
    assert ctypes.sizeof(f.Stats) == 11 * 8 + 10 * 8 + 8 + 8 + 8 * 8


def test_host_only_entry_points(pkg):
    """Entry points that do not touch the device behave without a GPU; compute entry points must refuse loudly."""
    import numpy as np
    prod = pkg.Product()
    assert "gfx950" in prod.version()
    q = prod.quantize_u8(np.array([0.0, 0.5, 0.999, 1.0, 2.0, -1.0, float("nan")], dtype=np.float32))
    assert q.tolist() == [0, 127, 254, 255, 255, 0, 0]      # `(p*255.0) as u8` truncation (renderer.rs:141-143)
    sc = prod.new_scene()
    bad = pkg.ffi.MaterialDesc(); bad.type = 77; bad.normal_tex = pkg.ffi.NONE
    try:
        sc.add_material(bad)
        raise AssertionError("bad material accepted")
    except RuntimeError as e:
        assert "not implemented" in str(e)
