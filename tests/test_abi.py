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
    assert ctypes.sizeof(f.MaterialDesc) == 4 + 20 + 4 + 4 + 4 + 20 + 4 + 4 + 5 * 4 + 20 + 20 + 12
    assert ctypes.sizeof(f.Camera) == 48
    assert ctypes.sizeof(f.Params) == 40
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


def test_invalid_inputs_return_error_codes(pkg):
    """Error behaviour of the boundary: bad ids / sizes / null data are refused with a message (never a crash, never an
    approximation).  Host-only: nothing here reaches the device."""
    import numpy as np
    import pytest
    f = pkg.ffi
    prod = pkg.Product()
    sc = prod.new_scene()
    with pytest.raises(AssertionError):                                 # LUT of the wrong length (caught by the binding)
        sc.add_lut470(np.zeros(100, np.float32))
    lut = sc.add_lut470(np.ones(470, np.float32))
    with pytest.raises(RuntimeError):                                   # texture id that does not exist
        d = f.MaterialDesc(); d.type = f.MAT_LAMBERT; d.color = f.Spectrum.constant(0.5); d.normal_tex = 5
        sc.add_material(d)
    with pytest.raises(RuntimeError):                                   # RGB albedo before the rgb2spec table is set
        d = f.MaterialDesc(); d.type = f.MAT_LAMBERT; d.color = f.Spectrum.rgb_albedo_srgb(0.5, 0.2, 0.1); d.normal_tex = f.NONE
        sc.add_material(d)
    with pytest.raises(RuntimeError):                                   # emissive radiance cannot be a texture
        d = f.MaterialDesc(); d.type = f.MAT_EMISSIVE; d.color = f.Spectrum.texture_albedo_srgb(0); d.normal_tex = f.NONE
        sc.add_material(d)
    d = f.MaterialDesc(); d.type = f.MAT_LAMBERT; d.color = f.Spectrum.constant(0.5); d.normal_tex = f.NONE
    mat = sc.add_material(d)
    tri = dict(pos=np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), nrm=np.array([[0, 0, 1]] * 3, np.float32), uv=None,
               idx=np.array([[0, 1, 2]], np.uint32), tangent=None)
    g = sc.add_mesh(tri)
    with pytest.raises(RuntimeError):                                   # index out of range
        bad = dict(tri); bad["idx"] = np.array([[0, 1, 7]], np.uint32)
        sc.add_mesh(bad)
    with pytest.raises(RuntimeError, match="non-finite"):               # NaN / inf vertex positions
        bad = dict(tri); bad["pos"] = tri["pos"].copy(); bad["pos"][1, 2] = np.inf
        sc.add_mesh(bad)
    with pytest.raises(RuntimeError):                                   # unknown geometry / material ids
        sc.add_instance(g + 3, mat)
    with pytest.raises(RuntimeError):
        sc.add_instance(g, mat + 9)
    with pytest.raises(RuntimeError):                                   # unknown light kind
        sc.add_delta_light(9, 1.0, f.Spectrum.lut(lut))
    with pytest.raises(RuntimeError):                                   # environment map with a bad illuminant id
        sc.add_environment_light(1.0, np.ones((4, 8, 3), np.float32), lut + 5)
    sc.set_bvh_builder("gpu"); sc.set_bvh_builder("auto")
    fn = sc.b.fn("scene_set_bvh_builder")
    assert fn(sc.h, 7) == -1                                            # unknown builder mode
    sc.add_instance(g, mat)
    cam = f.make_camera((0, 0, 3), (0, 0, -1), (0, 1, 0), 16, 16)
    with pytest.raises(RuntimeError) as e:                              # no GPU here: the product refuses, it does not fall back
        sc.build(cam)
    assert "HIP device" in str(e.value) or "gfx950" in str(e.value)
