"""ctypes view of include/mi355pt.h and a thin object wrapper over the C ABI.

This is plumbing for tests / bench / smoke: the product is libmi355pt.so (hand-written HIP for
gfx950 behind the C ABI).  There is no CPU fallback: if the shared library is missing this module
raises, and every compute entry point fails loudly when no gfx950 device is present.

`Backend` is prefix-agnostic so that the *oracle's* C entry points (oracle/libptoracle.so, prefix
`ptoracle_`, test infrastructure only) can be driven with the very same scene description — the
oracle binding itself lives in oracle/ptoracle.py, not here.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("MI355PT_LIB") or os.path.join(HERE, "csrc", "libmi355pt.so")   # env override: A/B builds while tuning

NONE = 0xFFFFFFFF
SPEC_CONSTANT, SPEC_RGB_ALBEDO_SRGB, SPEC_LUT470, SPEC_TEXTURE_ALBEDO_SRGB, SPEC_SIGMOID, SPEC_RGB_ALBEDO_SRGB_LINEAR = 0, 1, 2, 3, 4, 5
SPEC_TEXTURE_ILLUMINANT_SRGB, SPEC_TEXTURE_UNBOUNDED_SRGB = 6, 7
MAT_LAMBERT, MAT_EMISSIVE, MAT_GLASS, MAT_PLASTIC, MAT_CLEARCOAT, MAT_METAL, MAT_SIMPLE_PBR = 0, 1, 2, 3, 4, 5, 6
STRATEGY = {"pt": 0, "nee": 1, "mis": 2}
SAMPLER = {"random": 0, "sobol": 1}


class Spectrum(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("id", C.c_uint32), ("c", C.c_float * 3)]

    @staticmethod
    def constant(v):
        return Spectrum(SPEC_CONSTANT, 0, (C.c_float * 3)(v, 0, 0))

    @staticmethod
    def rgb_albedo_srgb(r, g, b):
        return Spectrum(SPEC_RGB_ALBEDO_SRGB, 0, (C.c_float * 3)(r, g, b))

    @staticmethod
    def rgb_albedo_srgb_linear(r, g, b):
        return Spectrum(SPEC_RGB_ALBEDO_SRGB_LINEAR, 0, (C.c_float * 3)(r, g, b))

    @staticmethod
    def lut(i):
        return Spectrum(SPEC_LUT470, i, (C.c_float * 3)(0, 0, 0))

    @staticmethod
    def texture_albedo_srgb(i):
        return Spectrum(SPEC_TEXTURE_ALBEDO_SRGB, i, (C.c_float * 3)(0, 0, 0))

    @staticmethod
    def texture_illuminant_srgb(i, illuminant_lut):
        """SpectrumParameter::texture(.., SpectrumType::Illuminant): c[0] carries the LUT470 id of presets::cie_illum_d6500()"""
        return Spectrum(SPEC_TEXTURE_ILLUMINANT_SRGB, i, (C.c_float * 3)(float(illuminant_lut), 0, 0))

    @staticmethod
    def texture_unbounded_srgb(i):
        return Spectrum(SPEC_TEXTURE_UNBOUNDED_SRGB, i, (C.c_float * 3)(0, 0, 0))


class MaterialDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("color", Spectrum), ("normal_tex", C.c_uint32), ("normal_flip_y", C.c_uint32),
                ("intensity", C.c_float), ("eta", Spectrum), ("thin", C.c_uint32), ("roughness", C.c_float),
                ("metallic", C.c_float), ("ior", C.c_float), ("clearcoat_ior", C.c_float), ("clearcoat_roughness", C.c_float),
                ("clearcoat_thickness", C.c_float), ("clearcoat_tint", Spectrum), ("k", Spectrum),
                ("metallic_tex", C.c_uint32), ("roughness_tex", C.c_uint32), ("clearcoat_thickness_tex", C.c_uint32),
                ("intensity_tex", C.c_uint32)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.metallic_tex = NONE; self.roughness_tex = NONE; self.clearcoat_thickness_tex = NONE; self.intensity_tex = NONE


LIGHT_POINT, LIGHT_SPOT, LIGHT_DIRECTIONAL = 1, 2, 3


class LightDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("intensity", C.c_float), ("angle_inner", C.c_float), ("angle_outer", C.c_float),
                ("spectrum", Spectrum), ("local_to_world", C.c_float * 16)]


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("direction", C.c_float * 3), ("up", C.c_float * 3),
                ("fov_deg", C.c_float), ("width", C.c_uint32), ("height", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("spp", C.c_uint32), ("seed", C.c_uint32), ("max_depth", C.c_uint32), ("strategy", C.c_uint32),
                ("sampler", C.c_uint32), ("exposure", C.c_float), ("shard_index", C.c_uint32),
                ("shard_count", C.c_uint32), ("collect_stats", C.c_uint32), ("rr_gate_slack", C.c_float), ("albedo_lut", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "closest_rays", "shadow_rays", "nodes_closest", "tris_closest",
                                          "nodes_shadow", "tris_shadow", "closest_hits", "bounces", "spectrum_evals",
                                          "textured_lookups")] + [("phase_cycles", C.c_uint64 * 10), ("kernel_ms", C.c_double), ("launches", C.c_uint32),
                                                ("wave_steps", C.c_uint64 * 8), ("busy_hist", C.c_uint64 * 16), ("divergence", C.c_uint64 * 12)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["phase_cycles"] = list(self.phase_cycles)
        d["wave_steps"] = list(self.wave_steps)
        d["busy_hist"] = [list(self.busy_hist)[:8], list(self.busy_hist)[8:]]
        d["divergence"] = list(self.divergence)
        return d


def make_camera(position, direction, up, width, height, fov_deg=45.0):
    return Camera((C.c_float * 3)(*position), (C.c_float * 3)(*direction), (C.c_float * 3)(*up), fov_deg, width, height)


def make_params(spp, strategy="mis", sampler="sobol", seed=0, max_depth=16, exposure=1.0, shard_index=0, shard_count=1,
                collect_stats=0, rr_gate_slack=0.0, albedo_lut=0):
    return Params(spp, seed, max_depth, STRATEGY[strategy], SAMPLER[sampler], exposure, shard_index, shard_count, collect_stats, rr_gate_slack,
                  albedo_lut)


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


# every symbol include/mi355pt.h declares (tests/test_abi.py checks the built library exports all of them) ...
DEBUG_SYMBOLS = ["debug_unlock", "scene_debug_set_lowering", "scene_export_bvh", "probe_bvh_collapse", "probe_bvh_collapse_nodes", "probe_sobol", "probe_sincos", "probe_intersect", "probe_occluded",
                 "sample_log_records", "render_sample_log", "probe_radiance"]     # ... and include/mi355pt_debug.h
ABI_SYMBOLS = [
    "scene_create", "scene_destroy", "scene_set_rgb2spec", "scene_add_lut470", "scene_add_tex_rgb8", "scene_add_mesh",
    "scene_add_material", "scene_add_instance", "scene_add_delta_light", "scene_add_environment_light", "scene_set_bvh_builder", "scene_build", "render", "render_accum_device", "film_resolve_device",
    "quantize_u8", "scene_info", "scene_build_multi", "render_multi", "coat_albedo_table",
    "last_error", "version",
]


class Backend:
    """Prefix-agnostic binding of the scene-construction / probe subset shared by product and oracle."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix
        f = self.fn
        f("scene_create").argtypes = [C.POINTER(C.c_void_p)]
        f("scene_destroy").argtypes = [C.c_void_p]; f("scene_destroy").restype = None
        f("scene_set_rgb2spec").argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t]
        f("scene_add_lut470").argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        f("scene_add_tex_rgb8").argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        f("scene_add_mesh").argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 4 + [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        f("scene_add_material").argtypes = [C.c_void_p, C.POINTER(MaterialDesc), C.POINTER(C.c_uint32)]
        f("scene_add_instance").argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        f("scene_add_delta_light").argtypes = [C.c_void_p, C.POINTER(LightDesc)]
        f("scene_add_environment_light").argtypes = [C.c_void_p, C.c_float, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_float),
                                                     C.c_uint32]
        f("scene_build").argtypes = [C.c_void_p, C.POINTER(Camera)]
        f("probe_sobol").argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32), C.c_uint32, C.c_char_p, C.POINTER(C.c_uint32)]
        f("probe_intersect").argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_float),
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        if prefix == "mi355pt_":                                              # (product only: the oracle IS the host libm)
            f("probe_sincos").argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        f("probe_occluded").argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32,
                                        C.POINTER(C.c_uint8)]
        f("probe_radiance").argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.POINTER(C.c_uint32), C.c_uint32,
                                        C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        f("quantize_u8").argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_uint8)]

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def check(self, rc, what):
        if rc != 0:
            msg = ""
            if self.prefix == "mi355pt_":
                self.lib.mi355pt_last_error.restype = C.c_char_p
                msg = (self.lib.mi355pt_last_error() or b"").decode()
            raise RuntimeError(f"{self.prefix}{what} failed with code {rc}: {msg}")

    def new_scene(self):
        return SceneHandle(self)

    def probe_sincos(self, first_bits, stride, n):
        """(compared, sin mismatches, cos mismatches) of the device's sin / cos against this host's libm (include/mi355pt_debug.h)"""
        out = (C.c_uint64 * 3)()
        self.check(self.fn("probe_sincos")(first_bits, stride, n, out), "probe_sincos")
        return int(out[0]), int(out[1]), int(out[2])

    def probe_sobol(self, width, height, spp, seed, xys, pattern):
        xys = np.ascontiguousarray(xys, dtype=np.uint32).reshape(-1, 3)
        per = sum(2 if ch == "2" else 1 for ch in pattern)
        out = np.zeros((xys.shape[0], per), dtype=np.uint32)
        self.check(self.fn("probe_sobol")(width, height, spp, seed, _ptr(xys, C.c_uint32), xys.shape[0], pattern.encode(),
                                           _ptr(out, C.c_uint32)), "probe_sobol")
        return out

    def quantize_u8(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        out = np.zeros(rgb.shape, dtype=np.uint8)
        self.check(self.fn("quantize_u8")(_ptr(rgb, C.c_float), rgb.size, _ptr(out, C.c_uint8)), "quantize_u8")
        return out


class SceneHandle:
    """Owns one opaque scene; mirrors scene::Scene's construction API (scene/src/scene.rs:43-76)."""

    def __init__(self, backend):
        self.b = backend
        self.h = C.c_void_p()
        backend.check(backend.fn("scene_create")(C.byref(self.h)), "scene_create")
        self.keep = []
        self.material_descs = []       # by material id (delta / environment lights add hidden materials on the library side, after these)

    def close(self):
        if self.h:
            self.b.fn("scene_destroy")(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_rgb2spec(self, table):
        table = np.ascontiguousarray(table, dtype=np.float32)
        self.b.check(self.b.fn("scene_set_rgb2spec")(self.h, _ptr(table, C.c_float), table.size), "scene_set_rgb2spec")

    def add_lut470(self, values):
        v = np.ascontiguousarray(values, dtype=np.float32)
        assert v.size == 470
        out = C.c_uint32()
        self.b.check(self.b.fn("scene_add_lut470")(self.h, _ptr(v, C.c_float), C.byref(out)), "scene_add_lut470")
        return out.value

    def add_tex_rgb8(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, c = img.shape
        assert c == 3
        out = C.c_uint32()
        self.b.check(self.b.fn("scene_add_tex_rgb8")(self.h, _ptr(img, C.c_uint8), w, h, C.byref(out)), "scene_add_tex_rgb8")
        return out.value

    def add_mesh(self, mesh):
        pos = np.ascontiguousarray(mesh["pos"], dtype=np.float32)
        nrm = np.ascontiguousarray(mesh["nrm"], dtype=np.float32)
        uv = None if mesh.get("uv") is None else np.ascontiguousarray(mesh["uv"], dtype=np.float32)
        tan = None if mesh.get("tangent") is None else np.ascontiguousarray(mesh["tangent"], dtype=np.float32)
        idx = np.ascontiguousarray(mesh["idx"], dtype=np.uint32).reshape(-1)
        out = C.c_uint32()
        self.b.check(self.b.fn("scene_add_mesh")(self.h, _ptr(pos, C.c_float), _ptr(nrm, C.c_float), _ptr(uv, C.c_float),
                                                  _ptr(tan, C.c_float), _ptr(idx, C.c_uint32), pos.shape[0], idx.size // 3,
                                                  C.byref(out)), "scene_add_mesh")
        return out.value

    def add_material(self, desc):
        out = C.c_uint32()
        self.b.check(self.b.fn("scene_add_material")(self.h, C.byref(desc), C.byref(out)), "scene_add_material")
        while len(self.material_descs) <= out.value:
            self.material_descs.append(None)
        self.material_descs[out.value] = desc
        return out.value

    def add_instance(self, geom, mat, local_to_world=None):
        m = np.eye(4, dtype=np.float32) if local_to_world is None else np.asarray(local_to_world, dtype=np.float32)
        cols = np.ascontiguousarray(m.T.reshape(-1))   # column-major
        self.b.check(self.b.fn("scene_add_instance")(self.h, geom, mat, _ptr(cols, C.c_float)), "scene_add_instance")

    def add_delta_light(self, kind, intensity, spectrum, local_to_world=None, angle_inner=0.0, angle_outer=0.0):
        m = np.eye(4, dtype=np.float32) if local_to_world is None else np.asarray(local_to_world, dtype=np.float32)
        d = LightDesc(); d.kind = kind; d.intensity = intensity; d.angle_inner = angle_inner; d.angle_outer = angle_outer
        d.spectrum = spectrum
        d.local_to_world = (C.c_float * 16)(*np.ascontiguousarray(m.T.reshape(-1)))   # column-major
        self.b.check(self.b.fn("scene_add_delta_light")(self.h, C.byref(d)), "scene_add_delta_light")

    def add_environment_light(self, intensity, rgb, illuminant_lut, local_to_world=None):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        m = np.eye(4, dtype=np.float32) if local_to_world is None else np.asarray(local_to_world, dtype=np.float32)
        cols = np.ascontiguousarray(m.T.reshape(-1))
        self.b.check(self.b.fn("scene_add_environment_light")(self.h, intensity, _ptr(rgb, C.c_float), rgb.shape[1], rgb.shape[0],
                                                              _ptr(cols, C.c_float), illuminant_lut), "scene_add_environment_light")

    def set_bvh_builder(self, mode):
        """mi355pt_scene_set_bvh_builder: "auto" | "host" | "gpu" (product only; the oracle has its own BVH)."""
        fn = self.b.fn("scene_set_bvh_builder")
        fn.argtypes = [C.c_void_p, C.c_int]
        self.b.check(fn(self.h, {"auto": 0, "host": 1, "gpu": 2}[mode]), "scene_set_bvh_builder")

    def debug_set_lowering(self, mode):
        """mi355pt_scene_debug_set_lowering (include/mi355pt_debug.h): "auto" | "no_local_tris" | "general" (product only)."""
        fn = self.b.fn("scene_debug_set_lowering")
        fn.argtypes = [C.c_void_p, C.c_int]
        self.b.check(fn(self.h, {"auto": 0, "no_local_tris": 1, "general": 2}[mode]), "scene_debug_set_lowering")

    def build(self, cam):
        self.b.check(self.b.fn("scene_build")(self.h, C.byref(cam)), "scene_build")

    # ---- probes ----
    def probe_intersect(self, origins, dirs):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.zeros(n, np.float32); inst = np.zeros(n, np.uint32); tri = np.zeros(n, np.uint32); nrm = np.zeros((n, 3), np.float32)
        self.b.check(self.b.fn("probe_intersect")(self.h, _ptr(o, C.c_float), _ptr(d, C.c_float), n, _ptr(t, C.c_float),
                                                   _ptr(inst, C.c_uint32), _ptr(tri, C.c_uint32), _ptr(nrm, C.c_float)), "probe_intersect")
        return t, inst, tri, nrm

    def probe_occluded(self, origins, dirs, t_max):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        tm = np.ascontiguousarray(t_max, dtype=np.float32).reshape(-1)
        out = np.zeros(o.shape[0], np.uint8)
        self.b.check(self.b.fn("probe_occluded")(self.h, _ptr(o, C.c_float), _ptr(d, C.c_float), _ptr(tm, C.c_float), o.shape[0],
                                                  _ptr(out, C.c_uint8)), "probe_occluded")
        return out

    def probe_radiance(self, cam, params, xys):
        xys = np.ascontiguousarray(xys, dtype=np.uint32).reshape(-1, 3)
        n = xys.shape[0]
        L = np.zeros((n, 4), np.float32); lam = np.zeros((n, 4), np.float32); pdf = np.zeros((n, 4), np.float32)
        self.b.check(self.b.fn("probe_radiance")(self.h, C.byref(cam), C.byref(params), _ptr(xys, C.c_uint32), n, _ptr(L, C.c_float),
                                                  _ptr(lam, C.c_float), _ptr(pdf, C.c_float)), "probe_radiance")
        return L, lam, pdf


class Product(Backend):
    """libmi355pt.so — the HIP product.  Raises if the extension has not been built."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
        lib = C.CDLL(path)
        super().__init__(lib, "mi355pt_")
        lib.mi355pt_version.restype = C.c_char_p
        lib.mi355pt_last_error.restype = C.c_char_p
        lib.mi355pt_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.POINTER(C.c_float), C.POINTER(Stats)]
        lib.mi355pt_render_accum_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_uint32, C.c_uint32,
                                                    C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        lib.mi355pt_film_resolve_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        if not hasattr(lib, "mi355pt_render_sample_log"):      # an older build loaded through MI355PT_LIB for an A/B timing run
            return
        lib.mi355pt_sample_log_records.argtypes = [C.POINTER(Camera), C.POINTER(Params), C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t)]
        lib.mi355pt_render_sample_log.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_uint32, C.c_uint32] + \
            [C.POINTER(C.c_float)] * 3 + [C.c_size_t, C.POINTER(C.c_float)]

    def version(self):
        return self.lib.mi355pt_version().decode()

    def debug_unlock(self, on=True):
        """mi355pt_debug_unlock (mi355pt_debug.h): lets params.rr_gate_slack through; returns the previous state.  The parity tests that
        relax the Russian-roulette gate switch it on for themselves (tests/conftest.py `product` fixture)."""
        if not hasattr(self.lib, "mi355pt_debug_unlock"):      # an older build loaded through MI355PT_LIB
            return True
        return bool(self.lib.mi355pt_debug_unlock(1 if on else 0))

    def scene_info(self, scene):
        buf = C.create_string_buffer(256)
        self.lib.mi355pt_scene_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        self.check(self.lib.mi355pt_scene_info(scene.h, buf, 256), "scene_info")
        return buf.value.decode()

    def render(self, scene, cam, params, want_stats=False):
        """RendererImage::render -> (H, W, 3) float32 tone-mapped sRGB in [0,1] (renderer.rs:120-134)."""
        out = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        st = Stats()
        self.check(self.lib.mi355pt_render(scene.h, C.byref(cam), C.byref(params), _ptr(out, C.c_float), C.byref(st)), "render")
        return (out, st) if want_stats else out

    def render_accum_device(self, scene, cam, params, s_begin, s_end, d_accum_ptr, stream=None, stats=None):
        self.check(self.lib.mi355pt_render_accum_device(scene.h, C.byref(cam), C.byref(params), s_begin, s_end, C.c_void_p(d_accum_ptr),
                                                        C.c_void_p(stream or 0), C.byref(stats) if stats is not None else None),
                   "render_accum_device")

    def render_sample_log(self, scene, cam, params, s_begin, s_end, want_accum=False):
        """mi355pt_render_sample_log: every finished path of the production launch(es) for [s_begin, s_end) of the shard in `params`.
        Returns L, lambda, pdf as (tiles_of_shard, 64, s_end - s_begin, 4) arrays (pixel (y & 7) * 8 + (x & 7) inside its 8x8 tile;
        tile k of the shard is frame tile shard_index + k * shard_count) and, on request, the (H, W, 3) linear film sums."""
        n = C.c_size_t()
        self.check(self.lib.mi355pt_sample_log_records(C.byref(cam), C.byref(params), s_begin, s_end, C.byref(n)), "sample_log_records")
        ns = s_end - s_begin
        shape = (n.value // (64 * ns), 64, ns, 4)
        L = np.zeros(shape, np.float32); lam = np.zeros(shape, np.float32); pdf = np.zeros(shape, np.float32)
        acc = np.zeros((cam.height, cam.width, 3), np.float32) if want_accum else None
        self.check(self.lib.mi355pt_render_sample_log(scene.h, C.byref(cam), C.byref(params), s_begin, s_end, _ptr(L, C.c_float),
                                                      _ptr(lam, C.c_float), _ptr(pdf, C.c_float), n.value, _ptr(acc, C.c_float)), "render_sample_log")
        return (L, lam, pdf, acc) if want_accum else (L, lam, pdf)

    def export_bvh(self, scene):
        """mi355pt_scene_export_bvh -> (nodes (n, 16) uint32 view of the 64-B records, tris (m, 12) uint32 view of the 48-B records, root)"""
        fn = self.lib.mi355pt_scene_export_bvh
        fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
        nn, nt, root = C.c_uint32(0), C.c_uint32(0), C.c_int32(0)
        self.check(fn(scene.h, None, C.byref(nn), None, C.byref(nt), C.byref(root)), "scene_export_bvh")
        nodes = np.zeros((nn.value, 16), np.uint32); tris = np.zeros((nt.value, 12), np.uint32)
        self.check(fn(scene.h, nodes.ctypes.data, C.byref(nn), tris.ctypes.data, C.byref(nt), C.byref(root)), "scene_export_bvh")
        return nodes, tris, root.value

    def probe_bvh_collapse(self, tri_pos, rays_od):
        """mi355pt_probe_bvh_collapse (host-only) -> (info dict, rays whose leaf sets differ between the BVH2 and the collapsed BVH4)"""
        tri = np.ascontiguousarray(tri_pos, dtype=np.float32).reshape(-1, 9); rays = np.ascontiguousarray(rays_od, dtype=np.float32).reshape(-1, 6)
        info = np.zeros(4, np.uint32); mism = C.c_uint32(0)
        fn = self.lib.mi355pt_probe_bvh_collapse
        fn.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        self.check(fn(_ptr(tri, C.c_float), tri.shape[0], _ptr(rays, C.c_float), rays.shape[0], _ptr(info, C.c_uint32), C.byref(mism)), "probe_bvh_collapse")
        return dict(nodes2=int(info[0]), nodes4=int(info[1]), depth2=int(info[2]), max_stack4=int(info[3])), mism.value

    def probe_bvh_collapse_nodes(self, nodes, root, n_tris):
        """mi355pt_probe_bvh_collapse_nodes (host-only): the collapse + validation SceneImpl::build runs, on a caller-supplied BVH2
        ((n, 16) uint32 view of the 64-B records, as export_bvh returns).  Raises RuntimeError when the tree is refused."""
        nodes = np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1, 16)
        info = np.zeros(4, np.uint32)
        fn = self.lib.mi355pt_probe_bvh_collapse_nodes
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_uint32)]
        self.check(fn(nodes.ctypes.data, nodes.shape[0], root, n_tris, _ptr(info, C.c_uint32)), "probe_bvh_collapse_nodes")
        return dict(nodes2=int(info[0]), nodes4=int(info[1]), dp=bool(info[2]), max_stack4=int(info[3]))

    def coat_albedo_table(self, alpha, r0):
        """mi355pt_coat_albedo_table: the 64-entry E(cos theta) table behind params.albedo_lut (host-only)."""
        out = np.zeros(64, np.float32)
        self.lib.mi355pt_coat_albedo_table.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float)]
        self.check(self.lib.mi355pt_coat_albedo_table(alpha, r0, _ptr(out, C.c_float)), "coat_albedo_table")
        return out

    def build_multi(self, scene, cam, device_ids):
        """mi355pt_scene_build_multi: replicate the (described, not yet built) scene on the listed devices."""
        ids = (C.c_int * len(device_ids))(*device_ids)
        self.lib.mi355pt_scene_build_multi.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int, C.POINTER(C.c_int)]
        self.check(self.lib.mi355pt_scene_build_multi(scene.h, C.byref(cam), len(device_ids), ids), "scene_build_multi")

    def render_multi(self, scene, cam, params):
        """mi355pt_render_multi -> (H, W, 3) float32, the frame mi355pt_render returns, rendered by all devices of the scene."""
        out = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        self.lib.mi355pt_render_multi.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.POINTER(C.c_float)]
        self.check(self.lib.mi355pt_render_multi(scene.h, C.byref(cam), C.byref(params), _ptr(out, C.c_float)), "render_multi")
        return out

    def film_resolve_device(self, d_accum_ptr, n_pixels, spp, d_out_ptr, stream=None):
        self.check(self.lib.mi355pt_film_resolve_device(C.c_void_p(d_accum_ptr), n_pixels, spp, C.c_void_p(d_out_ptr),
                                                        C.c_void_p(stream or 0)), "film_resolve_device")
