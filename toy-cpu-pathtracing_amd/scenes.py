"""Scene authoring: the four BASELINE configs' scenes, mirroring renderer/src/scene/scene_{3,8,10,17}.rs
call for call against the C ABI (works with any ffi.Backend).  Assets are the synthetic stand-ins of
assets.py (the reference's OBJ/PNG files are LFS stubs)."""
import json
import os
import subprocess

import numpy as np

from . import assets
from .ffi import (HERE, LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_SPOT, MAT_CLEARCOAT, MAT_EMISSIVE, MAT_GLASS, MAT_LAMBERT, MAT_METAL, MAT_PLASTIC, MAT_SIMPLE_PBR, NONE, MaterialDesc,
                  Spectrum, make_camera)

DATA = os.path.join(HERE, "data")


def presets():
    """{name: 470 x f32} baked by tools/bake_presets.py from spectrum/src/presets.rs."""
    names = json.load(open(os.path.join(DATA, "presets470.json")))["names"]
    d = np.fromfile(os.path.join(DATA, "presets470.bin"), dtype="<f4").reshape(len(names), 470)
    return {n: d[i] for i, n in enumerate(names)}


def cmf_xyz():
    p = presets()
    return np.ascontiguousarray(np.stack([p["cie_x"], p["cie_y"], p["cie_z"]]), dtype=np.float32)


def srgb_table():
    """The sRGB coefficient table (rgb_to_spec/tables/srgb_table.bin layout), generated on demand."""
    path = os.path.join(DATA, "srgb_table.bin")
    if not os.path.exists(path):
        root = os.path.dirname(HERE)
        exe = os.path.join(root, "tools", "rgb2spec_fit")
        if not os.path.exists(exe):
            subprocess.check_call(["g++", "-O3", "-std=c++17", "-pthread", "-o", exe, os.path.join(root, "tools", "rgb2spec_fit.cpp")])
        names = json.load(open(os.path.join(DATA, "presets470.json")))["names"]
        ix = [str(names.index(k)) for k in ("cie_x", "cie_y", "cie_z", "cie_illum_d6500")]
        subprocess.check_call([exe, os.path.join(DATA, "presets470.bin"), *ix, path, str(min(16, os.cpu_count() or 1))])
    t = np.fromfile(path, dtype="<f4")
    assert t.size == 64 + 3 * 64 ** 3 * 3
    return t


_ASSET_CACHE = {}


def _asset(name):
    if name not in _ASSET_CACHE:
        if name == "room":
            _ASSET_CACHE[name] = {k: assets.load_obj_semantics(v) for k, v in assets.cornell_room().items()}
        elif name == "bunny":
            _ASSET_CACHE[name] = assets.load_obj_semantics(assets.bunny_class())
        elif name == "dragon":
            _ASSET_CACHE[name] = assets.load_obj_semantics(assets.dragon_class())
        elif name.startswith("tex"):                      # bunny-material-0
            _ASSET_CACHE[name] = assets.bunny_textures(int(name[3:]))
        elif name.startswith("m1tex"):                    # bunny-material-1 (scene 4)
            _ASSET_CACHE[name] = assets.bunny_textures(int(name[5:]), seed=6)
        elif name.startswith("dtex"):                     # dragon-material: BaseColor, Normal
            _ASSET_CACHE[name] = assets.bunny_textures(int(name[4:]), seed=9)
        elif name.startswith("dmaps"):                    # dragon-material: Metallic, Roughness, ClearcoatThickness
            _ASSET_CACHE[name] = assets.material_maps(int(name[5:]))
    return _ASSET_CACHE[name]


def lambert(color, normal_tex=NONE):
    d = MaterialDesc(); d.type = MAT_LAMBERT; d.color = color; d.normal_tex = normal_tex; d.normal_flip_y = 0
    return d


def _rot_y(deg):
    """Mat4::from_quat(Quat::from_rotation_y(deg.to_radians())) in f32, glam's formula (1 - y*(y+y), w*(y+y))."""
    a = np.float32(np.float32(deg) * np.float32(np.pi / 180.0))
    y, w = np.float32(np.sin(a * np.float32(0.5))), np.float32(np.cos(a * np.float32(0.5)))
    y2 = np.float32(y + y); yy = np.float32(y * y2); wy = np.float32(w * y2)
    c, s = np.float32(np.float32(1.0) - yy), wy
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float32)


def _translate(x, y, z):
    m = np.eye(4, dtype=np.float32); m[:3, 3] = np.array([x, y, z], dtype=np.float32)
    return m


def _room(scene, p, with_box=True, with_light=True):
    """box/hidari/migi/yuka/oku/tenjou/light — scene_3.rs:33-107 (identical in scenes 8, 10, 17)."""
    room = _asset("room")
    grey = Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8)
    order = [("box", grey), ("hidari", Spectrum.rgb_albedo_srgb(0.9, 0.0, 0.0)), ("migi", Spectrum.rgb_albedo_srgb(0.0, 0.9, 0.0)),
             ("yuka", grey), ("oku", grey), ("tenjou", grey)]
    for name, col in order:
        if name == "box" and not with_box:
            continue
        g = scene.add_mesh(room[name])
        scene.add_instance(g, scene.add_material(lambert(col)))
    if not with_light:
        return
    d65 = scene.add_lut470(p["cie_illum_d6500"])
    em = MaterialDesc(); em.type = MAT_EMISSIVE; em.color = Spectrum.lut(d65); em.intensity = 10.0; em.normal_tex = NONE
    g = scene.add_mesh(room["light"])
    scene.add_instance(g, scene.add_material(em))


def _mat4_trs_scene17():
    """Transform::identity().rotate(Quat::from_euler(XYZ, 0, 120deg, 0)).scale(2.5).translate(0,0,0.5)
    = T * S * R  (scene_17.rs:61-69, transform.rs:128-145), float32."""
    a = np.float32(np.float32(120.0) * np.float32(np.pi / 180.0))          # f32::to_radians
    # glam: q = (0, sin(a/2), 0, cos(a/2)); Mat4::from_quat builds 1 - y*(y+y) and w*(y+y)
    y, w = np.float32(np.sin(a * np.float32(0.5))), np.float32(np.cos(a * np.float32(0.5)))
    y2 = np.float32(y + y); yy = np.float32(y * y2); wy = np.float32(w * y2)
    c, s = np.float32(np.float32(1.0) - yy), wy
    R = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float32)
    S = np.diag(np.array([2.5, 2.5, 2.5, 1], dtype=np.float32))
    T = np.eye(4, dtype=np.float32); T[2, 3] = 0.5
    return (T @ (S @ R)).astype(np.float32)


def load_scene(scene, scene_id, width, height, tex_size=1024, build=True):
    """load_scene_N(&mut scene, &mut camera) + scene.build(&camera) (main.rs:70-106).  Returns the camera.
    build=False: describe only (the caller builds, e.g. with Product.build_multi)."""
    p = presets()
    scene.set_rgb2spec(srgb_table())
    if scene_id == 3:      # scene_3.rs:13-31: textured + normal-mapped Lambert hero
        g = scene.add_mesh(_asset("bunny"))
        albedo, normal = _asset(f"tex{tex_size}")
        t_alb = scene.add_tex_rgb8(albedo)
        t_nrm = scene.add_tex_rgb8(normal)
        scene.add_instance(g, scene.add_material(lambert(Spectrum.texture_albedo_srgb(t_alb), t_nrm)))
        _room(scene, p)
        cam = make_camera((0.0, 3.15221, 6.0), (0.0, -0.9, -3.2), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 0:    # scene_0.rs: constant-colour Lambert hero (used by small tests)
        g = scene.add_mesh(_asset("bunny"))
        scene.add_instance(g, scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8))))
        _room(scene, p)
        cam = make_camera((0.0, 3.15221, 6.0), (0.0, -0.9, -3.2), (0.0, 1.0, 0.0), width, height)
    elif scene_id in (8, 11):    # scene_8.rs:17-27: SF11 glass hero; scene_11.rs:15-26: the same with roughness 0.2
        g = scene.add_mesh(_asset("bunny"))
        d = MaterialDesc(); d.type = MAT_GLASS; d.eta = Spectrum.lut(scene.add_lut470(p["glass_sf11_eta"]))
        d.normal_tex = NONE; d.thin = 0; d.roughness = 0.0 if scene_id == 8 else 0.2; d.color = Spectrum.constant(1.0)
        scene.add_instance(g, scene.add_material(d))
        _room(scene, p)
        cam = make_camera((0.0, 3.5, 6.0), (0.0, -1.0, -3.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 10:   # scene_10.rs:17-30: smooth thin-film plastic eta 1.8
        g = scene.add_mesh(_asset("bunny"))
        d = MaterialDesc(); d.type = MAT_PLASTIC; d.eta = Spectrum.constant(1.8); d.color = Spectrum.constant(1.0)
        d.normal_tex = NONE; d.thin = 1; d.roughness = 0.0
        scene.add_instance(g, scene.add_material(d))
        _room(scene, p)
        cam = make_camera((0.0, 3.5, 6.0), (0.0, -1.0, -3.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 1:    # scene_1.rs:12-87: two Lambert heroes, floor, one triangle, two D65 point lights (no area light)
        g = scene.add_mesh(_asset("bunny"))
        scene.add_instance(g, scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.5, 0.5, 0.8))))
        scene.add_instance(g, scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.5, 0.8, 0.5))),
                           (_translate(-1.0, 1.0, 3.0) @ _rot_y(30.0)).astype(np.float32))      # from_rotate(..).translate(..) = T * R
        room = _asset("room")
        scene.add_instance(scene.add_mesh(room["yuka"]), scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8))))
        tri = assets.load_obj_semantics(assets.single_triangle())                                # SingleTrianglePrimitive :46-67
        scene.add_instance(scene.add_mesh(tri), scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.5, 0.5))), _rot_y(60.0))
        d65 = Spectrum.lut(scene.add_lut470(p["cie_illum_d6500"]))
        scene.add_delta_light(LIGHT_POINT, 10.0, d65, _translate(0.0, 3.0, 0.0))
        scene.add_delta_light(LIGHT_POINT, 10.0, d65, _translate(3.0, 5.0, 0.0))
        cam = make_camera((0.0, 3.5, 7.0), (0.0, -1.0, -3.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 2:    # scene_2.rs:12-101: Lambert hero in the Cornell room lit by one D65 point light
        g = scene.add_mesh(_asset("bunny"))
        scene.add_instance(g, scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8))))
        _room(scene, p, with_light=False)
        scene.add_delta_light(LIGHT_POINT, 10.0, Spectrum.lut(scene.add_lut470(p["cie_illum_d6500"])), _translate(0.0, 3.0, 0.0))
        cam = make_camera((0.0, 3.5, 6.0), (0.0, -1.0, -3.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 19:   # scene_19.rs:17-153: floor + three heroes (SimplePbr, clearcoat, plastic) under an environment light.
        # Stand-ins: constant metallic/roughness instead of FloatTexture maps, synthetic sky.
        room = _asset("room")
        scene.add_instance(scene.add_mesh(room["yuka"]), scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8))))
        g = scene.add_mesh(_asset("dragon"))
        d = MaterialDesc(); d.type = MAT_SIMPLE_PBR; d.color = Spectrum.rgb_albedo_srgb(0.8, 0.6, 0.3)
        d.metallic = 0.5; d.roughness = 0.4; d.normal_tex = NONE; d.ior = 1.5
        scene.add_instance(g, scene.add_material(d))
        d = MaterialDesc(); d.type = MAT_CLEARCOAT; d.color = Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8)
        d.metallic = 1.0; d.roughness = 0.7; d.normal_tex = NONE; d.ior = 1.5; d.clearcoat_ior = 1.5
        d.clearcoat_roughness = 0.01; d.clearcoat_tint = Spectrum.rgb_albedo_srgb(0.7, 0.8, 1.0); d.clearcoat_thickness = 0.8
        scene.add_instance(g, scene.add_material(d), _translate(0.5, 0.0, 0.5))
        d = MaterialDesc(); d.type = MAT_PLASTIC; d.eta = Spectrum.constant(1.5); d.color = Spectrum.rgb_albedo_srgb_linear(0.4, 0.9, 1.0)
        d.normal_tex = NONE; d.thin = 0; d.roughness = 0.0
        scene.add_instance(g, scene.add_material(d), _translate(-0.5, 0.0, -0.5))
        scene.add_environment_light(1.0, assets.sky_envmap(), scene.add_lut470(p["cie_illum_d6500"]))
        cam = make_camera((-1.5, 0.8, 2.5), (1.5, -0.4, -2.5), (0.0, 1.0, 0.0), width, height)
    elif scene_id in (4, 5):     # scene_4.rs: scene 3 with bunny-material-1; scene_5.rs: grey Lambert + normal map, close-up camera
        g = scene.add_mesh(_asset("bunny"))
        if scene_id == 4:
            albedo, normal = _asset(f"m1tex{tex_size}")
            col = Spectrum.texture_albedo_srgb(scene.add_tex_rgb8(albedo))
        else:
            _, normal = _asset(f"tex{tex_size}")
            col = Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8)
        scene.add_instance(g, scene.add_material(lambert(col, scene.add_tex_rgb8(normal))))
        _room(scene, p)
        cam = (make_camera((0.0, 3.5, 6.0), (0.0, -1.0, -3.0), (0.0, 1.0, 0.0), width, height) if scene_id == 4 else
               make_camera((0.3, 1.6, 2.8), (0.0, -0.5, -2.0), (0.0, 1.0, 0.0), width, height))
    elif scene_id in (9, 13):    # scene_9.rs: plastic eta 1.8, not thin; scene_13.rs: eta 1.5, linear-sRGB (0.4, 0.9, 1.0) colour
        g = scene.add_mesh(_asset("bunny"))
        d = MaterialDesc(); d.type = MAT_PLASTIC; d.normal_tex = NONE; d.thin = 0; d.roughness = 0.0
        d.eta = Spectrum.constant(1.8 if scene_id == 9 else 1.5)
        d.color = Spectrum.constant(1.0) if scene_id == 9 else Spectrum.rgb_albedo_srgb_linear(0.4, 0.9, 1.0)
        scene.add_instance(g, scene.add_material(d))
        _room(scene, p)
        cam = make_camera((0.0, 3.5, 6.0), (0.0, -1.0, -3.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id in (12, 14):   # scene_12.rs: four BK7 glass heroes, roughness 0.05..0.75; scene_14.rs: four coloured plastic heroes
        g = scene.add_mesh(_asset("bunny"))
        positions = [(-1.3, 0.0, -0.5), (-0.5, 0.0, -0.5), (0.3, 0.0, -0.5), (1.1, 0.0, -0.5)]
        if scene_id == 12:
            bk7 = Spectrum.lut(scene.add_lut470(p["glass_bk7_eta"]))
            specs = [(MAT_GLASS, bk7, Spectrum.constant(1.0), r) for r in (0.05, 0.25, 0.5, 0.75)]
        else:
            cols = [(1.0, 0.5, 0.5), (0.5, 1.0, 0.5), (0.5, 0.5, 1.0), (1.0, 0.8, 0.4)]
            specs = [(MAT_PLASTIC, Spectrum.constant(1.5), Spectrum.rgb_albedo_srgb(*c), r) for c, r in zip(cols, (0.05, 0.1, 0.3, 0.5))]
        for pos, (mt, eta, col, rough) in zip(positions, specs):
            d = MaterialDesc(); d.type = mt; d.eta = eta; d.color = col; d.normal_tex = NONE; d.thin = 0; d.roughness = rough
            m = np.diag(np.array([0.6, 0.6, 0.6, 1.0], dtype=np.float32)); m[:3, 3] = np.array(pos, dtype=np.float32)
            scene.add_instance(g, scene.add_material(d), m)
        _room(scene, p)
        cam = make_camera((0.0, 3.5, 6.0), (0.0, -1.0, -3.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id in (15, 16, 18):   # the dragon hero of scene 17 with: SimplePbr + texture maps (scene_15.rs), a near-smooth coat
        # (clearcoat roughness 0.01, scene_16.rs), the same with a FloatTexture coat thickness (scene_18.rs)
        g = scene.add_mesh(_asset("dragon"))
        if scene_id == 15:
            albedo, normal = _asset(f"dtex{tex_size}")
            met, rgh, _ = _asset(f"dmaps{tex_size}")
            d = MaterialDesc(); d.type = MAT_SIMPLE_PBR; d.color = Spectrum.texture_albedo_srgb(scene.add_tex_rgb8(albedo))
            d.metallic_tex = scene.add_tex_rgb8(met); d.roughness_tex = scene.add_tex_rgb8(rgh)
            d.normal_tex = scene.add_tex_rgb8(normal); d.normal_flip_y = 0; d.ior = 1.5
        else:
            d = MaterialDesc(); d.type = MAT_CLEARCOAT; d.color = Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8)
            d.metallic = 1.0; d.roughness = 0.7; d.normal_tex = NONE; d.ior = 1.5; d.clearcoat_ior = 1.5
            d.clearcoat_roughness = 0.01; d.clearcoat_tint = Spectrum.rgb_albedo_srgb(0.7, 0.8, 1.0); d.clearcoat_thickness = 0.8
            if scene_id == 18:
                d.clearcoat_thickness_tex = scene.add_tex_rgb8(_asset(f"dmaps{tex_size}")[2])
        scene.add_instance(g, scene.add_material(d), _mat4_trs_scene17())
        _room(scene, p)
        cam = make_camera((0.0, 3.15221, 6.0), (0.0, -0.9, -3.2), (0.0, 1.0, 0.0), width, height)
    elif scene_id in (27, 28):   # not a reference scene: GlassMaterial / PlasticMaterial with FloatParameter::Texture roughness (28: the
        # same heroes with the constant roughness 0, to show that the map is what makes the difference)
        # (glass_material.rs:42,116, plastic_material.rs:43,104): BK7 glass and a coloured plastic hero whose roughness comes from a
        # grey map that covers 0 (effectively smooth: the specular branch) to 0.5 (microfacet branch) across the surface
        g = scene.add_mesh(_asset("bunny"))
        albedo, _ = _asset(f"tex{tex_size}")
        grey = lambda a: np.ascontiguousarray(np.repeat(a[..., None], 3, -1))
        rmap = np.where(albedo[..., 1] < 96, 0, albedo[..., 1] // 2).astype(np.uint8)
        t_rgh = scene.add_tex_rgb8(grey(rmap))
        bk7 = Spectrum.lut(scene.add_lut470(p["glass_bk7_eta"]))
        for pos, (mt, eta, col) in zip([(-0.7, 0.0, -0.5), (0.8, 0.0, -0.3)],
                                       [(MAT_GLASS, bk7, Spectrum.constant(1.0)), (MAT_PLASTIC, Spectrum.constant(1.5), Spectrum.rgb_albedo_srgb(0.5, 0.8, 1.0))]):
            d = MaterialDesc(); d.type = mt; d.eta = eta; d.color = col; d.normal_tex = NONE; d.thin = 0; d.roughness = 0.0; d.roughness_tex = t_rgh if scene_id == 27 else NONE
            m = np.diag(np.array([0.9, 0.9, 0.9, 1.0], dtype=np.float32)); m[:3, 3] = np.array(pos, dtype=np.float32)
            scene.add_instance(g, scene.add_material(d), m)
        _room(scene, p)
        cam = make_camera((0.0, 3.5, 6.0), (0.0, -1.0, -3.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 26:   # not a reference scene: no light at all (every strategy must return a black frame)
        scene.add_instance(scene.add_mesh(_asset("bunny")), scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8))))
        _room(scene, p, with_light=False)
        cam = make_camera((0.0, 3.5, 6.0), (0.0, -1.0, -3.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 33:   # not a reference scene: scene 25's two triangles with DEGENERATE triangles mixed into the first mesh (a collinear
        # triple and a repeated vertex: cross product exactly zero, math::intersect_triangle's first rejection, ray.rs:49-56).  The product leaves
        # them out of the tree; the frame must be scene 25's bit for bit.
        raw = assets.single_triangle()
        raw = dict(pos=np.concatenate([raw["pos"], np.array([[0.0, 0.0, 0.0]], np.float32)]).astype(np.float32),
                   nrm=np.concatenate([raw["nrm"], raw["nrm"][:1]]).astype(np.float32), uv=np.concatenate([raw["uv"], np.array([[0.5, 0.0]], np.float32)]).astype(np.float32),
                   idx=np.array([[0, 1, 3], [0, 1, 2], [1, 1, 2]], dtype=np.uint32))
        tri = assets.load_obj_semantics(raw)
        tri2 = assets.load_obj_semantics(assets.single_triangle())
        tri2 = dict(tri2); tri2["pos"] = (tri2["pos"] + np.array([0.5, 0.0, -1.0], dtype=np.float32)).astype(np.float32)
        scene.add_instance(scene.add_mesh(tri), scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.5, 0.5))))
        scene.add_instance(scene.add_mesh(tri2), scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.5, 0.8, 0.5))))
        scene.add_delta_light(LIGHT_POINT, 10.0, Spectrum.lut(scene.add_lut470(p["cie_illum_d6500"])), _translate(0.5, 2.0, 2.5))
        cam = make_camera((0.0, 1.5, 6.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id in (24, 25):   # not reference scenes: degenerate BVHs — one triangle (the root is a leaf), two triangles (one leaf) under a point light
        tri = assets.load_obj_semantics(assets.single_triangle())
        if scene_id == 25:
            tri2 = dict(tri); tri2["pos"] = (tri["pos"] + np.array([0.5, 0.0, -1.0], dtype=np.float32)).astype(np.float32)
        scene.add_instance(scene.add_mesh(tri), scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.5, 0.5))))
        if scene_id == 25:
            scene.add_instance(scene.add_mesh(tri2), scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.5, 0.8, 0.5))))
        scene.add_delta_light(LIGHT_POINT, 10.0, Spectrum.lut(scene.add_lut470(p["cie_illum_d6500"])), _translate(0.5, 2.0, 2.5))
        cam = make_camera((0.0, 1.5, 6.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 23:   # not a reference scene: furnace — a convex Lambert ellipsoid (albedo 0.5 at every wavelength) inside a
        # constant environment: every object pixel must show 0.5 x the background radiance
        ell = assets.load_obj_semantics(assets.blob_mesh(64, 57, seed=1, lobes=(1, 0.0, 1.0, 1, 0.0, 1.0), center=(0.0, 0.0, 0.0), scale=1.0, with_uv=False))
        scene.add_instance(scene.add_mesh(ell), scene.add_material(lambert(Spectrum.constant(0.5))))
        scene.add_environment_light(1.0, np.ones((8, 16, 3), dtype=np.float32), scene.add_lut470(p["cie_illum_d6500"]))
        cam = make_camera((0.0, 0.6, 4.0), (0.0, -0.15, -1.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 22:   # not a reference scene: scene 15's material shape — SimplePbr with textured base colour, normal map and
        # FloatTexture metallic / roughness maps (scene_15.rs:18-60), on the dragon-class hero of scene 17
        g = scene.add_mesh(_asset("dragon"))
        albedo, normal = _asset(f"tex{tex_size}")
        grey = lambda a: np.ascontiguousarray(np.repeat(a[..., None], 3, -1))
        t_met = scene.add_tex_rgb8(grey(albedo[..., 0]))               # stand-in Metallic.png / Roughness.png (grey, replicated)
        t_rgh = scene.add_tex_rgb8(grey((albedo[..., 1] // 2 + 40).astype(np.uint8)))
        d = MaterialDesc(); d.type = MAT_SIMPLE_PBR; d.color = Spectrum.texture_albedo_srgb(scene.add_tex_rgb8(albedo))
        d.normal_tex = scene.add_tex_rgb8(normal); d.normal_flip_y = 0; d.metallic = 0.0; d.roughness = 0.0; d.ior = 1.5
        d.metallic_tex = t_met; d.roughness_tex = t_rgh
        scene.add_instance(g, scene.add_material(d), _mat4_trs_scene17())
        _room(scene, p)
        cam = make_camera((0.0, 3.15221, 6.0), (0.0, -0.9, -3.2), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 21:   # not a reference scene: scene 2's room lit by a spot light, a directional light AND the area light
        g = scene.add_mesh(_asset("bunny"))
        scene.add_instance(g, scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8))))
        d65 = Spectrum.lut(scene.add_lut470(p["cie_illum_d6500"]))
        # spot light above the hero looking down (-y): local +z -> world -y is a rotation about x by +90 deg
        rx = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
        scene.add_delta_light(LIGHT_SPOT, 30.0, d65, (_translate(0.5, 3.5, -0.5) @ rx).astype(np.float32), angle_inner=0.9, angle_outer=0.6)
        _room(scene, p)
        scene.add_delta_light(LIGHT_DIRECTIONAL, 0.5, d65, (_rot_y(200.0) @ rx).astype(np.float32))
        cam = make_camera((0.0, 3.5, 6.0), (0.0, -1.0, -3.0), (0.0, 1.0, 0.0), width, height)
    elif scene_id in (6, 7):     # scene_6.rs:16-26: smooth gold hero; scene_7.rs:16-39: four gold heroes, roughness 0.05..0.75
        g = scene.add_mesh(_asset("bunny"))
        lut_eta, lut_k = scene.add_lut470(p["au_eta"]), scene.add_lut470(p["au_k"])

        def gold(roughness):
            d = MaterialDesc(); d.type = MAT_METAL; d.eta = Spectrum.lut(lut_eta); d.k = Spectrum.lut(lut_k)
            d.normal_tex = NONE; d.roughness = roughness; d.color = Spectrum.constant(1.0)
            return d
        if scene_id == 6:
            scene.add_instance(g, scene.add_material(gold(0.0)))
            _room(scene, p)
            cam = make_camera((0.0, 3.5, 6.0), (0.0, -1.0, -3.0), (0.0, 1.0, 0.0), width, height)
        else:
            for pos, rough in zip([(-1.3, 0.0, -0.5), (-0.5, 0.0, -0.5), (0.3, 0.0, -0.5), (1.1, 0.0, -0.5)], [0.05, 0.25, 0.5, 0.75]):
                # Transform::from_scale(0.6).translate(position) = T * S (transform.rs:119-129)
                m = np.diag(np.array([0.6, 0.6, 0.6, 1.0], dtype=np.float32)); m[:3, 3] = np.array(pos, dtype=np.float32)
                scene.add_instance(g, scene.add_material(gold(rough)), m)
            _room(scene, p)
            cam = make_camera((0.0, 2.5, 5.0), (0.0, -0.7, -2.5), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 20:   # not a reference scene: scene 17's hero with SimplePbrMaterial (mixed metallic) — exercises simple_pbr_material.rs
        g = scene.add_mesh(_asset("dragon"))
        d = MaterialDesc(); d.type = MAT_SIMPLE_PBR; d.color = Spectrum.rgb_albedo_srgb(0.8, 0.5, 0.3)
        d.metallic = 0.4; d.roughness = 0.5; d.normal_tex = NONE; d.ior = 1.5
        scene.add_instance(g, scene.add_material(d), _mat4_trs_scene17())
        _room(scene, p)
        cam = make_camera((0.0, 3.15221, 6.0), (0.0, -0.9, -3.2), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 17:   # scene_17.rs:20-70: rough clearcoat over rough metal
        g = scene.add_mesh(_asset("dragon"))
        d = MaterialDesc(); d.type = MAT_CLEARCOAT; d.color = Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8)
        d.metallic = 1.0; d.roughness = 0.7; d.normal_tex = NONE; d.ior = 1.5; d.clearcoat_ior = 1.5
        d.clearcoat_roughness = 0.75; d.clearcoat_tint = Spectrum.rgb_albedo_srgb(0.7, 0.8, 1.0); d.clearcoat_thickness = 0.8
        scene.add_instance(g, scene.add_material(d), _mat4_trs_scene17())
        _room(scene, p)
        cam = make_camera((0.0, 3.15221, 6.0), (0.0, -0.9, -3.2), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 29:   # not a reference scene: TWO environment lights (Scene sums every infinite light, scene.rs:185-231; the MIS weight
        # sums probability x direction pdf over them, mis_renderer.rs:205-214) over a floor, a Lambert and a smooth-gold hero
        room = _asset("room")
        scene.add_instance(scene.add_mesh(room["yuka"]), scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8))))
        g = scene.add_mesh(_asset("bunny"))
        scene.add_instance(g, scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.6, 0.3))), _translate(-0.6, 0.0, 0.0))
        d = MaterialDesc(); d.type = MAT_METAL; d.eta = Spectrum.lut(scene.add_lut470(p["au_eta"])); d.k = Spectrum.lut(scene.add_lut470(p["au_k"]))
        d.normal_tex = NONE; d.roughness = 0.0; d.color = Spectrum.constant(1.0)
        scene.add_instance(g, scene.add_material(d), _translate(0.6, 0.0, 0.0))
        d65 = scene.add_lut470(p["cie_illum_d6500"])
        scene.add_environment_light(1.0, assets.sky_envmap(), d65)
        scene.add_environment_light(0.35, assets.sky_envmap(64, 32, seed=11), d65, _rot_y(140.0))
        cam = make_camera((-1.5, 0.8, 2.5), (1.5, -0.4, -2.5), (0.0, 1.0, 0.0), width, height)
    elif scene_id == 30:   # not a reference scene: a TEXTURED emitter (SpectrumParameter::Texture radiance, emissive_material.rs:48-79: looked up at
        # the hit / sampled uv, its light-pick weight at uv (0.5, 0.5)) — a uv-mapped panel under the ceiling of the Cornell room
        g = scene.add_mesh(_asset("bunny"))
        scene.add_instance(g, scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8))))
        _room(scene, p, with_light=False)
        albedo, _ = _asset(f"tex{min(tex_size, 256)}")
        panel = assets.load_obj_semantics(dict(pos=np.array([[-0.9, 3.95, -0.9], [0.9, 3.95, -0.9], [0.9, 3.95, 0.9], [-0.9, 3.95, 0.9]], np.float32),
                                               nrm=np.array([[0, -1, 0]] * 4, np.float32), uv=np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32),
                                               idx=np.array([[0, 1, 2], [0, 2, 3]], np.uint32)))
        em = MaterialDesc(); em.type = MAT_EMISSIVE; em.color = Spectrum.texture_albedo_srgb(scene.add_tex_rgb8(albedo)); em.intensity = 12.0; em.normal_tex = NONE
        scene.add_instance(scene.add_mesh(panel), scene.add_material(em))
        cam = make_camera((0.0, 3.15221, 6.0), (0.0, -0.9, -3.2), (0.0, 1.0, 0.0), width, height)
    elif scene_id in (31, 32):   # not reference scenes: the textured emitter of scene 30 with an ILLUMINANT-type (31) / UNBOUNDED-type (32) radiance
        # texture (rgb_texture.rs:56-64: RgbIlluminantSpectrum / RgbUnboundedSpectrum of every texel) AND a FloatParameter::texture intensity
        # (emissive_material.rs:55-56: a horizontal ramp, read at the hit / sampled uv; the light-pick weight takes both at uv (0.5, 0.5), :63-79)
        g = scene.add_mesh(_asset("bunny"))
        scene.add_instance(g, scene.add_material(lambert(Spectrum.rgb_albedo_srgb(0.8, 0.8, 0.8))))
        _room(scene, p, with_light=False)
        albedo, _ = _asset(f"tex{min(tex_size, 256)}")
        panel = assets.load_obj_semantics(dict(pos=np.array([[-0.9, 3.95, -0.9], [0.9, 3.95, -0.9], [0.9, 3.95, 0.9], [-0.9, 3.95, 0.9]], np.float32),
                                               nrm=np.array([[0, -1, 0]] * 4, np.float32), uv=np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32),
                                               idx=np.array([[0, 1, 2], [0, 2, 3]], np.uint32)))
        ramp = np.repeat(np.tile(np.linspace(40, 255, 64).astype(np.uint8)[None, :], (16, 1))[..., None], 3, axis=2).copy()   # grey image, red channel read
        t_col, t_int = scene.add_tex_rgb8(albedo), scene.add_tex_rgb8(ramp)
        em = MaterialDesc(); em.type = MAT_EMISSIVE; em.normal_tex = NONE; em.intensity = 1.0; em.intensity_tex = t_int
        if scene_id == 31:
            em.color = Spectrum.texture_illuminant_srgb(t_col, scene.add_lut470(p["cie_illum_d6500"]))
        else:
            em.color = Spectrum.texture_unbounded_srgb(t_col)
        scene.add_instance(scene.add_mesh(panel), scene.add_material(em))
        cam = make_camera((0.0, 3.15221, 6.0), (0.0, -0.9, -3.2), (0.0, 1.0, 0.0), width, height)
    else:
        raise ValueError(f"scene {scene_id} is outside the hot-path scope (SURVEY.md §8)")
    if build:
        scene.build(cam)
    return cam
