// load_scene_N(&mut scene, &mut camera) of renderer/src/scene/scene_{0..19}.rs, call for call.
// Asset paths keep the reference's names under $MI355PT_ASSETS (default ./assets); the files are the synthetic
// stand-ins written by tools/export_assets.py (the reference's are git-LFS stubs), textures as binary PPM.
#pragma once
#include "renderer.hpp"

namespace renderer {

inline std::string asset(const std::string& rel) {
    const char* e = std::getenv("MI355PT_ASSETS");
    return std::string(e ? e : "./assets") + "/" + rel;
}

// box / hidari / migi / yuka / oku / tenjou / light — identical in scenes 3, 8, 10, 17 (scene_3.rs:33-107)
inline void load_room(Scene& scene, bool with_light = true) {
    struct Wall { const char* obj; ColorSrgb c; };
    const Wall walls[] = {{"box.obj", {0.8f, 0.8f, 0.8f}}, {"hidari.obj", {0.9f, 0.0f, 0.0f}}, {"migi.obj", {0.0f, 0.9f, 0.0f}},
                          {"yuka.obj", {0.8f, 0.8f, 0.8f}}, {"oku.obj", {0.8f, 0.8f, 0.8f}}, {"tenjou.obj", {0.8f, 0.8f, 0.8f}}};
    for (const Wall& w : walls) {
        GeometryIndex geom = scene.load_obj(asset(w.obj));
        Spectrum spectrum = RgbAlbedoSpectrum::create(w.c);
        scene.create_primitive(GeometryPrimitive{geom, LambertMaterial::create(SpectrumParameter::constant(spectrum), NormalParameter::none()),
                                                 Transform::identity()});
    }
    if (!with_light) return;
    GeometryIndex geom = scene.load_obj(asset("light.obj"));
    scene.create_primitive(GeometryPrimitive{geom, EmissiveMaterial::create(SpectrumParameter::constant(presets::cie_illum_d6500()), FloatParameter::constant(10.0f)),
                                             Transform::identity()});
}

inline void load_scene_3(Scene& scene, Camera& camera) {            // scene_3.rs:13-115
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    RgbTexture texture = RgbTexture::load_srgb(asset("bunny-material-0/BaseColor.ppm"));
    SpectrumParameter spectrum_param = SpectrumParameter::texture(texture, SpectrumType::Albedo);
    NormalTexture normal_texture = NormalTexture::load(asset("bunny-material-0/Normal.ppm"), false);
    NormalParameter normal_param = NormalParameter::texture(normal_texture);
    scene.create_primitive(GeometryPrimitive{geom, LambertMaterial::create(spectrum_param, normal_param), Transform::identity()});
    load_room(scene);
    camera.set_look_to({0.0f, 3.15221f, 6.0f}, {0.0f, -0.9f, -3.2f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_0(Scene& scene, Camera& camera) {            // scene_0.rs: constant grey Lambert hero
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    scene.create_primitive(GeometryPrimitive{geom, LambertMaterial::create(SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.8f, 0.8f, 0.8f})), NormalParameter::none()),
                                             Transform::identity()});
    load_room(scene);
    camera.set_look_to({0.0f, 3.15221f, 6.0f}, {0.0f, -0.9f, -3.2f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_1(Scene& scene, Camera& camera) {            // scene_1.rs:12-87: point lights only
    const float rad = 3.14159265358979323846f / 180.0f;
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    scene.create_primitive(GeometryPrimitive{geom, LambertMaterial::create(SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.5f, 0.5f, 0.8f})), NormalParameter::none()),
                                             Transform::identity()});
    scene.create_primitive(GeometryPrimitive{geom, LambertMaterial::create(SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.5f, 0.8f, 0.5f})), NormalParameter::none()),
                                             Transform::from_rotate_y(30.0f * rad).translate({-1.0f, 1.0f, 3.0f})});
    geom = scene.load_obj(asset("yuka.obj"));
    scene.create_primitive(GeometryPrimitive{geom, LambertMaterial::create(SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.8f, 0.8f, 0.8f})), NormalParameter::none()),
                                             Transform::identity()});
    scene.create_primitive(SingleTrianglePrimitive{{{-2.0f, 0.0f, 0.0f}, {2.0f, 0.0f, 0.0f}, {-2.0f, 4.0f, 0.0f}},
                                                   {{0, 0, 1}, {0, 0, 1}, {0, 0, 1}},
                                                   {{0.0f, 0.0f}, {1.0f, 0.0f}, {0.0f, 1.0f}},
                                                   LambertMaterial::create(SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.8f, 0.5f, 0.5f})), NormalParameter::none()),
                                                   Transform::from_rotate_y(60.0f * rad)});
    scene.create_primitive(PointLightPrimitive{10.0f, presets::cie_illum_d6500(), Transform::from_translate({0.0f, 3.0f, 0.0f})});
    scene.create_primitive(PointLightPrimitive{10.0f, presets::cie_illum_d6500(), Transform::from_translate({3.0f, 5.0f, 0.0f})});
    camera.set_look_to({0.0f, 3.5f, 7.0f}, {0.0f, -1.0f, -3.0f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_2(Scene& scene, Camera& camera) {            // scene_2.rs:12-101: Cornell room lit by one point light
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    scene.create_primitive(GeometryPrimitive{geom, LambertMaterial::create(SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.8f, 0.8f, 0.8f})), NormalParameter::none()),
                                             Transform::identity()});
    load_room(scene, false);
    scene.create_primitive(PointLightPrimitive{10.0f, presets::cie_illum_d6500(), Transform::from_translate({0.0f, 3.0f, 0.0f})});
    camera.set_look_to({0.0f, 3.5f, 6.0f}, {0.0f, -1.0f, -3.0f}, {0.0f, 1.0f, 0.0f});
}
// ---- the remaining Cornell scenes (scene_{4,5,9,12,13,14,15,16,18}.rs); texture files are the stand-ins' .ppm ----
inline Material textured_lambert(const std::string& dir, bool with_albedo) {
    NormalParameter normal_param = NormalParameter::texture(NormalTexture::load(asset(dir + "/Normal.ppm"), false));
    SpectrumParameter col = with_albedo ? SpectrumParameter::texture(RgbTexture::load_srgb(asset(dir + "/BaseColor.ppm")), SpectrumType::Albedo)
                                        : SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.8f, 0.8f, 0.8f}));
    return LambertMaterial::create(col, normal_param);
}
inline void load_scene_4(Scene& scene, Camera& camera) {            // scene_4.rs: scene 3 with bunny-material-1
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    scene.create_primitive(GeometryPrimitive{geom, textured_lambert("bunny-material-1", true), Transform::identity()});
    load_room(scene);
    camera.set_look_to({0.0f, 3.5f, 6.0f}, {0.0f, -1.0f, -3.0f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_5(Scene& scene, Camera& camera) {            // scene_5.rs: grey Lambert + normal map, close-up
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    scene.create_primitive(GeometryPrimitive{geom, textured_lambert("bunny-material-0", false), Transform::identity()});
    load_room(scene);
    camera.set_look_to({0.3f, 1.6f, 2.8f}, {0.0f, -0.5f, -2.0f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_9(Scene& scene, Camera& camera) {            // scene_9.rs: plastic eta 1.8 without the thin film
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    scene.create_primitive(GeometryPrimitive{geom, PlasticMaterial::create(1.8f, SpectrumParameter::constant(ConstantSpectrum::create(1.0f)), NormalParameter::none(), false,
                                                                           FloatParameter::constant(0.0f)), Transform::identity()});
    load_room(scene);
    camera.set_look_to({0.0f, 3.5f, 6.0f}, {0.0f, -1.0f, -3.0f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_13(Scene& scene, Camera& camera) {           // scene_13.rs: blue plastic (linear-sRGB colour), eta 1.5
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    scene.create_primitive(GeometryPrimitive{geom, PlasticMaterial::create(1.5f, SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgbLinear{0.4f, 0.9f, 1.0f})),
                                                                           NormalParameter::none(), false, FloatParameter::constant(0.0f)), Transform::identity()});
    load_room(scene);
    camera.set_look_to({0.0f, 3.5f, 6.0f}, {0.0f, -1.0f, -3.0f}, {0.0f, 1.0f, 0.0f});
}
inline void load_four_heroes(Scene& scene, Camera& camera, const Material mats[4]) {   // layout of scene_7/12/14.rs
    GeometryIndex bunny_geom = scene.load_obj(asset("bunny.obj"));
    const float scale = 0.6f;
    const Vec3 positions[4] = {{-1.3f, 0.0f, -0.5f}, {-0.5f, 0.0f, -0.5f}, {0.3f, 0.0f, -0.5f}, {1.1f, 0.0f, -0.5f}};
    for (int i = 0; i < 4; ++i)
        scene.create_primitive(GeometryPrimitive{bunny_geom, mats[i], Transform::from_scale({scale, scale, scale}).translate(positions[i])});
    load_room(scene);
    camera.set_look_to({0.0f, 3.5f, 6.0f}, {0.0f, -1.0f, -3.0f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_12(Scene& scene, Camera& camera) {           // scene_12.rs: four BK7 glass heroes, roughness 0.05 .. 0.75
    const float r[4] = {0.05f, 0.25f, 0.5f, 0.75f};
    Material m[4];
    for (int i = 0; i < 4; ++i) m[i] = GlassMaterial::create(GlassType::Bk7, NormalParameter::none(), false, FloatParameter::constant(r[i]));
    load_four_heroes(scene, camera, m);
}
inline void load_scene_14(Scene& scene, Camera& camera) {           // scene_14.rs: four coloured plastic heroes
    const float r[4] = {0.05f, 0.1f, 0.3f, 0.5f};
    const ColorSrgb c[4] = {{1.0f, 0.5f, 0.5f}, {0.5f, 1.0f, 0.5f}, {0.5f, 0.5f, 1.0f}, {1.0f, 0.8f, 0.4f}};
    Material m[4];
    for (int i = 0; i < 4; ++i)
        m[i] = PlasticMaterial::create(1.5f, SpectrumParameter::constant(RgbAlbedoSpectrum::create(c[i])), NormalParameter::none(), false, FloatParameter::constant(r[i]));
    load_four_heroes(scene, camera, m);
}
inline Transform dragon_transform() {                                // scene_15/16/17/18.rs
    return Transform::identity().rotate_y(120.0f * (3.14159265358979323846f / 180.0f)).scale({2.5f, 2.5f, 2.5f}).translate({0.0f, 0.0f, 0.5f});
}
inline void load_scene_15(Scene& scene, Camera& camera) {           // scene_15.rs: SimplePbr with BaseColor / Metallic / Roughness / Normal maps
    GeometryIndex geom = scene.load_obj(asset("dragon.min.obj"));
    const std::string d = "dragon-material/";
    scene.create_primitive(GeometryPrimitive{
        geom,
        SimplePbrMaterial::create(SpectrumParameter::texture(RgbTexture::load_srgb(asset(d + "BaseColor.ppm")), SpectrumType::Albedo),
                                  FloatParameter::texture(FloatTexture::load(asset(d + "Metallic.ppm"), false)),
                                  FloatParameter::texture(FloatTexture::load(asset(d + "Roughness.ppm"), false)),
                                  NormalParameter::texture(NormalTexture::load(asset(d + "Normal.ppm"), false)), FloatParameter::constant(1.5f)),
        dragon_transform()});
    load_room(scene);
    camera.set_look_to({0.0f, 3.15221f, 6.0f}, {0.0f, -0.9f, -3.2f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_16_18(Scene& scene, Camera& camera, bool thickness_map) {   // scene_16.rs / scene_18.rs: coat roughness 0.01
    GeometryIndex geom = scene.load_obj(asset("dragon.min.obj"));
    FloatParameter thickness = thickness_map ? FloatParameter::texture(FloatTexture::load(asset("dragon-material/ClearcoatThickness.ppm"), false))
                                             : FloatParameter::constant(0.8f);
    if (thickness_map) thickness.v = 0.8f;
    scene.create_primitive(GeometryPrimitive{
        geom,
        SimpleClearcoatPbrMaterial::create(SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.8f, 0.8f, 0.8f})), FloatParameter::constant(1.0f),
                                           FloatParameter::constant(0.7f), NormalParameter::none(), FloatParameter::constant(1.5f), FloatParameter::constant(1.5f),
                                           FloatParameter::constant(0.01f), SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.7f, 0.8f, 1.0f})), thickness),
        dragon_transform()});
    load_room(scene);
    camera.set_look_to({0.0f, 3.15221f, 6.0f}, {0.0f, -0.9f, -3.2f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_19(Scene& scene, Camera& camera) {           // scene_19.rs:17-153: three heroes under an environment light
    // stand-ins: constant metallic / roughness instead of the FloatTexture maps, synthetic sky (PFM unless the real EXR is present)
    GeometryIndex floor_geom = scene.load_obj(asset("yuka.obj"));
    scene.create_primitive(GeometryPrimitive{floor_geom, LambertMaterial::create(SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.8f, 0.8f, 0.8f})), NormalParameter::none()),
                                             Transform::identity()});
    GeometryIndex dragon_geom = scene.load_obj(asset("dragon.min.obj"));
    scene.create_primitive(GeometryPrimitive{dragon_geom,
                                             SimplePbrMaterial::create(SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.8f, 0.6f, 0.3f})), FloatParameter::constant(0.5f),
                                                                       FloatParameter::constant(0.4f), NormalParameter::none(), FloatParameter::constant(1.5f)),
                                             Transform::identity()});
    scene.create_primitive(GeometryPrimitive{
        dragon_geom,
        SimpleClearcoatPbrMaterial::create(SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.8f, 0.8f, 0.8f})), FloatParameter::constant(1.0f), FloatParameter::constant(0.7f),
                                           NormalParameter::none(), FloatParameter::constant(1.5f), FloatParameter::constant(1.5f), FloatParameter::constant(0.01f),
                                           SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.7f, 0.8f, 1.0f})), FloatParameter::constant(0.8f)),
        Transform::identity().translate({0.5f, 0.0f, 0.5f})});
    scene.create_primitive(GeometryPrimitive{dragon_geom,
                                             PlasticMaterial::create(1.5f, SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgbLinear{0.4f, 0.9f, 1.0f})), NormalParameter::none(), false,
                                                                     FloatParameter::constant(0.0f)),
                                             Transform::identity().translate({-0.5f, 0.0f, -0.5f})});
    {   // scene_19.rs reads sky/scythian_tombs_2_1k.exr (an LFS object); the synthetic stand-in is exported as .pfm
        std::string sky = asset("sky/scythian_tombs_2_1k.exr");
        if (!std::ifstream(sky)) sky = asset("sky/scythian_tombs_2_1k.pfm");
        scene.create_primitive(EnvironmentLightPrimitive{1.0f, sky, Transform::identity()});
    }
    camera.set_look_to({-1.5f, 0.8f, 2.5f}, {1.5f, -0.4f, -2.5f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_6(Scene& scene, Camera& camera) {            // scene_6.rs:13-110: smooth gold hero
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    scene.create_primitive(GeometryPrimitive{geom, MetalMaterial::create(MetalType::Gold, NormalParameter::none(), FloatParameter::constant(0.0f)), Transform::identity()});
    load_room(scene);
    camera.set_look_to({0.0f, 3.5f, 6.0f}, {0.0f, -1.0f, -3.0f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_7(Scene& scene, Camera& camera) {            // scene_7.rs:13-130: four gold heroes, roughness 0.05 .. 0.75
    GeometryIndex bunny_geom = scene.load_obj(asset("bunny.obj"));
    const float scale = 0.6f;
    const Vec3 positions[4] = {{-1.3f, 0.0f, -0.5f}, {-0.5f, 0.0f, -0.5f}, {0.3f, 0.0f, -0.5f}, {1.1f, 0.0f, -0.5f}};
    const float roughness_values[4] = {0.05f, 0.25f, 0.5f, 0.75f};
    for (int i = 0; i < 4; ++i)
        scene.create_primitive(GeometryPrimitive{bunny_geom, MetalMaterial::create(MetalType::Gold, NormalParameter::none(), FloatParameter::constant(roughness_values[i])),
                                                 Transform::from_scale({scale, scale, scale}).translate(positions[i])});
    load_room(scene);
    camera.set_look_to({0.0f, 2.5f, 5.0f}, {0.0f, -0.7f, -2.5f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_11(Scene& scene, Camera& camera) {           // scene_11.rs: SF11 glass hero with roughness 0.2
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    scene.create_primitive(GeometryPrimitive{geom, GlassMaterial::create(GlassType::Sf11, NormalParameter::none(), false, FloatParameter::constant(0.2f)),
                                             Transform::identity()});
    load_room(scene);
    camera.set_look_to({0.0f, 3.5f, 6.0f}, {0.0f, -1.0f, -3.0f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_8(Scene& scene, Camera& camera) {            // scene_8.rs:13-111
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    scene.create_primitive(GeometryPrimitive{geom, GlassMaterial::create(GlassType::Sf11, NormalParameter::none(), false, FloatParameter::constant(0.0f)),
                                             Transform::identity()});
    load_room(scene);
    camera.set_look_to({0.0f, 3.5f, 6.0f}, {0.0f, -1.0f, -3.0f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_10(Scene& scene, Camera& camera) {           // scene_10.rs:13-112
    GeometryIndex geom = scene.load_obj(asset("bunny.obj"));
    scene.create_primitive(GeometryPrimitive{geom, PlasticMaterial::create(1.8f, SpectrumParameter::constant(ConstantSpectrum::create(1.0f)), NormalParameter::none(),
                                                                           true, FloatParameter::constant(0.0f)),
                                             Transform::identity()});
    load_room(scene);
    camera.set_look_to({0.0f, 3.5f, 6.0f}, {0.0f, -1.0f, -3.0f}, {0.0f, 1.0f, 0.0f});
}
inline void load_scene_17(Scene& scene, Camera& camera) {           // scene_17.rs:13-155
    GeometryIndex geom = scene.load_obj(asset("dragon.min.obj"));
    SpectrumParameter base_color = SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.8f, 0.8f, 0.8f}));
    SpectrumParameter tint = SpectrumParameter::constant(RgbAlbedoSpectrum::create(ColorSrgb{0.7f, 0.8f, 1.0f}));
    Transform t = Transform::identity().rotate_y(120.0f * (3.14159265358979323846f / 180.0f)).scale({2.5f, 2.5f, 2.5f}).translate({0.0f, 0.0f, 0.5f});
    scene.create_primitive(GeometryPrimitive{
        geom,
        SimpleClearcoatPbrMaterial::create(base_color, FloatParameter::constant(1.0f), FloatParameter::constant(0.7f), NormalParameter::none(),
                                           FloatParameter::constant(1.5f), FloatParameter::constant(1.5f), FloatParameter::constant(0.75f), tint,
                                           FloatParameter::constant(0.8f)),
        t});
    load_room(scene);
    camera.set_look_to({0.0f, 3.15221f, 6.0f}, {0.0f, -0.9f, -3.2f}, {0.0f, 1.0f, 0.0f});
}

}  // namespace renderer
