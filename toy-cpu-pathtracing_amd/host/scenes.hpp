// load_scene_N(&mut scene, &mut camera) of renderer/src/scene/scene_{3,8,10,17}.rs, call for call.
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
inline void load_room(Scene& scene) {
    struct Wall { const char* obj; ColorSrgb c; };
    const Wall walls[] = {{"box.obj", {0.8f, 0.8f, 0.8f}}, {"hidari.obj", {0.9f, 0.0f, 0.0f}}, {"migi.obj", {0.0f, 0.9f, 0.0f}},
                          {"yuka.obj", {0.8f, 0.8f, 0.8f}}, {"oku.obj", {0.8f, 0.8f, 0.8f}}, {"tenjou.obj", {0.8f, 0.8f, 0.8f}}};
    for (const Wall& w : walls) {
        GeometryIndex geom = scene.load_obj(asset(w.obj));
        Spectrum spectrum = RgbAlbedoSpectrum::create(w.c);
        scene.create_primitive(GeometryPrimitive{geom, LambertMaterial::create(SpectrumParameter::constant(spectrum), NormalParameter::none()),
                                                 Transform::identity()});
    }
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
    SpectrumParameter base_color = SpectrumParameter::constant(RgbAlbedoSpectrum::create({0.8f, 0.8f, 0.8f}));
    SpectrumParameter tint = SpectrumParameter::constant(RgbAlbedoSpectrum::create({0.7f, 0.8f, 1.0f}));
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
