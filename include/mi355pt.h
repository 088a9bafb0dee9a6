/*
 * mi355pt.h — C ABI of the MI355X-native spectral path-tracing integrator.
 *
 * The reference (MatchaChoco010/toy-cpu-pathtracing) has no FFI: it is one generic Rust
 * binary.  The seam this library drops in behind is
 *     RendererImage::<R>::render::<S>()          renderer/src/renderer.rs:120-134
 * i.e. the rayon loop that calls BaseSrgbRenderer::render(p) for every pixel
 * (renderer/src/renderer/base_renderer.rs:146-280) against an already built
 * &Scene (scene/src/scene.rs:36-76) and &Camera (renderer/src/camera.rs:14-92).
 * Each entry point below names the reference interface it replaces.
 *
 * Conventions: every function returns 0 on success or a negative MI355PT_E_* code and
 * never unwinds; input buffers are borrowed for the duration of the call and copied;
 * all floats are IEEE binary32, all matrices column-major (glam::Mat4 layout).
 * The library requires a gfx950 device: there is no CPU fallback.
 */
#ifndef MI355PT_H
#define MI355PT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355PT_OK 0
#define MI355PT_E_INVALID (-1)   /* bad argument / bad id / wrong call order */
#define MI355PT_E_DEVICE (-2)    /* HIP error (see mi355pt_last_error)         */
#define MI355PT_E_NOT_BUILT (-3) /* scene used before mi355pt_scene_build      */
#define MI355PT_E_NO_DEVICE (-4) /* no gfx950 device / HIP runtime unavailable */

#define MI355PT_NONE 0xffffffffu

typedef struct mi355pt_scene mi355pt_scene; /* opaque; replaces scene::Scene<Id> (scene/src/scene.rs:36-41) */

/* ---- spectra: spectrum::Spectrum / scene::SpectrumParameter (scene/src/material/parameter.rs:13-47) ---- */
enum {
    MI355PT_SPEC_CONSTANT = 0,            /* ConstantSpectrum::new(c[0])            spectrum/constant_spectrum.rs */
    MI355PT_SPEC_RGB_ALBEDO_SRGB = 1,     /* RgbAlbedoSpectrum<ColorSrgb>::new(c)   spectrum/rgb_albedo_spectrum.rs (needs the table) */
    MI355PT_SPEC_LUT470 = 2,              /* DenselySampledSpectrum, id from add_lut470 (presets::cie_illum_d6500(), glass_sf11_eta(), ...) */
    MI355PT_SPEC_TEXTURE_ALBEDO_SRGB = 3, /* SpectrumParameter::texture(RgbTexture::load_srgb, SpectrumType::Albedo), id from add_tex_rgb8 */
    MI355PT_SPEC_SIGMOID = 4,             /* explicit sigmoid-polynomial coefficients c0,c1,c2 (rgb_sigmoid_polynomial.rs:179-182) */
    MI355PT_SPEC_RGB_ALBEDO_SRGB_LINEAR = 5, /* RgbAlbedoSpectrum<ColorSrgbLinear>::new(c): the same table, no EOTF inversion */
    /* SpectrumParameter::texture(RgbTexture::load_srgb, SpectrumType::Illuminant / Unbounded) (texture/rgb_texture.rs:56-64): per lookup
     * scale = 2 max(rgb), the sigmoid of rgb / scale, times scale — and for Illuminant times presets::cie_illum_d6500(), whose LUT470 id
     * goes in c[0] (as a number).  id = texture from add_tex_rgb8.  Accepted for the radiance of EMISSIVE materials (what the types are for;
     * rgb_illuminant_spectrum.rs:26-46, rgb_unbounded_spectrum.rs:23-42); a black texel gives 0 where the reference divides 0 / 0. */
    MI355PT_SPEC_TEXTURE_ILLUMINANT_SRGB = 6,
    MI355PT_SPEC_TEXTURE_UNBOUNDED_SRGB = 7
};
typedef struct mi355pt_spectrum {
    uint32_t kind;
    uint32_t id;
    float c[3];
} mi355pt_spectrum;

/* ---- materials: scene::{LambertMaterial, EmissiveMaterial, GlassMaterial, PlasticMaterial,
 *      SimpleClearcoatPbrMaterial, MetalMaterial, SimplePbrMaterial}::new  (scene/src/material/impls/) ---- */
enum {
    MI355PT_MAT_LAMBERT = 0,   /* color = albedo, normal_tex                       lambert_material.rs:17-27   */
    MI355PT_MAT_EMISSIVE = 1,  /* color = radiance, intensity                      emissive_material.rs:17-29  */
    MI355PT_MAT_GLASS = 2,     /* eta (LUT470), normal_tex, thin, roughness        glass_material.rs:34-49     */
    MI355PT_MAT_PLASTIC = 3,   /* eta (constant), color, normal_tex, thin, rough.  plastic_material.rs:17-38    */
    MI355PT_MAT_CLEARCOAT = 4, /* simple_pbr_clearcoat_material.rs:17-75                                       */
    MI355PT_MAT_METAL = 5,     /* MetalMaterial::new(MetalType, normal, roughness): eta + k (LUT470 = presets::au_eta()/au_k() ...),
                                * normal_tex, roughness (alpha = roughness^2)           metal_material.rs:17-92      */
    MI355PT_MAT_SIMPLE_PBR = 6 /* SimplePbrMaterial::new(base_color=color, metallic, roughness, normal, ior)
                                * = the clearcoat material's base layer                 simple_pbr_material.rs:14-53 */
};
typedef struct mi355pt_material_desc {
    uint32_t type;
    mi355pt_spectrum color;
    uint32_t normal_tex;    /* MI355PT_NONE = NormalParameter::none() */
    uint32_t normal_flip_y; /* NormalTexture::load(path, flip_y)      */
    float intensity;        /* Emissive FloatParameter::constant      */
    mi355pt_spectrum eta;
    uint32_t thin;
    float roughness;
    /* clearcoat only: SimpleClearcoatPbrMaterial::new(base_color=color, metallic, roughness, normal, ior,
     * clearcoat_ior, clearcoat_roughness, clearcoat_tint, clearcoat_thickness) */
    float metallic, ior, clearcoat_ior, clearcoat_roughness, clearcoat_thickness;
    mi355pt_spectrum clearcoat_tint;
    mi355pt_spectrum k;     /* metal only: extinction coefficient */
    /* FloatParameter::texture(FloatTexture::load(path, false)) for metallic / roughness (SimplePbr, clearcoat) and the metal's
     * roughness: texture ids from add_tex_rgb8 (a grey image replicated to RGB; the red channel is read with the same bilinear
     * rule, texture/sampler.rs:81-107), MI355PT_NONE = the constant above.  material/parameter.rs:58-83 */
    uint32_t metallic_tex, roughness_tex;   /* roughness_tex also on glass / plastic (glass_material.rs:42,116, plastic_material.rs:43,104) */
    uint32_t clearcoat_thickness_tex;   /* clearcoat only (scene_18.rs:37-42) */
    uint32_t intensity_tex;             /* emissive only: FloatParameter::texture intensity (emissive_material.rs:55-56: sampled at the hit /
                                         * sampled point; the light-pick weight takes it at uv (0.5, 0.5), :69-76); MI355PT_NONE = `intensity` */
} mi355pt_material_desc;

/* ---- delta lights: CreatePrimitiveDesc::{PointLightPrimitive, SpotLightPrimitive, DirectionalLightPrimitive}
 *      (scene/src/primitive/create_desc.rs:35-66; primitive/impls/{point,spot,directional}_light.rs).  The spot light
 *      looks down its local +z, the directional light shines along its local +z. ---- */
enum { MI355PT_LIGHT_POINT = 1, MI355PT_LIGHT_SPOT = 2, MI355PT_LIGHT_DIRECTIONAL = 3 };
typedef struct mi355pt_light_desc {
    uint32_t kind;
    float intensity;
    float angle_inner, angle_outer; /* spot only; compared against the cosine exactly as spot_light.rs:110-112 does */
    mi355pt_spectrum spectrum;      /* constant or LUT470 */
    float local_to_world[16];       /* column-major */
} mi355pt_light_desc;

/* ---- camera: renderer::Camera::new + set_look_to (renderer/src/camera.rs:27-49) ---- */
typedef struct mi355pt_camera {
    float position[3];  /* world space */
    float direction[3]; /* normalised by the library, as set_look_to does */
    float up[3];
    float fov_deg; /* vertical; main.rs:68 fixes 45 */
    uint32_t width, height;
} mi355pt_camera;

/* ---- renderer arguments: RendererArgs + CLI flags (renderer.rs:84-90, main.rs:20-53) ---- */
enum { MI355PT_STRATEGY_PT = 0, MI355PT_STRATEGY_NEE = 1, MI355PT_STRATEGY_MIS = 2 };
enum { MI355PT_SAMPLER_RANDOM = 0, MI355PT_SAMPLER_SOBOL = 1 };
typedef struct mi355pt_params {
    uint32_t spp;       /* --spp        */
    uint32_t seed;      /* --seed       */
    uint32_t max_depth; /* --max-depth  */
    uint32_t strategy;  /* --renderer pt|nee|mis */
    uint32_t sampler;   /* --sampler random|sobol */
    float exposure;     /* main.rs:191 fixes 1.0; tone map is Reinhard (main.rs:192) */
    /* multi-GPU sharding of the pixel loop (renderer.rs:121): this call renders only the 8x8
     * pixel tiles t with t % shard_count == shard_index; other pixels are left untouched. */
    uint32_t shard_index, shard_count; /* 0,1 (or 0,0) = whole frame */
    uint32_t collect_stats;            /* run the instrumented kernel variant and fill mi355pt_stats: 1 = with the reference's
                                        * traversal order (the canonical per-sample node/triangle counts of SURVEY.md 8d),
                                        * 2 = with the production traversal (cooperative, slightly more nodes; lane-use diagnostics; scenes without clearcoat / textured
                                        * emitters run the instrumented kernel whose traversal has the production form of their kernels) */
    float rr_gate_slack;               /* MUST BE 0 (the reference) — a non-zero value is REFUSED with MI355PT_E_INVALID unless the process called
                                        * mi355pt_debug_unlock(1) (mi355pt_debug.h), so that a caller's uninitialised struct can never change a
                                        * picture.  Diagnostic: apply_russian_roulette skips the roulette when
                                        * max(T) >= 1 (base_renderer.rs:76-92); with a slack s the gate is max(T) >= 1 - s.
                                        * Used by the parity tests to show that GPU / oracle path flips on solid constant-eta
                                        * dielectrics come from that gate sitting on T = F * (1 / pdf) = 1 +- 1 ulp */
    uint32_t albedo_lut;               /* SimpleClearcoatPbrMaterial's coat weight (simple_pbr_clearcoat_material.rs:190-192): 0 = the
                                        * reference's 64-sample Monte-Carlo directional albedo per shading vertex
                                        * (generalized_schlick.rs:893-918, counter RNG); 1 = its expectation E(cos theta) from a 64-entry
                                        * table per material (mi355pt_coat_albedo_table), deterministic and faster: an OPTION that deviates
                                        * from the reference's estimator by less than that estimator's own noise */
} mi355pt_params;

typedef struct mi355pt_stats {
    uint64_t samples, closest_rays, shadow_rays, nodes_closest, tris_closest, nodes_shadow, tris_shadow;
    uint64_t closest_hits, bounces, spectrum_evals, textured_lookups;
    /* diagnostic (collect_stats only): wave-cycles spent per phase of the wave state machine, summed over waves:
     * 0 regenerate (Sobol + camera ray), 1 closest-hit traversal, 2 shading + light sampling, 3 shadow traversal,
     * 4 film/sensor, 5 whole loop; shading split: 6 surface+emission+RR, 7 BSDF sample, 8 light sample (NEE), 9 NOT cycles: low 32 bits = closest-hit merges that met an exact tie in t between two triangles (the merge keeps the lower index), high 32 bits = those between triangles of different material or normal */
    uint64_t phase_cycles[10];
    double kernel_ms; /* device time of the path-tracing launch(es), HIP events on the launch stream */
    uint32_t launches;
    /* diagnostic (collect_stats only): wave-level step counts, to compare with the per-lane counts above
     * (lane utilisation of a stage = lane count / (64 * wave count)): 0 closest-hit node steps, 1 closest-hit triangle
     * steps, 2 any-hit node steps, 3 any-hit triangle steps, 4 iterations of the wave state machine, 5 lanes shaded
     * (summed over iterations), 6 lanes with a shadow ray, 7 steal rounds of the merged traversal */
    uint64_t wave_steps[8];
    /* diagnostic (collect_stats = 2): how many lanes of the wave were still walking when a node step was issued — wave-level node
     * steps of the cooperative traversals by busy-lane count, bucket k = 8k+1 .. 8k+8 lanes; [0..7] closest-hit, [8..15] any-hit (where one traversal walks both kinds of ray, its steps are in [0..7]).  The
     * low buckets are the tail a lock-step traversal pays for its deepest ray */
    uint64_t busy_hist[16];
    /* diagnostic (collect_stats = 2): material divergence of the shading stage, summed over wave iterations — 0 iterations with
     * at least one lane shading a surface, 1 distinct material classes among those lanes (MI355PT_MAT_* types; the ceiling of what a
     * material-key sort between bounces could remove is 1 - [0]/[1] of the shading stage), 2 lanes shading a surface, 3 lanes in
     * the largest class; 4..7 iterations in which the lanes that continue past emission (every class but EMISSIVE) hold 0, 1, 2, >= 3
     * distinct classes, 8..11 wave-cycles of the shading stage in those iterations (what a second or third class in the wave costs) */
    uint64_t divergence[12];
} mi355pt_stats;

/* ---------------- scene construction ---------------- */
int mi355pt_scene_create(mi355pt_scene** out);             /* scene::create_scene!()                     scene.rs:266-285 */
void mi355pt_scene_destroy(mi355pt_scene* s);
/* rgb_to_spec::SRGB_DATA: [64 z_nodes][3][64][64][64][3] f32   rgb_to_spec/src/lib.rs:1-4, rgb_sigmoid_polynomial.rs:35-84 */
int mi355pt_scene_set_rgb2spec(mi355pt_scene* s, const float* table, size_t n_floats);
/* a preset spectrum baked to 470 1-nm entries        spectrum/src/presets.rs:75-231, densely_sampled_spectrum.rs:40-53 */
int mi355pt_scene_add_lut470(mi355pt_scene* s, const float values[470], uint32_t* out_id);
/* RgbTexture::load_srgb / NormalTexture::load (8-bit RGB, row-major, top row first)  texture/loader.rs:44-64 */
int mi355pt_scene_add_tex_rgb8(mi355pt_scene* s, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t* out_id);
/* Scene::load_obj -> TriangleMesh{positions,normals,uvs,tangents,indices}  geometry/impls/triangle_mesh.rs:130-242.
 * uv may be NULL (no texcoords); tri_tangent (one vec3 per triangle) may be NULL only if uv is NULL. */
int mi355pt_scene_add_mesh(mi355pt_scene* s, const float* pos, const float* nrm, const float* uv, const float* tri_tangent,
                           const uint32_t* idx, uint32_t n_vert, uint32_t n_tri, uint32_t* out_geom);
int mi355pt_scene_add_material(mi355pt_scene* s, const mi355pt_material_desc* desc, uint32_t* out_mat);
/* Scene::create_primitive(CreatePrimitiveDesc::GeometryPrimitive{geometry_index, surface_material, transform})
 * scene.rs:57-61, primitive/create_desc.rs:10-15.  Emissive materials make the instance an area light. */
int mi355pt_scene_add_instance(mi355pt_scene* s, uint32_t geom, uint32_t mat, const float local_to_world[16]);
/* Scene::create_primitive(CreatePrimitiveDesc::{Point,Spot,Directional}LightPrimitive{..})  scene.rs:57-61.  Lights and
 * emissive instances enter the light sampler in creation order (light_sampler.rs:163-180). */
int mi355pt_scene_add_delta_light(mi355pt_scene* s, const mi355pt_light_desc* desc);
/* CreatePrimitiveDesc::EnvironmentLightPrimitive{intensity, texture, transform} (create_desc.rs:67-76;
 * primitive/impls/environment_light.rs): lat-long float RGB map (row 0 = +y pole, the reference reads an EXR through
 * image::open(..).to_rgb32f()), importance-sampled by luminance * sin(theta).  `illuminant_lut` is the LUT470 id of
 * presets::cie_illum_d6500() (rgb_illuminant_spectrum.rs:28).  Only the rotation of the transform matters. */
int mi355pt_scene_add_environment_light(mi355pt_scene* s, float intensity, const float* rgb, uint32_t width, uint32_t height,
                                        const float local_to_world[16], uint32_t illuminant_lut);
/* The table behind mi355pt_params.albedo_lut (host-only, deterministic; no reference counterpart): out[k], k < 64, is the expectation
 * of GeneralizedSchlickBsdf::directional_albedo's estimator f |cos i| / pdf (generalized_schlick.rs:893-918; mode R, scalar r0, r90 = 1)
 * at cos(theta_o) = (k + 0.5) / 64, integrated over a 256 x 256 midpoint grid of the (u, v) square in double precision.  The kernel
 * (and the oracle in its LUT mode) interpolate it linearly in |cos(theta_o)|.  mi355pt_scene_build computes it for every clearcoat
 * material. */
int mi355pt_coat_albedo_table(float alpha, float r0, float out[64]);
/* Which builder mi355pt_scene_build uses for the BVH (stands where Bvh::build is, scene/src/bvh.rs:92-230; no reference
 * counterpart for the choice).  AUTO: host sweep SAH below 131 072 triangles, the GPU binned-SAH builder from there on;
 * HOST / GPU force one (GPU fails with MI355PT_E_DEVICE rather than substituting the host builder).  The environment
 * variable MI355PT_BVH_BUILDER=auto|host|gpu overrides the setting.  Closest hits do not depend on the builder
 * (up to exact ties between triangles). */
enum { MI355PT_BVH_AUTO = 0, MI355PT_BVH_HOST = 1, MI355PT_BVH_GPU = 2 };
int mi355pt_scene_set_bvh_builder(mi355pt_scene* s, int mode);
/* Scene::build(&camera): world->render translation, BVH build, light list; uploads to the current HIP device.
 * scene.rs:64-76.  The camera POSITION is baked into the device records (render space = world - position,
 * camera.rs:84-86): every later render call must pass the same position (direction, up, fov and size are free) and run
 * with the same HIP device current, else it returns MI355PT_E_INVALID / MI355PT_E_DEVICE; move the camera = build again.
 * Threading contract: one host thread and one stream at a time per scene (the scene owns its launch scratch). */
int mi355pt_scene_build(mi355pt_scene* s, const mi355pt_camera* cam);

/* Diagnostic: "nodes=.. tris=.. depth=.. builder=host|gpu bvh_ms=.. bvh_device_ms=.. features=.." of the built scene
 * (no reference counterpart). */
int mi355pt_scene_info(const mi355pt_scene* s, char* buf, size_t buf_size);

/* ---------------- rendering ---------------- */
/* RendererImage::render::<S>() + Sensor::to_rgb: fills out_rgb (host, W*H*3, row-major, y down) with
 * tone-mapped sRGB-encoded values in [0,1] exactly like RendererImage.pixels.  renderer.rs:101-134, sensor.rs:81-88 */
int mi355pt_render(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p, float* out_rgb,
                   mi355pt_stats* stats /* NULL ok */);
/* The same pixel loop with inputs and outputs resident in HBM: adds the *linear* per-pixel RGB sums of
 * sample indices [sample_begin, sample_end) into d_accum (device, W*H*3 f32; Sensor.accumulated_rgb,
 * sensor.rs:12-20,77).  Asynchronous on `hip_stream` (a hipStream_t, NULL = default stream).  This is
 * what the multi-GPU path reduces with RCCL before resolving. */
int mi355pt_render_accum_device(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p,
                                uint32_t sample_begin, uint32_t sample_end, float* d_accum, void* hip_stream,
                                mi355pt_stats* stats /* NULL ok; non-NULL synchronises the stream */);
/* The seam as main.rs:228 calls it — ONE call from ONE process — over several GPUs of the node (north star: "pixel tiles shard
 * across the 8 GPUs of one node").  mi355pt_scene_build_multi replaces mi355pt_scene_build: the scene (< 30 MB) is replicated on
 * every listed device (ids may repeat, which rehearses the path on fewer GPUs).  mi355pt_render_multi = mi355pt_render: the frame's
 * 8x8 tiles are dealt round-robin to the devices (shard i of n, concurrent launches on per-device streams), the devices' linear
 * films (Sensor.accumulated_rgb, sensor.rs:12-20) are gathered onto the first device by xGMI peer copies and added — disjoint
 * tiles, so the sum is exact — then resolved there (Sensor::to_rgb) and copied to out_rgb.  The caller's current device is
 * restored.  One process per GPU instead: mi355pt_render_accum_device with shard_index = rank + one RCCL reduce. */
int mi355pt_scene_build_multi(mi355pt_scene* s, const mi355pt_camera* cam, int n_devices, const int* device_ids);
int mi355pt_render_multi(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p, float* out_rgb);
/* Sensor::to_rgb on device buffers: mean over spp, clip, Reinhard, sRGB OETF.  sensor.rs:81-88, tone_map.rs:20-28 */
int mi355pt_film_resolve_device(const float* d_accum, uint32_t n_pixels, uint32_t spp, float* d_out_rgb, void* hip_stream);
/* RendererImage::save quantisation `(p*255.0) as u8`  renderer.rs:137-148 (host helper) */
int mi355pt_quantize_u8(const float* rgb, size_t n, uint8_t* out);

/* (The probes the parity tests read — Sobol words, single rays, the per-sample log of a render launch, the BVH export and the collapse
 * check — and the switch that lets `rr_gate_slack` through live in mi355pt_debug.h: test and diagnosis surface of the same library, not
 * part of what a caller of renderer::render() needs.) */
const char* mi355pt_last_error(void); /* thread-local message of the last failing call */
const char* mi355pt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355PT_H */
