// Data layout shared by the host lowering (scene.cpp) and the gfx950 kernels (pt_kernels.hip).
//
// Everything the sample loop touches lives in HBM as flat, 16-byte-aligned records sized for
// whole dwordx4 gathers (lanes of a wave traverse independently, so node/triangle fetches are
// per-lane gathers; a record that is one or a few aligned 16 B pieces costs the fewest TA cycles):
//
//   DevNode      64 B  BVH2 node holding BOTH child boxes + child links      (SURVEY §8d "node 64 B")
//   DevTri       48 B  render-space triangle, leaf order                     (36 B of positions + pad)
//   DevTriShade 112 B  per-triangle shading attributes + geometric normal, fetched once per closest hit
//   DevMaterial  96 B  tagged material record
//   DevLightTri  64 B  emissive triangle (render space) + area CDF entry + normal
//   LUTs         470 f32 each; CIE x/y/z interleaved as float4 per nm
//   rgb2spec     [3][64][64][64] float4 (c0,c1,c2,0) + 64 z nodes
//   textures     RGBA8 (one dword per texel)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__
#else
#define PT_HD
#endif

namespace pt {

#ifndef PT_STACK_DEPTH
#define PT_STACK_DEPTH 24
#endif
constexpr int STACK_DEPTH = PT_STACK_DEPTH;        // per-lane traversal stack entries kept in LDS (6 KB per wave)
constexpr int MAX_LEAF_TRIS = 4;
constexpr int MAX_BUILD_DEPTH = 22;    // builder guarantees depth <= this (< STACK_DEPTH)
constexpr int HASH_TABLE_DIMS = 136;   // precomputed murmur(dimension, seed) entries: 3 + 16 bounces x 8 draws
                                       // LDS per wave: 7168 (stack) + 1088 (hash) + 544 (Sobol prefixes) = 8.8 KB -> 16 waves/CU fit in 160 KB

struct alignas(16) DevNode {
    // child i box: lo = (bx[i], by[i], bz[i]) hi = (bx[2+i], by[2+i], bz[2+i])
    float bx[4];   // lo0.x lo1.x hi0.x hi1.x
    float by[4];
    float bz[4];
    int32_t child[2];   // >= 0: node index; < 0: leaf, see leaf_first/leaf_count
    uint32_t pad[2];
};
static_assert(sizeof(DevNode) == 64, "node must be 64 B");
// The tree the wave-cooperative traversals walk: the BVH2 collapsed to up to FOUR children per node (scene.cpp collapse_bvh4).  A wave's
// node steps per ray are bounded below by the depth of the path it follows (stealing spreads the breadth of a traversal over idle lanes, not
// its depth): 18 steps for rays that need 10 on average in the BVH2.  Half the levels = half the dependent fetch -> test -> fetch round trips.
// Children are collapsed only where the per-lane LDS stack (STACK_DEPTH entries) provably cannot overflow: a node with k children
// leaves at most k - 1 pending siblings per level, and a subtree is collapsed only if the slots left cover its all-binary height.
struct alignas(16) DevNode4 {
    float lox[4], hix[4], loy[4], hiy[4], loz[4], hiz[4];   // child c box: lo = (lox[c], loy[c], loz[c]), hi likewise; unused slots: a point box at +FLT_MAX (never hit)
    int32_t child[4];                                       // >= 0: DevNode4 index; < 0: leaf (leaf_first / leaf_count)
    uint32_t pad[4];
};
static_assert(sizeof(DevNode4) == 128, "wide node must be 128 B");
// PT_NODE_FMA 1: the slab distances of the 4-wide step are ONE fma per plane, plane * (1/d) + (-(o * (1/d))), instead of a subtraction and a
// multiplication ((plane - o) * (1/d)).  The fma form cancels: its absolute error is u * |o / d| + u * |t| (u = 2^-24), i.e. up to 2 u * max(|o|, |plane|)
// measured in space, where the two-step form has 2 u * |t|.  The host therefore pads every box of the DevNode4 tree by NODE4_PAD_REL x the largest
// coordinate magnitude of the scene (2 x that bound) when it uploads the tree (scene.cpp): the padded test can only visit MORE nodes than the
// exact one, and which triangle a ray hits is decided by the unchanged watertight triangle test.  The BVH2 records (probes, canonical counts) stay exact.
#ifndef PT_NODE_FMA
#define PT_NODE_FMA 1        // measured (round 3, same box): pad 2^-20 +0.1...0.25 %, pad 2^-22 +0.55...0.75 % (scenes 3 / 8 / 10)
#endif
#ifndef PT_NODE_PAD_LOG2
#define PT_NODE_PAD_LOG2 22
#endif
constexpr float NODE4_PAD_REL = 1.0f / (float)(1u << PT_NODE_PAD_LOG2);   // 2^-22 = twice the error bound 2^-23 x R.  The pad must stay well below RAY_EPS (1e-5): 2^-20 of the
                                                                         // 12-unit Cornell room is 1.1e-5, and the flat leaf boxes of the walls then contain the origins of the rays
                                                                         // that leave them (every bounce tests the wall's own triangles again)
// PT_NODE_Q16 1: the cooperative traversals read a 64-BYTE node — the four child boxes as 16-bit planes on ONE grid over the scene's
// bounds (lo planes snapped down, hi planes up, so a quantised box contains its box), 4 x dwordx4 per lane and node step instead of 7;
// dequantisation is free (t = q * (cell / d) + (origin - o) / d: one fma on the converted integer, SDWA word select in the conversion).
// Layout: for each axis one 16-byte row {lo[0..3], hi[0..3]} as u16, then the four links; unused slots lo = 65535, hi = 0.
// Built, parity-green (160 GPU tests) and MEASURED SLOWER (round 3, same box): scene 3 2 076 -> 1 861, scenes 10 / 8 -10 %, scene 17 -8 %.
// Two reasons, both in the counters: (1) a grid cell of the Cornell room is 1.8e-4, eighteen times RAY_EPS — the flat leaf boxes of the
// walls become slabs that CONTAIN the origins of the rays leaving them, so every bounce off a wall tests the wall's own triangles again
// (triangle tests per sample 8.4 -> 12.1 closest, 3.1 -> 6.5 shadow: +2.2 dense triangle steps per iteration); (2) the node steps themselves
// got no faster with 43 % fewer L1 requests — like round 2's 32-byte nodes.  (An ablation that ADDS requests, PT_ABLATE_EXTRA_NODE_LOADS,
// loses 9 % per 3 requests, but it also adds 12 live registers; the two quantisation experiments say the request count is not the lever.)
#ifndef PT_NODE_Q16
#define PT_NODE_Q16 0
#endif
struct alignas(16) DevNode4Q {
    uint16_t q[3][2][4];     // [axis][0 = lo, 1 = hi][child]
    int32_t child[4];
};
static_assert(sizeof(DevNode4Q) == 64, "quantised wide node must be 64 B");
PT_HD inline int32_t make_leaf(uint32_t first, uint32_t count) { return (int32_t)(0x80000000u | (first << 3) | (count - 1)); }
PT_HD inline uint32_t leaf_first(int32_t c) { return ((uint32_t)c & 0x7fffffffu) >> 3; }
PT_HD inline uint32_t leaf_count(int32_t c) { return ((uint32_t)c & 7u) + 1; }

struct alignas(16) DevTri {
    float p0[3]; float p1x;
    float p1yz[2]; float p2xy[2];
    float p2z;
    uint32_t instance;            // DevInstance index
    uint32_t flags;               // bit 0: that instance is a pure translation (DevInstance::identity)
    uint32_t mclass;              // sort class of the triangle's material (MT_* | 8 if it has a spectrum texture) - known with the hit, one dependent fetch before
                                  // the shading record (the path queues sort on it, pt_kernel.hpp)
};
static_assert(sizeof(DevTri) == 48, "tri must be 48 B");

struct alignas(16) DevTriShade {
    float n0[3]; float n1x;        // LOCAL-space vertex normals (unit)
    float n1yz[2]; float n2xy[2];
    float n2z; float tangent[3];   // LOCAL-space per-triangle tangent (valid if flags & 1)
    float uv0[2]; float uv1[2];
    float uv2[2]; uint32_t material; uint32_t instance;
    uint32_t flags;                // bit0: has uv/tangent, bit1: emissive, bit2: instance transform has an identity linear part
    uint32_t light;                // light index if emissive else ~0
    uint32_t local_tri;            // triangle index inside its mesh
    float light_pdf_area;          // (1/area_i) * (cdf_i - cdf_{i-1}) for emissive tris (emissive_triangle_mesh.rs:334-353)
    float ng[3]; uint32_t pad_ng;  // RENDER-space geometric normal the reference's way: normalize(normalize(cross(p1-p0, p2-p0))) of the LOCAL vertices
                                   // (ray.rs:167-174) carried through Transform * Normal (samples.rs:135) - a function of the triangle alone,
                                   // computed once by the host with the arithmetic of the device code it replaces
};
static_assert(sizeof(DevTriShade) == 112, "shade record must be 112 B");

// The reference intersects every mesh in ITS OWN space: the ray goes through local_to_render.inverse() (a numeric Mat4 inverse, per call),
// the hit comes back through local_to_render (primitive/impls/triangle_mesh.rs:89-119, samples.rs:130-143).  Position, normals and even wo
// (= M * -(M^-1 * d)) therefore carry the roundings of that round trip, and rough GGX lobes / the Russian-roulette gate amplify their
// last bit into other paths.  The traversal walks render-space triangles (DevTri); the ONE triangle it returns is intersected again the
// reference's way (refine_hit, pt_path.hpp) with this record and the instance's two matrices, so that the shading point is the
// reference's bit for bit.
using DevTriLocal = DevTri;    // the same record with LOCAL-space positions in the mesh's own vertex order

struct alignas(16) DevInstance {
    float m[12];       // local_to_render: columns x, y, z, w (the xyz of each; the bottom row is 0 0 0 1)
    float inv[12];     // glam Mat4::inverse(local_to_render), same layout (scene.cpp mat4_inverse_glam)
    uint32_t identity; // 1: the 3x3 parts of both are exactly the identity (a translation): the multiplies are exact and skipped
    uint32_t pad[3];
};
static_assert(sizeof(DevInstance) == 112, "instance record");

enum : uint32_t { SPK_CONSTANT = 0, SPK_SIGMOID = 1, SPK_LUT = 2, SPK_TEXTURE = 3, SPK_ILLUM = 4 };   // ILLUM: pad[0] = scale bits, id = illuminant LUT
struct DevTexture {
    uint32_t offset;   // texel offset into the RGBA8 pool
    uint32_t w, h, pad;
};

struct DevSpectrum {
    uint32_t kind;
    uint32_t id;       // LUT index or texture index
    float c[3];        // constant in c[0] or sigmoid coefficients; SPK_TEXTURE: c[0] = SpectrumType of the texture as bits (0 Albedo, 1 Illuminant,
                       // 2 Unbounded: rgb_texture.rs:56-64), c[1] = the illuminant's LUT id as bits (Illuminant)
    uint32_t pad[3];   // SPK_ILLUM: pad[0] = the scale (float bits); SPK_TEXTURE: the texture's DevTexture {offset, w, h}, so that a lookup
                       // does not wait for a descriptor fetch between the material record and the texels
};
static_assert(sizeof(DevSpectrum) == 32, "spectrum param");

enum : uint32_t { MT_LAMBERT = 0, MT_EMISSIVE = 1, MT_GLASS = 2, MT_PLASTIC = 3, MT_CLEARCOAT = 4, MT_METAL = 5 };
struct alignas(16) DevMaterial {
    uint32_t type;
    uint32_t normal_tex;   // ~0 = none
    uint32_t normal_flip_y;
    uint32_t thin;
    float intensity, roughness, metallic, ior;
    float cc_ior, cc_roughness, cc_thickness;
    float intensity_avg;   // emissive: what EmissiveMaterial::average_intensity multiplies with (emissive_material.rs:69-76): `intensity`, or the
                           // intensity texture at uv (0.5, 0.5), sampled once by the host with the device's bilinear arithmetic
    DevSpectrum color;
    DevSpectrum eta;       // glass: LUT, plastic: constant
    DevSpectrum cc_tint;   // clearcoat tint; metal: extinction coefficient k
    uint32_t metallic_tex, roughness_tex;   // FloatParameter::Texture ids (red channel), ~0 = use the constants above; EMISSIVE materials:
                                            // metallic_tex is the INTENSITY texture (mi355pt_material_desc::intensity_tex)
    uint32_t cc_thickness_tex;
    uint32_t cc_albedo_lut;   // clearcoat: first entry of this material's 64-entry coat-albedo table in DevScene::cc_albedo
    DevTexture normal_desc;   // textures[normal_tex], embedded for the same reason as DevSpectrum::pad
};
static_assert(sizeof(DevMaterial) == 176, "material record");

struct alignas(16) DevLightTri {
    float p0[3]; float p1x;
    float p1yz[2]; float p2xy[2];
    float p2z; float cdf;   // area_table entry (normalised running sum)
    float n[3];             // geometric normal of the triangle (emissive_triangle_mesh.rs:213-221), precomputed like DevTriShade::ng
    uint32_t pad[3];
};
static_assert(sizeof(DevLightTri) == 64, "light tri");

enum : uint32_t { LK_AREA = 0, LK_POINT = 1, LK_SPOT = 2, LK_DIRECTIONAL = 3, LK_ENV = 4 };
struct alignas(16) DevLight {
    uint32_t first_tri, n_tris;   // area: into light_tris; environment light: first_tri = its index in DevScene::envs
    uint32_t material;            // emissive material (delta lights: a hidden one holding the spectrum, intensity 1)
    float area_sum;               // area: sum of triangle areas; delta: the scalar factor of phi (4 pi I, ...), so that
                                  // the light-pick weight is mean_lambda((spectrum * material.intensity) * area_sum) for every kind
    uint32_t kind;                // LK_*
    float intensity;              // delta lights
    float angle_inner, angle_outer;   // spot
    float pos[3];                 // point/spot: render-space position; directional: normalised render-space direction
    float pad0;
    float axis[3];                // spot: third row of the linear part of render_to_local ((inv * w).z = dot(axis, w))
    float pad1;
};
static_assert(sizeof(DevLight) == 64, "light record");

// EnvironmentLight (primitive/impls/environment_light.rs): DevScene::envs[DevScene::n_envs], in light-list order
struct DevEnv {
    const float* texels;          // [h][w][4] float RGB0, row 0 = +y pole
    const float* marginal;        // [h]
    const float* conditional;     // [h][w]
    uint32_t w, h;
    float total_weight, intensity;
    float l2r[9], r2l[9];         // linear parts, column-major 3x3
    uint32_t illuminant_lut;      // presets::cie_illum_d6500()
    uint32_t light_index;         // position in the light list
    uint32_t pad[2];
};

struct DevScene {
    const DevNode* nodes;         // BVH2: the plain traversals (probes, canonical step counts)
    const DevNode4* nodes4;       // the same tree collapsed to <= 4 children per node: the render path's cooperative traversals (PT_NODE_Q16 0)
    const DevNode4Q* nodes4q;     // ... with its boxes on the 16-bit scene grid (PT_NODE_Q16 1: what the traversals read)
    float grid_org[3], grid_cell[3];   // plane = grid_org + q * grid_cell
    const DevTri* tris;
    const DevTriShade* shade;
    const DevTriLocal* tris_local; // leaf order, like tris / shade
    const DevTri* tris_render;     // render-space positions (mi355pt_scene_export_bvh; = tris unless tris_are_local)
    // When EVERY instance of the scene is the same pure translation (meshes placed with the identity: the Cornell scenes of BASELINE configs 1-4)
    // the reference's local ray is the render ray with its origin shifted by one constant, for every primitive: `tris` then IS tris_local and
    // every triangle test of every traversal runs on (origin + tri_shift, direction) - the reference's own test, any-hit included, and
    // winner_hit has nothing left to redo.  Otherwise tri_shift = 0, tris = tris_render and winner_hit re-tests the one triangle found.
    float tri_shift[3]; uint32_t tris_are_local;
    float shared_mw[3]; uint32_t pad_mw;   // local_to_render's translation in that case (load_surface)
    const DevInstance* instances;
    const DevMaterial* materials;
    const DevLight* lights;
    const DevLightTri* light_tris;
    const float* light_uvs;       // [n light tris][6]: the emissive triangles' vertex uvs (textured emitters: radiance at the sampled point)
    const float* luts;            // [n_luts][470]
    const float* cmf;             // [470][4]  (xbar, ybar, zbar, 0)
    const float* rgb2spec;        // [3][64][64][64][4]
    const float* z_nodes;         // [64]
    const uint32_t* texels;       // RGBA8 pool
    const DevTexture* textures;
    const float* cc_albedo;       // [n clearcoat materials][64] E(cos theta) of the coat's directional albedo (mi355pt_params.albedo_lut)
    uint32_t n_nodes, n_tris, n_lights, n_materials;
    int32_t root;                 // root link (node index, or leaf if the scene has <= MAX_LEAF_TRIS tris)
    int32_t root4;                // root of nodes4
    uint32_t n_nodes4, pad1;
    const DevEnv* envs;           // every infinite light of the scene (Scene sums them all: scene.rs:185-231)
    uint32_t n_envs, pad_env;
};

struct DevCamera {
    float s[3], u[3], f[3];       // look_to_rh basis (camera.rs:58-62)
    float tan_half_fov, aspect;
    uint32_t width, height;
};

struct DevParams {
    uint32_t spp, seed, max_depth, strategy, sampler;
    float exposure;
    uint32_t log2_spp, n_base4_digits;     // ZSobolSampler::new (z_sobol_sampler.rs:179-196)
    uint32_t sample_begin, sample_end;
    uint32_t shard_index, shard_count;
    uint32_t tiles_x, tiles_y;
    uint32_t n_work;                        // tiles * sample chunks handled by this launch
    uint32_t chunks, chunk_size;            // sample-range split per tile (1 = none)
    uint32_t block_log2;                    // a work item covers a 2^b x 2^b pixel block of its 8x8 tile (3: the whole tile); fewer pixels
                                            // per item leave fewer per-lane Sobol digits to hash (pt_kernels.hip)
    uint32_t sample_prefix_digits;          // single-pixel items (b = 0) over aligned 4^m sample blocks: this many top base-4 digits of the
                                            // sample index are item-uniform too and join the Sobol prefix tables
    uint32_t stats_mode;                    // instrumented variant only: 1 = reference traversal order (canonical counts), 2 = production traversal
    uint32_t albedo_lut;                    // clearcoat coat weight from the per-material table instead of the 64-sample estimate
    float rr_gate;                          // Russian roulette is skipped when max(T) >= rr_gate (1 = the reference; mi355pt_params.rr_gate_slack)
    float xyz_to_rgb[9];                    // row-major sRGB matrix (gamut.rs:50-63)
};

// Scene feature bits: the host picks the smallest kernel specialisation that covers the scene's materials, so a
// Lambert-only Cornell box does not carry the registers and code of the clearcoat / dielectric / texture paths.
// FEAT_EMTEX: a textured emitter (radiance looked up at the hit / sampled uv) — its own bit so that textured SURFACES (scene 3, the headline
// config) do not carry the emitter-texture code in their register budget; FEAT_STD = every feature but that one.
enum : uint32_t { FEAT_TEX = 1, FEAT_DIEL = 2, FEAT_CC = 4, FEAT_MLIGHT = 8, FEAT_ROUGH = 16, FEAT_METAL = 32, FEAT_DELTA = 64, FEAT_ENV = 128, FEAT_EMTEX = 256,
                  FEAT_STD = 255, FEAT_ALL = 511 };

struct DevStats {
    unsigned long long samples, closest_rays, shadow_rays, nodes_closest, tris_closest, nodes_shadow, tris_shadow;
    unsigned long long closest_hits, bounces, spectrum_evals, textured_lookups;
    unsigned long long phase_cycles[10];
    unsigned long long wave_steps[8];
    unsigned long long busy_hist[2][8];
    unsigned long long divergence[12];
};

}  // namespace pt
