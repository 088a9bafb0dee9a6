// extern "C" boundary of libmi355pt.so (include/mi355pt.h).  Host C++ only; the compute lives in pt_kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi355pt.h"
#include "../../include/mi355pt_debug.h"
#include "layout.hpp"
#include "scene.hpp"

namespace pt {
struct PathOut { float* L; float* lam; float* pdf; uint32_t s_base, n_s; };   // per-sample log of a launch (pt_path.hpp)
hipError_t launch_pt(const DevScene&, const DevCamera&, const DevParams&, const uint64_t*, float*, float*, unsigned*, DevStats*, bool, uint32_t, int, hipStream_t,
                     const PathOut&, float*);
size_t query_defer_bytes_per_wave();
hipError_t launch_resolve(const float*, uint32_t, uint32_t, float*, hipStream_t);
hipError_t launch_film_add(float*, const float*, size_t, hipStream_t);
hipError_t launch_film_pack(const float*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, float*, hipStream_t);
hipError_t launch_film_unpack(float*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const float*, hipStream_t);
hipError_t launch_probe_sobol(uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t*, uint32_t, const uint8_t*, uint32_t, uint32_t, uint32_t*, hipStream_t);
hipError_t launch_probe_intersect(const DevScene&, const float*, const float*, uint32_t, float*, uint32_t*, uint32_t*, float*, hipStream_t);
hipError_t launch_probe_occluded(const DevScene&, const float*, const float*, const float*, uint32_t, uint8_t*, hipStream_t);
hipError_t launch_probe_sincos(uint32_t, uint32_t, uint32_t, float*, float*, hipStream_t);
uint64_t host_murmur_dim_seed(uint32_t dimension, uint32_t seed);
int query_resident_waves(bool stats, uint32_t feat, uint32_t sampler, uint32_t strategy);
}  // namespace pt

using namespace pt;

static thread_local std::string g_err;
static bool g_debug_unlocked = false;      // mi355pt_debug_unlock: lets mi355pt_params.rr_gate_slack through
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(MI355PT_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

static const uint32_t CIE_CMF_BITS[470 * 4] = {
#include "cie_cmf.inc"
};

// Per-scene launch resources, allocated once (no hipMalloc/hipFree/sync on the launch path, so a caller can queue
// launches on its own stream or capture them): the MurmurHash(dimension, seed) table per seed and a ring of work
// counters / stats blocks so that back-to-back asynchronous launches never share a counter.
constexpr int CTX_RING = 16;
struct LaunchCtx {
    int device = -1;                  // the device every buffer below lives on (= SceneImpl::device when the context was made)
    int waves[2][2][3] = {{{0}}};     // [instrumented][sampler][strategy]: resident waves of the kernel that combination launches (0: not asked yet)
    uint64_t* d_hash = nullptr;
    uint32_t hash_seed = 0;
    bool hash_valid = false;
    unsigned* d_counters = nullptr;   // CTX_RING counters
    DevStats* d_stats = nullptr;      // CTX_RING blocks
    float* d_defer = nullptr;         // the resident waves' deferral queues (pt_kernel.hpp PT_DEFER: 16 KB per wave), sized for the largest grid seen
    size_t defer_bytes = 0;
    float* d_partial = nullptr;       // per-chunk film tiles of split launches (tiles * chunks * 64 * 3 floats), grown on demand;
    size_t partial_floats = 0;        // reused by consecutive launches: one stream at a time per scene
    int next = 0;
    ~LaunchCtx() {
        // freed with the owning device current (a scene rebuilt on another device drops its context from there)
        int cur = -1;
        const bool swap = device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess;
        (void)hipFree(d_hash); (void)hipFree(d_counters); (void)hipFree(d_stats); (void)hipFree(d_partial); (void)hipFree(d_defer);
        if (swap) (void)hipSetDevice(cur);
    }
};
// One device's share of a multi-device scene (mi355pt_scene_build_multi): a full replica of the scene on that device plus
// the stream, film and event mi355pt_render_multi drives it with.  Replica 0 is the scene object itself.
struct MultiPart {
    int device = -1;
    mi355pt_scene* scene = nullptr;      // owned unless it is the parent (part 0)
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    float* d_film = nullptr;             // full-frame linear film of this device's tile shard
    float* d_pack = nullptr;             // parts 1..: the shard's tiles as a compact film (tile-major, 192 floats per tile) — what crosses xGMI
    float* d_stage = nullptr;            // part 0 only: one landing area per peer for those compact films
    float* d_out = nullptr;              // part 0 only: resolved frame
    size_t film_floats = 0, pack_floats = 0, stage_floats = 0;
};
struct mi355pt_scene {
    SceneImpl impl;
    mutable LaunchCtx* ctx = nullptr;
    mutable std::vector<MultiPart> parts;   // empty unless built with mi355pt_scene_build_multi
    ~mi355pt_scene();
};

namespace {

struct V3 { float x, y, z; };
V3 cross3(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
V3 norm3(V3 a) { float r = 1.0f / std::sqrt((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); return {a.x * r, a.y * r, a.z * r}; }

uint32_t log2_int(uint32_t v) { return v == 0 ? 0 : 31 - (uint32_t)__builtin_clz(v); }
uint32_t round_up_pow2(uint32_t v) { return v <= 1 ? 1 : 1u << (32 - __builtin_clz(v - 1)); }

DevCamera make_camera(const mi355pt_camera* c) {
    DevCamera d{};
    V3 f = norm3(V3{c->direction[0], c->direction[1], c->direction[2]});       // set_look_to normalises (camera.rs:46-48)
    V3 up = norm3(V3{c->up[0], c->up[1], c->up[2]});
    V3 s = norm3(cross3(f, up));                                                // glam Mat3::look_to_rh
    V3 u = cross3(s, f);
    d.s[0] = s.x; d.s[1] = s.y; d.s[2] = s.z; d.u[0] = u.x; d.u[1] = u.y; d.u[2] = u.z; d.f[0] = f.x; d.f[1] = f.y; d.f[2] = f.z;
    float fov_rad = c->fov_deg * (3.14159265358979323846f / 180.0f);
    d.tan_half_fov = std::tan(fov_rad / 2.0f);
    d.aspect = (float)c->width / (float)c->height;
    d.width = c->width; d.height = c->height;
    return d;
}

// GamutSrgb::new().xyz_to_rgb() (color/src/gamut.rs:29-63), glam Mat3 arithmetic in f32
void srgb_xyz_to_rgb(float out_rowmajor[9]) {
    auto xy = [](float x, float y) { return V3{x * 1.0f / y, 1.0f, (1.0f - x - y) * 1.0f / y}; };
    V3 r = xy(0.64f, 0.33f), g = xy(0.30f, 0.60f), b = xy(0.15f, 0.06f), w = xy(0.3127f, 0.3290f);
    auto inv = [](V3 x, V3 y, V3 z, V3 o[3]) {   // returns columns of the inverse
        V3 t0 = cross3(y, z), t1 = cross3(z, x), t2 = cross3(x, y);
        float det = (z.x * t2.x) + (z.y * t2.y) + (z.z * t2.z);
        float id = 1.0f / det;
        V3 r0{t0.x * id, t0.y * id, t0.z * id}, r1{t1.x * id, t1.y * id, t1.z * id}, r2{t2.x * id, t2.y * id, t2.z * id};
        o[0] = V3{r0.x, r1.x, r2.x}; o[1] = V3{r0.y, r1.y, r2.y}; o[2] = V3{r0.z, r1.z, r2.z};
    };
    auto mulv = [](const V3 m[3], V3 v) {
        return V3{m[0].x * v.x + m[1].x * v.y + m[2].x * v.z, m[0].y * v.x + m[1].y * v.y + m[2].y * v.z, m[0].z * v.x + m[1].z * v.y + m[2].z * v.z};
    };
    V3 rgb[3] = {r, g, b}, irgb[3];
    inv(r, g, b, irgb);
    V3 c = mulv(irgb, w);
    V3 r2x[3] = {V3{rgb[0].x * c.x, rgb[0].y * c.x, rgb[0].z * c.x}, V3{rgb[1].x * c.y, rgb[1].y * c.y, rgb[1].z * c.y},
                 V3{rgb[2].x * c.z, rgb[2].y * c.z, rgb[2].z * c.z}};
    V3 x2r[3];
    inv(r2x[0], r2x[1], r2x[2], x2r);
    // row-major: row i = (col0[i], col1[i], col2[i])
    out_rowmajor[0] = x2r[0].x; out_rowmajor[1] = x2r[1].x; out_rowmajor[2] = x2r[2].x;
    out_rowmajor[3] = x2r[0].y; out_rowmajor[4] = x2r[1].y; out_rowmajor[5] = x2r[2].y;
    out_rowmajor[6] = x2r[0].z; out_rowmajor[7] = x2r[1].z; out_rowmajor[8] = x2r[2].z;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
};

// the persistent grid of the exact (instrumented?, sampler, strategy) kernel of this scene's feature set on this scene's device
int resident_waves(LaunchCtx* lc, bool stats, uint32_t feat, uint32_t sampler, uint32_t strategy) {
    int& c = lc->waves[stats ? 1 : 0][sampler & 1u][strategy < 3u ? strategy : 0u];
    if (!c) c = query_resident_waves(stats, feat, sampler, strategy);
    return c;
}

DevParams make_params(const mi355pt_camera* cam, const mi355pt_params* p, uint32_t s_begin, uint32_t s_end) {
    DevParams d{};
    d.spp = p->spp; d.seed = p->seed; d.max_depth = p->max_depth; d.strategy = p->strategy; d.sampler = p->sampler;
    d.exposure = p->exposure;
    d.rr_gate = 1.0f - p->rr_gate_slack;
    d.albedo_lut = p->albedo_lut ? 1u : 0u;
    d.log2_spp = log2_int(p->spp);                                              // ZSobolSampler::new (:179-196)
    uint32_t res = round_up_pow2(std::max(cam->width, cam->height));
    d.n_base4_digits = log2_int(res) + (d.log2_spp + 1) / 2;
    d.sample_begin = s_begin; d.sample_end = s_end;
    d.shard_count = p->shard_count ? p->shard_count : 1;
    d.shard_index = p->shard_count ? p->shard_index : 0;
    d.tiles_x = (cam->width + 7) / 8; d.tiles_y = (cam->height + 7) / 8;
    srgb_xyz_to_rgb(d.xyz_to_rgb);
    return d;
}

// returns the scene's launch context with the hash table valid for `seed`; `slot` receives a fresh ring slot
int get_launch_ctx(const mi355pt_scene* sc, uint32_t seed, hipStream_t stream, LaunchCtx** out, int* slot) {
    // (check_args has made sure that the current device is the one the scene was built on)
    if (sc->ctx && sc->ctx->device != sc->impl.device) {
        // the scene was rebuilt on another device since its last render (mi355pt_scene_build after hipSetDevice, or build_multi with a
        // different first device): the hash table, counters, stats and partial film of the old context live in the OLD device's memory
        delete sc->ctx; sc->ctx = nullptr;
    }
    if (!sc->ctx) {
        LaunchCtx* lc = new LaunchCtx();
        lc->device = sc->impl.device;
        HIP_TRY(hipMalloc((void**)&lc->d_hash, sizeof(uint64_t) * HASH_TABLE_DIMS));
        HIP_TRY(hipMalloc((void**)&lc->d_counters, sizeof(unsigned) * CTX_RING));
        HIP_TRY(hipMalloc((void**)&lc->d_stats, sizeof(DevStats) * CTX_RING));
        sc->ctx = lc;
    }
    LaunchCtx* lc = sc->ctx;
    if (!lc->hash_valid || lc->hash_seed != seed) {
        std::vector<uint64_t> tab(HASH_TABLE_DIMS);
        for (int i = 0; i < HASH_TABLE_DIMS; ++i) tab[i] = host_murmur_dim_seed((uint32_t)i, seed);
        // a seed change is rare (the CLI renders one seed): synchronous copy, ordered after any in-flight launch
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(lc->d_hash, tab.data(), sizeof(uint64_t) * HASH_TABLE_DIMS, hipMemcpyHostToDevice));
        lc->hash_seed = seed; lc->hash_valid = true;
    }
    *slot = lc->next; lc->next = (lc->next + 1) % CTX_RING;
    *out = lc;
    return MI355PT_OK;
}

int check_args(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p) {
    if (!s || !cam || !p) return fail(MI355PT_E_INVALID, "null argument");
    if (!s->impl.built) return fail(MI355PT_E_NOT_BUILT, "scene not built");
    if (cam->width == 0 || cam->height == 0 || p->spp == 0) return fail(MI355PT_E_INVALID, "empty image or spp == 0");
    if (p->strategy > 2 || p->sampler > 1) return fail(MI355PT_E_INVALID, "bad strategy/sampler");
    if (p->max_depth > 1000u) return fail(MI355PT_E_INVALID, "max_depth > 1000 (the path records of the kernel's queues hold the depth in 10 bits and the sampler dimension in 15)");
    if (!(p->rr_gate_slack >= 0.0f && p->rr_gate_slack < 1.0f)) return fail(MI355PT_E_INVALID, "rr_gate_slack must be in [0, 1)");
    if (p->rr_gate_slack != 0.0f && !g_debug_unlocked)
        return fail(MI355PT_E_INVALID, "mi355pt_params.rr_gate_slack must be 0 (a diagnostic: mi355pt_debug_unlock(1) in mi355pt_debug.h enables it)");
    if (p->shard_count && p->shard_index >= p->shard_count) return fail(MI355PT_E_INVALID, "bad shard");
    // mi355pt_scene_build bakes the world -> render translation (render space = world - camera position, camera.rs:84-86) into every
    // device record and uploads to the device that was current then: a render call must name the same camera position and device
    if (std::memcmp(cam->position, s->impl.build_cam_pos, sizeof(float) * 3) != 0)
        return fail(MI355PT_E_INVALID, "camera position differs from the one given to mi355pt_scene_build: rebuild the scene");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != s->impl.device)
        return fail(MI355PT_E_DEVICE, "current HIP device is not the device the scene was built on");
    return MI355PT_OK;
}

}  // namespace

static void free_parts(std::vector<MultiPart>& parts) {
    for (size_t i = 0; i < parts.size(); ++i) {
        MultiPart& m = parts[i];
        (void)hipSetDevice(m.device);
        if (m.stream) (void)hipStreamDestroy(m.stream);
        if (m.done) (void)hipEventDestroy(m.done);
        (void)hipFree(m.d_film); (void)hipFree(m.d_pack); (void)hipFree(m.d_stage); (void)hipFree(m.d_out);
        if (i > 0) delete m.scene;
    }
    parts.clear();
}
mi355pt_scene::~mi355pt_scene() {
    int cur = 0;
    const bool have = !parts.empty() && hipGetDevice(&cur) == hipSuccess;
    free_parts(parts);
    if (have) (void)hipSetDevice(cur);
    delete ctx;
}

extern "C" {

const char* mi355pt_last_error(void) { return g_err.c_str(); }
int mi355pt_debug_unlock(int on) { const int was = g_debug_unlocked ? 1 : 0; g_debug_unlocked = on != 0; return was; }
#ifndef MI355PT_BUILD_ID
#define MI355PT_BUILD_ID "dev"
#endif
const char* mi355pt_version(void) { return "mi355pt 0.2.0 (gfx950) build " MI355PT_BUILD_ID; }

int mi355pt_scene_create(mi355pt_scene** out) {
    if (!out) return fail(MI355PT_E_INVALID, "null out");
    *out = new (std::nothrow) mi355pt_scene();
    return *out ? MI355PT_OK : fail(MI355PT_E_INVALID, "allocation failed");
}
void mi355pt_scene_destroy(mi355pt_scene* s) { delete s; }

int mi355pt_scene_set_rgb2spec(mi355pt_scene* s, const float* table, size_t n) {
    if (!s || !table || n != (size_t)(64 + 3 * 64 * 64 * 64 * 3)) return fail(MI355PT_E_INVALID, "rgb2spec table must have 64 + 3*64^3*3 floats");
    s->impl.table.assign(table, table + n);
    return MI355PT_OK;
}
int mi355pt_scene_add_lut470(mi355pt_scene* s, const float* v, uint32_t* id) {
    if (!s || !v || !id) return fail(MI355PT_E_INVALID, "null argument");
    s->impl.luts.emplace_back(v, v + 470);
    *id = (uint32_t)s->impl.luts.size() - 1;
    return MI355PT_OK;
}
int mi355pt_scene_add_tex_rgb8(mi355pt_scene* s, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t* id) {
    if (!s || !rgb || !id || w == 0 || h == 0) return fail(MI355PT_E_INVALID, "bad texture");
    SceneImpl::Tex t; t.w = w; t.h = h; t.rgb.assign(rgb, rgb + (size_t)w * h * 3);
    s->impl.textures.push_back(std::move(t));
    *id = (uint32_t)s->impl.textures.size() - 1;
    return MI355PT_OK;
}
int mi355pt_scene_add_mesh(mi355pt_scene* s, const float* pos, const float* nrm, const float* uv, const float* tri_tangent, const uint32_t* idx,
                           uint32_t nv, uint32_t nt, uint32_t* out) {
    if (!s || !pos || !nrm || !idx || !out || nv == 0 || nt == 0) return fail(MI355PT_E_INVALID, "bad mesh");
    if ((uv != nullptr) != (tri_tangent != nullptr)) return fail(MI355PT_E_INVALID, "uv and tri_tangent must be given together");
    for (size_t i = 0; i < (size_t)nt * 3; ++i) if (idx[i] >= nv) return fail(MI355PT_E_INVALID, "vertex index out of range");
    // a non-finite position would poison every box above it in the BVH: refuse it here rather than render garbage
    for (size_t i = 0; i < (size_t)nv * 3; ++i) if (!std::isfinite(pos[i])) return fail(MI355PT_E_INVALID, "non-finite vertex position");
    HostMesh m; m.n_vert = nv; m.n_tri = nt;
    m.pos.assign(pos, pos + (size_t)nv * 3);
    m.nrm.resize((size_t)nv * 3);
    for (uint32_t i = 0; i < nv; ++i) {   // Normal::new normalises and Normal::from renormalises (normal.rs:18-20,93-100)
        V3 n = norm3(norm3(V3{nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]}));
        m.nrm[3 * i] = n.x; m.nrm[3 * i + 1] = n.y; m.nrm[3 * i + 2] = n.z;
    }
    if (uv) { m.uv.assign(uv, uv + (size_t)nv * 2); m.tangent.assign(tri_tangent, tri_tangent + (size_t)nt * 3); }
    m.idx.assign(idx, idx + (size_t)nt * 3);
    s->impl.meshes.push_back(std::move(m));
    *out = (uint32_t)s->impl.meshes.size() - 1;
    return MI355PT_OK;
}
int mi355pt_scene_add_material(mi355pt_scene* s, const mi355pt_material_desc* d, uint32_t* out) {
    if (!s || !d || !out) return fail(MI355PT_E_INVALID, "null argument");
    SceneImpl& im = s->impl;
    DevMaterial m{};
    std::string err;
    int rc;
    m.type = d->type;
    m.normal_tex = d->normal_tex; m.normal_flip_y = d->normal_flip_y; m.thin = d->thin;
    m.intensity = d->intensity; m.roughness = d->roughness; m.metallic = d->metallic; m.ior = d->ior;
    m.cc_ior = d->clearcoat_ior; m.cc_roughness = d->clearcoat_roughness; m.cc_thickness = d->clearcoat_thickness;
    m.metallic_tex = m.roughness_tex = m.cc_thickness_tex = 0xffffffffu;
    m.intensity_avg = d->intensity;
    if (d->type == MI355PT_MAT_EMISSIVE && d->intensity_tex != MI355PT_NONE) {
        // FloatParameter::texture intensity (emissive_material.rs:55-56,69-76): the device reads it at the hit / sampled uv through the
        // material's metallic_tex slot; the light-pick weight uses ONE value, the texture at uv (0.5, 0.5), computed here with the device's
        // bilinear arithmetic (texture/sampler.rs:81-107: red channel of the gamma-encoded texel / 255)
        if (d->intensity_tex >= im.textures.size()) return fail(MI355PT_E_INVALID, "bad intensity texture id");
        m.metallic_tex = d->intensity_tex;
        const SceneImpl::Tex& t = im.textures[d->intensity_tex];
        const float u = std::fabs(0.5f - std::trunc(0.5f)), v = 1.0f - std::fabs(0.5f - std::trunc(0.5f));
        const float x = u * ((float)t.w - 1.0f), y = v * ((float)t.h - 1.0f);
        const uint32_t x0 = (uint32_t)std::floor(x), y0 = (uint32_t)std::floor(y), x1 = std::min(x0 + 1u, t.w - 1u), y1 = std::min(y0 + 1u, t.h - 1u);
        const float fx = x - (float)x0, fy = y - (float)y0;
        auto red = [&](uint32_t xx, uint32_t yy) { return (float)t.rgb[((size_t)yy * t.w + xx) * 3] / 255.0f; };
        const float top = red(x0, y0) * (1.0f - fx) + red(x1, y0) * fx, bottom = red(x0, y1) * (1.0f - fx) + red(x1, y1) * fx;
        m.intensity_avg = top * (1.0f - fy) + bottom * fy;
    }
    if (d->type == MI355PT_MAT_CLEARCOAT && d->clearcoat_thickness_tex != MI355PT_NONE) {
        if (d->clearcoat_thickness_tex >= im.textures.size()) return fail(MI355PT_E_INVALID, "bad clearcoat thickness texture id");
        m.cc_thickness_tex = d->clearcoat_thickness_tex;
    }
    if (d->type == MI355PT_MAT_GLASS || d->type == MI355PT_MAT_PLASTIC) {   // roughness: FloatParameter (glass_material.rs:42, plastic_material.rs:43)
        if (d->roughness_tex != MI355PT_NONE && d->roughness_tex >= im.textures.size()) return fail(MI355PT_E_INVALID, "bad roughness texture id");
        m.roughness_tex = d->roughness_tex;
    }
    if (d->type == MI355PT_MAT_SIMPLE_PBR || d->type == MI355PT_MAT_CLEARCOAT || d->type == MI355PT_MAT_METAL) {
        if ((d->metallic_tex != MI355PT_NONE && d->metallic_tex >= im.textures.size()) || (d->roughness_tex != MI355PT_NONE && d->roughness_tex >= im.textures.size()))
            return fail(MI355PT_E_INVALID, "bad metallic/roughness texture id");
        m.metallic_tex = d->type == MI355PT_MAT_METAL ? 0xffffffffu : d->metallic_tex; m.roughness_tex = d->roughness_tex;
    }
    if (d->normal_tex != MI355PT_NONE && d->normal_tex >= im.textures.size()) return fail(MI355PT_E_INVALID, "bad normal texture id");
    switch (d->type) {
        case MI355PT_MAT_LAMBERT:
            if ((rc = im.lower_spectrum(d->color, &m.color, true, &err))) return fail(rc, err);
            break;
        case MI355PT_MAT_EMISSIVE:
            // SpectrumParameter::Texture radiance (emissive_material.rs:48-79): an sRGB texture of any SpectrumType, looked up at the hit / sampled uv
            if ((rc = im.lower_spectrum(d->color, &m.color, 2, &err))) return fail(rc, "emissive radiance: " + err);
            break;
        case MI355PT_MAT_GLASS:
        case MI355PT_MAT_PLASTIC:
            if ((rc = im.lower_spectrum(d->eta, &m.eta, false, &err))) return fail(rc, "eta: " + err);
            if ((rc = im.lower_spectrum(d->color, &m.color, d->type == MI355PT_MAT_PLASTIC, &err))) return fail(rc, err);
            break;
        case MI355PT_MAT_CLEARCOAT:
            if ((rc = im.lower_spectrum(d->color, &m.color, true, &err))) return fail(rc, err);
            if ((rc = im.lower_spectrum(d->clearcoat_tint, &m.cc_tint, true, &err))) return fail(rc, "clearcoat tint: " + err);
            break;
        case MI355PT_MAT_SIMPLE_PBR:      // SimplePbrMaterial == the clearcoat material's base layer: thickness 0 takes exactly that path
            m.type = MT_CLEARCOAT; m.cc_thickness = 0.0f; m.cc_ior = 1.5f; m.cc_roughness = 0.0f;
            m.cc_tint.kind = SPK_CONSTANT; m.cc_tint.c[0] = 1.0f;
            if ((rc = im.lower_spectrum(d->color, &m.color, true, &err))) return fail(rc, err);
            break;
        case MI355PT_MAT_METAL:
            if ((rc = im.lower_spectrum(d->eta, &m.eta, false, &err))) return fail(rc, "eta: " + err);
            if ((rc = im.lower_spectrum(d->k, &m.cc_tint, false, &err))) return fail(rc, "k: " + err);
            break;
        default:
            return fail(MI355PT_E_INVALID, "material type not implemented on the device yet");
    }
    im.materials.push_back(m);
    im.mat_descs.push_back(*d);
    *out = (uint32_t)im.materials.size() - 1;
    return MI355PT_OK;
}
int mi355pt_scene_add_delta_light(mi355pt_scene* s, const mi355pt_light_desc* d) {
    if (!s || !d) return fail(MI355PT_E_INVALID, "null argument");
    if (d->kind < MI355PT_LIGHT_POINT || d->kind > MI355PT_LIGHT_DIRECTIONAL) return fail(MI355PT_E_INVALID, "bad light kind");
    SceneImpl& im = s->impl;
    DevMaterial m{};                        // hidden emissive material: carries the light's spectrum with intensity 1
    std::string err;
    int rc;
    m.type = MT_EMISSIVE; m.normal_tex = 0xffffffffu; m.metallic_tex = m.roughness_tex = m.cc_thickness_tex = 0xffffffffu; m.intensity = 1.0f; m.intensity_avg = 1.0f;
    if ((rc = im.lower_spectrum(d->spectrum, &m.color, false, &err))) return fail(rc, "light spectrum: " + err);
    im.materials.push_back(m);
    mi355pt_material_desc md{}; md.type = MI355PT_MAT_EMISSIVE; md.color = d->spectrum; md.intensity = 1.0f; md.normal_tex = MI355PT_NONE; md.intensity_tex = MI355PT_NONE;
    im.mat_descs.push_back(md);
    HostDeltaLight hl{*d, (uint32_t)im.materials.size() - 1, (uint32_t)im.instances.size()};
    im.delta_lights.push_back(hl);
    return MI355PT_OK;
}
int mi355pt_scene_add_environment_light(mi355pt_scene* s, float intensity, const float* rgb, uint32_t w, uint32_t h, const float* l2w,
                                        uint32_t illuminant_lut) {
    if (!s || !rgb || !l2w || w == 0 || h == 0) return fail(MI355PT_E_INVALID, "bad environment light arguments");
    SceneImpl& im = s->impl;
    if (illuminant_lut >= im.luts.size()) return fail(MI355PT_E_INVALID, "bad illuminant LUT id");
    SceneImpl::HostEnv he;
    he.intensity = intensity; he.w = w; he.h = h; he.illuminant_lut = illuminant_lut;
    he.rgb.assign(rgb, rgb + (size_t)w * h * 3);
    std::memcpy(he.l2w, l2w, sizeof(float) * 16);
    im.envs.push_back(std::move(he));
    DevMaterial m{};                         // hidden emissive material: the integrated RgbIlluminantSpectrum, filled in at build()
    m.type = MT_EMISSIVE; m.normal_tex = 0xffffffffu; m.metallic_tex = m.roughness_tex = m.cc_thickness_tex = 0xffffffffu; m.intensity = 1.0f; m.intensity_avg = 1.0f; m.color.kind = SPK_CONSTANT;
    im.materials.push_back(m);
    mi355pt_material_desc md{}; md.type = MI355PT_MAT_EMISSIVE; md.intensity = 1.0f; md.normal_tex = MI355PT_NONE; md.intensity_tex = MI355PT_NONE;
    im.mat_descs.push_back(md);
    mi355pt_light_desc ld{}; ld.kind = LK_ENV; ld.intensity = intensity; std::memcpy(ld.local_to_world, l2w, sizeof(float) * 16);
    im.delta_lights.push_back(HostDeltaLight{ld, (uint32_t)im.materials.size() - 1, (uint32_t)im.instances.size(), (uint32_t)im.envs.size() - 1});
    return MI355PT_OK;
}
int mi355pt_scene_add_instance(mi355pt_scene* s, uint32_t geom, uint32_t mat, const float* l2w) {
    if (!s || !l2w) return fail(MI355PT_E_INVALID, "null argument");
    if (geom >= s->impl.meshes.size() || mat >= s->impl.materials.size()) return fail(MI355PT_E_INVALID, "bad geometry/material id");
    HostInstance hi; hi.geom = geom; hi.mat = mat; std::memcpy(hi.l2w, l2w, sizeof(float) * 16);
    s->impl.instances.push_back(hi);
    return MI355PT_OK;
}
int mi355pt_coat_albedo_table(float alpha, float r0, float* out) {
    if (!out || !(alpha >= 0.0f) || !(r0 >= 0.0f)) return fail(MI355PT_E_INVALID, "bad argument");
    coat_albedo_table(alpha, r0, out);
    return MI355PT_OK;
}
int mi355pt_scene_set_bvh_builder(mi355pt_scene* s, int mode) {
    if (!s) return fail(MI355PT_E_INVALID, "null argument");
    if (mode != MI355PT_BVH_AUTO && mode != MI355PT_BVH_HOST && mode != MI355PT_BVH_GPU) return fail(MI355PT_E_INVALID, "unknown BVH builder mode");
    s->impl.bvh_builder = mode;
    return MI355PT_OK;
}
int mi355pt_scene_debug_set_lowering(mi355pt_scene* s, int mode) {
    if (!s) return fail(MI355PT_E_INVALID, "null argument");
    if (mode < 0 || mode > 2) return fail(MI355PT_E_INVALID, "unknown lowering mode");
    s->impl.lowering = mode;
    return MI355PT_OK;
}
int mi355pt_scene_build(mi355pt_scene* s, const mi355pt_camera* cam) {
    if (!s || !cam) return fail(MI355PT_E_INVALID, "null argument");
    std::string err;
    float cmf[470 * 4];
    std::memcpy(cmf, CIE_CMF_BITS, sizeof(cmf));
    int rc = s->impl.build(cam, cmf, &err);
    // the launch context belongs to the old build: another feature set launches other kernels (cached grid sizes), another device makes its
    // buffers foreign memory — get_launch_ctx makes a new one, on the build's device, at the next render
    delete s->ctx; s->ctx = nullptr;
    return rc ? fail(rc, err) : MI355PT_OK;
}

static constexpr uint32_t PT_MAX_LAUNCH_SAMPLES = 4096;
static int render_accum_range(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p, uint32_t s_begin, uint32_t s_end,
                              float* d_accum, void* hip_stream, mi355pt_stats* stats, const PathOut& pout) {
    int rc = check_args(s, cam, p);
    if (rc) return rc;
    if (!d_accum || s_end > p->spp || s_begin >= s_end) return fail(MI355PT_E_INVALID, "bad sample range or null accumulator");
    if (p->sampler == MI355PT_SAMPLER_SOBOL && !stats && s_end - s_begin > PT_MAX_LAUNCH_SAMPLES) {
        // long Sobol ranges go out as aligned blocks of 4096 sample indices: single-pixel work items over an aligned 4^6 block hash the
        // fewest digits per draw (the digits above the block join the prefix tables), and no launch runs for minutes
        for (uint32_t b = s_begin; b < s_end;) {
            const uint32_t e = std::min(s_end, (b / PT_MAX_LAUNCH_SAMPLES + 1u) * PT_MAX_LAUNCH_SAMPLES);
            if ((rc = render_accum_range(s, cam, p, b, e, d_accum, hip_stream, nullptr, pout))) return rc;
            b = e;
        }
        return MI355PT_OK;
    }
    hipStream_t stream = (hipStream_t)hip_stream;
    DevCamera dc = make_camera(cam);
    DevParams dp = make_params(cam, p, s_begin, s_end);
    uint32_t n_tiles_total = dp.tiles_x * dp.tiles_y;
    uint32_t n_tiles = n_tiles_total > dp.shard_index ? (n_tiles_total - dp.shard_index + dp.shard_count - 1) / dp.shard_count : 0;
    if (n_tiles == 0) return MI355PT_OK;
    LaunchCtx* lc; int slot;
    if ((rc = get_launch_ctx(s, p->seed, stream, &lc, &slot))) return rc;
    if (lc->device != s->impl.device) return fail(MI355PT_E_DEVICE, "launch context and scene live on different devices");
    int waves = resident_waves(lc, stats && p->collect_stats, s->impl.features, dp.sampler, dp.strategy);
    // Work items.  A work item is a 2^b x 2^b pixel block of an 8x8 tile times a range of sample indices, its (pixel, sample)
    // pairs handed to the lanes as a pool.  Sobol: the fewer pixels an item has, the fewer Morton digits vary inside it, and only
    // varying digits (minus the two that have block-level tables) are hashed per draw (pt_device.hpp sampler_index): take the
    // smallest block that still gives the pool >= PT_MIN_ITEM_SAMPLES pairs, so lanes keep finding new paths and the
    // per-item prefix tables stay amortised.
    uint32_t n_samples = s_end - s_begin;
    uint32_t block_log2 = 3;
    uint64_t PT_MIN_ITEM_SAMPLES = 2048;
#ifdef MI355PT_TUNING   // launch-shape sweeps (tools/block_sweep.sh, chunk_sweep.sh): not in the shipped library
    if (const char* e = getenv("MI355PT_MIN_ITEM")) PT_MIN_ITEM_SAMPLES = (uint64_t)std::max(64, atoi(e));
#endif
    if (dp.sampler == MI355PT_SAMPLER_SOBOL) {
        while (block_log2 > 0 && ((uint64_t)n_samples << (2u * (block_log2 - 1u))) >= PT_MIN_ITEM_SAMPLES) --block_log2;
    }
#ifdef MI355PT_TUNING
    if (const char* e = getenv("MI355PT_BLOCK")) { int b = atoi(e); if (b >= 0 && b <= 3) block_log2 = (uint32_t)b; }
#endif
    // the permuted block-uniform digits (everything above bit hi_shift of the 2 n - odd bit sample index) are packed into 27 bits
    // of a table word: large frames (>= 16384 pixels wide) need a larger block
    {
        const uint32_t odd = dp.log2_spp & 1u, index_bits = 2u * dp.n_base4_digits - odd;
        auto hi_shift = [&](uint32_t b) { return 2u * ((dp.log2_spp + 1u) / 2u + b) - odd; };
        while (block_log2 < 3 && (hi_shift(block_log2) < 6u || index_bits > hi_shift(block_log2) + 27u)) ++block_log2;
    }
    dp.block_log2 = block_log2;
    const uint32_t n_items = n_tiles * (64u >> (2u * block_log2));
    // split the sample range only when there are too few items to fill the chip (small images / many shards): about 8 work
    // items per resident wave, but no chunk under 16 samples (every work item rebuilds its Sobol prefix tables; measured
    // with tools/chunk_sweep.sh: one shard of 4 / 8 at 1080p is 2.2 % / 0.9 % faster with 16-sample than with 8-sample chunks)
    // (while some resident waves would have no item at all, chunks may go down to 8 samples: a 256x256 frame has 1 024 tiles)
    uint32_t chunks = 1;
    while (n_items * chunks < (uint32_t)waves * 8 && chunks * 2 <= n_samples &&
           (n_samples / (chunks * 2)) >= (n_items * chunks >= (uint32_t)waves ? 16u : 8u)) chunks *= 2;
#ifdef MI355PT_TUNING
    if (const char* e = getenv("MI355PT_CHUNKS")) { uint32_t c = (uint32_t)atoi(e); if (c >= 1 && c <= n_samples) chunks = c; }
#endif
    dp.chunks = chunks; dp.chunk_size = (n_samples + chunks - 1) / chunks;
    dp.n_work = n_items * chunks;
    // single-pixel items whose sample ranges are aligned blocks of 4^m indices: the sample digits above m are item-uniform as well
    dp.sample_prefix_digits = 0;
    if (dp.sampler == MI355PT_SAMPLER_SOBOL && block_log2 == 0 && (dp.log2_spp & 1u) == 0u && n_samples % chunks == 0) {
        const uint32_t cs = dp.chunk_size;
        uint32_t m = 0;
        while ((1u << (2u * (m + 1u))) <= cs) ++m;
        if ((1u << (2u * m)) == cs && s_begin % cs == 0 && m >= 3 && m <= dp.log2_spp / 2u &&
            2u * dp.n_base4_digits <= 2u * m + 27u) dp.sample_prefix_digits = dp.log2_spp / 2u - m;   // prefix above bit 2m must fit 27 bits
    }
    unsigned* d_counter = lc->d_counters + slot;
    DevStats* d_stats = lc->d_stats + slot;
    HIP_TRY(hipMemsetAsync(d_counter, 0, sizeof(unsigned), stream));
    bool want_stats = stats && p->collect_stats;
    dp.stats_mode = p->collect_stats;
    if (want_stats) HIP_TRY(hipMemsetAsync(d_stats, 0, sizeof(DevStats), stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (stats) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventRecord(e0, stream)); }
    int grid = (int)std::min<uint32_t>(dp.n_work, (uint32_t)waves);
    if (dp.chunks > 1) {
        const size_t need = (size_t)n_tiles * dp.chunks * 64u * 3u;   // one slot per 8x8 tile and chunk, whatever the block size
        if (need > lc->partial_floats) {
            HIP_TRY(hipStreamSynchronize(stream));                 // an earlier launch may still be reading the old buffer
            (void)hipFree(lc->d_partial); lc->d_partial = nullptr; lc->partial_floats = 0;
            HIP_TRY(hipMalloc((void**)&lc->d_partial, need * sizeof(float)));
            lc->partial_floats = need;
        }
    }
    if (const size_t per_wave = query_defer_bytes_per_wave()) {
        // (one stream at a time per scene, like d_partial: the queues are empty between launches, so consecutive launches share them)
        const size_t need = per_wave * (size_t)grid;
        if (need > lc->defer_bytes) {
            HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(lc->d_defer); lc->d_defer = nullptr; lc->defer_bytes = 0;
            HIP_TRY(hipMalloc((void**)&lc->d_defer, need));
            lc->defer_bytes = need;
        }
    }
    HIP_TRY(launch_pt(s->impl.dev, dc, dp, lc->d_hash, d_accum, lc->d_partial, d_counter, d_stats, want_stats, s->impl.features, grid, stream, pout, lc->d_defer));
    if (stats) {
        HIP_TRY(hipEventRecord(e1, stream));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        std::memset(stats, 0, sizeof(*stats));
        stats->kernel_ms = ms; stats->launches = 1;
        if (want_stats) {
            DevStats h;
            HIP_TRY(hipMemcpy(&h, d_stats, sizeof(h), hipMemcpyDeviceToHost));
            stats->samples = h.samples; stats->closest_rays = h.closest_rays; stats->shadow_rays = h.shadow_rays;
            stats->nodes_closest = h.nodes_closest; stats->tris_closest = h.tris_closest; stats->nodes_shadow = h.nodes_shadow;
            stats->tris_shadow = h.tris_shadow; stats->closest_hits = h.closest_hits; stats->bounces = h.bounces;
            stats->spectrum_evals = h.spectrum_evals; stats->textured_lookups = h.textured_lookups;
            for (int i = 0; i < 10; ++i) stats->phase_cycles[i] = h.phase_cycles[i];
            for (int i = 0; i < 8; ++i) stats->wave_steps[i] = h.wave_steps[i];
            for (int i = 0; i < 16; ++i) stats->busy_hist[i] = h.busy_hist[i >> 3][i & 7];
            for (int i = 0; i < 12; ++i) stats->divergence[i] = h.divergence[i];
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    return MI355PT_OK;   // stats == NULL: fully asynchronous on `stream`
}

int mi355pt_render_accum_device(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p, uint32_t s_begin, uint32_t s_end,
                                float* d_accum, void* hip_stream, mi355pt_stats* stats) {
    return render_accum_range(s, cam, p, s_begin, s_end, d_accum, hip_stream, stats, PathOut{nullptr, nullptr, nullptr, 0u, 0u});
}

// tiles of the frame that belong to the shard of `p`
static uint32_t shard_tiles(const mi355pt_camera* cam, const mi355pt_params* p) {
    const uint32_t total = ((cam->width + 7) / 8) * ((cam->height + 7) / 8);
    const uint32_t cnt = p->shard_count ? p->shard_count : 1u, idx = p->shard_count ? p->shard_index : 0u;
    return total > idx ? (total - idx + cnt - 1) / cnt : 0u;
}

int mi355pt_sample_log_records(const mi355pt_camera* cam, const mi355pt_params* p, uint32_t s_begin, uint32_t s_end, size_t* out_records) {
    if (!cam || !p || !out_records || s_end <= s_begin) return fail(MI355PT_E_INVALID, "bad argument");
    *out_records = (size_t)shard_tiles(cam, p) * 64u * (s_end - s_begin);
    return MI355PT_OK;
}

int mi355pt_render_sample_log(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p, uint32_t s_begin, uint32_t s_end,
                              float* out_L, float* out_lambda, float* out_pdf, size_t n_records, float* out_accum) {
    int rc = check_args(s, cam, p);
    if (rc) return rc;
    if (!out_L || !out_lambda || !out_pdf || s_end > p->spp || s_begin >= s_end) return fail(MI355PT_E_INVALID, "bad sample range or null output");
    if (p->spp & (p->spp - 1u)) return fail(MI355PT_E_INVALID, "the per-sample log needs a power-of-two spp (the sample index is read back from the Morton index)");
    if (p->collect_stats) return fail(MI355PT_E_INVALID, "the per-sample log is written by the production kernel, not the instrumented one");
    const size_t need = (size_t)shard_tiles(cam, p) * 64u * (s_end - s_begin);
    if (n_records != need) return fail(MI355PT_E_INVALID, "n_records must be tiles of the shard * 64 * (sample_end - sample_begin)");
    if (need == 0) return MI355PT_OK;
    const size_t n_film = (size_t)cam->width * cam->height * 3;
    DevBuf<float> d_L, d_lam, d_pdf, d_acc;
    HIP_TRY(d_L.alloc(need * 4)); HIP_TRY(d_lam.alloc(need * 4)); HIP_TRY(d_pdf.alloc(need * 4)); HIP_TRY(d_acc.alloc(n_film));
    HIP_TRY(hipMemset(d_L.p, 0, need * 16)); HIP_TRY(hipMemset(d_lam.p, 0, need * 16)); HIP_TRY(hipMemset(d_pdf.p, 0, need * 16));
    HIP_TRY(hipMemset(d_acc.p, 0, n_film * sizeof(float)));
    if ((rc = render_accum_range(s, cam, p, s_begin, s_end, d_acc.p, nullptr, nullptr, PathOut{d_L.p, d_lam.p, d_pdf.p, s_begin, s_end - s_begin}))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_L, d_L.p, need * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_lambda, d_lam.p, need * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_pdf, d_pdf.p, need * 16, hipMemcpyDeviceToHost));
    if (out_accum) HIP_TRY(hipMemcpy(out_accum, d_acc.p, n_film * sizeof(float), hipMemcpyDeviceToHost));
    return MI355PT_OK;
}

// ---------------- several GPUs of one node behind ONE call (single process) ----------------
// The reference calls the seam once from one process (renderer/src/main.rs:228).  mi355pt_scene_build_multi replicates the scene
// on every listed device (the working set is < 30 MB); mi355pt_render_multi deals the frame's 8x8 tiles round-robin to the devices
// (each launch on its own stream, concurrently), gathers the rank-local linear films onto the first device over xGMI peer copies
// and adds them there — the tile shards are disjoint, so the sum is exact and its order fixed — then resolves and copies out.
// No communicator is needed inside one process; the one-process-per-GPU path (bench.py, torch.distributed) reduces the same films
// with RCCL.
int mi355pt_scene_build_multi(mi355pt_scene* s, const mi355pt_camera* cam, int n_devices, const int* device_ids) {
    if (!s || !cam || !device_ids || n_devices < 1 || n_devices > 64) return fail(MI355PT_E_INVALID, "bad device list");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MI355PT_E_NO_DEVICE, "no HIP device: the product path requires a gfx950 GPU");
    for (int i = 0; i < n_devices; ++i) if (device_ids[i] < 0 || device_ids[i] >= ndev) return fail(MI355PT_E_INVALID, "device id out of range");
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    free_parts(s->parts);
    int rc = MI355PT_OK;
    std::vector<MultiPart> parts((size_t)n_devices);
    for (int i = 0; i < n_devices && rc == MI355PT_OK; ++i) {
        MultiPart& m = parts[(size_t)i];
        m.device = device_ids[i];
        if (hipSetDevice(m.device) != hipSuccess) { rc = fail(MI355PT_E_DEVICE, "hipSetDevice failed"); break; }
        if (i == 0) m.scene = s;
        else {
            m.scene = new (std::nothrow) mi355pt_scene();
            if (!m.scene) { rc = fail(MI355PT_E_INVALID, "allocation failed"); break; }
            SceneImpl& d = m.scene->impl; const SceneImpl& o = s->impl;       // the description, not the lowered state
            d.table = o.table; d.luts = o.luts; d.textures = o.textures; d.meshes = o.meshes; d.mat_descs = o.mat_descs; d.materials = o.materials;
            d.instances = o.instances; d.envs = o.envs; d.delta_lights = o.delta_lights; d.bvh_builder = o.bvh_builder; d.lowering = o.lowering;
        }
        rc = mi355pt_scene_build(m.scene, cam);
        if (rc == MI355PT_OK && hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking) != hipSuccess) rc = fail(MI355PT_E_DEVICE, "hipStreamCreate failed");
        if (rc == MI355PT_OK && hipEventCreateWithFlags(&m.done, hipEventDisableTiming) != hipSuccess) rc = fail(MI355PT_E_DEVICE, "hipEventCreate failed");
        if (rc == MI355PT_OK && i > 0 && m.device != parts[0].device) {
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, parts[0].device, m.device);
            if (can) { (void)hipSetDevice(parts[0].device); (void)hipDeviceEnablePeerAccess(m.device, 0); (void)hipGetLastError(); }   // already enabled is fine
        }
    }
    (void)hipSetDevice(cur);
    if (rc != MI355PT_OK) { std::string keep = g_err; free_parts(parts); g_err = keep; return rc; }
    s->parts = std::move(parts);
    return MI355PT_OK;
}

int mi355pt_render_multi(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p, float* out_rgb) {
    if (!s || !cam || !p || !out_rgb) return fail(MI355PT_E_INVALID, "null argument");
    if (s->parts.empty()) return fail(MI355PT_E_NOT_BUILT, "scene not built with mi355pt_scene_build_multi");
    if (p->shard_count > 1) return fail(MI355PT_E_INVALID, "mi355pt_render_multi shards the frame itself: pass shard_count 0 or 1");
    if (p->collect_stats) return fail(MI355PT_E_INVALID, "collect_stats is a single-device diagnostic");
    if (cam->width == 0 || cam->height == 0 || p->spp == 0) return fail(MI355PT_E_INVALID, "empty image or spp == 0");
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    std::vector<MultiPart>& parts = s->parts;
    const uint32_t n = (uint32_t)parts.size();
    const size_t film = (size_t)cam->width * cam->height * 3, film_pad = (film + 3) / 4 * 4;
    int rc = MI355PT_OK;
    auto hip_ok = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == MI355PT_OK) rc = fail(MI355PT_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); return e == hipSuccess; };
    // the shards: device i renders frame tiles i, i + n, i + 2 n, ...; its share of the film is tiles_of(i) * 192 floats
    const uint32_t tiles_total = ((cam->width + 7u) / 8u) * ((cam->height + 7u) / 8u);
    auto tiles_of = [&](uint32_t i) { return tiles_total > i ? (tiles_total - i + n - 1u) / n : 0u; };
    std::vector<size_t> stage_off(n, 0);
    size_t stage_total = 0;
    for (uint32_t i = 1; i < n; ++i) { stage_off[i] = stage_total; stage_total += (size_t)tiles_of(i) * 192u; }
    // 1. every device renders its tile shard into its own zeroed film, concurrently; a peer then packs its tiles and PUSHES the compact
    //    film (film / n bytes) into its own landing area on the first device, on its own stream: the n - 1 transfers overlap each other
    //    and the first device's rendering
    {
        MultiPart& r = parts[0];
        if (hip_ok(hipSetDevice(r.device), "hipSetDevice") && r.stage_floats < stage_total) {
            (void)hipFree(r.d_stage); r.d_stage = nullptr; r.stage_floats = 0;
            if (hip_ok(hipMalloc((void**)&r.d_stage, stage_total * sizeof(float)), "hipMalloc stage")) r.stage_floats = stage_total;
        }
    }
    for (uint32_t i = 0; i < n && rc == MI355PT_OK; ++i) {
        MultiPart& m = parts[i];
        if (!hip_ok(hipSetDevice(m.device), "hipSetDevice")) break;
        if (m.film_floats != film_pad) {
            (void)hipFree(m.d_film); (void)hipFree(m.d_out); m.d_film = m.d_out = nullptr; m.film_floats = 0;
            if (!hip_ok(hipMalloc((void**)&m.d_film, film_pad * sizeof(float)), "hipMalloc film")) break;
            if (i == 0 && !hip_ok(hipMalloc((void**)&m.d_out, film_pad * sizeof(float)), "hipMalloc out")) break;
            m.film_floats = film_pad;
        }
        const size_t pack = (size_t)tiles_of(i) * 192u;
        if (i > 0 && m.pack_floats < pack) {
            (void)hipFree(m.d_pack); m.d_pack = nullptr; m.pack_floats = 0;
            if (!hip_ok(hipMalloc((void**)&m.d_pack, std::max<size_t>(pack, 1) * sizeof(float)), "hipMalloc pack")) break;
            m.pack_floats = pack;
        }
        if (!hip_ok(hipMemsetAsync(m.d_film, 0, film_pad * sizeof(float), m.stream), "hipMemsetAsync")) break;
        mi355pt_params q = *p;
        q.shard_index = i; q.shard_count = n;
        if ((rc = mi355pt_render_accum_device(m.scene, cam, &q, 0, p->spp, m.d_film, (void*)m.stream, nullptr))) break;
        if (i > 0 && pack) {
            if (!hip_ok(launch_film_pack(m.d_film, cam->width, cam->height, i, n, tiles_of(i), m.d_pack, m.stream), "film pack")) break;
            if (!hip_ok(hipMemcpyPeerAsync(parts[0].d_stage + stage_off[i], parts[0].device, m.d_pack, m.device, pack * sizeof(float), m.stream), "hipMemcpyPeerAsync")) break;
        }
        hip_ok(hipEventRecord(m.done, m.stream), "hipEventRecord");
    }
    // 2. on the first device: each landed shard is written into the film (disjoint tiles: stores in a fixed order, exact and
    //    deterministic), then resolve and copy out
    if (rc == MI355PT_OK && hip_ok(hipSetDevice(parts[0].device), "hipSetDevice")) {
        MultiPart& r = parts[0];
        for (uint32_t i = 1; i < n && rc == MI355PT_OK; ++i) {
            if (!tiles_of(i)) continue;
            if (!hip_ok(hipStreamWaitEvent(r.stream, parts[i].done, 0), "hipStreamWaitEvent")) break;
            hip_ok(launch_film_unpack(r.d_film, cam->width, cam->height, i, n, tiles_of(i), r.d_stage + stage_off[i], r.stream), "film unpack");
        }
        if (rc == MI355PT_OK) rc = mi355pt_film_resolve_device(r.d_film, cam->width * cam->height, p->spp, r.d_out, (void*)r.stream);
        if (rc == MI355PT_OK) hip_ok(hipMemcpyAsync(out_rgb, r.d_out, film * sizeof(float), hipMemcpyDeviceToHost, r.stream), "hipMemcpyAsync");
        if (rc == MI355PT_OK) hip_ok(hipStreamSynchronize(r.stream), "hipStreamSynchronize");
    }
    if (rc != MI355PT_OK) for (MultiPart& m : parts) { (void)hipSetDevice(m.device); (void)hipStreamSynchronize(m.stream); }   // nothing of this call stays in flight
    (void)hipSetDevice(cur);
    return rc;
}

int mi355pt_scene_info(const mi355pt_scene* s, char* buf, size_t n) {
    if (!s || !buf || n == 0) return fail(MI355PT_E_INVALID, "null argument");
    if (!s->impl.built) return fail(MI355PT_E_INVALID, "scene not built");
    std::snprintf(buf, n, "%s features=%u", s->impl.info.c_str(), s->impl.features);
    return MI355PT_OK;
}
int mi355pt_film_resolve_device(const float* d_accum, uint32_t n_pixels, uint32_t spp, float* d_out, void* hip_stream) {
    if (!d_accum || !d_out || spp == 0) return fail(MI355PT_E_INVALID, "bad resolve arguments");
    HIP_TRY(launch_resolve(d_accum, n_pixels * 3, spp, d_out, (hipStream_t)hip_stream));
    return MI355PT_OK;
}

int mi355pt_render(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p, float* out_rgb, mi355pt_stats* stats) {
    int rc = check_args(s, cam, p);
    if (rc) return rc;
    if (!out_rgb) return fail(MI355PT_E_INVALID, "null output");
    size_t n = (size_t)cam->width * cam->height * 3;
    DevBuf<float> d_acc, d_out;
    HIP_TRY(d_acc.alloc(n));
    HIP_TRY(d_out.alloc(n));
    HIP_TRY(hipMemset(d_acc.p, 0, n * sizeof(float)));
    if ((rc = mi355pt_render_accum_device(s, cam, p, 0, p->spp, d_acc.p, nullptr, stats))) return rc;
    if ((rc = mi355pt_film_resolve_device(d_acc.p, cam->width * cam->height, p->spp, d_out.p, nullptr))) return rc;
    HIP_TRY(hipMemcpy(out_rgb, d_out.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return MI355PT_OK;
}

int mi355pt_quantize_u8(const float* rgb, size_t n, uint8_t* out) {
    if (!rgb || !out) return fail(MI355PT_E_INVALID, "null argument");
    for (size_t i = 0; i < n; ++i) {   // Rust `as u8`: saturating, NaN -> 0 (renderer.rs:141-143)
        float v = rgb[i] * 255.0f;
        out[i] = std::isnan(v) ? 0 : (v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (uint8_t)v));
    }
    return MI355PT_OK;
}

// ---------------- probes ----------------

int mi355pt_probe_sobol(uint32_t width, uint32_t height, uint32_t spp, uint32_t seed, const uint32_t* xys, uint32_t n, const char* pattern,
                        uint32_t* out_bits) {
    if (!xys || !pattern || !out_bits || spp == 0) return fail(MI355PT_E_INVALID, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MI355PT_E_NO_DEVICE, "no HIP device");
    uint32_t n_pat = (uint32_t)std::strlen(pattern), per = 0;
    for (uint32_t i = 0; i < n_pat; ++i) per += pattern[i] == '2' ? 2 : 1;
    if (n == 0 || per == 0) return MI355PT_OK;
    uint32_t log2_spp = log2_int(spp);
    uint32_t nb4 = log2_int(round_up_pow2(std::max(width, height))) + (log2_spp + 1) / 2;
    DevBuf<uint32_t> d_xys, d_out; DevBuf<uint8_t> d_pat;
    HIP_TRY(d_xys.alloc((size_t)n * 3)); HIP_TRY(d_out.alloc((size_t)n * per)); HIP_TRY(d_pat.alloc(n_pat));
    HIP_TRY(hipMemcpy(d_xys.p, xys, sizeof(uint32_t) * 3 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_pat.p, pattern, n_pat, hipMemcpyHostToDevice));
    HIP_TRY(launch_probe_sobol(width, seed, log2_spp, nb4, d_xys.p, n, d_pat.p, n_pat, per, d_out.p, nullptr));
    HIP_TRY(hipMemcpy(out_bits, d_out.p, sizeof(uint32_t) * (size_t)n * per, hipMemcpyDeviceToHost));
    return MI355PT_OK;
}

// Host-only: sweep-SAH BVH2 over n triangles + the BVH4 collapse, both walked on the CPU for n_rays rays (closest hit over the triangle
// boxes' entry distances is not the point — the point is that both trees return the SAME set of leaves for every ray, i.e. the collapse
// loses nothing, and that the collapsed tree's worst-case stack need stays inside STACK_DEPTH).  No device involved.
int mi355pt_probe_bvh_collapse_nodes(const void* bvh2_nodes, uint32_t n_nodes, int32_t root, uint32_t n_tris, uint32_t* out_info) {
    if (!bvh2_nodes || !out_info || n_nodes == 0) return fail(MI355PT_E_INVALID, "bad argument");
    // the same checks SceneImpl::build makes before it uploads a tree: links in range (collapse_bvh4 indexes with them), then the collapse
    // with its validation of the result (every triangle in one leaf, no cycle, worst-case per-lane stack need < STACK_DEPTH)
    std::vector<DevNode> n2(n_nodes);
    std::memcpy(n2.data(), bvh2_nodes, sizeof(DevNode) * n_nodes);
    if (root >= 0 && (uint32_t)root >= n_nodes) return fail(MI355PT_E_INVALID, "root out of range");
    for (const DevNode& n : n2) for (int c = 0; c < 2; ++c) if (n.child[c] >= 0 && (uint32_t)n.child[c] >= n_nodes) return fail(MI355PT_E_INVALID, "child link out of range");
    {   // a cycle or a shared child would make the height computation run forever: every node may be reached once
        std::vector<uint8_t> seen(n_nodes, 0); std::vector<int32_t> st; if (root >= 0) st.push_back(root);
        while (!st.empty()) { const int32_t v = st.back(); st.pop_back(); if (seen[(size_t)v]++) return fail(MI355PT_E_INVALID, "BVH2 is not a tree"); for (int c = 0; c < 2; ++c) if (n2[(size_t)v].child[c] >= 0) st.push_back(n2[(size_t)v].child[c]); }
    }
    std::vector<DevNode4> n4; int32_t root4 = 0; int max_stack = 0; std::string err; const char* method = "";
    if (!collapse_bvh4(n2, root, n_tris, &n4, &root4, &max_stack, &err, &method)) return fail(MI355PT_E_INVALID, err);
    out_info[0] = n_nodes; out_info[1] = (uint32_t)n4.size(); out_info[2] = (uint32_t)(method[0] == 'd' ? 1 : 0); out_info[3] = (uint32_t)max_stack;
    return MI355PT_OK;
}

int mi355pt_probe_bvh_collapse(const float* tri_pos, uint32_t n_tris, const float* rays_od, uint32_t n_rays, uint32_t* out_info, uint32_t* out_mismatch) {
    if (!tri_pos || !out_info || n_tris == 0) return fail(MI355PT_E_INVALID, "bad argument");
    std::vector<BuildTri> bt(n_tris);
    for (uint32_t i = 0; i < n_tris; ++i) for (int a = 0; a < 3; ++a) {
        const float v0 = tri_pos[9 * i + a], v1 = tri_pos[9 * i + 3 + a], v2 = tri_pos[9 * i + 6 + a];
        bt[i].lo[a] = std::fmin(v0, std::fmin(v1, v2)); bt[i].hi[a] = std::fmax(v0, std::fmax(v1, v2)); bt[i].c[a] = 0.5f * (bt[i].lo[a] + bt[i].hi[a]);
    }
    BvhOut bvh; build_bvh(bt, &bvh);
    std::vector<DevNode4> n4; int32_t root4 = 0; int max_stack = 0; std::string err;
    if (!collapse_bvh4(bvh.nodes, bvh.root, bvh.order.size(), &n4, &root4, &max_stack, &err)) return fail(MI355PT_E_INVALID, err);
    out_info[0] = (uint32_t)bvh.nodes.size(); out_info[1] = (uint32_t)n4.size(); out_info[2] = (uint32_t)bvh.max_depth; out_info[3] = (uint32_t)max_stack;
    uint32_t mism = 0;
    auto slab = [](const float lo[3], const float hi[3], const float* o, const float* inv) {
        float tn = 0.0f, tf = 3.0e38f;
        for (int a = 0; a < 3; ++a) { float l = (lo[a] - o[a]) * inv[a], h = (hi[a] - o[a]) * inv[a]; tn = std::fmax(tn, std::fmin(l, h)); tf = std::fmin(tf, std::fmax(l, h)); }
        return tn <= tf;
    };
    for (uint32_t r = 0; rays_od && r < n_rays; ++r) {
        const float* o = rays_od + 6 * r; const float* d = o + 3;
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        std::vector<int32_t> leaves2, leaves4, st;
        st.push_back(bvh.root);
        while (!st.empty()) {
            int32_t c = st.back(); st.pop_back();
            if (c < 0) { leaves2.push_back(c); continue; }
            const DevNode& n = bvh.nodes[(size_t)c];
            for (int k = 0; k < 2; ++k) { float lo[3] = {n.bx[k], n.by[k], n.bz[k]}, hi[3] = {n.bx[2 + k], n.by[2 + k], n.bz[2 + k]}; if (slab(lo, hi, o, inv)) st.push_back(n.child[k]); }
        }
        st.push_back(root4);
        size_t deepest = 0;
        while (!st.empty()) {
            deepest = std::max(deepest, st.size());
            int32_t c = st.back(); st.pop_back();
            if (c < 0) { leaves4.push_back(c); continue; }
            const DevNode4& n = n4[(size_t)c];
            for (int k = 0; k < 4; ++k) { float lo[3] = {n.lox[k], n.loy[k], n.loz[k]}, hi[3] = {n.hix[k], n.hiy[k], n.hiz[k]}; if (slab(lo, hi, o, inv)) st.push_back(n.child[k]); }
        }
        std::sort(leaves2.begin(), leaves2.end()); std::sort(leaves4.begin(), leaves4.end());
        if (leaves2 != leaves4) ++mism;
    }
    if (out_mismatch) *out_mismatch = mism;
    return MI355PT_OK;
}

int mi355pt_scene_export_bvh(const mi355pt_scene* s, void* out_nodes, uint32_t* n_nodes, void* out_tris, uint32_t* n_tris, int32_t* root) {
    if (!s || !n_nodes || !n_tris) return fail(MI355PT_E_INVALID, "null argument");
    if (!s->impl.built) return fail(MI355PT_E_NOT_BUILT, "scene not built");
    const DevScene& d = s->impl.dev;
    if (out_nodes) { if (*n_nodes < d.n_nodes) return fail(MI355PT_E_INVALID, "node buffer too small"); if (d.n_nodes) HIP_TRY(hipMemcpy(out_nodes, d.nodes, sizeof(DevNode) * d.n_nodes, hipMemcpyDeviceToHost)); }
    if (out_tris) { if (*n_tris < d.n_tris) return fail(MI355PT_E_INVALID, "triangle buffer too small"); HIP_TRY(hipMemcpy(out_tris, d.tris_render, sizeof(DevTri) * d.n_tris, hipMemcpyDeviceToHost)); }
    *n_nodes = d.n_nodes; *n_tris = d.n_tris;
    if (root) *root = d.root;
    return MI355PT_OK;
}

int mi355pt_probe_intersect(const mi355pt_scene* s, const float* o, const float* d, uint32_t n, float* out_t, uint32_t* out_inst, uint32_t* out_tri,
                            float* out_n) {
    if (!s || !o || !d || !out_t || !out_inst || !out_tri) return fail(MI355PT_E_INVALID, "null argument");
    if (!s->impl.built) return fail(MI355PT_E_NOT_BUILT, "scene not built");
    if (n == 0) return MI355PT_OK;
    DevBuf<float> d_o, d_d, d_t, d_n; DevBuf<uint32_t> d_i, d_tr;
    HIP_TRY(d_o.alloc((size_t)n * 3)); HIP_TRY(d_d.alloc((size_t)n * 3)); HIP_TRY(d_t.alloc(n)); HIP_TRY(d_n.alloc((size_t)n * 3));
    HIP_TRY(d_i.alloc(n)); HIP_TRY(d_tr.alloc(n));
    HIP_TRY(hipMemcpy(d_o.p, o, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d.p, d, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    HIP_TRY(launch_probe_intersect(s->impl.dev, d_o.p, d_d.p, n, d_t.p, d_i.p, d_tr.p, d_n.p, nullptr));
    HIP_TRY(hipMemcpy(out_t, d_t.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_inst, d_i.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_tri, d_tr.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    if (out_n) HIP_TRY(hipMemcpy(out_n, d_n.p, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
    return MI355PT_OK;
}

int mi355pt_probe_occluded(const mi355pt_scene* s, const float* o, const float* d, const float* tmax, uint32_t n, uint8_t* out) {
    if (!s || !o || !d || !tmax || !out) return fail(MI355PT_E_INVALID, "null argument");
    if (!s->impl.built) return fail(MI355PT_E_NOT_BUILT, "scene not built");
    if (n == 0) return MI355PT_OK;
    DevBuf<float> d_o, d_d, d_t; DevBuf<uint8_t> d_out;
    HIP_TRY(d_o.alloc((size_t)n * 3)); HIP_TRY(d_d.alloc((size_t)n * 3)); HIP_TRY(d_t.alloc(n)); HIP_TRY(d_out.alloc(n));
    HIP_TRY(hipMemcpy(d_o.p, o, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d.p, d, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_t.p, tmax, sizeof(float) * n, hipMemcpyHostToDevice));
    HIP_TRY(launch_probe_occluded(s->impl.dev, d_o.p, d_d.p, d_t.p, n, d_out.p, nullptr));
    HIP_TRY(hipMemcpy(out, d_out.p, n, hipMemcpyDeviceToHost));
    return MI355PT_OK;
}

int mi355pt_probe_sincos(uint32_t first_bits, uint32_t stride, uint32_t n, uint64_t* out_counts) {
    if (!out_counts || stride == 0) return fail(MI355PT_E_INVALID, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MI355PT_E_NO_DEVICE, "no HIP device");
    out_counts[0] = out_counts[1] = out_counts[2] = 0;
    if (n == 0) return MI355PT_OK;
    if ((uint64_t)first_bits + (uint64_t)(n - 1) * stride > 0xffffffffull) return fail(MI355PT_E_INVALID, "bit patterns wrap");
    DevBuf<float> d_s, d_c;
    HIP_TRY(d_s.alloc(n)); HIP_TRY(d_c.alloc(n));
    HIP_TRY(launch_probe_sincos(first_bits, stride, n, d_s.p, d_c.p, nullptr));
    std::vector<float> hs(n), hc(n);
    HIP_TRY(hipMemcpy(hs.data(), d_s.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hc.data(), d_c.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t b = first_bits + i * stride;
        float x; std::memcpy(&x, &b, 4);
        const float ls = sinf(x), lc = cosf(x);                                   // the libm of this host: what f32::sin / f32::cos call
        out_counts[0]++;
        out_counts[1] += std::memcmp(&ls, &hs[i], 4) != 0 && !(std::isnan(ls) && std::isnan(hs[i]));
        out_counts[2] += std::memcmp(&lc, &hc[i], 4) != 0 && !(std::isnan(lc) && std::isnan(hc[i]));
    }
    return MI355PT_OK;
}

int mi355pt_probe_radiance(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p, const uint32_t* xys, uint32_t n, float* out_L,
                           float* out_lambda, float* out_pdf) {
    int rc = check_args(s, cam, p);
    if (rc) return rc;
    if (!xys || !out_L || !out_lambda || !out_pdf) return fail(MI355PT_E_INVALID, "null argument");
    if (n == 0) return MI355PT_OK;
    for (uint32_t i = 0; i < n; ++i)
        if (xys[3 * i] >= cam->width || xys[3 * i + 1] >= cam->height || xys[3 * i + 2] >= p->spp) return fail(MI355PT_E_INVALID, "query outside the frame or the sample range");
    // the whole frame, every sample index, in the launch shape mi355pt_render takes for this job — then pick the queried records
    mi355pt_params q = *p;
    q.shard_index = 0; q.shard_count = 1; q.collect_stats = 0;
    const size_t recs = (size_t)shard_tiles(cam, &q) * 64u * p->spp;
    if (recs > ((size_t)1 << 26)) return fail(MI355PT_E_INVALID, "frame x spp too large for mi355pt_probe_radiance: use mi355pt_render_sample_log on a sparse shard");
    std::vector<float> L(recs * 4), lam(recs * 4), pdf(recs * 4);
    if ((rc = mi355pt_render_sample_log(s, cam, &q, 0, p->spp, L.data(), lam.data(), pdf.data(), recs, nullptr))) return rc;
    const uint32_t tiles_x = (cam->width + 7) / 8;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t x = xys[3 * i], y = xys[3 * i + 1], k = xys[3 * i + 2];
        const size_t slot = ((size_t)((y / 8) * tiles_x + x / 8) * 64u + ((y & 7u) * 8u + (x & 7u))) * p->spp + k;
        std::memcpy(out_L + 4 * (size_t)i, &L[4 * slot], 16); std::memcpy(out_lambda + 4 * (size_t)i, &lam[4 * slot], 16);
        std::memcpy(out_pdf + 4 * (size_t)i, &pdf[4 * slot], 16);
    }
    return MI355PT_OK;
}

}  // extern "C"
