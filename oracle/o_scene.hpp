// ORACLE — TEST INFRASTRUCTURE ONLY.
// Restatement of scene/src/{scene,samples,light_sampler}.rs, scene/src/primitive/{bvh,impls/
// triangle_mesh,impls/emissive_triangle_mesh}.rs and scene/src/geometry/impls/triangle_mesh.rs.
#pragma once
#include <algorithm>
#include <memory>
#include <vector>
#include "o_bvh.hpp"
#include "o_spectrum.hpp"

namespace oracle {

// ---------------- parameters / materials (material/parameter.rs) ----------------
enum SpectrumParamKind : uint32_t { SP_CONSTANT = 0, SP_TEXTURE_ALBEDO_SRGB = 1, SP_TEXTURE_ILLUMINANT_SRGB = 2, SP_TEXTURE_UNBOUNDED_SRGB = 3 };
struct SpectrumParameter {
    uint32_t kind = SP_CONSTANT;
    Spectrum constant;
    int texture = -1;
    const float* illuminant = nullptr;     // SP_TEXTURE_ILLUMINANT_SRGB: presets::cie_illum_d6500() (470 entries)
};

enum MaterialType : uint32_t { MAT_LAMBERT = 0, MAT_EMISSIVE = 1, MAT_GLASS = 2, MAT_PLASTIC = 3, MAT_CLEARCOAT = 4, MAT_METAL = 5 };

struct Material {
    uint32_t type = MAT_LAMBERT;
    SpectrumParameter color;       // Lambert albedo / Emissive radiance / Plastic colour / Clearcoat base colour
    int normal_tex = -1;           // NormalParameter::Texture
    bool normal_flip_y = false;
    float intensity = 1.0f;        // Emissive FloatParameter::Constant
    Spectrum eta;                  // Glass: LUT spectrum; Plastic: constant; Metal: real part of the index
    Spectrum k;                    // Metal: extinction coefficient (presets::au_k() ...)
    int metallic_tex = -1, roughness_tex = -1, cc_thickness_tex = -1;   // FloatParameter::Texture (grey image in the red channel), -1 = constant
    int intensity_tex = -1;        // emissive: FloatParameter::Texture intensity (emissive_material.rs:55-56,69-76)
    bool thin = false;
    float roughness = 0.0f;
    // clearcoat (simple_pbr_clearcoat_material.rs): filled by the API when type == MAT_CLEARCOAT
    float cc_metallic = 0, cc_base_ior = 1.5f, cc_ior = 1.5f, cc_roughness = 0, cc_thickness = 0;
    SpectrumParameter cc_tint;
    std::vector<float> cc_albedo_lut;   // 64 entries: the coat's E(cos theta) table (Scene::cc_draws == 2), handed over by the test
    bool is_emissive() const { return type == MAT_EMISSIVE; }
};

// ---------------- geometry (geometry/impls/triangle_mesh.rs) ----------------
struct GeomHit {            // geometry::Intersection
    V3 position, normal, shading_normal, tangent; V2 uv; uint32_t index; float t_hit;
};

struct TriangleMesh {
    std::vector<V3> positions, normals, tangents;   // tangents: one per triangle (or empty)
    std::vector<V2> uvs;                            // per vertex (or empty)
    std::vector<uint32_t> indices;
    Bounds bounds;
    Bvh bvh;

    void tri_positions(uint32_t t, V3 ps[3]) const {
        ps[0] = positions[indices[t * 3]]; ps[1] = positions[indices[t * 3 + 1]]; ps[2] = positions[indices[t * 3 + 2]];
    }
    Bounds tri_bounds(uint32_t t) const {             // :31-39
        V3 ps[3]; tri_positions(t, ps);
        return Bounds{vmin(vmin(ps[0], ps[1]), ps[2]), vmax(vmax(ps[0], ps[1]), ps[2])};
    }
    void build() {                                    // :244-251
        bounds = Bounds{positions[0], positions[0]};
        for (auto& p : positions) bounds = Bounds{vmin(bounds.mn, p), vmax(bounds.mx, p)};
        uint32_t nt = (uint32_t)(indices.size() / 3);
        bvh = Bvh::build(nt, [this](uint32_t t) { return tri_bounds(t); });
    }
    bool tri_intersect(uint32_t t, const Ray& ray, float t_max, float* t_out, GeomHit* out) const {   // :41-109
        V3 ps[3]; tri_positions(t, ps);
        TriHit h;
        if (!intersect_triangle(ray, t_max, ps, &h)) return false;
        V3 n0 = normals[indices[t * 3]], n1 = normals[indices[t * 3 + 1]], n2 = normals[indices[t * 3 + 2]];
        V3 sn = normalize(n0 * h.b[0] + n1 * h.b[1] + n2 * h.b[2]);           // normal.rs:68-78
        V2 uv{0, 0};
        if (!uvs.empty()) {
            V2 u0 = uvs[indices[t * 3]], u1 = uvs[indices[t * 3 + 1]], u2 = uvs[indices[t * 3 + 2]];
            uv = V2{u0.x * h.b[0] + u1.x * h.b[1] + u2.x * h.b[2], u0.y * h.b[0] + u1.y * h.b[1] + u2.y * h.b[2]};
        }
        V3 tangent = tangents.empty() ? generate_tangent(sn) : orthogonalize_vector(sn, tangents[t]);
        *t_out = h.t;
        *out = GeomHit{h.p, h.n, sn, tangent, uv, t, h.t};
        return true;
    }
    bool tri_intersect_p(uint32_t t, const Ray& ray, float t_max) const {     // :111-126
        V3 ps[3]; tri_positions(t, ps);
        return intersect_triangle(ray, t_max, ps, nullptr);
    }
};

// ---------------- SurfaceInteraction / Intersection (samples.rs, primitive/bvh.rs) ----------------
struct SurfaceInteraction {
    V3 position, normal, shading_normal, tangent; V2 uv; int material;
};
static inline SurfaceInteraction transform_interaction(const M4& m, const SurfaceInteraction& s) {   // samples.rs:130-143
    return SurfaceInteraction{transform_point3(m, s.position), transform_normal(m, s.normal),
                              transform_normal(m, s.shading_normal), transform_vector3(m, s.tangent), s.uv, s.material};
}
struct Intersection {
    float t_hit; V3 wo; int primitive; uint32_t triangle; SurfaceInteraction interaction;
};

// ---------------- primitives (primitive/impls/{triangle_mesh,emissive_triangle_mesh}.rs) ----------------
struct Primitive {
    int geometry = -1;
    int material = -1;
    M4 local_to_world = M4::identity();
    M4 local_to_render = M4::identity();
    M4 render_to_local = M4::identity();   // used only by fast mode (hoisted inverse)
    // DIAGNOSTIC lowering (Scene::render_space_lowering): this primitive's mesh with positions carried to render space once, the way the
    // product lowers instances (csrc/scene.cpp), so that the two lowerings can be told apart in parity tests
    std::shared_ptr<TriangleMesh> rs_mesh;
    bool is_light = false;
    uint32_t seq = 0;                      // creation order among all primitives
    std::vector<float> area_list, area_table;
    float area_sum = 0.0f;
};

// PointLight / SpotLight / DirectionalLight primitives (primitive/impls/{point,spot,directional}_light.rs)
enum DeltaKind : uint32_t { DL_POINT = 1, DL_SPOT = 2, DL_DIRECTIONAL = 3 };
struct DeltaLight {
    uint32_t kind = DL_POINT;
    float intensity = 1.0f, angle_inner = 0.0f, angle_outer = 0.0f;
    Spectrum spectrum;
    M4 local_to_world = M4::identity(), local_to_render = M4::identity();
    float area = 0.0f;            // DirectionalLight::preprocess (directional_light.rs:46-54)
    uint32_t seq = 0;             // creation order among all primitives (light_list order, light_sampler.rs:163-180)
};

// EnvironmentLight (primitive/impls/environment_light.rs): lat-long float RGB texture, luminance * sin(theta) 2-D CDF
struct EnvLight {
    float intensity = 1.0f;
    uint32_t w = 0, h = 0;
    std::vector<float> rgb;                         // h * w * 3, row 0 = theta 0 (+y)
    std::vector<float> marginal, conditional;       // h, h * w
    float total_weight = 0.0f;
    Spectrum integrated;                            // RgbIlluminantSpectrum of the mean colour (:44-61)
    const float* illuminant = nullptr;              // presets::cie_illum_d6500()
    M4 local_to_world = M4::identity(), local_to_render = M4::identity();
    uint32_t seq = 0;

    static Spectrum rgb_illuminant(const Rgb2SpecTable& table, const float rgb[3], const float* illum) {   // rgb_illuminant_spectrum.rs:26-41
        Spectrum s; s.kind = SPEC_RGB_ILLUMINANT; s.lut = illum;
        float mx = std::fmax(rgb[0], std::fmax(rgb[1], rgb[2]));
        s.scale = 2.0f * mx;
        if (s.scale == 0.0f) { s.kind = SPEC_CONSTANT; s.c[0] = 0.0f; return s; }   // black texel: 0 instead of the reference's 0/0
        float scaled[3] = {rgb[0] / s.scale, rgb[1] / s.scale, rgb[2] / s.scale};
        table.get_srgb_encoded(scaled, s.c);
        return s;
    }
    void build(const Rgb2SpecTable& table) {                                                      // :28-75,153-199
        float tot[3] = {0, 0, 0};
        for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) for (int c = 0; c < 3; ++c) tot[c] += rgb[((size_t)y * w + x) * 3 + c];
        float n = (float)(w * h);
        for (int c = 0; c < 3; ++c) tot[c] /= n;
        integrated = rgb_illuminant(table, tot, illuminant);
        std::vector<float> row_w(h, 0.0f);
        conditional.assign((size_t)w * h, 0.0f); marginal.assign(h, 0.0f);
        for (uint32_t y = 0; y < h; ++y) {
            float row_sum = 0.0f;
            for (uint32_t x = 0; x < w; ++x) {
                float v = ((float)y + 0.5f) / (float)h;
                float theta = v * PI_F;
                const float* p = &rgb[((size_t)y * w + x) * 3];
                float lum = 0.299f * p[0] + 0.587f * p[1] + 0.114f * p[2];
                float wgt = lum * std::fmax(std::sin(theta), 1e-8f);
                row_sum += wgt;
                conditional[(size_t)y * w + x] = row_sum;
            }
            row_w[y] = row_sum;
            if (row_sum > 0.0f) for (uint32_t x = 0; x < w; ++x) conditional[(size_t)y * w + x] /= row_sum;
        }
        total_weight = 0.0f;
        for (float r : row_w) total_weight += r;
        float cum = 0.0f;
        for (uint32_t y = 0; y < h; ++y) { cum += row_w[y]; marginal[y] = total_weight > 0.0f ? cum / total_weight : (float)(y + 1) / (float)h; }
    }
    static uint32_t sample_cdf(const float* cdf, uint32_t n, float u) {          // binary_search_by: insertion point, clamped (:201-206)
        uint32_t lo = 0, hi = n;
        while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (cdf[mid] < u) lo = mid + 1; else hi = mid; }
        return std::min(lo, n - 1);
    }
    static void dir_to_spherical(V3 d, float* theta, float* phi) {               // :96-103
        *theta = std::fmin(std::fmax(std::acos(d.y), 0.0f), PI_F);
        float p = std::atan2(d.z, d.x);
        if (p < 0.0f) p += 2.0f * PI_F;
        *phi = p;
    }
    void texel(uint32_t x, uint32_t y, float out[3]) const { const float* p = &rgb[((size_t)y * w + x) * 3]; out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; }
    void sample_texture(float u, float v, float out[3]) const {                   // :113-151
        u = std::fmin(std::fmax(u, 0.0f), 1.0f); v = std::fmin(std::fmax(v, 0.0f), 1.0f);
        float x = u * (float)(w - 1), y = v * (float)(h - 1);
        uint32_t x0 = (uint32_t)std::floor(x), y0 = (uint32_t)std::floor(y);
        uint32_t x1 = std::min(x0 + 1, w - 1), y1 = std::min(y0 + 1, h - 1);
        float fx = x - (float)x0, fy = y - (float)y0;
        float p00[3], p01[3], p10[3], p11[3];
        texel(x0, y0, p00); texel(x1, y0, p01); texel(x0, y1, p10); texel(x1, y1, p11);
        for (int c = 0; c < 3; ++c) {
            float p0 = p00[c] * (1.0f - fx) + p01[c] * fx, p1 = p10[c] * (1.0f - fx) + p11[c] * fx;
            out[c] = p0 * (1.0f - fy) + p1 * fy;
        }
    }
    float direction_pdf(V3 dir_render) const {                                   // :212-238
        if (total_weight <= 0.0f) return 0.0f;
        V3 dl = transform_vector3(inverse(local_to_render), dir_render);
        float theta, phi; dir_to_spherical(dl, &theta, &phi);
        float u = phi / (2.0f * PI_F), v = theta / PI_F;
        uint32_t x = std::min((uint32_t)std::floor(u * (float)w), w - 1), y = std::min((uint32_t)std::floor(v * (float)h), h - 1);
        float p[3]; texel(x, y, p);
        float lum = 0.299f * p[0] + 0.587f * p[1] + 0.114f * p[2];
        float st = std::fmax(std::sin(theta), 1e-8f);
        float pdf_tex = (lum * st) / total_weight;
        float jac = (float)w * (float)h / (2.0f * PI_F * PI_F * st);
        return pdf_tex * jac;
    }
    SS direction_radiance(const Rgb2SpecTable& table, V3 dir_render, const Wavelengths& wl) const {   // :292-305
        V3 dl = transform_vector3(inverse(local_to_render), dir_render);
        float theta, phi; dir_to_spherical(dl, &theta, &phi);
        float rgbv[3];
        sample_texture(phi / (2.0f * PI_F), theta / PI_F, rgbv);
        return rgb_illuminant(table, rgbv, illuminant).sample(wl) * intensity;
    }
    void sample(float ux, float uy, V3* wi, float* pdf_dir) const {              // :317-340
        uint32_t y = sample_cdf(marginal.data(), h, ux);
        uint32_t x = sample_cdf(&conditional[(size_t)y * w], w, uy);
        float u = ((float)x + 0.5f) / (float)w, v = ((float)y + 0.5f) / (float)h;
        float theta = v * PI_F, phi = u * 2.0f * PI_F;
        V3 wl{std::sin(theta) * std::cos(phi), std::cos(theta), std::sin(theta) * std::sin(phi)};
        *wi = transform_vector3(local_to_render, wl);
        *pdf_dir = direction_pdf(*wi);
    }
};

// The PRODUCT's flat BVH2 (csrc/layout.hpp DevNode / DevTri, handed over by mi355pt_scene_export_bvh) walked in the plain order of
// csrc/pt_device.hpp trace_closest / trace_any — near child first, pruned by the best distance, a per-ray stack; any-hit: child 0 first,
// first hit ends the walk — ONLY to count node and triangle steps for the rays the oracle traces.  It checks the instrumented kernel's
// self-counted work (SURVEY 8d "counts must agree within 2 %") with an independent implementation; hits still come from the oracle's
// own two-level BVH.
struct FlatBvh {
    struct Node { float bx[4], by[4], bz[4]; int32_t child[2]; uint32_t pad[2]; };      // 64 B, = DevNode
    struct Tri { float v[9]; uint32_t pad[3]; };                                         // 48 B, = DevTri (p0 p1 p2 packed)
    std::vector<Node> nodes; std::vector<Tri> tris; int32_t root = 0;
    bool valid() const { return !tris.empty(); }
    static uint32_t leaf_first(int32_t c) { return ((uint32_t)c & 0x7fffffffu) >> 3; }
    static uint32_t leaf_count(int32_t c) { return ((uint32_t)c & 7u) + 1; }
    static bool slab(const Node& n, int i, const Ray& r, V3 inv, float t_max, float* t_near) {
        float lx = (n.bx[i] - r.o.x) * inv.x, hx = (n.bx[2 + i] - r.o.x) * inv.x;
        float ly = (n.by[i] - r.o.y) * inv.y, hy = (n.by[2 + i] - r.o.y) * inv.y;
        float lz = (n.bz[i] - r.o.z) * inv.z, hz = (n.bz[2 + i] - r.o.z) * inv.z;
        float tn = std::fmax(std::fmax(std::fmin(lx, hx), std::fmin(ly, hy)), std::fmax(std::fmin(lz, hz), 0.0f));
        float tf = std::fmin(std::fmin(std::fmax(lx, hx), std::fmax(ly, hy)), std::fmin(std::fmax(lz, hz), t_max));
        *t_near = tn;
        return tn <= tf;
    }
    void walk(const Ray& r, float t_max, bool any_hit, TraversalCounters* c) const {
        V3 inv{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
        float t_best = t_max;
        int32_t stack[64]; int sp = 0;
        int32_t cur = root;
        for (;;) {
            if (cur >= 0) {
                const Node& n = nodes[(size_t)cur];
                c->nodes++;
                float n0, n1;
                bool h0 = slab(n, 0, r, inv, t_best, &n0), h1 = slab(n, 1, r, inv, t_best, &n1);
                if (h0 && h1) {
                    bool first0 = any_hit ? true : n0 <= n1;
                    stack[sp++] = first0 ? n.child[1] : n.child[0];
                    cur = first0 ? n.child[0] : n.child[1];
                    continue;
                } else if (h0) { cur = n.child[0]; continue; }
                else if (h1) { cur = n.child[1]; continue; }
            } else {
                uint32_t first = leaf_first(cur), cnt = leaf_count(cur);
                for (uint32_t i = 0; i < cnt; ++i) {
                    const float* v = tris[first + i].v;
                    V3 ps[3] = {V3{v[0], v[1], v[2]}, V3{v[3], v[4], v[5]}, V3{v[6], v[7], v[8]}};
                    TriHit h;
                    c->items++;
                    if (intersect_triangle(r, t_best, ps, &h)) {
                        if (any_hit) return;
                        if (h.t < t_best) t_best = h.t;
                    }
                }
            }
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
};

struct Counters {
    TraversalCounters closest_tlas, closest_blas, any_tlas, any_blas;
    TraversalCounters flat_closest, flat_any;     // steps through the product's exported tree (FlatBvh)
    uint64_t closest_rays = 0, shadow_rays = 0, closest_hits = 0, bounces = 0, samples = 0, sampler_draws = 0;
    uint64_t spectrum_evals = 0, textured_lookups = 0;
    void add(const Counters& o) {
        auto a = [](TraversalCounters& x, const TraversalCounters& y) { x.nodes += y.nodes; x.items += y.items; };
        a(closest_tlas, o.closest_tlas); a(closest_blas, o.closest_blas); a(any_tlas, o.any_tlas); a(any_blas, o.any_blas);
        a(flat_closest, o.flat_closest); a(flat_any, o.flat_any);
        closest_rays += o.closest_rays; shadow_rays += o.shadow_rays; closest_hits += o.closest_hits;
        bounces += o.bounces; samples += o.samples; sampler_draws += o.sampler_draws;
        spectrum_evals += o.spectrum_evals; textured_lookups += o.textured_lookups;
    }
};

struct Scene {
    Rgb2SpecTable table;
    std::vector<std::vector<float>> luts;
    std::vector<TextureRgb8> textures;
    std::vector<std::unique_ptr<TriangleMesh>> geometries;
    std::vector<Material> materials;
    std::vector<Primitive> primitives;
    std::vector<DeltaLight> delta_lights;
    std::vector<EnvLight> env_lights;             // light_list entries <= ENV_BASE - k
    uint32_t next_seq = 0;
    std::vector<int> light_list;                  // LightSamplerFactory::light_list: >= 0 primitive index, < 0 delta light -1-k
    Bvh tlas;
    FlatBvh flat;                                 // the product's tree, for step counting only (empty unless handed over)
    bool faithful = true;                         // see o_bvh.hpp
    // DIAGNOSTIC (tests only; never the CPU baseline): intersect pre-transformed render-space triangles with the untransformed ray instead
    // of transforming the ray into each primitive's local space (primitive/impls/triangle_mesh.rs:89-119).  Mathematically the same hit;
    // numerically it rounds like the product's flat render-space BVH, which lets a test attribute GPU-vs-oracle path flips to the lowering.
    bool render_space_lowering = false;
    int cc_draws = 0;                             // clearcoat coat-weight mode, see Clearcoat::coat_weight (o_materials.hpp)
    bool built = false;

    // EmissiveTriangleMesh::new (emissive_triangle_mesh.rs:28-68)
    void init_light(Primitive& p) {
        const TriangleMesh& m = *geometries[p.geometry];
        p.area_list.clear(); p.area_table.clear();
        for (size_t t = 0; t < m.indices.size() / 3; ++t) {
            V3 p0 = transform_point3(p.local_to_world, m.positions[m.indices[t * 3]]);
            V3 p1 = transform_point3(p.local_to_world, m.positions[m.indices[t * 3 + 1]]);
            V3 p2 = transform_point3(p.local_to_world, m.positions[m.indices[t * 3 + 2]]);
            V3 e0 = p0 - p1, e1 = p0 - p2;                       // p1.vector_to(p0), p2.vector_to(p0)
            p.area_list.push_back(length(cross(e0, e1)) * 0.5f);
        }
        float s = 0.0f;
        for (float a : p.area_list) { s += a; p.area_table.push_back(s); }
        p.area_sum = s;
        for (float& a : p.area_table) a /= s;
    }

    // Scene::build (scene.rs:64-76): Render = World - camera position
    void build(V3 cam_pos) {
        M4 world_to_render = M4::from_translation(-cam_pos);     // camera.rs:84-86
        light_list.clear();
        std::vector<std::pair<uint32_t, int>> order;     // (creation sequence, light_list entry)
        for (size_t i = 0; i < primitives.size(); ++i) {
            Primitive& p = primitives[i];
            p.local_to_render = world_to_render * p.local_to_world;
            p.render_to_local = inverse(p.local_to_render);
            p.is_light = materials[p.material].is_emissive();
            if (p.is_light) { init_light(p); order.push_back({p.seq, (int)i}); }
        }
        for (auto& g : geometries) if (g->bvh.nodes.empty()) g->build();
        for (auto& p : primitives) {
            p.rs_mesh.reset();
            if (!render_space_lowering) continue;
            const TriangleMesh& g = *geometries[p.geometry];
            p.rs_mesh = std::make_shared<TriangleMesh>();
            p.rs_mesh->normals = g.normals; p.rs_mesh->tangents = g.tangents; p.rs_mesh->uvs = g.uvs; p.rs_mesh->indices = g.indices;
            for (const V3& q : g.positions) p.rs_mesh->positions.push_back(transform_point3(p.local_to_render, q));
            p.rs_mesh->build();
        }
        tlas = Bvh::build((uint32_t)primitives.size(), [this](uint32_t i) {
            return transform_bounds(primitives[i].local_to_render, geometries[primitives[i].geometry]->bounds);
        });
        // scene bounds for DirectionalLight::preprocess: union of the primitives' render-space bounds (scene.rs:70-73)
        Bounds sb = transform_bounds(primitives[0].local_to_render, geometries[primitives[0].geometry]->bounds);
        for (size_t i = 1; i < primitives.size(); ++i) {
            Bounds b = transform_bounds(primitives[i].local_to_render, geometries[primitives[i].geometry]->bounds);
            sb.mn = vmin(sb.mn, b.mn); sb.mx = vmax(sb.mx, b.mx);
        }
        V3 center = (sb.mn + sb.mx) * 0.5f;                       // bounds.rs:59-77
        float radius = length(center - sb.mx);
        for (size_t k = 0; k < delta_lights.size(); ++k) {
            DeltaLight& d = delta_lights[k];
            d.local_to_render = world_to_render * d.local_to_world;
            d.area = PI_F * radius * radius;
            order.push_back({d.seq, -1 - (int)k});
        }
        for (size_t k = 0; k < env_lights.size(); ++k) {
            EnvLight& e = env_lights[k];
            e.local_to_render = world_to_render * e.local_to_world;
            e.build(table);
            order.push_back({e.seq, ENV_BASE - (int)k});
        }
        std::sort(order.begin(), order.end());
        for (auto& o : order) light_list.push_back(o.second);
        built = true;
    }

    // primitive/impls/triangle_mesh.rs:89-119 — ray to local (Mat4 inverse per call), hit back to render.
    bool primitive_intersect(uint32_t pi, const Ray& ray, float t_max, float* t_out, Intersection* out, Counters* c) const {
        const Primitive& p = primitives[pi];
        if (p.rs_mesh) {                                         // diagnostic lowering: render-space triangles, the ray as it is
            const TriangleMesh& g = *p.rs_mesh;
            GeomHit gh;
            auto item = [&](uint32_t t, const Ray& r, float tm, float* to, GeomHit* ho) { return g.tri_intersect(t, r, tm, to, ho); };
            if (!g.bvh.intersect<GeomHit>(ray, t_max, item, !faithful, &gh, c ? &c->closest_blas : nullptr)) return false;
            out->t_hit = gh.t_hit; out->wo = -ray.d; out->primitive = (int)pi; out->triangle = gh.index;
            // positions and the geometric normal are already in render space; the shading normal / tangent come from LOCAL vertex data
            out->interaction = SurfaceInteraction{gh.position, gh.normal, transform_normal(p.local_to_render, gh.shading_normal),
                                                  transform_vector3(p.local_to_render, gh.tangent), gh.uv, p.material};
            *t_out = gh.t_hit;
            return true;
        }
        const TriangleMesh& g = *geometries[p.geometry];
        M4 inv = faithful ? inverse(p.local_to_render) : p.render_to_local;
        Ray lr = transform_ray(inv, ray);
        GeomHit gh;
        auto item = [&](uint32_t t, const Ray& r, float tm, float* to, GeomHit* ho) { return g.tri_intersect(t, r, tm, to, ho); };
        if (!g.bvh.intersect<GeomHit>(lr, t_max, item, !faithful, &gh, c ? &c->closest_blas : nullptr)) return false;
        SurfaceInteraction si{gh.position, gh.normal, gh.shading_normal, gh.tangent, gh.uv, p.material};
        out->t_hit = gh.t_hit;
        out->wo = transform_vector3(p.local_to_render, -lr.d);
        out->primitive = (int)pi;
        out->triangle = gh.index;
        out->interaction = transform_interaction(p.local_to_render, si);
        *t_out = gh.t_hit;
        return true;
    }
    bool primitive_intersect_p(uint32_t pi, const Ray& ray, float t_max, Counters* c) const {
        const Primitive& p = primitives[pi];
        if (p.rs_mesh) {
            const TriangleMesh& g = *p.rs_mesh;
            auto item = [&](uint32_t t, const Ray& r, float tm) { return g.tri_intersect_p(t, r, tm); };
            return g.bvh.intersect_p(ray, t_max, item, c ? &c->any_blas : nullptr);
        }
        const TriangleMesh& g = *geometries[p.geometry];
        M4 inv = faithful ? inverse(p.local_to_render) : p.render_to_local;
        Ray lr = transform_ray(inv, ray);
        auto item = [&](uint32_t t, const Ray& r, float tm) { return g.tri_intersect_p(t, r, tm); };
        return g.bvh.intersect_p(lr, t_max, item, c ? &c->any_blas : nullptr);
    }
    // Scene::intersect / intersect_p (scene.rs:80-103)
    bool intersect(const Ray& ray, float t_max, Intersection* out, Counters* c) const {
        if (c) c->closest_rays++;
        if (c && flat.valid()) flat.walk(ray, t_max, false, &c->flat_closest);
        auto item = [&](uint32_t pi, const Ray& r, float tm, float* to, Intersection* ho) {
            return primitive_intersect(pi, r, tm, to, ho, c);
        };
        bool hit = tlas.intersect<Intersection>(ray, t_max, item, !faithful, out, c ? &c->closest_tlas : nullptr);
        if (hit && c) c->closest_hits++;
        return hit;
    }
    bool intersect_p(const Ray& ray, float t_max, Counters* c) const {
        if (c) c->shadow_rays++;
        if (c && flat.valid()) flat.walk(ray, t_max, true, &c->flat_any);
        auto item = [&](uint32_t pi, const Ray& r, float tm) { return primitive_intersect_p(pi, r, tm, c); };
        return tlas.intersect_p(ray, t_max, item, c ? &c->any_tlas : nullptr);
    }

    // ---- spectrum parameter evaluation (parameter.rs:38-47, rgb_texture.rs:48-66) ----
    SS sample_spectrum_param(const SpectrumParameter& sp, V2 uv, const Wavelengths& w, Counters* c) const {
        if (c) c->spectrum_evals++;
        if (sp.kind == SP_CONSTANT) return sp.constant.sample(w);
        if (c) c->textured_lookups++;
        float rgb[3];
        bilinear_sample_rgb(textures[sp.texture], uv, rgb);
        if (sp.kind == SP_TEXTURE_ALBEDO_SRGB) {
            Spectrum s; s.kind = SPEC_SIGMOID;
            table.get_srgb_encoded(rgb, s.c);
            return s.sample(w);
        }
        // SpectrumType::{Illuminant, Unbounded} (texture/rgb_texture.rs:56-64): RgbIlluminantSpectrum::new / RgbUnboundedSpectrum::new of the texel
        // (rgb_illuminant_spectrum.rs:26-46, rgb_unbounded_spectrum.rs:23-42): scale = 2 max(rgb), the sigmoid of rgb / scale
        float mx = std::fmax(rgb[0], std::fmax(rgb[1], rgb[2]));
        float scale = 2.0f * mx;
        if (scale == 0.0f) return SS::zero();                                // black texel: 0 instead of the reference's 0 / 0
        float scaled[3] = {rgb[0] / scale, rgb[1] / scale, rgb[2] / scale};
        Spectrum s; s.kind = SPEC_SIGMOID;
        table.get_srgb_encoded(scaled, s.c);
        SS out = SS::zero();
        for (int i = 0; i < 4; ++i) {
            if (i > 0 && w.is_secondary_terminated()) break;
            float v = s.value(w.lambda[i]);
            out.v[i] = sp.kind == SP_TEXTURE_ILLUMINANT_SRGB ? (scale * v) * Spectrum::lut_value(sp.illuminant, w.lambda[i]) : scale * v;
        }
        return out;
    }

    // FloatParameter::sample (parameter.rs:65-72) -> FloatTexture::sample, gamma_corrected = false (float_texture.rs:33-52)
    float sample_float_param(float constant, int tex, V2 uv) const {
        if (tex < 0) return constant;
        float rgb[3];
        bilinear_sample_rgb(textures[tex], uv, rgb);
        return rgb[0];
    }

    // ---- lights ----
    // EmissiveMaterial::average_intensity * area_sum (emissive_material.rs:63-79, emissive_triangle_mesh.rs:166-173)
    static constexpr int ENV_BASE = -1000000;       // light_list entry of environment light k is ENV_BASE - k
    static bool is_env(int entry) { return entry <= ENV_BASE; }
    SS light_phi(int prim, const Wavelengths& w) const {
        if (is_env(prim)) { const EnvLight& e = env_lights[ENV_BASE - prim]; return e.intensity * e.integrated.sample(w); }   // :287-290
        if (prim < 0) {
            const DeltaLight& d = delta_lights[-1 - prim];
            SS s = d.spectrum.sample(w);
            if (d.kind == DL_POINT) return (4.0f * PI_F * d.intensity) * s;                        // point_light.rs:78-81
            if (d.kind == DL_SPOT)                                                                  // spot_light.rs:84-96
                return ((d.intensity * s) * 2.0f * PI_F) *
                       ((1.0f - std::cos(d.angle_inner)) + (std::cos(d.angle_inner) - std::cos(d.angle_outer)) / 2.0f);
            return (d.intensity * d.area) * s;                                                      // directional_light.rs:40-44
        }
        const Primitive& p = primitives[prim];
        const Material& m = materials[p.material];
        SS rad = sample_spectrum_param(m.color, V2{0.5f, 0.5f}, w, nullptr);
        return (rad * sample_float_param(m.intensity, m.intensity_tex, V2{0.5f, 0.5f})) * p.area_sum;   // average_intensity: textures at the centre (:63-79)
    }
};

// LightSampler (light_sampler.rs:26-62,190-220) — rebuilt per call, as the reference does.
struct LightSampler {
    const Scene* scene;
    std::vector<float> weights, table;
    float weight_sum = 0.0f;
    LightSampler(const Scene& s, const Wavelengths& w) : scene(&s) {
        for (int pi : s.light_list) {
            float wt = s.light_phi(pi, w).average();
            weight_sum += wt;
            weights.push_back(wt);
        }
        table.assign(weights.size(), 0.0f);
        float cum = 0.0f;
        for (size_t i = 0; i < table.size(); ++i) { cum += weights[i]; table[i] = cum / weight_sum; }
    }
    bool sample_light(float u, int* prim, float* prob) const {
        if (table.empty() || weight_sum == 0.0f) return false;
        for (size_t i = 0; i < table.size(); ++i)
            if (u < table[i]) { *prim = scene->light_list[i]; *prob = weights[i] / weight_sum; return true; }
        size_t l = table.size() - 1;
        *prim = scene->light_list[l]; *prob = weights[l] / weight_sum;
        return true;
    }
    // probability_infinite_light (light_sampler.rs:115-153): weight / sum of the INFINITE lights' weights
    float probability_infinite(int entry) const {
        if (table.empty() || weight_sum == 0.0f) return 0.0f;
        float inf_sum = 0.0f;
        for (size_t i = 0; i < scene->light_list.size(); ++i) if (Scene::is_env(scene->light_list[i])) inf_sum += weights[i];
        if (inf_sum == 0.0f) return 0.0f;
        for (size_t i = 0; i < scene->light_list.size(); ++i) if (scene->light_list[i] == entry) return weights[i] / inf_sum;
        return 0.0f;
    }
    float probability(int prim) const {
        if (table.empty() || weight_sum == 0.0f) return 0.0f;
        for (size_t i = 0; i < scene->light_list.size(); ++i)
            if (scene->light_list[i] == prim) return weights[i] / weight_sum;
        return 0.0f;
    }
};

}  // namespace oracle
