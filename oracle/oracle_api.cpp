// ORACLE — TEST INFRASTRUCTURE ONLY.
// C entry points of the CPU restatement, shaped like include/mi355pt.h so the parity tests feed
// the same scene description to the oracle and to the HIP product.  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() load this library; the product never does.
//
// Parity status: the reference cannot be built here (no Rust toolchain, LFS assets are stubs) and
// holds no known-answer vectors below image level => "parity unpinned" except for the Sobol
// generator matrices, which are pinned to the reference's own table words.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>

#include "../include/mi355pt.h"
#include "o_integrator.hpp"

using namespace oracle;

struct ptoracle_scene {
    Scene scene;
    Counters counters;
    std::string err;
};

static bool lower_spectrum(ptoracle_scene* h, const mi355pt_spectrum& in, Spectrum* out) {
    Scene& s = h->scene;
    switch (in.kind) {
        case MI355PT_SPEC_CONSTANT: out->kind = SPEC_CONSTANT; out->c[0] = in.c[0]; return true;
        case MI355PT_SPEC_SIGMOID: out->kind = SPEC_SIGMOID; std::memcpy(out->c, in.c, sizeof(float) * 3); return true;
        case MI355PT_SPEC_RGB_ALBEDO_SRGB:
            if (!s.table.valid()) return false;
            out->kind = SPEC_SIGMOID; s.table.get_srgb_encoded(in.c, out->c); return true;   // RgbSigmoidPolynomial::from
        case MI355PT_SPEC_RGB_ALBEDO_SRGB_LINEAR:
            if (!s.table.valid()) return false;
            out->kind = SPEC_SIGMOID; s.table.get_srgb_encoded(in.c, out->c, true); return true;
        case MI355PT_SPEC_LUT470:
            if (in.id >= s.luts.size()) return false;
            out->kind = SPEC_LUT470; out->lut = s.luts[in.id].data(); return true;
        default: return false;
    }
}
static bool lower_param(ptoracle_scene* h, const mi355pt_spectrum& in, SpectrumParameter* out) {
    if (in.kind == MI355PT_SPEC_TEXTURE_ALBEDO_SRGB) {
        if (in.id >= h->scene.textures.size() || !h->scene.table.valid()) return false;
        out->kind = SP_TEXTURE_ALBEDO_SRGB; out->texture = (int)in.id; return true;
    }
    if (in.kind == MI355PT_SPEC_TEXTURE_ILLUMINANT_SRGB || in.kind == MI355PT_SPEC_TEXTURE_UNBOUNDED_SRGB) {
        if (in.id >= h->scene.textures.size() || !h->scene.table.valid()) return false;
        out->texture = (int)in.id;
        if (in.kind == MI355PT_SPEC_TEXTURE_UNBOUNDED_SRGB) { out->kind = SP_TEXTURE_UNBOUNDED_SRGB; return true; }
        const uint32_t lut = (uint32_t)in.c[0];
        if (!(in.c[0] >= 0.0f) || lut >= h->scene.luts.size()) return false;
        out->kind = SP_TEXTURE_ILLUMINANT_SRGB; out->illuminant = h->scene.luts[lut].data(); return true;
    }
    out->kind = SP_CONSTANT;
    return lower_spectrum(h, in, &out->constant);
}

extern "C" {

int ptoracle_scene_create(ptoracle_scene** out) { *out = new ptoracle_scene(); return 0; }
void ptoracle_scene_destroy(ptoracle_scene* s) { delete s; }
int ptoracle_scene_set_faithful(ptoracle_scene* s, int faithful) { s->scene.faithful = faithful != 0; return 0; }
// clearcoat coat-weight mode (Clearcoat::coat_weight): 0 shared estimate per vertex (default, = the product), 1 three independent estimates
// (the reference's structure), 2 table lookup; the table of material `mat` (64 floats, mi355pt_coat_albedo_table) for mode 2
int ptoracle_scene_set_clearcoat_mode(ptoracle_scene* s, int mode) { s->scene.cc_draws = mode; return 0; }
int ptoracle_scene_set_coat_albedo_lut(ptoracle_scene* s, uint32_t mat, const float* table64) {
    if (mat >= s->scene.materials.size()) return -1;
    s->scene.materials[mat].cc_albedo_lut.assign(table64, table64 + 64);
    return 0;
}
// diagnostic, call before scene_build: 1 = pre-transformed render-space triangles (the product's lowering) instead of the reference's
// per-primitive ray transform (o_scene.hpp Scene::render_space_lowering)
int ptoracle_scene_set_lowering(ptoracle_scene* s, int render_space) { s->scene.render_space_lowering = render_space != 0; return 0; }

int ptoracle_scene_set_rgb2spec(ptoracle_scene* s, const float* table, size_t n) {
    if (n != (size_t)(TBL + 3 * TBL * TBL * TBL * 3)) return -1;
    s->scene.table.data.assign(table, table + n);
    return 0;
}
int ptoracle_scene_add_lut470(ptoracle_scene* s, const float* v, uint32_t* id) {
    s->scene.luts.emplace_back(v, v + NLUT);
    *id = (uint32_t)s->scene.luts.size() - 1;
    return 0;
}
int ptoracle_scene_add_tex_rgb8(ptoracle_scene* s, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t* id) {
    TextureRgb8 t; t.w = w; t.h = h; t.data.assign(rgb, rgb + (size_t)w * h * 3);
    s->scene.textures.push_back(std::move(t));
    *id = (uint32_t)s->scene.textures.size() - 1;
    return 0;
}
int ptoracle_scene_add_mesh(ptoracle_scene* s, const float* pos, const float* nrm, const float* uv, const float* tri_tangent,
                            const uint32_t* idx, uint32_t nv, uint32_t nt, uint32_t* out) {
    auto m = std::make_unique<TriangleMesh>();
    m->positions.resize(nv); m->normals.resize(nv);
    for (uint32_t i = 0; i < nv; ++i) {
        m->positions[i] = V3{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
        // Normal::new(n) normalises, Normal::from renormalises (triangle_mesh.rs:166-170, normal.rs:18-20)
        m->normals[i] = normalize(normalize(V3{nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]}));
    }
    if (uv) { m->uvs.resize(nv); for (uint32_t i = 0; i < nv; ++i) m->uvs[i] = V2{uv[2 * i], uv[2 * i + 1]}; }
    if (tri_tangent) { m->tangents.resize(nt); for (uint32_t i = 0; i < nt; ++i) m->tangents[i] = V3{tri_tangent[3 * i], tri_tangent[3 * i + 1], tri_tangent[3 * i + 2]}; }
    if ((uv != nullptr) != (tri_tangent != nullptr)) return -1;
    m->indices.assign(idx, idx + (size_t)nt * 3);
    for (uint32_t i : m->indices) if (i >= nv) return -1;
    s->scene.geometries.push_back(std::move(m));
    *out = (uint32_t)s->scene.geometries.size() - 1;
    return 0;
}
int ptoracle_scene_add_material(ptoracle_scene* s, const mi355pt_material_desc* d, uint32_t* out) {
    Material m;
    m.type = d->type;
    if (!lower_param(s, d->color, &m.color)) return -1;
    m.normal_tex = d->normal_tex == MI355PT_NONE ? -1 : (int)d->normal_tex;
    if (m.normal_tex >= (int)s->scene.textures.size()) return -1;
    m.normal_flip_y = d->normal_flip_y != 0;
    m.intensity = d->intensity;
    m.intensity_tex = (d->type != MI355PT_MAT_EMISSIVE || d->intensity_tex == MI355PT_NONE) ? -1 : (int)d->intensity_tex;
    if (m.intensity_tex >= (int)s->scene.textures.size()) return -1;
    if (d->type == MI355PT_MAT_GLASS || d->type == MI355PT_MAT_PLASTIC) { if (!lower_spectrum(s, d->eta, &m.eta)) return -1; }
    m.thin = d->thin != 0; m.roughness = d->roughness;
    m.metallic_tex = d->metallic_tex == MI355PT_NONE ? -1 : (int)d->metallic_tex;
    m.roughness_tex = d->roughness_tex == MI355PT_NONE ? -1 : (int)d->roughness_tex;
    m.cc_thickness_tex = (d->type != MI355PT_MAT_CLEARCOAT || d->clearcoat_thickness_tex == MI355PT_NONE) ? -1 : (int)d->clearcoat_thickness_tex;
    if (m.metallic_tex >= (int)s->scene.textures.size() || m.roughness_tex >= (int)s->scene.textures.size() ||
        m.cc_thickness_tex >= (int)s->scene.textures.size()) return -1;
    if (d->type == MI355PT_MAT_METAL) { if (!lower_spectrum(s, d->eta, &m.eta) || !lower_spectrum(s, d->k, &m.k)) return -1; }
    if (d->type == MI355PT_MAT_SIMPLE_PBR) {     // SimplePbrMaterial == the clearcoat material's base layer (thickness 0 takes that path)
        m.type = MAT_CLEARCOAT; m.cc_metallic = d->metallic; m.cc_base_ior = d->ior; m.cc_thickness = 0.0f;
        m.cc_tint = SpectrumParameter{}; 
    }
    if (d->type == MI355PT_MAT_CLEARCOAT) {
        m.cc_metallic = d->metallic; m.cc_base_ior = d->ior; m.cc_ior = d->clearcoat_ior; m.cc_roughness = d->clearcoat_roughness;
        m.cc_thickness = d->clearcoat_thickness;
        if (!lower_param(s, d->clearcoat_tint, &m.cc_tint)) return -1;
    }
    s->scene.materials.push_back(m);
    *out = (uint32_t)s->scene.materials.size() - 1;
    return 0;
}
int ptoracle_scene_add_instance(ptoracle_scene* s, uint32_t geom, uint32_t mat, const float* l2w) {
    if (geom >= s->scene.geometries.size() || mat >= s->scene.materials.size()) return -1;
    Primitive p; p.geometry = (int)geom; p.material = (int)mat; p.local_to_world = M4::from_cols16(l2w);
    p.seq = s->scene.next_seq++;
    s->scene.primitives.push_back(p);
    return 0;
}
int ptoracle_scene_add_delta_light(ptoracle_scene* s, const mi355pt_light_desc* d) {
    if (d->kind < MI355PT_LIGHT_POINT || d->kind > MI355PT_LIGHT_DIRECTIONAL) return -1;
    DeltaLight l; l.kind = d->kind; l.intensity = d->intensity; l.angle_inner = d->angle_inner; l.angle_outer = d->angle_outer;
    if (!lower_spectrum(s, d->spectrum, &l.spectrum)) return -1;
    l.local_to_world = M4::from_cols16(d->local_to_world);
    l.seq = s->scene.next_seq++;
    s->scene.delta_lights.push_back(l);
    return 0;
}
int ptoracle_scene_add_environment_light(ptoracle_scene* s, float intensity, const float* rgb, uint32_t w, uint32_t h, const float* l2w,
                                         uint32_t illuminant_lut) {
    if (!rgb || w == 0 || h == 0 || illuminant_lut >= s->scene.luts.size()) return -1;
    EnvLight e; e.intensity = intensity; e.w = w; e.h = h; e.rgb.assign(rgb, rgb + (size_t)w * h * 3);
    e.illuminant = s->scene.luts[illuminant_lut].data();
    e.local_to_world = M4::from_cols16(l2w);
    e.seq = s->scene.next_seq++;
    s->scene.env_lights.push_back(std::move(e));
    return 0;
}
int ptoracle_scene_build(ptoracle_scene* s, const mi355pt_camera* cam) {
    if (s->scene.primitives.empty()) return -1;
    s->scene.build(V3{cam->position[0], cam->position[1], cam->position[2]});
    return 0;
}

static Camera make_camera(const mi355pt_camera* c) {
    Camera cam; cam.fov = c->fov_deg; cam.width = c->width; cam.height = c->height;
    cam.set_look_to(V3{c->position[0], c->position[1], c->position[2]}, V3{c->direction[0], c->direction[1], c->direction[2]},
                    V3{c->up[0], c->up[1], c->up[2]});
    return cam;
}
static RenderParams make_params(const mi355pt_camera* c, const mi355pt_params* p) {
    RenderParams r; r.width = c->width; r.height = c->height; r.spp = p->spp; r.seed = p->seed; r.max_depth = p->max_depth; r.rr_gate = 1.0f - p->rr_gate_slack;
    r.strategy = p->strategy; r.sampler = p->sampler; r.exposure = p->exposure;
    return r;
}

// cmf: 3*470 floats (X, Y, Z LUTs) — the sensor's presets::x()/y()/z() (sensor.rs:66-68)
// Renders linear accumulators (W*H*3) for samples [s_begin, s_end) with n_threads workers over a
// dynamic row queue (stands in for rayon's par_iter_mut, renderer.rs:121).  Returns seconds.
double ptoracle_render_accum(ptoracle_scene* s, const mi355pt_camera* c, const mi355pt_params* p, const float* cmf,
                             uint32_t s_begin, uint32_t s_end, float* accum, int n_threads, int collect_counters) {
    Camera cam = make_camera(c);
    PathTracer pt{s->scene, cam, make_params(c, p), {cmf, cmf + NLUT, cmf + 2 * NLUT}};
    uint32_t W = c->width, H = c->height;
    uint32_t sc = p->shard_count ? p->shard_count : 1, si = p->shard_count ? p->shard_index : 0;
    uint32_t tiles_x = (W + 7) / 8;
    std::atomic<uint32_t> next_row{0};
    std::mutex mu;
    if (n_threads < 1) n_threads = 1;
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&]() {
        Counters local;
        for (;;) {
            uint32_t y = next_row.fetch_add(1);
            if (y >= H) break;
            for (uint32_t x = 0; x < W; ++x) {
                uint32_t tile = (y / 8) * tiles_x + (x / 8);
                if (tile % sc != si) continue;
                V3 a = pt.render_pixel_accum(x, y, s_begin, s_end, collect_counters ? &local : nullptr);
                float* o = accum + ((size_t)y * W + x) * 3;
                o[0] += a.x; o[1] += a.y; o[2] += a.z;
            }
        }
        std::lock_guard<std::mutex> g(mu);
        s->counters.add(local);
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n_threads; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
int ptoracle_film_resolve(const float* accum, uint32_t n_pixels, uint32_t spp, float* out) {
    for (uint32_t i = 0; i < n_pixels; ++i) Sensor::resolve(V3{accum[3 * i], accum[3 * i + 1], accum[3 * i + 2]}, spp, out + 3 * i);
    return 0;
}
int ptoracle_quantize_u8(const float* rgb, size_t n, uint8_t* out) {      // renderer.rs:141-143 `(p*255.0) as u8` (saturating)
    for (size_t i = 0; i < n; ++i) {
        float v = rgb[i] * 255.0f;
        out[i] = std::isnan(v) ? 0 : (v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (uint8_t)v));
    }
    return 0;
}
void ptoracle_get_counters(ptoracle_scene* s, uint64_t* out /* 20 */) {
    const Counters& c = s->counters;
    uint64_t v[20] = {c.samples, c.closest_rays, c.shadow_rays, c.closest_tlas.nodes, c.closest_tlas.items, c.closest_blas.nodes,
                      c.closest_blas.items, c.any_tlas.nodes, c.any_tlas.items, c.any_blas.nodes, c.any_blas.items,
                      c.closest_hits, c.bounces, c.spectrum_evals, c.textured_lookups, c.sampler_draws,
                      c.flat_closest.nodes, c.flat_closest.items, c.flat_any.nodes, c.flat_any.items};
    std::memcpy(out, v, sizeof(v));
}
// the product's flat BVH (mi355pt_scene_export_bvh: 64-B node records, 48-B triangle records, root link) for step counting (FlatBvh)
int ptoracle_scene_set_flat_bvh(ptoracle_scene* s, const void* nodes, uint32_t n_nodes, const void* tris, uint32_t n_tris, int32_t root) {
    static_assert(sizeof(FlatBvh::Node) == 64 && sizeof(FlatBvh::Tri) == 48, "record sizes of csrc/layout.hpp");
    s->scene.flat.nodes.assign((const FlatBvh::Node*)nodes, (const FlatBvh::Node*)nodes + n_nodes);
    s->scene.flat.tris.assign((const FlatBvh::Tri*)tris, (const FlatBvh::Tri*)tris + n_tris);
    s->scene.flat.root = root;
    return 0;
}
void ptoracle_reset_counters(ptoracle_scene* s) { s->counters = Counters(); }

// ---- probes ----
int ptoracle_probe_sobol(uint32_t width, uint32_t height, uint32_t spp, uint32_t seed, const uint32_t* xys, uint32_t n,
                         const char* pattern, uint32_t* out) {
    size_t per = 0;
    for (const char* q = pattern; *q; ++q) per += (*q == '2') ? 2 : 1;
    for (uint32_t i = 0; i < n; ++i) {
        Sampler s = Sampler::create(1, spp, width, height, seed);
        s.start_pixel_sample(xys[3 * i], xys[3 * i + 1], xys[3 * i + 2], width);
        uint32_t* o = out + (size_t)i * per;
        for (const char* q = pattern; *q; ++q) {
            if (*q == '2') { s.get_2d_bits(o); o += 2; } else { *o++ = s.get_1d_bits(); }
        }
    }
    return 0;
}
int ptoracle_probe_sobol_index(uint32_t width, uint32_t height, uint32_t spp, uint32_t x, uint32_t y, uint32_t sample,
                               uint32_t dimension, uint64_t* out) {
    Sampler s = Sampler::create(1, spp, width, height, 0);
    s.start_pixel_sample(x, y, sample, width);
    s.dimension = dimension;
    *out = s.get_sample_index();
    return 0;
}
int ptoracle_sobol_matrices(uint32_t* out104) { std::memcpy(out104, sobol_matrices().m, sizeof(uint32_t) * 104); return 0; }

int ptoracle_probe_intersect(ptoracle_scene* s, const float* o, const float* d, uint32_t n, float* out_t, uint32_t* out_inst,
                             uint32_t* out_tri, float* out_n) {
    for (uint32_t i = 0; i < n; ++i) {
        Ray r{V3{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, V3{d[3 * i], d[3 * i + 1], d[3 * i + 2]}};
        Intersection h;
        if (s->scene.intersect(r, std::numeric_limits<float>::max(), &h, nullptr)) {
            out_t[i] = h.t_hit; out_inst[i] = (uint32_t)h.primitive; out_tri[i] = h.triangle;
            if (out_n) { out_n[3 * i] = h.interaction.normal.x; out_n[3 * i + 1] = h.interaction.normal.y; out_n[3 * i + 2] = h.interaction.normal.z; }
        } else {
            out_t[i] = -1.0f; out_inst[i] = MI355PT_NONE; out_tri[i] = MI355PT_NONE;
            if (out_n) { out_n[3 * i] = out_n[3 * i + 1] = out_n[3 * i + 2] = 0.0f; }
        }
    }
    return 0;
}
int ptoracle_probe_occluded(ptoracle_scene* s, const float* o, const float* d, const float* tmax, uint32_t n, uint8_t* out) {
    for (uint32_t i = 0; i < n; ++i) {
        Ray r{V3{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, V3{d[3 * i], d[3 * i + 1], d[3 * i + 2]}};
        out[i] = s->scene.intersect_p(r, tmax[i], nullptr) ? 1 : 0;
    }
    return 0;
}
int ptoracle_probe_radiance(ptoracle_scene* s, const mi355pt_camera* c, const mi355pt_params* p, const uint32_t* xys, uint32_t n,
                            float* out_L, float* out_lambda, float* out_pdf) {
    Camera cam = make_camera(c);
    static const float zeros[3 * NLUT] = {0};
    PathTracer pt{s->scene, cam, make_params(c, p), {zeros, zeros, zeros}};
    for (uint32_t i = 0; i < n; ++i) {
        Wavelengths wl; SS L = pt.trace(xys[3 * i], xys[3 * i + 1], xys[3 * i + 2], &wl, nullptr);
        for (int k = 0; k < NS; ++k) { out_L[4 * i + k] = L.v[k]; out_lambda[4 * i + k] = wl.lambda[k]; out_pdf[4 * i + k] = wl.pdf[k]; }
    }
    return 0;
}
// probe_radiance + per-query diagnostic flags of PathTracer::trace (bit 0: a Russian-roulette gate on a knife edge)
int ptoracle_probe_radiance_flags(ptoracle_scene* s, const mi355pt_camera* c, const mi355pt_params* p, const uint32_t* xys, uint32_t n,
                                  float* out_L, float* out_lambda, float* out_pdf, uint8_t* out_flags) {
    Camera cam = make_camera(c);
    static const float zeros[3 * NLUT] = {0};
    PathTracer pt{s->scene, cam, make_params(c, p), {zeros, zeros, zeros}};
    for (uint32_t i = 0; i < n; ++i) {
        Wavelengths wl; uint32_t fl = 0;
        SS L = pt.trace(xys[3 * i], xys[3 * i + 1], xys[3 * i + 2], &wl, nullptr, &fl);
        for (int k = 0; k < NS; ++k) { out_L[4 * i + k] = L.v[k]; out_lambda[4 * i + k] = wl.lambda[k]; out_pdf[4 * i + k] = wl.pdf[k]; }
        out_flags[i] = (uint8_t)fl;
    }
    return 0;
}
// table lookup probe: RgbSigmoidPolynomial::from(ColorSrgb(rgb)) -> c0,c1,c2
int ptoracle_probe_rgb2spec(ptoracle_scene* s, const float* rgb_enc, uint32_t n, float* out_c) {
    if (!s->scene.table.valid()) return -1;
    for (uint32_t i = 0; i < n; ++i) s->scene.table.get_srgb_encoded(rgb_enc + 3 * i, out_c + 3 * i);
    return 0;
}
// Trowbridge-Reitz probes (bsdf/dielectric.rs:29-112 = conductor.rs:151-250): D(wm), visible-normal density Dw(wo, wm), G(wo, wi)
// for wm = wi = dirs[i]; sample_wm(wo, u) for the n (u.x, u.y) pairs
void ptoracle_probe_ggx(float alpha_x, float alpha_y, const float* wo3, const float* dirs, uint32_t n, float* out_D, float* out_Dw, float* out_G) {
    DielectricBsdf g(SS::one(), true, false, alpha_x, alpha_y);
    V3 wo{wo3[0], wo3[1], wo3[2]};
    for (uint32_t i = 0; i < n; ++i) {
        V3 w{dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
        out_D[i] = g.D(w); out_Dw[i] = g.Dw(wo, w); out_G[i] = g.G(wo, w);
    }
}
void ptoracle_probe_ggx_sample(float alpha_x, float alpha_y, const float* wo3, const float* u2, uint32_t n, float* out_wm) {
    DielectricBsdf g(SS::one(), true, false, alpha_x, alpha_y);
    V3 wo{wo3[0], wo3[1], wo3[2]};
    for (uint32_t i = 0; i < n; ++i) { V3 w = g.sample_wm(wo, V2{u2[2 * i], u2[2 * i + 1]}); out_wm[3 * i] = w.x; out_wm[3 * i + 1] = w.y; out_wm[3 * i + 2] = w.z; }
}
// fresnel_complex probe (bsdf/conductor.rs:92-124), one wavelength lane replicated
float ptoracle_probe_fresnel_complex(float cos_i, float eta, float k) { return fresnel_complex(cos_i, SS::constant(eta), SS::constant(k)).v[0]; }
int ptoracle_bvh_stats(ptoracle_scene* s, uint64_t* out /* tlas nodes, total blas nodes */) {
    out[0] = s->scene.tlas.nodes.size(); out[1] = 0;
    for (auto& g : s->scene.geometries) out[1] += g->bvh.nodes.size();
    return 0;
}

}  // extern "C"
