// ORACLE — TEST INFRASTRUCTURE ONLY.
// Restatement of renderer/src/renderer.rs, renderer/src/renderer/{base_renderer,common,
// pt_renderer,nee_renderer,mis_renderer}.rs and renderer/src/{camera,filter,sensor,tone_map}.rs.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "o_materials.hpp"
#include "o_sampler.hpp"

namespace oracle {
static thread_local bool g_trace_on = false;

enum Strategy : uint32_t { STRAT_PT = 0, STRAT_NEE = 1, STRAT_MIS = 2 };

struct Camera {     // camera.rs:14-92
    V3 position{0, 0, 0}, direction{0, 0, -1}, up{0, 1, 0};
    float fov = 45.0f; uint32_t width = 0, height = 0;
    void set_look_to(V3 p, V3 d, V3 u) { position = p; direction = normalize(d); up = normalize(u); }
    Ray generate_ray(float x, float y) const {                               // :51-65
        float aspect = (float)width / (float)height;
        float fov_rad = fov * (PI_F / 180.0f);                                // f32::to_radians
        float scale = std::tan(fov_rad / 2.0f);
        float dx = (2.0f * x / (float)width - 1.0f) * aspect * scale;
        float dy = (1.0f - 2.0f * y / (float)height) * scale;
        V3 rd = normalize(V3{dx, dy, -1.0f});
        // glam Mat3::look_to_rh(dir, up).transpose(): columns s, u, -f
        V3 f = direction;
        V3 s = normalize(cross(f, up));
        V3 u = cross(s, f);
        V3 wd = normalize(s * rd.x + u * rd.y + (-f) * rd.z);
        return Ray{V3{0, 0, 0}, wd};
    }
    Ray sample_ray(uint32_t px, uint32_t py, V2 uv) const {                   // :68-81, filter.rs:24-29 (width 1.0)
        float fx = uv.x * 1.0f - 1.0f * 0.5f, fy = uv.y * 1.0f - 1.0f * 0.5f;
        float x = (float)px + fx + 0.5f, y = (float)py + fy + 0.5f;
        return generate_ray(x, y);
    }
};

struct RenderParams {
    uint32_t width = 0, height = 0, spp = 1, seed = 0, max_depth = 16;
    float rr_gate = 1.0f;   // roulette is skipped when max(T) >= rr_gate; 1 = the reference (mi355pt_params.rr_gate_slack, diagnostic)
    uint32_t strategy = STRAT_MIS, sampler = 1;
    float exposure = 1.0f;
};

struct Sensor {     // sensor.rs:12-88
    V3 acc{0, 0, 0};
    const float* cmf_x; const float* cmf_y; const float* cmf_z;
    void add_sample(const Wavelengths& wl, const SS& s, float exposure) {
        int count = wl.is_secondary_terminated() ? 1 : NS;
        V3 xyz{0, 0, 0};
        for (int k = 0; k < count; ++k) {
            float l = wl.lambda[k];
            int i = (int)std::floor(l - LAMBDA_MIN);
            if (i == NLUT) i = 0;
            float c = s.v[k] / wl.pdf[k] / (float)NS;   // plain f32 divisions (sensor.rs:62-63)
            xyz.x += c * cmf_x[i]; xyz.y += c * cmf_y[i]; xyz.z += c * cmf_z[i];
        }
        static const M3 m = srgb_xyz_to_rgb();    // the reference rebuilds GamutSrgb per sample (sensor.rs:72)
        V3 rgb = m3_mul(m, xyz);
        acc = acc + rgb * exposure;
    }
    static void resolve(V3 acc, uint32_t spp, float out[3]) {               // to_rgb :81-88 + tone_map.rs:20-28
        V3 avg = acc / (float)spp;
        float a[3] = {avg.x, avg.y, avg.z};
        for (int i = 0; i < 3; ++i) {
            float c = std::fmax(a[i], 0.0f);
            c = c / (1.0f + c);
            out[i] = srgb_oetf(c);
        }
    }
};

struct PathTracer {
    const Scene& scene; const Camera& cam; RenderParams prm;
    const float* cmf[3];
    static constexpr float RAY_EPS = 1e-5f;        // base_renderer.rs:34
    static constexpr float SHADOW_EPS = 1e-4f;     // common.rs:12

    static float balance_heuristic(float a, float b) {                      // common.rs:15-20
        if (a == 0.0f && b == 0.0f) return 0.0f;
        return a / (a + b);
    }

    // evaluate_emissive_surface (base_renderer.rs:54-73): UniformEdf ignores direction.
    bool emissive_radiance(const SurfaceInteraction& si, const Wavelengths& wl, Counters* c, SS* out) const {
        const Material& m = scene.materials[si.material];
        if (!m.is_emissive()) return false;
        MaterialEval me{scene, c};
        *out = me.emissive_radiance(m, wl, si.uv);
        return true;
    }

    // EmissiveTriangleMesh::sample_radiance (emissive_triangle_mesh.rs:176-308)
    struct AreaSample { SS radiance; float pdf; V3 light_normal; float pdf_dir; V3 position; };
    AreaSample sample_area_light(int prim, V3 shading_pos, const Wavelengths& wl, float s, V2 uv, Counters* c) const {
        const Primitive& p = scene.primitives[prim];
        const TriangleMesh& g = *scene.geometries[p.geometry];
        size_t index = 0;
        for (size_t i = 0; i < p.area_table.size(); ++i) if (s < p.area_table[i]) { index = i; break; }
        float b0, b1;
        if (uv.x < uv.y) { b0 = uv.x / 2.0f; b1 = uv.y - b0; } else { b1 = uv.y / 2.0f; b0 = uv.x - b1; }
        float b2 = 1.0f - b0 - b1;
        V3 p0 = transform_point3(p.local_to_render, g.positions[g.indices[index * 3]]);
        V3 p1 = transform_point3(p.local_to_render, g.positions[g.indices[index * 3 + 1]]);
        V3 p2 = transform_point3(p.local_to_render, g.positions[g.indices[index * 3 + 2]]);
        V3 pt = p0 * b0 + p1 * b1 + p2 * b2;
        V3 normal = normalize(normalize(cross(p1 - p0, p2 - p0)));            // .normalize().to_normal()
        V2 tuv{0, 0};
        if (!g.uvs.empty()) {
            V2 u0 = g.uvs[g.indices[index * 3]], u1 = g.uvs[g.indices[index * 3 + 1]], u2 = g.uvs[g.indices[index * 3 + 2]];
            tuv = V2{u0.x * b0 + u1.x * b1 + u2.x * b2, u0.y * b0 + u1.y * b1 + u2.y * b2};
        }
        V3 wi = normalize(pt - shading_pos);
        MaterialEval me{scene, c};
        AreaSample a;
        a.radiance = me.emissive_radiance(scene.materials[p.material], wl, tuv);
        a.pdf = 1.0f / p.area_sum;
        float distance = length(pt - shading_pos);
        a.pdf_dir = a.pdf * (distance * distance) / std::fmax(std::fabs(dot(normal, -wi)), 1e-8f);
        a.light_normal = normal; a.position = pt;
        return a;
    }
    // EmissiveTriangleMesh::pdf_light_sample (:334-353)
    float pdf_light_area(int prim, uint32_t tri) const {
        const Primitive& p = scene.primitives[prim];
        float probability = tri == 0 ? p.area_table[0] : p.area_table[tri] - p.area_table[tri - 1];
        return 1.0f / p.area_list[tri] * probability;
    }
    // Scene::pdf_light_sample (scene.rs:156-182)
    float pdf_light_sample(const LightSampler& ls, V3 shading_pos, const Intersection& isect) const {
        if (!scene.primitives[isect.primitive].is_light) return 0.0f;
        float probability = ls.probability(isect.primitive);
        float pdf_area = pdf_light_area(isect.primitive, isect.triangle);
        V3 dv = shading_pos - isect.interaction.position;
        float distance = length(dv);
        V3 wo = -normalize(dv);
        float pdf_dir = pdf_area * (distance * distance) / std::fabs(dot(isect.interaction.normal, wo));
        return probability * pdf_dir;
    }

    // evaluate_area_light{,_with_mis} (common.rs:82-171)
    void eval_area_light(const SurfaceInteraction& sp, const AreaSample& rad, const Material& mat, const Wavelengths& wl,
                         V3 wo_render, const M4& r2t, float light_prob, bool mis, SS* contrib, float* weight, Counters* c, uint64_t mc_key) const {
        V3 dv = rad.position - sp.position;
        Ray shadow = move_forward(Ray{sp.position, normalize(dv)}, SHADOW_EPS);
        float t = length(dv) - 2.0f * SHADOW_EPS;
        bool visible = !scene.intersect_p(shadow, t, c);
        *contrib = SS::zero(); *weight = 1.0f;
        if (!visible) return;
        V3 wo = transform_vector3(r2t, wo_render);
        V3 wi = transform_vector3(r2t, normalize(dv));
        ShadingPoint spt{transform_normal(r2t, sp.normal), sp.uv};
        MaterialEval me{scene, c};
        me.mc_key = mc_key;   // same inner Monte-Carlo stream as the vertex's sample() call
        SS f = me.evaluate(mat, wl, wo, wi, spt);
        if (g_trace_on) {
            V3 ln_ = transform_normal(r2t, rad.light_normal);
            std::fprintf(stderr, "[oracle] nee wo=(%.9g %.9g %.9g) wi=(%.9g %.9g %.9g) f=(%.9g %.9g %.9g %.9g) pdf_dir=%.9g pdf_bsdf=%.9g rad=(%.9g %.9g %.9g %.9g) g=%.9g den=%.9g\n", wo.x, wo.y, wo.z, wi.x, wi.y, wi.z,
                         f.v[0], f.v[1], f.v[2], f.v[3], rad.pdf_dir, me.pdf(mat, wl, wo, wi, spt), rad.radiance.v[0], rad.radiance.v[1], rad.radiance.v[2], rad.radiance.v[3],
                         std::fabs(dot(ln_, -wi)) / length_squared(dv), rad.pdf * light_prob);
        }
        float distance2 = length_squared(dv);
        V3 ln = transform_normal(r2t, rad.light_normal);
        float cos_light = std::fabs(dot(ln, -wi));
        float g = cos_light / distance2;
        if (mis) {
            float pdf_bsdf = me.pdf(mat, wl, wo, wi, spt);
            *weight = balance_heuristic(rad.pdf_dir, pdf_bsdf);
        }
        *contrib = f * rad.radiance * g / (rad.pdf * light_prob);
    }

    // Scene::evaluate_infinite_light_radiance (scene.rs:208-224)
    SS infinite_radiance(V3 dir, const Wavelengths& wl) const {
        SS total = SS::zero();
        for (const EnvLight& e : scene.env_lights) total = total + e.direction_radiance(scene.table, dir, wl);
        return total;
    }
    // evaluate_infinite_light{,_with_mis} (common.rs:174-241) after EnvironmentLight::sample_infinite_light (:317-340)
    void eval_env_light(const EnvLight& e, V2 luv, const SurfaceInteraction& sp, const Material& mat, const Wavelengths& wl, V3 wo_render,
                        const M4& r2t, float light_prob, bool mis, SS* contrib, float* weight, Counters* c, uint64_t mc_key) const {
        V3 wi_r; float pdf_dir;
        e.sample(luv.x, luv.y, &wi_r, &pdf_dir);
        SS radiance = e.direction_radiance(scene.table, wi_r, wl);
        *contrib = SS::zero(); *weight = 1.0f;
        Ray shadow = move_forward(Ray{sp.position, wi_r}, SHADOW_EPS);
        if (scene.intersect_p(shadow, std::numeric_limits<float>::max(), c)) return;
        MaterialEval me{scene, c};
        me.mc_key = mc_key;
        ShadingPoint spt{transform_normal(r2t, sp.normal), sp.uv};
        V3 wo = transform_vector3(r2t, wo_render), wi = transform_vector3(r2t, wi_r);
        SS f = me.evaluate(mat, wl, wo, wi, spt);
        if (mis) *weight = balance_heuristic(pdf_dir, me.pdf(mat, wl, wo, wi, spt));
        *contrib = f * radiance / (pdf_dir * light_prob);
    }

    // evaluate_delta_point_light / evaluate_delta_directional_light (common.rs:23-79) for PointLight, SpotLight
    // (PrimitiveDeltaPointLight, {point,spot}_light.rs) and DirectionalLight (directional_light.rs:95-107); MIS weight 1.
    SS eval_delta_light(const DeltaLight& d, const SurfaceInteraction& sp, const Material& mat, const Wavelengths& wl, V3 wo_render,
                        const M4& r2t, float light_prob, Counters* c, uint64_t mc_key) const {
        MaterialEval me{scene, c};
        me.mc_key = mc_key;
        ShadingPoint spt{transform_normal(r2t, sp.normal), sp.uv};
        V3 wo = transform_vector3(r2t, wo_render);
        if (d.kind == DL_DIRECTIONAL) {
            V3 direction = normalize(transform_vector3(d.local_to_render, V3{0.0f, 0.0f, 1.0f}));
            SS intensity = d.intensity * d.spectrum.sample(wl);
            Ray shadow{sp.position, direction};                                             // not moved forward (:60-61)
            if (scene.intersect_p(shadow, std::numeric_limits<float>::max(), c)) return SS::zero();
            V3 wi = transform_vector3(r2t, normalize(direction));
            SS f = me.evaluate(mat, wl, wo, wi, spt);
            return f * intensity / light_prob;
        }
        V3 position = transform_point3(d.local_to_render, V3{0.0f, 0.0f, 0.0f});
        SS intensity = d.intensity * d.spectrum.sample(wl);
        if (d.kind == DL_SPOT) {                                                            // spot_light.rs:99-122
            V3 wi_l = normalize(position - sp.position);
            float a = d.angle_outer, b = d.angle_inner;
            float theta = transform_vector3(inverse(d.local_to_render), wi_l).z;
            float t = std::fmin(std::fmax((theta - a) / (b - a), 0.0f), 1.0f);
            float falloff = t * t * (3.0f - 2.0f * t);
            intensity = intensity * falloff;
        }
        V3 dv = position - sp.position;
        Ray shadow = move_forward(Ray{sp.position, normalize(dv)}, SHADOW_EPS);
        float t = length(dv) - 2.0f * SHADOW_EPS;
        if (scene.intersect_p(shadow, t, c)) return SS::zero();
        V3 wi = transform_vector3(r2t, normalize(dv));
        SS f = me.evaluate(mat, wl, wo, wi, spt);
        float distance2 = length_squared(dv);
        return f * intensity / (distance2 * light_prob);
    }

    // one camera path; returns L and the (possibly terminated) wavelengths
    // `flags` (diagnostic, optional): bit 0 = some Russian-roulette gate of this path sat on a knife edge, max(T) within 2 ulp of 1.
    // The gate `p >= 1` (base_renderer.rs:76-92) decides whether a random number is drawn at all; a reflection off a constant-eta
    // dielectric has f = F and pdf = F / (F + (1 - F)), so T = F * (1 / pdf) lands on 1 or 1 - ulp depending on the last bit of F.
    // An implementation whose cos(theta) differs in the last ulp takes the other branch, and every later Sobol dimension shifts.
    SS trace(uint32_t px, uint32_t py, uint32_t sample_index, Wavelengths* wl_out, Counters* c, uint32_t* flags = nullptr) const {
        Sampler smp = Sampler::create((int)prm.sampler, prm.spp, prm.width, prm.height, prm.seed);
        smp.start_pixel_sample(px, py, sample_index, prm.width);
        {   // PTORACLE_TRACE="x,y,s": print the BSDF samples and light connections of that sample (debugging aid, tools/trace_sample.py); read once
            static const struct TraceTarget { bool on = false; unsigned x = 0, y = 0, s = 0;
                                              TraceTarget() { if (const char* tr = std::getenv("PTORACLE_TRACE")) on = std::sscanf(tr, "%u,%u,%u", &x, &y, &s) == 3; } } target;
            g_trace_on = target.on && target.x == px && target.y == py && target.s == sample_index;
        }
        SS T = SS::one(), L = SS::zero();
        float u = smp.get_1d();
        Wavelengths wl = Wavelengths::new_uniform(u);
        V2 uvp = smp.get_2d();
        Ray ray = move_forward(cam.sample_ray(px, py, uvp), RAY_EPS);
        Intersection hit;
        if (c) c->samples++;
        auto finish = [&]() { if (c) c->sampler_draws += smp.draws; *wl_out = wl; return L; };
        if (!scene.intersect(ray, std::numeric_limits<float>::max(), &hit, c)) {                  // base_renderer.rs:180-187
            L = L + T * infinite_radiance(ray.d, wl);
            return finish();
        }
        SS le;
        if (emissive_radiance(hit.interaction, wl, c, &le)) L = L + T * le;
        MaterialEval me{scene, c};
        for (uint32_t depth = 1; depth <= prm.max_depth; ++depth) {
            const Material& mat = scene.materials[hit.interaction.material];
            if (mat.is_emissive()) break;                                                   // as_bsdf_material() == None
            if (c) c->bounces++;
            M4 r2t = from_shading_normal_tangent(hit.interaction.shading_normal, hit.interaction.tangent);
            V3 wo = transform_vector3(r2t, hit.wo);
            ShadingPoint spt{transform_normal(r2t, hit.interaction.normal), hit.interaction.uv};
            float uc = smp.get_1d();
            V2 uv = smp.get_2d();
            // key of the clearcoat's inner Monte-Carlo stream: one stream per path vertex (see McRng)
            me.mc_key = mix_bits((((uint64_t)smp.morton_index) << 32) | (uint64_t)smp.dimension) ^ 0xD1B54A32D192ED03ull;
            MaterialSample ms = me.sample(mat, uc, uv, wl, wo, spt);
            if (g_trace_on) std::fprintf(stderr, "[oracle] sample depth=%u sampled=%d spec=%d wo=(%.9g %.9g %.9g) wi=(%.9g %.9g %.9g) f0=%.9g pdf=%.9g uc=%.9g uv=(%.9g %.9g) T0=%.9g\n", depth - 1, (int)ms.is_sampled,
                                         (int)ms.is_specular(), wo.x, wo.y, wo.z, ms.wi.x, ms.wi.y, ms.wi.z, ms.f.v[0], ms.pdf, uc, uv.x, uv.y, T.v[0]);
            if (ms.is_non_specular() && prm.strategy != STRAT_PT) {                         // base_renderer.rs:218-228
                LightSampler ls(scene, wl);                                                 // mis_renderer.rs:40
                float ul = smp.get_1d();
                int lprim; float lprob;
                if (ls.sample_light(ul, &lprim, &lprob)) {
                    float s = smp.get_1d();
                    V2 luv = smp.get_2d();
                    if (Scene::is_env(lprim)) {
                        SS contrib; float w;
                        eval_env_light(scene.env_lights[Scene::ENV_BASE - lprim], luv, hit.interaction, mat, wl, hit.wo, r2t, lprob,
                                       prm.strategy == STRAT_MIS, &contrib, &w, c, me.mc_key);
                        L = L + (T * contrib) * w;
                    } else if (lprim < 0) {                                                 // Scene::calculate_light (scene.rs:114-139)
                        SS contrib = eval_delta_light(scene.delta_lights[-1 - lprim], hit.interaction, mat, wl, hit.wo, r2t, lprob, c, me.mc_key);
                        L = L + (T * contrib) * 1.0f;
                    } else {
                        AreaSample as = sample_area_light(lprim, hit.interaction.position, wl, s, luv, c);
                        SS contrib; float w;
                        eval_area_light(hit.interaction, as, mat, wl, hit.wo, r2t, lprob, prm.strategy == STRAT_MIS, &contrib, &w, c, me.mc_key);
                        L = L + (T * contrib) * w;
                    }
                }
            }
            // process_bsdf_sampling (base_renderer.rs:95-139)
            if (!ms.is_sampled) break;   // infinite-light arm contributes nothing without infinite lights
            float tf = 1.0f / ms.pdf;
            V3 wi_render = transform_vector3(inverse(r2t), ms.wi);
            V3 n = hit.interaction.normal;
            float sign = dot(n, wi_render) < 0.0f ? -1.0f : 1.0f;
            V3 origin = hit.interaction.position + sign * n * RAY_EPS;
            Ray next = move_forward(Ray{origin, wi_render}, RAY_EPS);
            Intersection nh;
            if (!scene.intersect(next, std::numeric_limits<float>::max(), &nh, c)) {
                // calculate_bsdf_infinite_light_contribution: PT adds f*Le/pdf (pt_renderer.rs:50-82), NEE nothing
                // (nee_renderer.rs:139-153), MIS weights it with the summed infinite-light pdf (mis_renderer.rs:183-230)
                if (!scene.env_lights.empty() && prm.strategy != STRAT_NEE) {
                    SS radiance = infinite_radiance(next.d, wl);
                    if (prm.strategy == STRAT_PT) L = L + T * ms.f * radiance / ms.pdf;
                    else {
                        LightSampler ls(scene, wl);
                        float light_pdf = 0.0f;
                        for (size_t k = 0; k < scene.env_lights.size(); ++k)
                            light_pdf += ls.probability_infinite(Scene::ENV_BASE - (int)k) * scene.env_lights[k].direction_pdf(next.d);
                        float w = balance_heuristic(ms.pdf, light_pdf);
                        L = L + T * ms.f * radiance * tf * w;
                    }
                }
                break;
            }
            SS next_emissive = SS::zero();
            SS nle;
            if (emissive_radiance(nh.interaction, wl, c, &nle)) next_emissive = ms.f * nle * tf;
            SS modifier = ms.f * tf;
            // calculate_bsdf_contribution
            if (prm.strategy == STRAT_PT) {
                L = L + T * next_emissive;                                                  // pt_renderer.rs:33-47
            } else if (prm.strategy == STRAT_NEE) {
                if (ms.is_specular()) L = L + T * next_emissive;                            // nee_renderer.rs:120-137
            } else {
                if (ms.is_specular()) L = L + T * next_emissive;                            // mis_renderer.rs:151-181
                else {
                    LightSampler ls(scene, wl);
                    float pdf_light = pdf_light_sample(ls, hit.interaction.position, nh);
                    float w = balance_heuristic(ms.pdf, pdf_light);
                    L = L + (T * next_emissive) * w;
                }
            }
            T = T * modifier;
            hit = nh;
            // apply_russian_roulette (base_renderer.rs:76-92)
            float p = T.max_value();
            if (flags && std::fabs(p - 1.0f) <= 2.4e-7f) *flags |= 1u;
            if (!(p >= prm.rr_gate)) {
                float ur = smp.get_1d();
                if (ur < p) div_assign(T, p); else break;
            }
        }
        return finish();
    }

    // BaseSrgbRenderer::render(p) (base_renderer.rs:146-280): linear RGB accumulator for one pixel
    V3 render_pixel_accum(uint32_t px, uint32_t py, uint32_t s_begin, uint32_t s_end, Counters* c) const {
        Sensor sensor; sensor.cmf_x = cmf[0]; sensor.cmf_y = cmf[1]; sensor.cmf_z = cmf[2];
        for (uint32_t s = s_begin; s < s_end; ++s) {
            Wavelengths wl; SS L = trace(px, py, s, &wl, c);
            sensor.add_sample(wl, L, prm.exposure);
        }
        return sensor.acc;
    }
};

}  // namespace oracle
