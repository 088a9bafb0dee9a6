// Path state and the per-vertex stages of BaseSrgbRenderer::render (renderer/src/renderer/base_renderer.rs:146-280) as
// device functions shared by the kernel variants in pt_kernels.hip.  All state lives in registers.
#pragma once
#include "pt_device.hpp"
#include "pt_ggx.hpp"

namespace pt {

constexpr float RAY_EPS = 1e-5f;       // base_renderer.rs:34
constexpr float SHADOW_EPS = 1e-4f;    // common.rs:12

struct Frame { f3 t, b, n; };          // rows of the Render -> VertexNormalTangent matrix
PT_DEV f3 to_local(const Frame& f, f3 v) { return mk3(dot(f.t, v), dot(f.b, v), dot(f.n, v)); }
PT_DEV f3 to_world(const Frame& f, f3 v) { return f.t * v.x + f.b * v.y + f.n * v.z; }
// Transform::from_shading_normal_tangent (math/src/transform.rs:186-203); the inverse of an orthonormal basis is its transpose
PT_DEV Frame shading_frame(f3 shading_normal, f3 tangent) {
    Frame f;
    f.n = normalize(shading_normal);
    f.b = normalize(cross(normalize(f.n), tangent));
    f.t = normalize(cross(f.b, f.n));
    return f;
}
// The reference does not transpose: Transform::from_shading_normal_tangent returns glam's NUMERIC Mat4::inverse of [T B N 0; 0 0 0 1]
// (math/src/transform.rs:186-203), and the way back (base_renderer.rs:111 `render_to_tangent.inverse()`, Transform * Normal's inverse-
// transpose, transform.rs:45-51) is the numeric inverse of THAT.  The entries differ from the transpose in the last ulp; where a decision
// hangs on an ulp — the Russian-roulette gate on T = F * (1 / pdf) = 1 +- ulp after a specular reflection off a constant-eta dielectric,
// DESIGN.md 2.1 — a path then takes another, equally valid, turn.  PT_FRAME_INVERSE builds both matrices with glam's cofactor arithmetic
// (o_math.hpp `inverse`, specialised for the zero / one entries of an affine matrix without translation: every dropped term is an exact
// 0): 1 = in the kernels that hold such materials (FEAT_DIEL), 2 = in every kernel, 0 = transpose (DEFAULT).
// Measured (round 3, tools/frame_inverse_probe.py, profiles/r03_frame_inverse_probe.jsonl; share of 196 608 samples of a 64x48x64 frame whose
// radiance leaves the oracle's under the reference's own gate): scene 9 0.286 % -> 0.263 %, scene 13 0.0885 % -> 0.0880 %, scene 19 0.040 % ->
// 0.042 % — the frames alone explain almost nothing; together with an oracle that intersects pre-translated render-space triangles like the
// product (Oracle.set_render_space_lowering; alone: 0.216 % / 0.076 % / 0.045 %) 0.053 % / 0.018 % / 0.002 %: the two roundings COMPOUND, and
// a fifth of the flips has yet another source.  Cost: -1.2...-1.3 % on the dielectric kernels (scenes 8 / 9 / 10), -5.7 % on C2's kernel if
// applied everywhere.  It does not by itself let the solid-plastic frames pass with the reference's gate, so it is not the default.
// PT_EXACT_DIV 1: the divisions of the light connection and of the sensor are the reference's divisions; 0 (default): a reciprocal shared by the
// four wavelengths, x * (1 / PI) and the like (<= 1 ulp each: no path decision hangs on them, but half of the samples' radiance then differs
// from the reference's in its last bit).  Together with PT_SIGMOID_EXACT (pt_device.hpp) this is the BIT-EXACT build: measured share of
// samples whose spectral radiance equals the oracle's bit for bit 0.50 -> 0.83 (this macro, -0.6 %) -> 0.9999 (both, -2.2 ... 2.8 % more) on
// untextured scenes; profiles/r03_bit_exact_options.log, DESIGN.md 2.1.
#ifndef PT_EXACT_DIV
#define PT_EXACT_DIV 0
#endif
#ifndef PT_FRAME_INVERSE
#define PT_FRAME_INVERSE 2
#endif
template <uint32_t FEAT> constexpr bool numeric_frames() { return PT_FRAME_INVERSE == 2 || (PT_FRAME_INVERSE == 1 && (FEAT & FEAT_DIEL) != 0u); }
// glam Mat4::inverse of the matrix with columns c0, c1, c2 (and w = (0, 0, 0, 1)): the inverse's columns
PT_DEV void inverse3_glam(f3 c0, f3 c1, f3 c2, f3& r0, f3& r1, f3& r2) {
    const float m00 = c0.x, m01 = c0.y, m02 = c0.z, m10 = c1.x, m11 = c1.y, m12 = c1.z, m20 = c2.x, m21 = c2.y, m22 = c2.z;
    const f3 i0 = mk3(m11 * m22 - m12 * m21, m01 * m22 - m02 * m21, m01 * m12 - m02 * m11);
    const f3 i1 = mk3(m10 * m22 - m12 * m20, m00 * m22 - m02 * m20, m00 * m12 - m02 * m10);
    const f3 i2 = mk3(m10 * m21 - m11 * m20, m00 * m21 - m01 * m20, m00 * m11 - m01 * m10);
    const f3 a0 = mk3(i0.x, -i0.y, i0.z), a1 = mk3(-i1.x, i1.y, -i1.z), a2 = mk3(i2.x, -i2.y, i2.z);
    const float det = (m00 * a0.x + m01 * a1.x) + m02 * a2.x;
    const float rcp = 1.0f / det;
    r0 = a0 * rcp; r1 = a1 * rcp; r2 = a2 * rcp;
}
// render -> tangent as the rows to_local() dots with (`fr`), and tangent -> render as the columns to_world() combines (`fw`, also the rows
// that take a NORMAL to tangent space: inverse-transpose of render -> tangent)
PT_DEV void shading_frames_numeric(f3 shading_normal, f3 tangent, Frame& fr, Frame& fw) {
    const Frame o = shading_frame(shading_normal, tangent);               // T, B, N as the reference orthonormalises them
    f3 r0, r1, r2;
    inverse3_glam(o.t, o.b, o.n, r0, r1, r2);                              // render -> tangent, columns
    fr.t = mk3(r0.x, r1.x, r2.x); fr.b = mk3(r0.y, r1.y, r2.y); fr.n = mk3(r0.z, r1.z, r2.z);
    inverse3_glam(r0, r1, r2, fw.t, fw.b, fw.n);                           // and back
}
// Transform::from_normal_map (transform.rs:216-244)
PT_DEV Frame normal_map_frame(f3 nm) {
    Frame f;
    f.n = normalize(nm);
    f3 cx = fabsf(f.n.x) < 0.9f ? mk3(1, 0, 0) : mk3(0, 1, 0);
    f.t = normalize(cx - dot(f.n, cx) * f.n);
    f.b = normalize(cross(f.n, f.t));
    return f;
}
// ... and like the shading frame the reference INVERTS that basis numerically (transform.rs:243) and inverts the inverse again for the way
// back (`transform.inverse()` in every material's sample, e.g. lambert_material.rs): nfr = rows to_local() dots with, nfw = columns of the way back
PT_DEV void normal_map_frames_numeric(f3 nm, Frame& nfr, Frame& nfw) {
    const Frame o = normal_map_frame(nm);
    f3 r0, r1, r2;
    inverse3_glam(o.t, o.b, o.n, r0, r1, r2);
    nfr.t = mk3(r0.x, r1.x, r2.x); nfr.b = mk3(r0.y, r1.y, r2.y); nfr.n = mk3(r0.z, r1.z, r2.z);
    inverse3_glam(r0, r1, r2, nfw.t, nfw.b, nfw.n);
}
PT_DEV f3 mat3_mul(const float* m, f3 v) {   // column-major 3x3
    return mk3(m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z, m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
PT_DEV f3 orthogonalize(f3 n, f3 v) { return normalize(v - n * dot(n, v)); }   // normal.rs:45-51
PT_DEV f3 generate_tangent(f3 n) { return orthogonalize(n, fabsf(n.x) > 0.999f ? mk3(0, 1, 0) : mk3(1, 0, 0)); }   // normal.rs:55-65

struct Surface {           // SurfaceInteraction<Render> + what the integrator needs
    f3 p, ng, ns, tangent;
    f3 wo;                 // Intersection.wo: local_to_render * -(render_to_local * d), the ray direction after the reference's round trip
    f2 uv;
    uint32_t material, flags, light;
    float light_pdf_area;
};

PT_DEV Surface load_surface(const DevScene& sc, const Hit& h, f3 rd) {
    Surface s;
    uint32_t inst; bool ident;
    const TriVerts tv = load_tri_local(sc.tris_local, h.tri, &inst, &ident);
    const f3 p_l = tv.p0 * h.b0 + tv.p1 * h.b1 + tv.p2 * h.b2;                       // ray.rs:161-165, in LOCAL space
    const float4* q = (const float4*)(sc.shade + h.tri);
    float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6];
    s.ng = mk3(g.x, g.y, g.z);                                                       // ray.rs:167-174 + samples.rs:135, precomputed per triangle (layout.hpp)
    f3 n0 = mk3(a.x, a.y, a.z), n1 = mk3(a.w, b.x, b.y), n2 = mk3(b.z, b.w, c.x);
    f3 tan_l = mk3(c.y, c.z, c.w);
    s.material = __float_as_uint(e.z);
    s.flags = __float_as_uint(f.x); s.light = __float_as_uint(f.y); s.light_pdf_area = f.w;
    // geometry/impls/triangle_mesh.rs:73-97 in LOCAL space, then primitive transform (samples.rs:130-143)
    f3 sn_l = normalize(n0 * h.b0 + n1 * h.b1 + n2 * h.b2);
    f3 tg_l;
    if (s.flags & 1u) {
        s.uv = f2{d.x * h.b0 + d.z * h.b1 + e.x * h.b2, d.y * h.b0 + d.w * h.b1 + e.y * h.b2};
        tg_l = orthogonalize(sn_l, tan_l);
    } else {
        s.uv = f2{0.0f, 0.0f};
        tg_l = generate_tangent(sn_l);
    }
    if (ident) {                                                                     // a translation: the 3x3 products are exact
        s.p = p_l + (sc.tris_are_local ? mk3(sc.shared_mw[0], sc.shared_mw[1], sc.shared_mw[2]) : load_instance_mw(sc.instances + inst));   // (uniform)
        s.ns = normalize(sn_l); s.tangent = tg_l;
        s.wo = -rd;
    } else {
        const InstXf x = load_instance(sc.instances + inst);
        s.p = xf_point(x.mx, x.my, x.mz, x.mw, p_l);
        s.ns = xf_normal(x, sn_l);
        s.tangent = xf_vector(x.mx, x.my, x.mz, tg_l);
        s.wo = xf_vector(x.mx, x.my, x.mz, -xf_vector(x.ix, x.iy, x.iz, rd));
    }
    return s;
}

PT_DEV DevSpectrum load_spectrum(const DevSpectrum* p) {
    const float4* q = (const float4*)p;
    float4 a = q[0], b = q[1];
    DevSpectrum s;
    s.kind = __float_as_uint(a.x); s.id = __float_as_uint(a.y); s.c[0] = a.z; s.c[1] = a.w; s.c[2] = b.x;
    s.pad[0] = __float_as_uint(b.y); s.pad[1] = __float_as_uint(b.z); s.pad[2] = __float_as_uint(b.w);
    return s;
}

// ---- EnvironmentLight (primitive/impls/environment_light.rs) ----
PT_DEV void env_spherical(const DevEnv& e, f3 dir_render, float& theta, float& phi) {          // :96-103 after the rotation to local
    f3 dl = mat3_mul(e.r2l, dir_render);
    theta = fminf(fmaxf(acosf(dl.y), 0.0f), PI_F);
    phi = atan2f(dl.z, dl.x);
    if (phi < 0.0f) phi += 2.0f * PI_F;
}
// direction_radiance (:292-305): bilinear texel -> RgbIlluminantSpectrum (scale * sigmoid * D65) -> * intensity
PT_DEV void env_radiance(const DevScene& sc, const DevEnv& e, f3 dir_render, const Wl& wl, float out[4]) {
    float theta, phi;
    env_spherical(e, dir_render, theta, phi);
    float u = fminf(fmaxf(phi / (2.0f * PI_F), 0.0f), 1.0f), v = fminf(fmaxf(theta / PI_F, 0.0f), 1.0f);
    const uint32_t w = e.w, h = e.h;
    float x = u * (float)(w - 1), y = v * (float)(h - 1);
    uint32_t x0 = (uint32_t)floorf(x), y0 = (uint32_t)floorf(y);
    uint32_t x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    float fx = x - (float)x0, fy = y - (float)y0;
    const float4* t = (const float4*)e.texels;
    float4 p00 = t[(size_t)y0 * w + x0], p01 = t[(size_t)y0 * w + x1], p10 = t[(size_t)y1 * w + x0], p11 = t[(size_t)y1 * w + x1];
    float rgb[3];
    { float a = p00.x * (1.0f - fx) + p01.x * fx, b = p10.x * (1.0f - fx) + p11.x * fx; rgb[0] = a * (1.0f - fy) + b * fy; }
    { float a = p00.y * (1.0f - fx) + p01.y * fx, b = p10.y * (1.0f - fx) + p11.y * fx; rgb[1] = a * (1.0f - fy) + b * fy; }
    { float a = p00.z * (1.0f - fx) + p01.z * fx, b = p10.z * (1.0f - fx) + p11.z * fx; rgb[2] = a * (1.0f - fy) + b * fy; }
    float scale = 2.0f * fmaxf(rgb[0], fmaxf(rgb[1], rgb[2]));                                     // rgb_illuminant_spectrum.rs:26-41
    if (scale == 0.0f) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }                        // black texel: 0 (the reference divides 0/0)
    float enc[3] = {rgb[0] / scale, rgb[1] / scale, rgb[2] / scale}, c[3];
    rgb2spec_lookup(sc, enc, c);
    const float* lut = sc.luts + (size_t)e.illuminant_lut * 470;
    float lam[4];
    wl_lams(wl, lam);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float val = (scale * sigmoid_value(c[0], c[1], c[2], lam[i])) * lut_value(lut, lam[i]);
        out[i] = ((i > 0 && wl.term) ? 0.0f : val) * e.intensity;
    }
}
PT_DEV float env_pdf(const DevEnv& e, f3 dir_render) {                                          // calculate_direction_pdf :212-238
    if (!(e.total_weight > 0.0f)) return 0.0f;
    float theta, phi;
    env_spherical(e, dir_render, theta, phi);
    const uint32_t w = e.w, h = e.h;
    float u = phi / (2.0f * PI_F), v = theta / PI_F;
    uint32_t x = min((uint32_t)floorf(u * (float)w), w - 1), y = min((uint32_t)floorf(v * (float)h), h - 1);
    float4 p = ((const float4*)e.texels)[(size_t)y * w + x];
    float lum = 0.299f * p.x + 0.587f * p.y + 0.114f * p.z;
    float st = fmaxf(ref_sinf(theta), 1e-8f);
    float pdf_tex = (lum * st) / e.total_weight;
    float jac = (float)w * (float)h / (2.0f * PI_F * PI_F * st);
    return pdf_tex * jac;
}
PT_DEV uint32_t env_sample_cdf(const float* cdf, uint32_t n, float u) {                            // binary_search_by insertion point (:201-206)
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (cdf[mid] < u) lo = mid + 1; else hi = mid; }
    return min(lo, n - 1);
}
PT_DEV void env_sample(const DevEnv& e, f2 uv, f3& wi, float& pdf_dir) {                        // sample_infinite_light :317-340
    const uint32_t w = e.w, h = e.h;
    uint32_t y = env_sample_cdf(e.marginal, h, uv.x);
    uint32_t x = env_sample_cdf(e.conditional + (size_t)y * w, w, uv.y);
    float u = ((float)x + 0.5f) / (float)w, v = ((float)y + 0.5f) / (float)h;
    float theta = v * PI_F, phi = u * 2.0f * PI_F;
    float s_t, c_t, s_p, c_p; ref_sincosf(theta, &s_t, &c_t); ref_sincosf(phi, &s_p, &c_p);
    f3 wl = mk3(s_t * c_p, c_t, s_t * s_p);
    wi = mat3_mul(e.l2r, wl);
    pdf_dir = env_pdf(e, wi);
}

// Scene::evaluate_infinite_light_radiance (scene.rs:208-231): every infinite light's radiance along the direction, summed
PT_DEV void env_radiance_all(const DevScene& sc, f3 dir_render, const Wl& wl, float out[4]) {
    env_radiance(sc, sc.envs[0], dir_render, wl, out);
    for (uint32_t k = 1; k < sc.n_envs; ++k) {
        float r[4];
        env_radiance(sc, sc.envs[k], dir_render, wl, r);
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = out[i] + r[i];
    }
}
PT_DEV float balance_heuristic(float a, float b) { return (a == 0.0f && b == 0.0f) ? 0.0f : a / (a + b); }   // common.rs:15-20

// fresnel_dielectric for one wavelength lane (material/common.rs:87-105); spectrum '/' maps x/0 -> 0
PT_DEV float sdiv(float a, float b) { return b == 0.0f ? 0.0f : a / b; }
PT_DEV float fresnel_dielectric1(float cos_i, float eta) {
    cos_i = fminf(fmaxf(cos_i, 0.0f), 1.0f);
    float sin2_i = 1.0f - cos_i * cos_i;
    float sin2_t = sdiv(sin2_i, eta * eta);
    float cos_t = sqrtf(fminf(fmaxf(1.0f - sin2_t, 0.0f), 1.0f));
    float r_parl = sdiv(eta * cos_i - cos_t, eta * cos_i + cos_t);
    float r_perp = sdiv(cos_i - eta * cos_t, cos_i + eta * cos_t);
    return (r_parl * r_parl + r_perp * r_perp) * 0.5f;
}
PT_DEV bool refract(f3 wi, f3 n, float eta, f3& wt) {                         // common.rs:117-139
    float cos_i = dot(n, wi);
    float sin2_i = fmaxf(1.0f - cos_i * cos_i, 0.0f);
    float sin2_t = sin2_i / (eta * eta);
    if (sin2_t >= 1.0f) return false;
    float cos_t = sqrtf(fmaxf(1.0f - sin2_t, 0.0f));
    f3 w = mk3(-wi.x / eta, -wi.y / eta, -wi.z / eta) + n * (cos_i / eta - cos_t);
    if (dot(w, w) < 1e-12f) return false;
    wt = normalize(w);
    return true;
}

enum : uint32_t { ST_DIFFUSE = 0, ST_SPEC_REFL = 1, ST_SPEC_TRANS = 2, ST_GLOSSY_REFL = 3, ST_GLOSSY_TRANS = 4 };



struct Path {
    Sampler smp;
    Wl wl;
    float T[4], L[4];
    f3 ro, rd;                 // next closest-hit ray
    float pf[4];               // f of the BSDF sample that spawned that ray
    float p_pdf;               // its pdf
    f3 prev_pos;               // the vertex it leaves (MIS weight of an emissive hit needs it)
    uint32_t depth;
    bool from_camera, prev_spec;
};
struct ShadowReq { bool on; f3 o, d; float t; float c[4]; };   // pending light connection: ray + contribution if unoccluded

// base_renderer.rs:160-177: start sample `s_cur` of pixel (px, py)
template <bool STATS>
PT_DEV void regen_path(Path& P, const SamplerCtx& sctx, const DevCamera& cam, uint32_t px, uint32_t py, uint32_t s_cur, StatCounters& st) {
    Sampler& smp = P.smp; Wl& wl = P.wl; float* T = P.T; float* L = P.L; f3& ro = P.ro; f3& rd = P.rd;
    bool& from_camera = P.from_camera; uint32_t& depth = P.depth; bool need_new = true;

    // base_renderer.rs:160-177: wavelengths (dim 0), pixel sample (dims 1-2), camera ray
    sampler_start(smp, sctx, px, py, s_cur);
    float u = get_1d(smp, sctx);
    wl_init(wl, u);
    f2 uv = get_2d(smp, sctx);
    float fx = (float)px + (uv.x * 1.0f - 1.0f * 0.5f) + 0.5f;             // filter.rs:24-29, camera.rs:68-72
    float fy = (float)py + (uv.y * 1.0f - 1.0f * 0.5f) + 0.5f;
    float dx = (2.0f * fx / (float)cam.width - 1.0f) * cam.aspect * cam.tan_half_fov;   // camera.rs:51-65
    float dy = (1.0f - 2.0f * fy / (float)cam.height) * cam.tan_half_fov;
    f3 dc = normalize(mk3(dx, dy, -1.0f));
    f3 s = mk3(cam.s[0], cam.s[1], cam.s[2]), uu = mk3(cam.u[0], cam.u[1], cam.u[2]), ff = mk3(cam.f[0], cam.f[1], cam.f[2]);
    rd = normalize(s * dc.x + uu * dc.y + (-ff) * dc.z);
    ro = mk3(0, 0, 0) + rd * RAY_EPS;                                       // move_forward
#pragma unroll
    for (int i = 0; i < 4; ++i) { T[i] = 1.0f; L[i] = 0.0f; }
    from_camera = true; depth = 0; need_new = false;
    if (STATS) st.samples++;
    (void)need_new;
}

// One path vertex in two halves.  The closest-hit result of P.ro/P.rd arrives (got/hit):
//   shade_vertex_head  accounts emission (with the strategy's weight), applies throughput + Russian roulette, builds the shading frames
//                      and draws the BSDF's random numbers; returns true when the path ends here, else C.cont is set;
//   shade_vertex_tail  samples the BSDF and the light: P.ro/P.rd hold the next ray, `sh` the light connection (it may be set even
//                      when the path ends).
// Kernels with the clearcoat material estimate the coat's directional albedo wave-cooperatively BETWEEN the halves (pt_kernel.hpp);
// all other kernels call the two halves back to back inside one divergent region (shade_vertex below), which compiles to the code of
// a single function: ShadeCtx never crosses a re-convergence point there.  One source for every kernel, probe included.
// What the first half hands to the second half:
struct ShadeCtx {
    Surface sf; const DevMaterial* mat; uint32_t mtype;
    Frame fr, fw, nf, nfw; f3 wo, wo_nm, ng_t; float geo_wo, uc; f2 uv;   // nf / nfw: the normal-map basis, there and back (nfw only with numeric frames in texture kernels)
    //   // fr: render -> tangent rows; fw: tangent -> render columns (= fr unless numeric_frames)
    bool is_diel, rough_diel, cont;
    float d_alpha;                 // dielectric roughness at the shading point (constant or FloatTexture)
    // clearcoat inputs / result of the cooperative estimate
    bool need_cc; float cc_alpha_c, cc_r0c, cc_thick, cc_metallic, cc_rough_b, cc_fc; uint64_t mc_key;
};

// PHASE (the tail queue of pt_kernel.hpp, PT_TAILQ): 0 = the whole first half; 1 = only its front — emission with the strategy's weight,
// throughput, Russian roulette, the depth test: what decides whether the path goes on —; 2 = only its back for a path that went on — the
// surface again from the hit, shading frames, the BSDF's random numbers — whose result feeds shade_vertex_tail.  1 then 2 on the same
// (P, hit) compute exactly what 0 computes: the sampler's dimensions are consumed in the same order.
template <bool STATS, uint32_t FEAT, int PHASE = 0>
PT_DEV bool shade_vertex_head(Path& P, const DevScene& sc, const DevParams& prm, const SamplerCtx& sctx, bool got, const Hit& hit, ShadowReq& sh,
                           StatCounters& st, unsigned long long& tsa, ShadeCtx& C) {
    Sampler& smp = P.smp; Wl& wl = P.wl; float* T = P.T; float* L = P.L; f3& ro = P.ro; f3& rd = P.rd;
    bool& from_camera = P.from_camera; bool& prev_spec = P.prev_spec; float* pf = P.pf; float& p_pdf = P.p_pdf; f3& prev_pos = P.prev_pos;
    uint32_t& depth = P.depth;
    bool end_path = false;
    bool& do_shadow = sh.on; f3& sh_o = sh.o; f3& sh_d = sh.d; float& sh_t = sh.t; float* sh_c = sh.c;
    do_shadow = false;

    if (PHASE != 2 && !got) {
        end_path = true;
        if ((FEAT & FEAT_ENV) && sc.n_envs != 0u) {
            float rad[4];
            env_radiance_all(sc, rd, wl, rad);
            if (from_camera) {                                               // base_renderer.rs:180-187
#pragma unroll
                for (int i = 0; i < 4; ++i) L[i] = L[i] + T[i] * rad[i];
            } else if (prm.strategy == 0u) {                                 // pt_renderer.rs:50-82: T * f * Le / pdf
#pragma unroll
                for (int i = 0; i < 4; ++i) L[i] = L[i] + sdiv((T[i] * pf[i]) * rad[i], p_pdf);
            } else if (prm.strategy == 2u) {                                 // mis_renderer.rs:183-230 (also for specular samples)
                // Scene::pdf_infinite_light_sample (scene.rs:185-206): sum over the infinite lights of (probability among the INFINITE
                // lights, light_sampler.rs:115-153) x (direction pdf).  One environment light with a non-zero weight: probability 1.
                float light_pdf = 0.0f;
                if (sc.n_envs == 1u) light_pdf = 1.0f * env_pdf(sc.envs[0], rd);
                else {
                    float inf_sum = 0.0f;
                    for (uint32_t li = 0; li < sc.n_lights; ++li) {
                        DevLight lt = sc.lights[li];
                        if (lt.kind != LK_ENV) continue;
                        const DevMaterial* lm = sc.materials + lt.material;
                        float ph[4];
                        DevSpectrum ls = load_spectrum(&lm->color);
                        eval_spectrum<false, (FEAT & FEAT_EMTEX) != 0, (FEAT & FEAT_EMTEX) != 0>(sc, ls, wl, f2{0.5f, 0.5f}, ph, st);
                        float sum = 0.0f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum += (ph[i] * lm->intensity_avg) * lt.area_sum;
                        inf_sum += sum / 4.0f;
                    }
                    for (uint32_t li = 0; li < sc.n_lights; ++li) {
                        DevLight lt = sc.lights[li];
                        if (lt.kind != LK_ENV) continue;
                        const DevMaterial* lm = sc.materials + lt.material;
                        float ph[4];
                        DevSpectrum ls = load_spectrum(&lm->color);
                        eval_spectrum<false, (FEAT & FEAT_EMTEX) != 0, (FEAT & FEAT_EMTEX) != 0>(sc, ls, wl, f2{0.5f, 0.5f}, ph, st);
                        float sum = 0.0f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum += (ph[i] * lm->intensity_avg) * lt.area_sum;
                        const float probability = inf_sum == 0.0f ? 0.0f : (sum / 4.0f) / inf_sum;
                        light_pdf += probability * env_pdf(sc.envs[lt.first_tri], rd);
                    }
                }
                float w = balance_heuristic(p_pdf, light_pdf);
                float tf = 1.0f / p_pdf;
#pragma unroll
                for (int i = 0; i < 4; ++i) L[i] = L[i] + ((((T[i] * pf[i]) * rad[i]) * tf) * w);
            }                                                                // nee_renderer.rs:139-153: nothing
        }
    } else {
        // PHASE 1 (the front of the tail queue's vertex) knows the material TYPE from the hit (DevTri::pad[0], Hit::mclass) and needs the surface
        // for emitters only: every other lane skips the ten loads and the normalisations here — the back of the vertex computes the surface once
        Surface sf; const DevMaterial* mat = nullptr; uint32_t mtype;
        if (PHASE == 1) {
            mtype = hit.mclass & 7u;
            if (mtype == MT_EMISSIVE) { sf = load_surface(sc, hit, rd); mat = sc.materials + sf.material; }
        } else { sf = load_surface(sc, hit, rd); mat = sc.materials + sf.material; mtype = mat->type; }
        const bool emissive = PHASE != 2 && mtype == MT_EMISSIVE;            // (PHASE 2: the path went on, so the surface is not an emitter)
        float Le[4] = {0, 0, 0, 0};
        if constexpr (PHASE != 2) {
        if (emissive) {                                                      // evaluate_emissive_surface :54-73
            DevSpectrum rs = load_spectrum(&mat->color);
            eval_spectrum<STATS, (FEAT & FEAT_EMTEX) != 0, (FEAT & FEAT_EMTEX) != 0>(sc, rs, wl, sf.uv, Le, st);
            float inten = mat->intensity;
            if ((FEAT & FEAT_EMTEX) && mat->metallic_tex != 0xffffffffu) { float t3[3]; bilinear_rgb(sc, mat->metallic_tex, sf.uv, t3); inten = t3[0]; }   // FloatParameter::texture intensity at the hit (emissive_material.rs:55-56)
#pragma unroll
            for (int i = 0; i < 4; ++i) Le[i] = Le[i] * inten;
        }
        if (from_camera) {
            if (emissive) {
#pragma unroll
                for (int i = 0; i < 4; ++i) L[i] = L[i] + T[i] * Le[i];      // :190-194
            }
        } else {
            // calculate_bsdf_contribution (pt :33-47, nee :120-137, mis :151-181)
            float tf = 1.0f / p_pdf;
            if (emissive) {
                float w = 1.0f;
                if (prm.strategy == 1u) w = prev_spec ? 1.0f : 0.0f;
                else if (prm.strategy == 2u && !prev_spec) {
                    // Scene::pdf_light_sample (scene.rs:156-182); light probability = phi-weighted pick
                    float wsum = 0.0f, wme = 0.0f;
                    if (!(FEAT & FEAT_MLIGHT) || sc.n_lights == 1u) {
                        // one light (every BASELINE config): phi(lambda) is Le * area, already evaluated — unless the radiance is a
                        // texture, whose phi is taken at uv (0.5, 0.5) (EmissiveMaterial::average_intensity, emissive_material.rs:61-79)
                        float sum = 0.0f;
                        float area = sc.lights[0].area_sum;
                        if ((FEAT & FEAT_EMTEX) && (mat->color.kind == SPK_TEXTURE || mat->metallic_tex != 0xffffffffu)) {
                            float ph[4];
                            DevSpectrum ls = load_spectrum(&mat->color);
                            eval_spectrum<false, true, true>(sc, ls, wl, f2{0.5f, 0.5f}, ph, st);
#pragma unroll
                            for (int i = 0; i < 4; ++i) sum += (ph[i] * mat->intensity_avg) * area;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) sum += Le[i] * area;
                        }
                        wsum = wme = sum / 4.0f;
                    } else
                    for (uint32_t li = 0; li < sc.n_lights; ++li) {
                        DevLight lt = sc.lights[li];
                        const DevMaterial* lm = sc.materials + lt.material;
                        float ph[4];
                        DevSpectrum ls = load_spectrum(&lm->color);
                        eval_spectrum<false, (FEAT & FEAT_EMTEX) != 0, (FEAT & FEAT_EMTEX) != 0>(sc, ls, wl, f2{0.5f, 0.5f}, ph, st);
                        float inten = lm->intensity_avg;
                        float sum = 0.0f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum += (ph[i] * inten) * lt.area_sum;
                        float wt = sum / 4.0f;
                        wsum += wt;
                        if (li == sf.light) wme = wt;
                    }
                    float probability = wsum == 0.0f ? 0.0f : wme / wsum;
                    f3 dv = prev_pos - sf.p;
                    float distance = length(dv);
                    f3 wo_l = -normalize(dv);
                    float pdf_dir = sf.light_pdf_area * (distance * distance) / fabsf(dot(sf.ng, wo_l));
                    w = balance_heuristic(p_pdf, probability * pdf_dir);
                }
                if (w != 0.0f || prm.strategy != 1u) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) L[i] = L[i] + (T[i] * ((pf[i] * Le[i]) * tf)) * w;
                }
            } else if (prm.strategy != 1u || prev_spec) {
                // The reference adds T * next_emissive_contribution * w even when the next vertex is not a light, and the contribution is
                // then SampledSpectrum::zero() (base_renderer.rs:124-131) — an exact zero, NOT f * 0 / pdf (pt_renderer.rs:43,
                // mis_renderer.rs:163-180 with pdf_light = 0).  That is +0 unless the throughput or the MIS weight is inf / NaN (rough
                // transmission at grazing half vectors: pdf = inf), in which case it poisons the sample with NaN exactly like the CPU
                // path; keep that behaviour bit for bit.  (Until round 3 this read (pf * 0) * tf: a specular sample with f = 0 and
                // pdf = NaN — a thin-film hero hit edge-on, wo.z = -7e-9, two samples of C3's 8.5e9 — went NaN here and not in the reference.)
                float w = (prm.strategy == 2u && !prev_spec) ? balance_heuristic(p_pdf, 0.0f) : 1.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) L[i] = L[i] + (T[i] * 0.0f) * w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) T[i] = T[i] * (pf[i] * tf);
            // apply_russian_roulette (:76-92)
            float p = fmaxf(fmaxf(fmaxf(fmaxf(-INFINITY, T[0]), T[1]), T[2]), T[3]);
            if (!(p >= prm.rr_gate)) {                                         // rr_gate = 1 (the reference) unless a test asks for slack
                float ur = get_1d(smp, sctx);
                if (ur < p) {
                    if (p != 0.0f) {
                        // four divisions, not one reciprocal: T_max / p must be exactly 1 so that the next vertex skips its roulette
                        // draw like the reference does (x * (1/x) can be 0.99999994, and a draw more shifts every later dimension)
#pragma unroll
                        for (int i = 0; i < 4; ++i) T[i] = T[i] / p;
                    }
                } else end_path = true;
            }
        }
        if (!end_path) {
            depth += 1;                                                       // for _ in 1..=max_depth (:197)
            if (depth > prm.max_depth || emissive) end_path = true;           // as_bsdf_material() == None (:199-202)
        }
        }   // PHASE != 2
        if (PHASE != 1 && !end_path) {
            if (STATS) { st.bounces++; tsa = __builtin_amdgcn_s_memtime(); }
            Frame fr, fw_num;
            if constexpr (numeric_frames<FEAT>()) shading_frames_numeric(sf.ns, sf.tangent, fr, fw_num);
            else fr = shading_frame(sf.ns, sf.tangent);
            const Frame& fw = numeric_frames<FEAT>() ? fw_num : fr;           // (the transpose form has ONE matrix: no second copy in ShadeCtx)
            f3 wo_r = sf.wo;                                                  // Intersection.wo
            f3 wo = to_local(fr, wo_r);
            f3 ng_t = normalize(to_local(fw, sf.ng));                         // Transform * Normal: inverse-transpose, then renormalised
            // base_renderer.rs:212-213 draws uc then uv for every material.  A draw whose value the material
            // never reads only has to advance the sampler's dimension (Lambert ignores uc, lambert_material.rs:44;
            // the smooth dielectric ignores uv, dielectric.rs:179-180): the Sobol digit loop is ~30 % of this
            // kernel's VALU time, so unused values are not computed.
            const bool is_diel = (FEAT & FEAT_DIEL) && (mtype == MT_GLASS || mtype == MT_PLASTIC);
            float uc = 0.0f;
            f2 uv = f2{0.0f, 0.0f};
            float d_alpha = mat->roughness;                                    // FloatParameter::sample(uv) (glass_material.rs:116, plastic_material.rs:104)
            if ((FEAT & FEAT_TEX) && (FEAT & FEAT_ROUGH) && is_diel && mat->roughness_tex != 0xffffffffu) { float t3[3]; bilinear_rgb(sc, mat->roughness_tex, sf.uv, t3); d_alpha = t3[0]; }
            const bool rough_diel = (FEAT & FEAT_ROUGH) && is_diel && d_alpha >= 1e-3f;            // !effectively_smooth (dielectric.rs:25-27)
            if (is_diel) {
                uc = get_1d(smp, sctx);
                // Plastic indexes a *textured* colour with the random uv (plastic_material.rs:123-126, Q15)
                if (rough_diel || (mtype == MT_PLASTIC && mat->color.kind == SPK_TEXTURE)) uv = get_2d(smp, sctx); else smp.dimension += 2;
            }
            else if ((FEAT & FEAT_CC) && mtype == MT_CLEARCOAT) { uc = get_1d(smp, sctx); uv = get_2d(smp, sctx); }
            else { smp.dimension += 1; uv = get_2d(smp, sctx); }
            // normal map frame (identity without a normal texture)
            Frame nf, nfw_num;
            if ((FEAT & FEAT_TEX) && mat->normal_tex != 0xffffffffu) {
                float rgb[3];
                bilinear_rgb(sc, mat->normal_desc, sf.uv, rgb);               // normal_texture.rs:39-66
                float nx = rgb[0] * 2.0f - 1.0f, ny = rgb[1] * 2.0f - 1.0f, nz = rgb[2] * 2.0f - 1.0f;
                if (mat->normal_flip_y) ny = -ny;
                float len = sqrtf(nx * nx + ny * ny + nz * nz);
                f3 nm = mk3(0, 0, 1);
                if (len > 0.0f) nm = normalize(normalize(mk3(nx / len, ny / len, nz / len)));
                if constexpr (numeric_frames<FEAT>()) normal_map_frames_numeric(nm, nf, nfw_num);
                else nf = normal_map_frame(nm);
            } else {
                nf.t = mk3(1, 0, 0); nf.b = mk3(0, 1, 0); nf.n = mk3(0, 0, 1);
                if constexpr (numeric_frames<FEAT>()) nfw_num = nf;           // (the inverse of the identity is the identity, exactly)
            }
            f3 wo_nm = to_local(nf, wo);
            // ---- hand-over to the second half ----
            C.sf = sf; C.mat = mat; C.mtype = mtype; C.fr = fr; if constexpr (numeric_frames<FEAT>() && !(FEAT & FEAT_CC)) C.fw = fw_num; C.nf = nf; if constexpr (numeric_frames<FEAT>() && (FEAT & FEAT_TEX) != 0u && !(FEAT & FEAT_CC)) C.nfw = nfw_num; C.wo = wo; C.wo_nm = wo_nm; C.ng_t = ng_t;
            C.geo_wo = dot(ng_t, wo); C.uc = uc; C.uv = uv; C.is_diel = is_diel; C.rough_diel = rough_diel; C.d_alpha = d_alpha; C.cont = true;
            if ((FEAT & FEAT_CC) && mtype == MT_CLEARCOAT) {
                // SimpleClearcoatPbrMaterial: FloatParameter values at the shading point + the inputs of the coat's directional albedo
                float metallic = mat->metallic; float thick = mat->cc_thickness;
                float rough_b = mat->roughness;
                if (FEAT & FEAT_TEX) {                                          // FloatParameter::Texture (parameter.rs:65-72)
                    float t3[3];
                    if (mat->metallic_tex != 0xffffffffu) { bilinear_rgb(sc, mat->metallic_tex, sf.uv, t3); metallic = t3[0]; }
                    if (mat->roughness_tex != 0xffffffffu) { bilinear_rgb(sc, mat->roughness_tex, sf.uv, t3); rough_b = t3[0]; }
                    if (mat->cc_thickness_tex != 0xffffffffu) { bilinear_rgb(sc, mat->cc_thickness_tex, sf.uv, t3); thick = t3[0]; }
                }
                C.cc_metallic = metallic; C.cc_thick = thick; C.cc_rough_b = rough_b;
                C.cc_alpha_c = mat->cc_roughness * mat->cc_roughness;               // roughness_to_alpha :76-78
                { float r = (mat->cc_ior - 1.0f) / (mat->cc_ior + 1.0f); C.cc_r0c = r * r; }   // compute_dielectric_r0 :81-84
                // coat weight: 64-sample Monte Carlo of the coat's directional albedo, ONE stream per path vertex.
                // The reference re-draws it from the thread RNG in sample(), evaluate() and pdf() (:190-192,318-320,416-418);
                // sharing one estimate per vertex keeps every marginal identical and is 3x cheaper.
                C.mc_key = mix_bits(((uint64_t)smp.morton << 32) | (uint64_t)smp.dimension) ^ 0xD1B54A32D192ED03ull;
                C.need_cc = thick > 0.0f;
                if (prm.albedo_lut != 0u && thick > 0.0f) {                     // the estimate's expectation from the material's table (an option, see mi355pt.h)
                    const float* tab = sc.cc_albedo + mat->cc_albedo_lut;
                    const float x = fminf(fmaxf(fabsf(wo_nm.z) * 64.0f - 0.5f, 0.0f), 63.0f);
                    const int i0 = min((int)x, 62);
                    const float tt = fminf(x - (float)i0, 1.0f);
                    C.cc_fc = tab[i0] + (tab[i0 + 1] - tab[i0]) * tt;
                    C.need_cc = false;
                }
            }
        }
    }
    return end_path;
}

template <bool STATS, uint32_t FEAT>
PT_DEV bool shade_vertex_tail(Path& P, const DevScene& sc, const DevParams& prm, const SamplerCtx& sctx, ShadowReq& sh, StatCounters& st,
                           unsigned long long& tsb, const ShadeCtx& C) {
    Sampler& smp = P.smp; Wl& wl = P.wl; float* T = P.T; float* L = P.L; f3& ro = P.ro; f3& rd = P.rd;
    bool& from_camera = P.from_camera; bool& prev_spec = P.prev_spec; float* pf = P.pf; float& p_pdf = P.p_pdf; f3& prev_pos = P.prev_pos;
    bool end_path = false;
    bool& do_shadow = sh.on; f3& sh_o = sh.o; f3& sh_d = sh.d; float& sh_t = sh.t; float* sh_c = sh.c;
    (void)L; (void)from_camera;
    const Surface& sf = C.sf; const DevMaterial* mat = C.mat; const uint32_t mtype = C.mtype;
    // (clearcoat kernels: ShadeCtx crosses the cooperative albedo estimate between the halves; the way back is recomputed here from the
    //  way there — the same inverse of the same matrix — instead of holding nine more registers across it: +3 % on scene 17)
    Frame fw_re;
    if constexpr (numeric_frames<FEAT>() && (FEAT & FEAT_CC) != 0u)
        inverse3_glam(mk3(C.fr.t.x, C.fr.b.x, C.fr.n.x), mk3(C.fr.t.y, C.fr.b.y, C.fr.n.y), mk3(C.fr.t.z, C.fr.b.z, C.fr.n.z), fw_re.t, fw_re.b, fw_re.n);
    const Frame& fr = C.fr; const Frame& fw = numeric_frames<FEAT>() ? ((FEAT & FEAT_CC) ? fw_re : C.fw) : C.fr; const Frame& nf = C.nf; const f3 wo = C.wo, wo_nm = C.wo_nm, ng_t = C.ng_t;
    Frame nfw_re;
    if constexpr (numeric_frames<FEAT>() && (FEAT & FEAT_TEX) != 0u && (FEAT & FEAT_CC) != 0u)
        inverse3_glam(mk3(C.nf.t.x, C.nf.b.x, C.nf.n.x), mk3(C.nf.t.y, C.nf.b.y, C.nf.n.y), mk3(C.nf.t.z, C.nf.b.z, C.nf.n.z), nfw_re.t, nfw_re.b, nfw_re.n);
    const Frame& nfw = (numeric_frames<FEAT>() && (FEAT & FEAT_TEX) != 0u) ? ((FEAT & FEAT_CC) ? nfw_re : C.nfw) : C.nf;   // the way back out of the normal-map basis
    const float uc = C.uc; const f2 uv = C.uv; const bool is_diel = C.is_diel, rough_diel = C.rough_diel;
    {
        {
            bool sampled = false, specular = false;
            f3 wi_sh = mk3(0, 0, 1);
            float s_f[4] = {0, 0, 0, 0}, s_pdf = 0.0f;
            float geo_wo = dot(ng_t, wo);

            // state the light connection needs to evaluate the BSDF again (BsdfSurfaceMaterial::{evaluate,pdf})
            uint32_t nee_kind = 0;            // 0 none, 1 Lambert, 2 clearcoat, 3 rough dielectric
            float d_er[4] = {1, 1, 1, 1};     // rough dielectric: relative eta per wavelength, flags
            bool d_thin = false, d_plastic = false;
            float albedo[4] = {0, 0, 0, 0};   // Lambert albedo / clearcoat base colour
            float cc_tint[4] = {1, 1, 1, 1};
            float cc_fc = 0.0f, cc_alpha_c = 0.0f, cc_alpha_b = 0.0f, cc_r0c = 0.0f, cc_r0d = 0.0f, cc_metallic = 0.0f, cc_thick = 0.0f;

            if (mtype == MT_LAMBERT) {
                // LambertMaterial::sample (lambert_material.rs:42-97) + NormalizedLambertBsdf (lambert.rs:38-75)
                nee_kind = 1;
                DevSpectrum cs = load_spectrum(&mat->color);
                eval_spectrum<STATS, (FEAT & FEAT_TEX) != 0>(sc, cs, wl, sf.uv, albedo, st);
                if (wo_nm.z != 0.0f) {
                    float r = sqrtf(uv.x), th = 2.0f * PI_F * uv.y;
                    float sn_th, cs_th; ref_sincosf(th, &sn_th, &cs_th);             // one range reduction for both (same values as sinf / cosf)
                    f3 wi = mk3(r * cs_th, r * sn_th, sqrtf(1.0f - uv.x));
                    if (wo_nm.z < 0.0f) wi.z = -wi.z;
                    if (wi.z != 0.0f && sgn1(wo_nm.z) == sgn1(wi.z)) {
                        f3 w = to_world(nfw, wi);
                        float gwi = dot(ng_t, w);
                        if (sgn1(gwi) == sgn1(geo_wo) && !(gwi != gwi) && !(geo_wo != geo_wo)) {
                            sampled = true; wi_sh = w; s_pdf = fabsf(wi.z) / PI_F;   // sampled f and pdf stay IEEE: for albedo 1 their ratio must round like the reference's (the `p >= 1` roulette gate)
#pragma unroll
                            for (int i = 0; i < 4; ++i) s_f[i] = albedo[i] * fabsf(wi.z) / PI_F;
                        }
                    }
                }
            } else if (is_diel) {
                // GlassMaterial/PlasticMaterial::sample -> DielectricBsdf::sample_specular (dielectric.rs:380-466)
                float eta[4];
                DevSpectrum es = load_spectrum(&mat->eta);
                eval_spectrum<STATS, (FEAT & FEAT_TEX) != 0>(sc, es, wl, sf.uv, eta, st);
                if (mtype == MT_PLASTIC) eta[1] = eta[2] = eta[3] = eta[0];   // SampledSpectrum::constant(self.eta) (plastic_material.rs:107-109): NOT sampled through the
                                                                              // wavelengths, so all four lanes keep the value after a glass vertex terminated the secondary ones
                bool eta_const = (eta[1] == eta[0]) && (eta[2] == eta[0]) && (eta[3] == eta[0]);
                if (eta[0] == 0.0f) { eta[0] = eta[1] = eta[2] = eta[3] = 1.0f; eta_const = true; }   // DielectricBsdf::new :144-148
                bool entering = geo_wo > 0.0f;
                bool thin = mat->thin != 0;
                if (rough_diel) {
                    // DielectricBsdf::sample_microfacet (dielectric.rs:217-365), alpha = roughness (glass_material.rs:120-126)
                    nee_kind = 3; d_thin = thin; d_plastic = mtype == MT_PLASTIC;
                    const float alpha = C.d_alpha;
#pragma unroll
                    for (int i = 0; i < 4; ++i) d_er[i] = (thin || entering) ? eta[i] : sdiv(1.0f, eta[i]);
                    if (wo_nm.z != 0.0f) {
                        f3 wm = ggx_sample_wm(alpha, alpha, wo_nm, uv);
                        float wodm = dot(wo_nm, wm);
                        float fr4[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) fr4[i] = fresnel_dielectric1(fabsf(wodm), d_er[i]);
                        float favg = (((0.0f + fr4[0]) + fr4[1]) + fr4[2] + fr4[3]) / 4.0f;
                        float pr = favg, pt = 1.0f - favg;
                        if (thin) { float r = favg, t = 1.0f - r, r2 = r * r; pr = r2 > 1.0f ? 1.0f : r + (t * t * r) / (1.0f - r2); pt = t; }
                        f3 wi = mk3(0, 0, 1);
                        if (uc < pr / (pr + pt)) {                                // sample_microfacet_reflection :284-314
                            f3 w = wm * (2.0f * wodm) - wo_nm;
                            float cd = fabsf(wodm);
                            if (wo_nm.z * w.z > 0.0f && !(cd < 1e-6f)) {
                                sampled = true; wi = w;
                                s_pdf = ggx_Dw(alpha, alpha, wo_nm, wm) / (4.0f * cd) * (pr / (pr + pt));
                                float dg = ggx_D(alpha, alpha, wm) * ggx_G(alpha, alpha, wo_nm, w);
#pragma unroll
                                for (int i = 0; i < 4; ++i) s_f[i] = fr4[i] * dg * fabsf(w.z) / (4.0f * fabsf(wo_nm.z));   // extra |cos wi| (Q8)
                            }
                        } else if (thin) {                                        // sample_specular_transmission_thin_surface :346-365
                            sampled = true; wi = mk3(-wo_nm.x, -wo_nm.y, -wo_nm.z); s_pdf = pt / (pr + pt);
#pragma unroll
                            for (int i = 0; i < 4; ++i) s_f[i] = 1.0f - fr4[i];
                        } else {                                                  // sample_microfacet_transmission :316-344
                            if (!eta_const) wl.term = true;
                            const float etap = d_er[0];
                            f3 wmr = entering ? wm : -wm;
                            f3 w;
                            if (refract(wo_nm, wmr, etap, w) && !(wo_nm.z * w.z > 0.0f) && fabsf(w.z) != 0.0f) {
                                float sden = dot(w, wm) + wodm / etap;
                                float denom = sden * sden;
                                float dwm = fabsf(dot(w, wm)) / denom;
                                sampled = true; wi = w;
                                s_pdf = ggx_Dw(alpha, alpha, wo_nm, wm) * dwm * (pt / (pr + pt));
                                float dg = ggx_D(alpha, alpha, wm) * ggx_G(alpha, alpha, wo_nm, w);
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    s_f[i] = sdiv((1.0f - fr4[i]) * dg * fabsf(dot(w, wm)) * fabsf(wodm), denom * fabsf(wo_nm.z) * etap * etap);   // spectrum / f32: x/0 -> 0
                            }
                        }
                        if (sampled) {
                            if (d_plastic && dot(wi, wo_nm) < 0.0f) {
                                float col[4];
                                DevSpectrum cs = load_spectrum(&mat->color);
                                eval_spectrum<STATS, (FEAT & FEAT_TEX) != 0>(sc, cs, wl, uv, col, st);
#pragma unroll
                                for (int i = 0; i < 4; ++i) s_f[i] = s_f[i] * col[i];
                            }
                            wi_sh = to_world(nfw, wi);
                        }
                    }
                } else
                if (wo_nm.z != 0.0f) {
                    float er[4], fr4[4], favg;
#pragma unroll
                    for (int i = 0; i < 4; ++i) er[i] = (thin || entering) ? eta[i] : sdiv(1.0f, eta[i]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) fr4[i] = fresnel_dielectric1(fabsf(wo_nm.z), er[i]);
                    favg = (((0.0f + fr4[0]) + fr4[1]) + fr4[2] + fr4[3]) / 4.0f;
                    float pr, pt;
                    if (thin) {                                                   // calculate_thin_surface_coefficients :367-378
                        float r = favg, t = 1.0f - r, r2 = r * r;
                        pr = r2 > 1.0f ? 1.0f : r + (t * t * r) / (1.0f - r2);
                        pt = t;
                    } else { pr = favg; pt = 1.0f - pr; }
                    f3 wi = mk3(0, 0, 1);
                    if (uc < pr / (pr + pt)) {
                        if (!(fabsf(wo_nm.z) < 1e-6f)) {
                            sampled = true; specular = true; wi = mk3(-wo_nm.x, -wo_nm.y, wo_nm.z); s_pdf = pr / (pr + pt);
#pragma unroll
                            for (int i = 0; i < 4; ++i) s_f[i] = fr4[i];
                        }
                    } else if (thin) {
                        wi = mk3(-wo_nm.x, -wo_nm.y, -wo_nm.z);
                        if (wi.z != 0.0f) {
                            sampled = true; specular = true; s_pdf = pt / (pr + pt);
#pragma unroll
                            for (int i = 0; i < 4; ++i) s_f[i] = 1.0f - fr4[i];
                        }
                    } else {
                        if (!eta_const) wl.term = true;                           // terminate_secondary :446-448
                        f3 n = entering ? mk3(0, 0, 1) : mk3(0, 0, -1);
                        f3 wt;
                        if (refract(wo_nm, n, er[0], wt) && wt.z != 0.0f) {
                            sampled = true; specular = true; wi = wt; s_pdf = pt / (pr + pt);
                            float e2 = er[0] * er[0];
#pragma unroll
                            for (int i = 0; i < 4; ++i) s_f[i] = sdiv(1.0f - fr4[i], e2);
                        }
                    }
                    if (sampled) {
                        if (mtype == MT_PLASTIC && dot(wi, wo_nm) < 0.0f) {          // plastic_material.rs:123-126 (random uv, Q15)
                            float col[4];
                            DevSpectrum cs = load_spectrum(&mat->color);
                            eval_spectrum<STATS, (FEAT & FEAT_TEX) != 0>(sc, cs, wl, uv, col, st);
#pragma unroll
                            for (int i = 0; i < 4; ++i) s_f[i] = s_f[i] * col[i];
                        }
                        wi_sh = to_world(nfw, wi);
                    }
                }
                // failed dielectric samples are "Diffuse" (non-specular): the reference then runs NEE with
                // f == 0 and ends the path; nothing observable happens, so it is skipped here.
            }
            else if ((FEAT & FEAT_CC) && mtype == MT_CLEARCOAT) {
                // SimpleClearcoatPbrMaterial::sample (simple_pbr_clearcoat_material.rs:137-260)
                nee_kind = 2;
                DevSpectrum cs = load_spectrum(&mat->color);
                eval_spectrum<STATS, (FEAT & FEAT_TEX) != 0>(sc, cs, wl, sf.uv, albedo, st);
                DevSpectrum ts = load_spectrum(&mat->cc_tint);
                eval_spectrum<STATS, (FEAT & FEAT_TEX) != 0>(sc, ts, wl, sf.uv, cc_tint, st);
                // FloatParameter values, coat parameters and the coat weight come from the first half / the wave-cooperative
                // estimate between the halves (ShadeCtx)
                const float metallic = C.cc_metallic, thick = C.cc_thick;
                cc_metallic = metallic; cc_thick = thick;
                cc_alpha_c = C.cc_alpha_c;
                cc_alpha_b = C.cc_rough_b * C.cc_rough_b;
                cc_r0c = C.cc_r0c;
                { float r = (mat->ior - 1.0f) / (mat->ior + 1.0f); cc_r0d = r * r; }
                if (thick > 0.0f) cc_fc = C.cc_fc;
                bool base = true;
                float ucb = uc;
                if (thick > 0.0f) {
                    if (uc < cc_fc) {
                        base = false;
                        GsSample g = gs_sample_R(cc_alpha_c, wo_nm, uv);
                        if (g.ok) {
                            sampled = true; specular = g.specular; wi_sh = to_world(nfw, g.wi); s_pdf = g.pdf * cc_fc;
#pragma unroll
                            for (int i = 0; i < 4; ++i) s_f[i] = (cc_r0c + (1.0f - cc_r0c) * g.p5) * g.dg;
                        }
                    } else ucb = (uc - cc_fc) / (1.0f - cc_fc);
                }
                if (base) {
                    // sample_base_material (:336-383): metallic / dielectric(+Lambert) / mixed
                    bool use_metal = metallic >= 1.0f || (metallic > 0.0f && ucb <= metallic);
                    float ucd = (metallic > 0.0f && metallic < 1.0f) ? (ucb - metallic) / (1.0f - metallic) : ucb;
                    f3 wi = mk3(0, 0, 1); bool okb = false; float fb[4] = {0, 0, 0, 0}, pb = 0.0f; bool specb = false;
                    if (use_metal) {
                        GsSample g = gs_sample_R(cc_alpha_b, wo_nm, uv);
                        if (g.ok) {
                            okb = true; wi = g.wi; pb = g.pdf; specb = g.specular;
#pragma unroll
                            for (int i = 0; i < 4; ++i) fb[i] = (albedo[i] + (1.0f - albedo[i]) * g.p5) * g.dg;
                        }
                    } else {
                        float frd = cc_r0d + (1.0f - cc_r0d) * schlick_p5(fabsf(wo_nm.z));
                        if (ucd < frd) {
                            GsSample g = gs_sample_R(cc_alpha_b, wo_nm, uv);
                            if (g.ok) {
                                okb = true; wi = g.wi; pb = g.pdf * frd; specb = g.specular;
#pragma unroll
                                for (int i = 0; i < 4; ++i) fb[i] = (cc_r0d + (1.0f - cc_r0d) * g.p5) * g.dg;
                            }
                        } else if (wo_nm.z != 0.0f) {
                            float r = sqrtf(uv.x), th = 2.0f * PI_F * uv.y;
                            float sn_th, cs_th; ref_sincosf(th, &sn_th, &cs_th);
                            f3 w = mk3(r * cs_th, r * sn_th, sqrtf(1.0f - uv.x));
                            if (wo_nm.z < 0.0f) w.z = -w.z;
                            if (w.z != 0.0f && sgn1(wo_nm.z) == sgn1(w.z)) {
                                okb = true; wi = w; pb = (fabsf(w.z) / PI_F) * (1.0f - frd);
#pragma unroll
                                for (int i = 0; i < 4; ++i) fb[i] = (albedo[i] * fabsf(w.z) / PI_F) * (1.0f - frd);
                            }
                        }
                    }
                    if (okb) {
                        sampled = true; specular = specb; wi_sh = to_world(nfw, wi);
                        if (thick > 0.0f) {
                            s_pdf = pb * (1.0f - cc_fc);
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                s_f[i] = fb[i] * (cc_attenuation1(cc_tint[i], thick, wo_nm.z) * cc_attenuation1(cc_tint[i], thick, wi_sh.z));   // Q14
                        } else {
                            s_pdf = pb;
#pragma unroll
                            for (int i = 0; i < 4; ++i) s_f[i] = fb[i];
                        }
                    }
                }
            }

            else if ((FEAT & FEAT_METAL) && mtype == MT_METAL) {
                // MetalMaterial::sample -> ConductorBsdf::sample (metal_material.rs:96-148, conductor.rs:257-329);
                // eta lives in albedo[], k in cc_tint[] for the light connection below
                nee_kind = 4;
                DevSpectrum es = load_spectrum(&mat->eta), ks = load_spectrum(&mat->cc_tint);
                eval_spectrum<STATS, false>(sc, es, wl, sf.uv, albedo, st);
                eval_spectrum<STATS, false>(sc, ks, wl, sf.uv, cc_tint, st);
                float rough_m = mat->roughness;
                if ((FEAT & FEAT_TEX) && mat->roughness_tex != 0xffffffffu) { float t3[3]; bilinear_rgb(sc, mat->roughness_tex, sf.uv, t3); rough_m = t3[0]; }
                const float alpha = rough_m * rough_m;                          // roughness_to_alpha :69-71
                cc_alpha_b = alpha;                                             // kept for the light connection
                if (wo_nm.z != 0.0f) {
                    f3 wi = mk3(0, 0, 1); bool okm = false;
                    if (alpha < 1e-3f) {                                        // sample_specular :276-298
                        wi = mk3(-wo_nm.x, -wo_nm.y, wo_nm.z);
                        okm = true; specular = true; s_pdf = 1.0f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) s_f[i] = fresnel_complex1(fabsf(wi.z), albedo[i], cc_tint[i]);
                    } else {                                                    // sample_microfacet_reflection :300-329
                        f3 wm = ggx_sample_wm(alpha, alpha, wo_nm, uv);
                        float wodm = dot(wo_nm, wm);
                        wi = wm * (2.0f * wodm) - wo_nm;
                        if (wo_nm.z * wi.z > 0.0f) {
                            okm = true;
                            float co = fabsf(wo_nm.z), ci = fabsf(wi.z);
                            if (co != 0.0f && ci != 0.0f) {                         // evaluate_torrance_sparrow :331-354
                                const float dd = ggx_D(alpha, alpha, wm), gg = ggx_G(alpha, alpha, wo_nm, wi);
#pragma unroll
                                for (int i = 0; i < 4; ++i) s_f[i] = fresnel_complex1(fabsf(wodm), albedo[i], cc_tint[i]) * dd * gg / (4.0f * co);
                            }
                            // pdf_microfacet recomputes the half vector from (wo, wi) (:414-439)
                            f3 h = wo_nm + wi;
                            s_pdf = 0.0f;
                            if (dot(h, h) != 0.0f) {
                                f3 hm = normalize(h);
                                float jac = 4.0f * fabsf(dot(wo_nm, hm));
                                if (jac != 0.0f) s_pdf = ggx_Dw(alpha, alpha, wo_nm, hm) / jac;
                            }
                        }
                    }
                    if (okm) {
                        f3 w = to_world(nfw, wi);
                        float gwi = dot(ng_t, w);
                        if (sgn1(gwi) == sgn1(geo_wo) && !(gwi != gwi) && !(geo_wo != geo_wo)) { sampled = true; wi_sh = w; }
                        else specular = false;
                    }
                }
            }

            // a specular sample skips the light connection (base_renderer.rs:218); failed samples are 'Diffuse' and do not
            if (specular) nee_kind = 0;
            if (STATS) tsb = __builtin_amdgcn_s_memtime();
            // NEE runs for every non-specular sample *including failed ones* (samples.rs:63-71, base_renderer.rs:218)
            if (nee_kind != 0u && prm.strategy != 0u) {
                // light pick: LightSampler (light_sampler.rs:26-43,190-220)
                // with one light any u picks it (light_sampler.rs:31-42): only the dimension advances
                float ul = 0.0f;
                if (!(FEAT & FEAT_MLIGHT) || sc.n_lights == 1u) smp.dimension += 1; else ul = get_1d(smp, sctx);
                // phi-weighted light pick.  The picked light's radiance is evaluated ONCE and doubles as its
                // phi weight (emissive radiance is never a texture, so it does not depend on uv).
                uint32_t pick = 0; float wsum = 0.0f, wpick = 0.0f;
                float lrad[4];
                if (!(FEAT & FEAT_MLIGHT) || sc.n_lights == 1u) {
                    const DevMaterial* lm0 = sc.materials + sc.lights[0].material;
                    DevSpectrum ls0 = load_spectrum(&lm0->color);
                    // (a textured radiance: phi at uv (0.5, 0.5), emissive_material.rs:61-79; the radiance itself follows at the sampled point)
                    eval_spectrum<STATS, (FEAT & FEAT_EMTEX) != 0, (FEAT & FEAT_EMTEX) != 0>(sc, ls0, wl, f2{0.5f, 0.5f}, lrad, st);
                    float sum = 0.0f, area = sc.lights[0].area_sum, inten = lm0->intensity_avg;
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum += (lrad[i] * inten) * area;
                    wsum = wpick = sum / 4.0f;
                } else {
                    for (uint32_t li = 0; li < sc.n_lights; ++li) {
                        DevLight lt = sc.lights[li];
                        const DevMaterial* lm = sc.materials + lt.material;
                        float ph[4];
                        DevSpectrum ls = load_spectrum(&lm->color);
                        eval_spectrum<false, (FEAT & FEAT_EMTEX) != 0, (FEAT & FEAT_EMTEX) != 0>(sc, ls, wl, f2{0.5f, 0.5f}, ph, st);
                        float sum = 0.0f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum += (ph[i] * lm->intensity_avg) * lt.area_sum;
                        wsum += sum / 4.0f;
                    }
                    float cum = 0.0f; bool chosen = false;
                    pick = sc.n_lights - 1;
                    for (uint32_t li = 0; li < sc.n_lights; ++li) {
                        DevLight lt = sc.lights[li];
                        const DevMaterial* lm = sc.materials + lt.material;
                        float ph[4];
                        DevSpectrum ls = load_spectrum(&lm->color);
                        eval_spectrum<false, (FEAT & FEAT_EMTEX) != 0, (FEAT & FEAT_EMTEX) != 0>(sc, ls, wl, f2{0.5f, 0.5f}, ph, st);
                        float sum = 0.0f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum += (ph[i] * lm->intensity_avg) * lt.area_sum;
                        float wt = sum / 4.0f;
                        cum += wt;
                        if (!chosen && (ul < cum / wsum || li == sc.n_lights - 1)) {
                            chosen = true; pick = li; wpick = wt;
#pragma unroll
                            for (int i = 0; i < 4; ++i) lrad[i] = ph[i];
                        }
                    }
                }
                if (sc.n_lights > 0 && wsum != 0.0f) {
                    float lprob = wpick / wsum;
                    float s1 = get_1d(smp, sctx);
                    f2 luv = get_2d(smp, sctx);
                    DevLight lt = sc.lights[pick];
                    const DevMaterial* lm = sc.materials + lt.material;
                    f3 dv, wi_r, ln = mk3(0, 0, 1);
                    float pdf_a = 1.0f, pdf_dir = 0.0f;
                    float dl_scale = 1.0f;                                      // delta lights: falloff
                    float l_inten = lm->intensity;                              // area lights: the emitter's intensity at the sampled point
                    float env_rad[4] = {0, 0, 0, 0};
                    if ((FEAT & FEAT_ENV) && lt.kind == LK_ENV) {               // sample_infinite_light (environment_light.rs:317-340)
                        const DevEnv& e = sc.envs[lt.first_tri];
                        env_sample(e, luv, wi_r, pdf_dir);
                        dv = wi_r;
                        env_radiance(sc, e, wi_r, wl, env_rad);
                    } else if ((FEAT & FEAT_DELTA) && lt.kind != LK_AREA) {
                        // Scene::calculate_light for PrimitiveDelta{Point,Directional}Light (scene.rs:114-139)
                        if (lt.kind == LK_DIRECTIONAL) {                        // directional_light.rs:95-107
                            dv = mk3(lt.pos[0], lt.pos[1], lt.pos[2]);
                            wi_r = normalize(dv);
                        } else {                                                // point_light.rs:83-93, spot_light.rs:99-122
                            dv = mk3(lt.pos[0], lt.pos[1], lt.pos[2]) - sf.p;
                            wi_r = normalize(dv);
                            if (lt.kind == LK_SPOT) {
                                float theta = lt.axis[0] * wi_r.x + lt.axis[1] * wi_r.y + lt.axis[2] * wi_r.z;
                                float t = fminf(fmaxf((theta - lt.angle_outer) / (lt.angle_inner - lt.angle_outer), 0.0f), 1.0f);
                                dl_scale = t * t * (3.0f - 2.0f * t);
                            }
                        }
                    } else {
                    // EmissiveTriangleMesh::sample_radiance (emissive_triangle_mesh.rs:176-308)
                    // first k with s < cdf[k] (else 0, :185-191).  The cdf is non-decreasing, so that index is the
                    // number of entries <= s: independent loads instead of a chain of dependent ones.
                    uint32_t cnt = 0;
                    for (uint32_t k = 0; k < lt.n_tris; ++k) cnt += (s1 < sc.light_tris[lt.first_tri + k].cdf) ? 0u : 1u;
                    uint32_t tsel = cnt < lt.n_tris ? cnt : 0u;
                    const float4* q = (const float4*)(sc.light_tris + lt.first_tri + tsel);
                    float4 qa = q[0], qb = q[1], qc = q[2], qd = q[3];
                    f3 p0 = mk3(qa.x, qa.y, qa.z), p1 = mk3(qa.w, qb.x, qb.y), p2 = mk3(qb.z, qb.w, qc.x);
                    float b0, b1;
                    if (luv.x < luv.y) { b0 = luv.x / 2.0f; b1 = luv.y - b0; } else { b1 = luv.y / 2.0f; b0 = luv.x - b1; }
                    float b2 = 1.0f - b0 - b1;
                    f3 lp = p0 * b0 + p1 * b1 + p2 * b2;
                    if ((FEAT & FEAT_EMTEX) && (lm->color.kind == SPK_TEXTURE || lm->metallic_tex != 0xffffffffu)) {    // EmissiveMaterial::radiance at the sampled point's uv (:48-59, emissive_triangle_mesh.rs:237-247)
                        const float* tu = sc.light_uvs + (size_t)(lt.first_tri + tsel) * 6;
                        const f2 suv = f2{tu[0] * b0 + tu[2] * b1 + tu[4] * b2, tu[1] * b0 + tu[3] * b1 + tu[5] * b2};
                        if (lm->color.kind == SPK_TEXTURE) {
                            DevSpectrum lsx = load_spectrum(&lm->color);
                            eval_spectrum<STATS, true, true>(sc, lsx, wl, suv, lrad, st);
                        }
                        if (lm->metallic_tex != 0xffffffffu) { float t3[3]; bilinear_rgb(sc, lm->metallic_tex, suv, t3); l_inten = t3[0]; }   // its intensity texture there
                    }
                    ln = mk3(qc.z, qc.w, qd.x);                                  // normalize(normalize(cross(p1 - p0, p2 - p0))), precomputed
                    pdf_a = 1.0f / lt.area_sum;
                    dv = lp - sf.p;
                    wi_r = normalize(dv);
                    float distance = length(lp - sf.p);
                    pdf_dir = pdf_a * (distance * distance) / fmaxf(fabsf(dot(ln, -wi_r)), 1e-8f);
                    }
                    // evaluate_area_light{,_with_mis} (common.rs:82-171)
                    f3 wi_t = to_local(fr, wi_r);
                    f3 wi_nm = to_local(nf, wi_t);
                    float fl[4] = {0, 0, 0, 0}; float pdf_b = 0.0f;
                    float gwi = dot(ng_t, wi_t);
                    if ((FEAT & FEAT_ROUGH) && nee_kind == 3u) {                 // DielectricBsdf::{evaluate_microfacet,pdf_microfacet} (:468-645)
                        const float alpha = C.d_alpha;
                        const float es = d_er[0];
                        // The reference connects to the light AFTER it sampled the BSDF (base_renderer.rs:203-228).  If that sample was a dispersive
                        // transmission it terminated the secondary wavelengths, and the evaluation's eta.sample(lambda) now returns (eta0, 0, 0, 0)
                        // (spectrum.rs:43-47): eta_rel is 0 in those lanes, their Fresnel term 1, and pr = (F0 + 3) / 4 in the pdf of the MIS weight.
                        // (Plastic builds its eta without the wavelengths — plastic_material.rs:107-109 — and keeps all four lanes.)
                        if (wl.term && !d_plastic) d_er[1] = d_er[2] = d_er[3] = 0.0f;
                        float co = wo_nm.z, ci = wi_nm.z;
                        bool refl = ci * co > 0.0f;
                        float etap = !refl ? (co > 0.0f ? es : 1.0f / es) : 1.0f;
                        f3 wm = wi_nm * etap + wo_nm;
                        bool ok = !(ci == 0.0f || co == 0.0f || dot(wm, wm) == 0.0f);
                        if (ok) {
                            wm = normalize(wm);
                            if (wm.z < 0.0f) wm = -wm;
                            if (dot(wm, wi_nm) * ci < 0.0f || dot(wm, wo_nm) * co < 0.0f) ok = false;
                        }
                        if (ok) {
                            float wodm = dot(wo_nm, wm), widm = dot(wi_nm, wm);
                            float fr4[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) fr4[i] = fresnel_dielectric1(fabsf(wodm), d_er[i]);
                            float pr = (((0.0f + fr4[0]) + fr4[1]) + fr4[2] + fr4[3]) / 4.0f, pt = 1.0f - pr;
                            float dg = ggx_D(alpha, alpha, wm) * ggx_G(alpha, alpha, wo_nm, wi_nm);
                            if (refl) {
                                pdf_b = ggx_Dw(alpha, alpha, wo_nm, wm) / (4.0f * fabsf(wodm)) * pr / (pr + pt);
#pragma unroll
                                for (int i = 0; i < 4; ++i) fl[i] = fr4[i] * dg / (4.0f * fabsf(wo_nm.z));
                            } else {
                                float sden = widm + wodm / es;
                                float denom = sden * sden;
                                pdf_b = d_thin ? pt / (pr + pt) : ggx_Dw(alpha, alpha, wo_nm, wm) * (fabsf(widm) / denom) * pt / (pr + pt);
#pragma unroll
                                for (int i = 0; i < 4; ++i) fl[i] = sdiv((1.0f - fr4[i]) * dg * fabsf(widm) * fabsf(wodm), denom * fabsf(wo_nm.z) * es * es);
                            }
                        }
                        if (d_plastic && dot(wi_nm, wo_nm) < 0.0f) {              // plastic_material.rs:169-172 (surface uv here)
                            float col[4];
                            DevSpectrum cs = load_spectrum(&mat->color);
                            eval_spectrum<STATS, (FEAT & FEAT_TEX) != 0>(sc, cs, wl, sf.uv, col, st);
#pragma unroll
                            for (int i = 0; i < 4; ++i) fl[i] = fl[i] * col[i];
                        }
                    } else if ((FEAT & FEAT_METAL) && nee_kind == 4u) {          // MetalMaterial::{evaluate,pdf} (metal_material.rs:150-229)
                        const float alpha = cc_alpha_b;
                        if (sgn1(gwi) == sgn1(geo_wo) && !(alpha < 1e-3f) && fabsf(wo_nm.z) != 0.0f && fabsf(wi_nm.z) != 0.0f && wo_nm.z * wi_nm.z > 0.0f) {
                            f3 h = wo_nm + wi_nm;
                            if (dot(h, h) != 0.0f) {
                                f3 wm = normalize(h);
                                float wodm = dot(wo_nm, wm);
                                const float dd = ggx_D(alpha, alpha, wm), gg = ggx_G(alpha, alpha, wo_nm, wi_nm);
#pragma unroll
                                for (int i = 0; i < 4; ++i) fl[i] = fresnel_complex1(fabsf(wodm), albedo[i], cc_tint[i]) * dd * gg / (4.0f * fabsf(wo_nm.z));
                                float jac = 4.0f * fabsf(wodm);
                                if (jac != 0.0f) pdf_b = ggx_Dw(alpha, alpha, wo_nm, wm) / jac;
                            }
                        }
                    } else if (!(FEAT & FEAT_CC) || nee_kind == 1u) {                                     // LambertMaterial::{evaluate,pdf} (lambert_material.rs:99-159)
                        if (sgn1(gwi) == sgn1(geo_wo) && wo_nm.z != 0.0f && wi_nm.z != 0.0f && sgn1(wo_nm.z) == sgn1(wi_nm.z)) {
                            // (lambert.rs: albedo * |cos| / PI.  PT_EXACT_DIV 0 multiplies by 1 / PI instead: <= 1 ulp apart)
                            pdf_b = PT_EXACT_DIV ? fabsf(wi_nm.z) / PI_F : fabsf(wi_nm.z) * INV_PI_F;
#pragma unroll
                            for (int i = 0; i < 4; ++i) fl[i] = PT_EXACT_DIV ? (albedo[i] * fabsf(wi_nm.z)) / PI_F : (albedo[i] * fabsf(wi_nm.z)) * INV_PI_F;
                        }
                    } else {                                                  // SimpleClearcoatPbrMaterial::{evaluate,pdf} (:261-433)
                        float dgc, p5c, pdfc, dgb, p5b, pdfb;
                        gs_eval_R(cc_alpha_c, wo_nm, wi_nm, dgc, p5c, pdfc);
                        gs_eval_R(cc_alpha_b, wo_nm, wi_nm, dgb, p5b, pdfb);
                        float metallic = cc_metallic;
                        // Lambert lobe of the dielectric base (lambert.rs:77-120)
                        float lam_f = 0.0f, lam_pdf = 0.0f;
                        if (wo_nm.z != 0.0f && wi_nm.z != 0.0f && sgn1(wo_nm.z) == sgn1(wi_nm.z)) { lam_f = fabsf(wi_nm.z) / PI_F; lam_pdf = lam_f; }
                        float frd = cc_r0d + (1.0f - cc_r0d) * schlick_p5(fabsf(wo_nm.z));     // fresnel(wo).average(), scalar r0
                        float pdf_met = pdfb, pdf_die = frd * pdfb + (1.0f - frd) * lam_pdf;
                        float pdf_base = metallic >= 1.0f ? pdf_met : (metallic <= 0.0f ? pdf_die : pdf_met * metallic + pdf_die * (1.0f - metallic));
                        float thick = cc_thick;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float f_met = (albedo[i] + (1.0f - albedo[i]) * p5b) * dgb;
                            float f_die = (cc_r0d + (1.0f - cc_r0d) * p5b) * dgb + (1.0f - frd) * (albedo[i] * lam_f);
                            float f_base = metallic >= 1.0f ? f_met : (metallic <= 0.0f ? f_die : f_met * metallic + f_die * (1.0f - metallic));
                            if (thick <= 0.0f) fl[i] = f_base;
                            else {
                                float att = cc_attenuation1(cc_tint[i], thick, wo_nm.z) * cc_attenuation1(cc_tint[i], thick, wi_nm.z);
                                fl[i] = ((cc_r0c + (1.0f - cc_r0c) * p5c) * dgc) * cc_fc + f_base * att * (1.0f - cc_fc);
                            }
                        }
                        pdf_b = thick <= 0.0f ? pdf_base : pdfc * cc_fc + pdf_base * (1.0f - cc_fc);
                    }
                    float dist2 = dot(dv, dv);
                    do_shadow = true;
                    if ((FEAT & FEAT_ENV) && lt.kind == LK_ENV) {               // evaluate_infinite_light{,_with_mis} (common.rs:174-241)
                        float wgt = prm.strategy == 2u ? balance_heuristic(pdf_dir, pdf_b) : 1.0f;
                        sh_d = wi_r; sh_o = sf.p + wi_r * SHADOW_EPS; sh_t = 3.402823466e+38f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) sh_c[i] = (T[i] * sdiv(fl[i] * env_rad[i], pdf_dir * lprob)) * wgt;
                    } else if ((FEAT & FEAT_DELTA) && lt.kind != LK_AREA) {
                        // evaluate_delta_{point,directional}_light (common.rs:23-79): no MIS weight
                        if (lt.kind == LK_DIRECTIONAL) {
                            sh_d = dv; sh_o = sf.p; sh_t = 3.402823466e+38f;      // the ray is not moved forward (:60-61)
#pragma unroll
                            for (int i = 0; i < 4; ++i) sh_c[i] = (T[i] * ((fl[i] * (lt.intensity * lrad[i])) / lprob)) * 1.0f;
                        } else {
                            sh_d = wi_r; sh_o = sf.p + wi_r * SHADOW_EPS; sh_t = length(dv) - 2.0f * SHADOW_EPS;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                float inten = lt.intensity * lrad[i];
                                if (lt.kind == LK_SPOT) inten = inten * dl_scale;
                                sh_c[i] = (T[i] * ((fl[i] * inten) / (dist2 * lprob))) * 1.0f;
                            }
                        }
                    } else {
                    f3 ln_t = normalize(to_local(fw, ln));
                    float g = fabsf(dot(ln_t, -wi_t)) / dist2;
                    float wgt = prm.strategy == 2u ? balance_heuristic(pdf_dir, pdf_b) : 1.0f;
                    sh_d = wi_r; sh_o = sf.p + wi_r * SHADOW_EPS; sh_t = length(dv) - 2.0f * SHADOW_EPS;
#ifdef PT_TRACE_MORTON
                    if (smp.morton == PT_TRACE_MORTON) printf("[gpu] nee depth=%u wo=(%.9g %.9g %.9g) wi=(%.9g %.9g %.9g) f=(%.9g %.9g %.9g %.9g) pdf_dir=%.9g pdf_bsdf=%.9g rad=(%.9g %.9g %.9g %.9g) g=%.9g den=%.9g T=(%.9g %.9g %.9g %.9g) alpha=%.9g kind=%u\n", P.depth, wo_nm.x, wo_nm.y, wo_nm.z,
                                                              wi_nm.x, wi_nm.y, wi_nm.z, fl[0], fl[1], fl[2], fl[3], pdf_dir, pdf_b, lrad[0] * l_inten, lrad[1] * l_inten, lrad[2] * l_inten, lrad[3] * l_inten, g, pdf_a * lprob, T[0], T[1], T[2], T[3], C.d_alpha, nee_kind);
#endif
                    const float rden = 1.0f / (pdf_a * lprob);   // PT_EXACT_DIV 0: one division for the four wavelengths (<= 1 ulp from x / (pdf_a * lprob))
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        sh_c[i] = (T[i] * (PT_EXACT_DIV ? (((fl[i] * (lrad[i] * l_inten)) * g) / (pdf_a * lprob)) : (((fl[i] * (lrad[i] * l_inten)) * g) * rden))) * wgt;
                    }
                }
            }

#ifdef PT_TRACE_MORTON
            if (smp.morton == PT_TRACE_MORTON) printf("[gpu] sample depth=%u sampled=%d spec=%d wo=(%.9g %.9g %.9g) wi_sh=(%.9g %.9g %.9g) f0=%.9g pdf=%.9g uc=%.9g uv=(%.9g %.9g) T0=%.9g\n", P.depth, (int)sampled, (int)specular,
                                                      wo_nm.x, wo_nm.y, wo_nm.z, wi_sh.x, wi_sh.y, wi_sh.z, s_f[0], s_pdf, uc, uv.x, uv.y, T[0]);
#endif
            if (!sampled) {
                end_path = true;                                              // process_bsdf_sampling -> None (:102-104,240-253)
            } else {
                // spawn the next ray (:106-121)
                f3 wi_r = to_world(fw, wi_sh);
                float sg = dot(sf.ng, wi_r) < 0.0f ? -1.0f : 1.0f;
                f3 org = sf.p + (sg * sf.ng) * RAY_EPS;
                rd = wi_r; ro = org + rd * RAY_EPS;
#pragma unroll
                for (int i = 0; i < 4; ++i) pf[i] = s_f[i];
                p_pdf = s_pdf; prev_spec = specular; prev_pos = sf.p; from_camera = false;
            }
        }
    }
    return end_path;
}

// Per-sample log (mi355pt_render_sample_log): when L != nullptr every finished path of the launch also writes its spectral radiance,
// wavelengths and wavelength pdfs to slot ((tile_k * 64 + pixel in tile) * n_s + (sample index - s_base)).  A wave-uniform branch at
// path end in the PRODUCTION kernel: the per-sample parity tests read what the benchmarked binary computed, in its own launch shape.
struct PathOut { float* L; float* lam; float* pdf; uint32_t s_base, n_s; };

PT_DEV void sample_log(const Path& P, const PathOut& pout, size_t slot) {
    const float pdf0 = 1.0f / (LAMBDA_MAX - LAMBDA_MIN);
    float lam[4];
    wl_lams(P.wl, lam);
    float4 l = make_float4(P.L[0], P.L[1], P.L[2], P.L[3]), w = make_float4(lam[0], lam[1], lam[2], lam[3]);
    float4 q = P.wl.term ? make_float4(pdf0 / 4.0f, 0.0f, 0.0f, 0.0f) : make_float4(pdf0, pdf0, pdf0, pdf0);   // sampled_spectrum.rs:346-365
    ((float4*)pout.L)[slot] = l; ((float4*)pout.lam)[slot] = w; ((float4*)pout.pdf)[slot] = q;
}

// Sensor::add_sample (sensor.rs:41-78): the RGB contribution of one finished path, added to the work item's LDS film tile
PT_DEV void film_rgb(const Path& P, const DevScene& sc, const DevParams& prm, float& r_out, float& g_out, float& b_out) {
    const Wl& wl = P.wl; const float* L = P.L;
    const float pdf0 = 1.0f / (LAMBDA_MAX - LAMBDA_MIN);
    float X = 0.0f, Y = 0.0f, Z = 0.0f;
    const float4* cmf = (const float4*)sc.cmf;
    float lam[4];
    wl_lams(wl, lam);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k == 0 || !wl.term) {
            int idx = (int)floorf(lam[k] - LAMBDA_MIN);
            if (idx == 470) idx = 0;
            // sensor.rs:52-66 divides by the wavelength pdf (1/470, or 1/1880 for a terminated sample) and by 4: the pdf is one of two
            // constants, so its reciprocal is folded at compile time (<= 1 ulp from L / pdf)
            const float inv_pdf = wl.term ? 1.0f / (pdf0 / 4.0f) : 1.0f / pdf0;
            float c = PT_EXACT_DIV ? (L[k] / (wl.term ? pdf0 / 4.0f : pdf0)) / 4.0f : (L[k] * inv_pdf) / 4.0f;
            float4 m = cmf[idx];
            X += c * m.x; Y += c * m.y; Z += c * m.z;
        }
    }
    const float* M = prm.xyz_to_rgb;   // row-major; glam Mat3*Vec3 = col0*x + col1*y + col2*z
    float r = M[0] * X + M[1] * Y + M[2] * Z;
    float g = M[3] * X + M[4] * Y + M[5] * Z;
    float b = M[6] * X + M[7] * Y + M[8] * Z;
    r_out = 0.0f + r * prm.exposure; g_out = 0.0f + g * prm.exposure; b_out = 0.0f + b * prm.exposure;
}

// The vertex as ONE call, for the kernels without the clearcoat material (see ShadeCtx above)
template <bool STATS, uint32_t FEAT>
PT_DEV bool shade_vertex(Path& P, const DevScene& sc, const DevParams& prm, const SamplerCtx& sctx, bool got, const Hit& hit, ShadowReq& sh,
                         StatCounters& st, unsigned long long& tsa, unsigned long long& tsb) {
    static_assert((FEAT & FEAT_CC) == 0u, "clearcoat kernels run the cooperative albedo estimate between the halves");
    ShadeCtx C;
    C.cont = false; C.need_cc = false; C.cc_fc = 0.0f;
    bool end_path = shade_vertex_head<STATS, FEAT>(P, sc, prm, sctx, got, hit, sh, st, tsa, C);
    if (C.cont) end_path = shade_vertex_tail<STATS, FEAT>(P, sc, prm, sctx, sh, st, tsb, C);
    return end_path;
}

}  // namespace pt
