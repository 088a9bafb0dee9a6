// Trowbridge-Reitz (GGX) microfacet helpers, GeneralizedSchlickBsdf in ScatterMode::R and the
// SimpleClearcoatPbrMaterial layering (scene/src/material/bsdf/generalized_schlick.rs,
// scene/src/material/impls/simple_pbr_clearcoat_material.rs) for gfx950.
// The clearcoat material only instantiates GeneralizedSchlick with mode R, entering, non-thin, tint = 1
// (simple_pbr_clearcoat_material.rs:121-133,445-456,...), so only those arms exist here; with tint = 1 the
// Lazanyi term of generalized_schlick.rs:104-114 is exactly zero.
#pragma once
#include "pt_device.hpp"

namespace pt {

PT_DEV float tan2_theta(f3 w) { float c2 = w.z * w.z; return c2 == 0.0f ? INFINITY : (1.0f - c2) / c2; }   // common.rs:19-26
PT_DEV float cos_phi(f3 w) { float st = sqrtf(fmaxf(1.0f - w.z * w.z, 0.0f)); return st == 0.0f ? 1.0f : fminf(fmaxf(w.x / st, -1.0f), 1.0f); }
PT_DEV float sin_phi(f3 w) { float st = sqrtf(fmaxf(1.0f - w.z * w.z, 0.0f)); return st == 0.0f ? 0.0f : fminf(fmaxf(w.y / st, -1.0f), 1.0f); }

// Every material of the reference passes alpha_x == alpha_y (glass_material.rs:120-126, metal_material.rs, the clearcoat layers).
// The anisotropic expressions then reduce ALGEBRAICALLY to (x^2 + y^2) / (z^2 alpha^2): the (1 - z^2) of tan^2 and the sin^2(theta) under
// cos / sin cancel, rounding error included — unless cos_phi / sin_phi CLAMP.  Near the normal 1 - z^2 is a difference of two numbers next
// to 1 (z^2 = 1 - 1.5e-6 has a spacing of 6e-8: 4 % of the result); when the rounded value comes out smaller than x^2, x / sin(theta) > 1 is
// clamped to 1 and the error no longer cancels: the reference's D then differs from the algebraic form by up to several per cent.  For a
// near-mirror metal (alpha = 0.0025: e = tan^2 / alpha^2 of order 1 at theta = 1e-3) that is a 1 ... 6 % step in f or pdf of a sampled
// direction, the throughput f / pdf lands on the other side of 1, the Russian-roulette gate draws (or not) and the path parts from the
// reference's (measured: 0.04 % of the samples of scene 7, round 3).  PT_GGX_EXACT = 1 (default) therefore restates the reference's
// expressions operation for operation; 0 keeps the algebraic forms of rounds 1-2 (two sqrt and four divisions fewer per D + Lambda).
#ifndef PT_GGX_EXACT
#define PT_GGX_EXACT 1
#endif
PT_DEV float tan2_theta_xy(f3 w) { float c2 = w.z * w.z; return c2 == 0.0f ? INFINITY : (w.x * w.x + w.y * w.y) / c2; }
PT_DEV float ggx_D(float ax, float ay, f3 wm) {                                   // dielectric.rs:35-47 = generalized_schlick.rs:119-131
    const float t2 = tan2_theta(wm);
    if (!isfinite(t2)) return 0.0f;
    float c2 = wm.z * wm.z, c4 = c2 * c2;
#if PT_GGX_EXACT
    const float st = sqrtf(fmaxf(1.0f - c2, 0.0f));
    const float cp = st == 0.0f ? 1.0f : fminf(fmaxf(wm.x / st, -1.0f), 1.0f), sp = st == 0.0f ? 0.0f : fminf(fmaxf(wm.y / st, -1.0f), 1.0f);
    float e = t2 * ((cp * cp) / (ax * ax) + (sp * sp) / (ay * ay));
#else
    float e = tan2_theta_xy(wm) / (ax * ay);
#endif
    return 1.0f / (PI_F * ax * ay * c4 * ((1.0f + e) * (1.0f + e)));
}
PT_DEV float ggx_lambda(float ax, float ay, f3 w) {                               // dielectric.rs:50-58 = :132-140
    const float t2 = tan2_theta(w);
    if (isinf(t2)) return 0.0f;
#if PT_GGX_EXACT
    const float st = sqrtf(fmaxf(1.0f - w.z * w.z, 0.0f));
    const float cp = st == 0.0f ? 1.0f : fminf(fmaxf(w.x / st, -1.0f), 1.0f), sp = st == 0.0f ? 0.0f : fminf(fmaxf(w.y / st, -1.0f), 1.0f);
    const float a = cp * ax, b = sp * ay;
    const float alpha2 = a * a + b * b;
    return (sqrtf(1.0f + alpha2 * t2) - 1.0f) / 2.0f;
#else
    return (sqrtf(1.0f + (ax * ay) * tan2_theta_xy(w)) - 1.0f) / 2.0f;
#endif
}
PT_DEV float ggx_G(float ax, float ay, f3 wo, f3 wi) { return 1.0f / (1.0f + ggx_lambda(ax, ay, wo) + ggx_lambda(ax, ay, wi)); }
PT_DEV float ggx_Dw(float ax, float ay, f3 w, f3 wm) {                            // :154-164
    float c = fabsf(w.z);
    if (c == 0.0f) return 0.0f;
    return (1.0f / (1.0f + ggx_lambda(ax, ay, w))) / c * ggx_D(ax, ay, wm) * fabsf(dot(w, wm));
}
// SINCOS: one sincosf instead of cosf + sinf (same values).  Measured per kernel, because at the 128-VGPR cap the register
// allocation decides more than the instruction count: +4.7 % in the clearcoat kernels (the 64-sample coat albedo, gs_sample_R),
// -3.5 % in the metal kernel — so only the coat asks for it.
template <bool SINCOS = false>
PT_DEV f3 ggx_sample_wm(float ax, float ay, f3 w, f2 u) {                          // :165-199 (PBRT-v4 Sample_wm, polar disk)
    f3 wh = normalize(mk3(ax * w.x, ay * w.y, w.z));
    if (wh.z < 0.0f) wh = -wh;
    f3 t1 = wh.z < 0.99999f ? normalize(cross(mk3(0, 0, 1), wh)) : mk3(1, 0, 0);
    f3 t2 = cross(wh, t1);
    float r = sqrtf(u.x), th = 2.0f * PI_F * u.y;
    float cs_th, sn_th;
    (void)SINCOS; ref_sincosf(th, &sn_th, &cs_th);
    float px = r * cs_th, pyy = r * sn_th;
    float h = sqrtf(fmaxf(1.0f - px * px, 0.0f));
    float lf = (1.0f + wh.z) / 2.0f;
    float py = h * (1.0f - lf) + pyy * lf;
    float pz = sqrtf(fmaxf(1.0f - px * px - py * py, 0.0f));
    f3 nh = t1 * px + t2 * py + wh * pz;
    return normalize(mk3(ax * nh.x, ay * nh.y, fmaxf(1e-6f, nh.z)));
}
// generalized_schlick_fresnel with exponent 5, r90 = 1, tint = 1: r0 + (1 - r0) (1 - c)^5    (:92-116)
PT_DEV float schlick_p5(float cos_theta) { float c = fminf(fmaxf(cos_theta, 0.0f), 1.0f); float o = 1.0f - c; float o2 = o * o; return o2 * o2 * o; }

// fresnel_complex for one wavelength lane (bsdf/conductor.rs:14-124): complex arithmetic spelled out like the reference
struct Cplx { float re, im; };
PT_DEV Cplx cmul(Cplx a, Cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
PT_DEV Cplx cdiv(Cplx a, Cplx b) {
    float den = b.re * b.re + b.im * b.im;
    if (den == 0.0f) return {0.0f, 0.0f};
    return {(a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den};
}
PT_DEV float fresnel_complex1(float cos_i, float eta, float k) {
    cos_i = fminf(fmaxf(cos_i, 0.0f), 1.0f);
    Cplx ce{eta, k};
    float sin2_i = 1.0f - cos_i * cos_i;
    Cplx s2t = cdiv(Cplx{sin2_i, 0.0f}, cmul(ce, ce));
    Cplx a{1.0f - s2t.re, 0.0f - s2t.im};
    float r = sqrtf(a.re * a.re + a.im * a.im), theta = atan2f(a.im, a.re);
    float sr = sqrtf(r), ht = theta * 0.5f;
    float sn_ht, cs_ht; ref_sincosf(ht, &sn_ht, &cs_ht);
    Cplx ct{sr * cs_ht, sr * sn_ht};
    Cplx ec{ce.re * cos_i, ce.im * cos_i};
    Cplx rp = cdiv(Cplx{ec.re - ct.re, ec.im - ct.im}, Cplx{ec.re + ct.re, ec.im + ct.im});
    Cplx et = cmul(ce, ct);
    Cplx rs = cdiv(Cplx{cos_i - et.re, 0.0f - et.im}, Cplx{cos_i + et.re, 0.0f + et.im});
    return ((rp.re * rp.re + rp.im * rp.im) + (rs.re * rs.re + rs.im * rs.im)) * 0.5f;
}

struct GsSample { f3 wi; float pdf; float dg; float p5; bool ok; bool specular; };   // f = F(p5) * dg
// GeneralizedSchlickBsdf::sample(.., ScatterMode::R) minus the Fresnel colour (:212-251, 342-372):
// returns wi, pdf, the colourless factor D*G/(4|cos o|) and (1-cos)^5 so the caller applies its r0.
PT_DEV GsSample gs_sample_R(float alpha, f3 wo, f2 uv) {
    GsSample s; s.ok = false; s.specular = false; s.wi = mk3(0, 0, 1); s.pdf = 0.0f; s.dg = 0.0f; s.p5 = 0.0f;
    if (wo.z == 0.0f) return s;
    if (alpha < 1e-3f) {                                                          // effectively_smooth -> sample_specular R
        s.wi = mk3(-wo.x, -wo.y, wo.z);
        if (s.wi.z == 0.0f) return s;
        s.p5 = schlick_p5(fabsf(wo.z)); s.dg = 1.0f; s.pdf = 1.0f; s.ok = true; s.specular = true;
        return s;
    }
    f3 wm = ggx_sample_wm<true>(alpha, alpha, wo, uv);
    float wodm = dot(wo, wm);
    f3 wi = wm * (2.0f * wodm) - wo;                                              // reflect (common.rs:59-64)
    if (!(wo.z * wi.z > 0.0f)) return s;
    float cd = fabsf(wodm);
    if (cd < 1e-6f) return s;
    float ci = fabsf(wi.z), co = fabsf(wo.z);
    float d = ggx_D(alpha, alpha, wm);
    s.pdf = ggx_Dw(alpha, alpha, wo, wm) / (4.0f * cd) * 1.0f;
    if (ci == 0.0f || co == 0.0f) return s;
    s.dg = d * ggx_G(alpha, alpha, wo, wi) / (4.0f * co);
    // note the reference multiplies (fresnel * d) * g / (4 co); the factor order differs by rounding only
    s.p5 = schlick_p5(cd); s.wi = wi; s.ok = true;
    return s;
}
// evaluate / pdf in mode R (:438-505, 640-700, 769-785): colourless factor + (1-cos)^5, and the pdf
PT_DEV void gs_eval_R(float alpha, f3 wo, f3 wi, float& dg, float& p5, float& pdf) {
    dg = 0.0f; p5 = 0.0f; pdf = 0.0f;
    if (alpha < 1e-3f) return;
    if (!(wo.z * wi.z > 0.0f)) return;                                            // same_hemisphere
    f3 wm = wo + wi;
    if (dot(wm, wm) == 0.0f) return;
    wm = normalize(wm);
    float co = fabsf(wo.z), ci = fabsf(wi.z);
    float wodm = fabsf(dot(wo, wm));
    float jac = 4.0f * wodm;
    if (jac != 0.0f) pdf = ggx_Dw(alpha, alpha, wo, wm) / jac;
    if (co == 0.0f || ci == 0.0f) return;
    dg = ggx_D(alpha, alpha, wm) * ggx_G(alpha, alpha, wo, wi) / (4.0f * co);
    p5 = schlick_p5(wodm);
}

// ---- the same sample + evaluation for the coat's directional-albedo ESTIMATE only (coat_albedo_term below) ----
// There the sampled direction never becomes a path direction: it is one quadrature point of a scalar factor, evaluated consistently
// at that point.  Hardware sin/cos (argument in revolutions, so sin(2 pi u) is one instruction), sqrt, rsq and rcp (~1 ulp) replace
// the correctly rounded forms: the estimate moves by ~1e-6 relative, the 64-sample loop loses a third of its instructions.
PT_DEV f3 est_normalize(f3 a) { return a * __builtin_amdgcn_rsqf(dot(a, a)); }
PT_DEV float est_lambda(float a2, f3 w) {
    float c2 = w.z * w.z;
    if (c2 == 0.0f) return 0.0f;
    float t2 = (w.x * w.x + w.y * w.y) * __builtin_amdgcn_rcpf(c2);
    return (__builtin_amdgcn_sqrtf(1.0f + a2 * t2) - 1.0f) * 0.5f;
}
PT_DEV float est_D(float a2, f3 wm) {
    float c2 = wm.z * wm.z;
    if (c2 == 0.0f) return 0.0f;
    float e = ((wm.x * wm.x + wm.y * wm.y) * __builtin_amdgcn_rcpf(c2)) * __builtin_amdgcn_rcpf(a2);
    return __builtin_amdgcn_rcpf(PI_F * a2 * (c2 * c2) * ((1.0f + e) * (1.0f + e)));
}
PT_DEV GsSample gs_sample_R_estimate(float alpha, f3 wo, f2 uv) {
    GsSample s; s.ok = false; s.specular = false; s.wi = mk3(0, 0, 1); s.pdf = 0.0f; s.dg = 0.0f; s.p5 = 0.0f;
    if (wo.z == 0.0f) return s;
    if (alpha < 1e-3f) {
        s.wi = mk3(-wo.x, -wo.y, wo.z);
        if (s.wi.z == 0.0f) return s;
        s.p5 = schlick_p5(fabsf(wo.z)); s.dg = 1.0f; s.pdf = 1.0f; s.ok = true; s.specular = true;
        return s;
    }
    const float a2 = alpha * alpha;
    // ggx_sample_wm
    f3 wh = est_normalize(mk3(alpha * wo.x, alpha * wo.y, wo.z));
    if (wh.z < 0.0f) wh = -wh;
    f3 t1 = wh.z < 0.99999f ? est_normalize(cross(mk3(0, 0, 1), wh)) : mk3(1, 0, 0);
    f3 t2 = cross(wh, t1);
    float r = __builtin_amdgcn_sqrtf(uv.x);
    float px = r * __builtin_amdgcn_cosf(uv.y), pyy = r * __builtin_amdgcn_sinf(uv.y);
    float h = __builtin_amdgcn_sqrtf(fmaxf(1.0f - px * px, 0.0f));
    float lf = (1.0f + wh.z) * 0.5f;
    float py = h * (1.0f - lf) + pyy * lf;
    float pz = __builtin_amdgcn_sqrtf(fmaxf(1.0f - px * px - py * py, 0.0f));
    f3 nh = t1 * px + t2 * py + wh * pz;
    f3 wm = est_normalize(mk3(alpha * nh.x, alpha * nh.y, fmaxf(1e-6f, nh.z)));
    // reflect + evaluate
    float wodm = dot(wo, wm);
    f3 wi = wm * (2.0f * wodm) - wo;
    if (!(wo.z * wi.z > 0.0f)) return s;
    float cd = fabsf(wodm);
    if (cd < 1e-6f) return s;
    float ci = fabsf(wi.z), co = fabsf(wo.z);
    float d = est_D(a2, wm);
    float lo = est_lambda(a2, wo);
    s.pdf = (__builtin_amdgcn_rcpf(1.0f + lo) * __builtin_amdgcn_rcpf(co)) * d * cd * __builtin_amdgcn_rcpf(4.0f * cd);
    if (ci == 0.0f || co == 0.0f) return s;
    s.dg = (d * __builtin_amdgcn_rcpf(1.0f + lo + est_lambda(a2, wi))) * __builtin_amdgcn_rcpf(4.0f * co);
    s.p5 = schlick_p5(cd); s.wi = wi; s.ok = true;
    return s;
}

// directional_albedo (:893-918): 64-sample Monte Carlo of f * |cos_i| / pdf for a *scalar* r0 (the coat).
// The reference seeds it from the thread RNG on every call; here it is one counter stream per path vertex (key),
// shared bit for bit with the oracle (oracle/o_materials.hpp McRng).
// One term of that estimate: sample k (0..63) of the stream `key` (the k-th triple of draws: uc unused, u, v).
PT_DEV float coat_albedo_term(float alpha, float r0, f3 wo, uint64_t key, uint32_t k) {
    uint64_t h1 = mix_bits(key + 0x632be59bd9b4e019ull * (uint64_t)(3u * k + 2u));
    uint64_t h2 = mix_bits(key + 0x632be59bd9b4e019ull * (uint64_t)(3u * k + 3u));
    f2 uv = f2{(float)(uint32_t)(h1 >> 40) * 5.9604644775390625e-8f, (float)(uint32_t)(h2 >> 40) * 5.9604644775390625e-8f};
    GsSample s = gs_sample_R_estimate(alpha, wo, uv);
    float term = 0.0f;
    if (s.ok) {
        float ci = fabsf(s.wi.z);
        float f = (r0 + (1.0f - r0) * s.p5) * s.dg;
        if (ci > 0.0f && s.pdf > 0.0f) term = (f * ci) * __builtin_amdgcn_rcpf(s.pdf);
    }
    return term;
}
// Wave-cooperative form (every lane of the wave must call it): for each lane that needs an estimate, the 64 lanes evaluate
// one sample each of that lane's stream and sum them with a 6-step xor butterfly = the balanced pairwise tree over the sample
// index (the oracle sums in the same order).  A wave pays 64 samples per requesting lane instead of 64 x 64 whenever any lane
// asks — with a clearcoat hero covering a fraction of the tile, most of the 64-sample loops used to run for a few lanes only.
PT_DEV float coat_directional_albedo_coop(bool need, float alpha, float r0, f3 wo, uint64_t key, uint32_t lane) {
    float result = 0.0f;
    unsigned long long m = __ballot(need);
    while (m != 0ull) {
        const int i = (int)__ffsll((long long)m) - 1;
        m &= m - 1ull;
        const float a = __shfl(alpha, i), r = __shfl(r0, i);
        const f3 w = mk3(__shfl(wo.x, i), __shfl(wo.y, i), __shfl(wo.z, i));
        const uint64_t kk = ((uint64_t)__shfl((uint32_t)(key >> 32), i) << 32) | (uint64_t)__shfl((uint32_t)key, i);
        float t = coat_albedo_term(a, r, w, kk, lane);
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) t = t + __shfl_xor(t, s);
        if ((int)lane == i) result = t / 64.0f;
    }
    return result;
}
// compute_attenuation (simple_pbr_clearcoat_material.rs:88-107) for one wavelength lane
PT_DEV float cc_attenuation1(float tint, float thickness, float cos_theta) {
    float log_tint = logf(fmaxf(tint, 1e-10f));
    float sigma = (-1.0f * log_tint) / 0.001f;
    float l = (thickness * 0.001f) / fmaxf(cos_theta, 1e-4f);
    return expf((-1.0f * sigma) * l);
}

}  // namespace pt
