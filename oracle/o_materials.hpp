// ORACLE — TEST INFRASTRUCTURE ONLY.
// Restatement of scene/src/material/{common,edf}.rs, bsdf/{lambert,dielectric}.rs and
// impls/{lambert,emissive,glass,plastic}_material.rs.
#pragma once
#include "o_sampler.hpp"
#include "o_scene.hpp"

namespace oracle {



enum BsdfSampleType : uint32_t { ST_DIFFUSE = 0, ST_SPEC_REFL = 1, ST_SPEC_TRANS = 2, ST_GLOSSY_REFL = 3, ST_GLOSSY_TRANS = 4 };

struct MaterialSample {     // samples.rs:35-91
    SS f = SS::zero(); V3 wi{0, 0, 1}; float pdf = 0.0f; uint32_t sample_type = ST_DIFFUSE; bool is_sampled = false;
    bool is_specular() const { return sample_type == ST_SPEC_REFL || sample_type == ST_SPEC_TRANS; }
    bool is_non_specular() const { return !is_specular(); }   // Diffuse | GlossyReflection | GlossyTransmission
};
struct BsdfSample { SS f; V3 wi; float pdf; uint32_t type; };

// ---------------- material/common.rs ----------------
static inline float cos2_theta(V3 w) { return w.z * w.z; }
static inline float abs_cos_theta(V3 w) { return std::fabs(w.z); }
static inline float tan2_theta(V3 w) {
    float c2 = cos2_theta(w);
    return c2 == 0.0f ? std::numeric_limits<float>::infinity() : (1.0f - c2) / c2;
}
static inline float cos_phi(V3 w) {
    float st = std::sqrt(std::fmax(1.0f - cos2_theta(w), 0.0f));
    return st == 0.0f ? 1.0f : clampf(w.x / st, -1.0f, 1.0f);
}
static inline float sin_phi(V3 w) {
    float st = std::sqrt(std::fmax(1.0f - cos2_theta(w), 0.0f));
    return st == 0.0f ? 0.0f : clampf(w.y / st, -1.0f, 1.0f);
}
static inline V3 reflect(V3 wo, V3 n) { return n * (2.0f * dot(wo, n)) - wo; }
static inline bool same_hemisphere(V3 a, V3 b) { return a.z * b.z > 0.0f; }
static inline V2 sample_uniform_disk_polar(V2 u) {
    float r = std::sqrt(u.x), th = 2.0f * PI_F * u.y;
    return V2{r * std::cos(th), r * std::sin(th)};
}
static inline SS fresnel_dielectric(float cos_theta_i, SS eta) {          // common.rs:87-105
    cos_theta_i = clampf(cos_theta_i, 0.0f, 1.0f);
    float sin2_i = 1.0f - cos_theta_i * cos_theta_i;
    SS sin2_t = SS::constant(sin2_i) / (eta * eta);
    SS cos_t = ss_sqrt(ss_clamp(SS::one() - sin2_t, 0.0f, 1.0f));
    SS ci = SS::constant(cos_theta_i);
    SS r_parl = (eta * ci - cos_t) / (eta * ci + cos_t);
    SS r_perp = (ci - eta * cos_t) / (ci + eta * cos_t);
    return (r_parl * r_parl + r_perp * r_perp) * 0.5f;
}
static inline bool refract(V3 wi, V3 n, float eta, V3* wt_out) {          // common.rs:117-139
    float cos_i = dot(n, wi);
    float sin2_i = std::fmax(1.0f - cos_i * cos_i, 0.0f);
    float sin2_t = sin2_i / (eta * eta);
    if (sin2_t >= 1.0f) return false;
    float cos_t = std::sqrt(std::fmax(1.0f - sin2_t, 0.0f));
    V3 wt = (-wi) / eta + n * (cos_i / eta - cos_t);
    if (length_squared(wt) < 1e-12f) return false;
    *wt_out = normalize(wt);
    return true;
}

// ---------------- NormalizedLambertBsdf (bsdf/lambert.rs) ----------------
static inline V3 sample_cosine_hemisphere(V2 uv) {
    float r = std::sqrt(uv.x), th = 2.0f * PI_F * uv.y;
    return V3{r * std::cos(th), r * std::sin(th), std::sqrt(1.0f - uv.x)};
}
static inline bool lambert_sample(SS albedo, V3 wo, V2 uv, BsdfSample* out) {
    if (wo.z == 0.0f) return false;
    V3 wi = sample_cosine_hemisphere(uv);
    if (wo.z < 0.0f) wi = V3{wi.x, wi.y, -wi.z};
    if (wi.z == 0.0f) return false;
    if (signum(wo.z) != signum(wi.z)) return false;
    out->f = albedo * std::fabs(wi.z) / PI_F;
    out->wi = wi; out->pdf = std::fabs(wi.z) / PI_F; out->type = ST_DIFFUSE;
    return true;
}
static inline SS lambert_evaluate(SS albedo, V3 wo, V3 wi) {
    if (wo.z == 0.0f || wi.z == 0.0f) return SS::zero();
    if (signum(wo.z) != signum(wi.z)) return SS::zero();
    return albedo * std::fabs(wi.z) / PI_F;
}
static inline float lambert_pdf(V3 wo, V3 wi) {
    if (wo.z == 0.0f || wi.z == 0.0f) return 0.0f;
    if (signum(wo.z) != signum(wi.z)) return 0.0f;
    return std::fabs(wi.z) / PI_F;
}

// ---------------- DielectricBsdf (bsdf/dielectric.rs) ----------------
struct DielectricBsdf {
    SS eta; bool entering, thin; float ax, ay;
    DielectricBsdf(SS e, bool ent, bool th, float a_x, float a_y) : eta(e), entering(ent), thin(th), ax(a_x), ay(a_y) {
        if (eta.v[0] == 0.0f) eta = SS::constant(1.0f);                   // :144-148
    }
    bool effectively_smooth() const { return std::fmax(ax, ay) < 1e-3f; }
    float D(V3 wm) const {                                                // :29-41
        float t2 = tan2_theta(wm);
        if (!std::isfinite(t2)) return 0.0f;
        float c4 = cos2_theta(wm) * cos2_theta(wm);
        float cp = cos_phi(wm), sp = sin_phi(wm);
        float e = t2 * ((cp * cp) / (ax * ax) + (sp * sp) / (ay * ay));
        return 1.0f / (PI_F * ax * ay * c4 * ((1.0f + e) * (1.0f + e)));
    }
    float lambda(V3 w) const {                                            // :43-51
        float t2 = tan2_theta(w);
        if (std::isinf(t2)) return 0.0f;
        float a = cos_phi(w) * ax, b = sin_phi(w) * ay;
        float alpha2 = a * a + b * b;
        return (std::sqrt(1.0f + alpha2 * t2) - 1.0f) / 2.0f;
    }
    float G1(V3 w) const { return 1.0f / (1.0f + lambda(w)); }
    float G(V3 wo, V3 wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
    float Dw(V3 w, V3 wm) const {                                         // :65-75
        float c = std::fabs(w.z);
        if (c == 0.0f) return 0.0f;
        return G1(w) / c * D(wm) * std::fabs(dot(w, wm));
    }
    V3 sample_wm(V3 w, V2 u) const {                                      // :77-112
        V3 wh = normalize(V3{ax * w.x, ay * w.y, w.z});
        if (wh.z < 0.0f) wh = -wh;
        V3 t1 = wh.z < 0.99999f ? normalize(cross(V3{0, 0, 1}, wh)) : V3{1, 0, 0};
        V3 t2 = cross(wh, t1);
        V2 p = sample_uniform_disk_polar(u);
        float h = std::sqrt(std::fmax(1.0f - p.x * p.x, 0.0f));
        float lf = (1.0f + wh.z) / 2.0f;
        float py = h * (1.0f - lf) + p.y * lf;
        float pz = std::sqrt(std::fmax(1.0f - p.x * p.x - py * py, 0.0f));
        V3 nh = t1 * p.x + t2 * py + wh * pz;
        return normalize(V3{ax * nh.x, ay * nh.y, std::fmax(1e-6f, nh.z)});
    }
    SS eta_rel() const { return (thin || entering) ? eta : SS::one() / eta; }
    static void thin_coeffs(float fr, float* r_out, float* t_out) {        // :367-378
        float r = fr, t = 1.0f - r, r2 = r * r;
        r = r2 > 1.0f ? 1.0f : r + (t * t * r) / (1.0f - r2);
        *r_out = r; *t_out = t;
    }
    bool half_vector(V3 wo, V3 wi, float e, V3* wm_out) const {            // :184-215
        float co = wo.z, ci = wi.z;
        bool refl = ci * co > 0.0f;
        float etap = !refl ? (co > 0.0f ? e : 1.0f / e) : 1.0f;
        V3 wm = wi * etap + wo;
        if (ci == 0.0f || co == 0.0f || length_squared(wm) == 0.0f) return false;
        wm = normalize(wm);
        if (wm.z < 0.0f) wm = -wm;
        if (dot(wm, wi) * ci < 0.0f || dot(wm, wo) * co < 0.0f) return false;
        *wm_out = wm; return true;
    }
    bool sample_specular(V3 wo, float uc, Wavelengths& wl, BsdfSample* out) const {   // :380-466
        float wo_cos = wo.z;
        V3 n = entering ? V3{0, 0, 1} : V3{0, 0, -1};
        SS er = eta_rel();
        float etap = er.v[0];
        SS fr = fresnel_dielectric(std::fabs(wo_cos), er);
        if (thin) {
            float pr, pt; thin_coeffs(fr.average(), &pr, &pt);
            if (uc < pr / (pr + pt)) {
                if (std::fabs(wo_cos) < 1e-6f) return false;
                *out = BsdfSample{fr, V3{-wo.x, -wo.y, wo.z}, pr / (pr + pt), ST_SPEC_REFL};
                return true;
            }
            V3 wi{-wo.x, -wo.y, -wo.z};
            if (wi.z == 0.0f) return false;
            *out = BsdfSample{SS::one() - fr, wi, pt / (pr + pt), ST_SPEC_TRANS};
            return true;
        }
        float pr = fr.average(), pt = 1.0f - pr;
        if (uc < pr / (pr + pt)) {
            if (std::fabs(wo_cos) < 1e-6f) return false;
            *out = BsdfSample{fr, V3{-wo.x, -wo.y, wo.z}, pr / (pr + pt), ST_SPEC_REFL};
            return true;
        }
        if (!eta.is_constant()) wl.terminate_secondary();
        V3 wt;
        if (!refract(wo, n, etap, &wt)) return false;
        if (wt.z == 0.0f) return false;
        SS f = (SS::one() - fr) / (etap * etap);
        *out = BsdfSample{f, wt, pt / (pr + pt), ST_SPEC_TRANS};
        return true;
    }
    bool sample_microfacet(V3 wo, V2 u, float uc, Wavelengths& wl, BsdfSample* out) const {   // :217-282
        V3 wm = sample_wm(wo, u);
        SS er = eta_rel();
        float es = er.v[0];
        SS fr = fresnel_dielectric(std::fabs(dot(wo, wm)), er);
        float pr = fr.average(), pt = 1.0f - pr;
        if (thin) {
            float tr, tt; thin_coeffs(fr.average(), &tr, &tt);
            if (uc < tr / (tr + tt)) return sample_mf_reflection(wo, wm, fr, tr / (tr + tt), out);
            *out = BsdfSample{SS::one() - fr, V3{-wo.x, -wo.y, -wo.z}, tt / (tr + tt), ST_GLOSSY_TRANS};   // :346-365
            return true;
        }
        if (uc < pr / (pr + pt)) return sample_mf_reflection(wo, wm, fr, pr / (pr + pt), out);
        if (!eta.is_constant()) wl.terminate_secondary();
        return sample_mf_transmission(wo, wm, SS::one() - fr, pt / (pr + pt), es, out);
    }
    bool sample_mf_reflection(V3 wo, V3 wm, SS fr, float prob, BsdfSample* out) const {       // :284-314
        V3 wi = reflect(wo, wm);
        if (!same_hemisphere(wo, wi)) return false;
        float cd = std::fabs(dot(wo, wm));
        if (cd < 1e-6f) return false;
        float pdf = Dw(wo, wm) / (4.0f * cd) * prob;
        float d = D(wm), g = G(wo, wi);
        SS f = fr * d * g * abs_cos_theta(wi) / (4.0f * abs_cos_theta(wo));   // extra |cos wi| (Q8)
        *out = BsdfSample{f, wi, pdf, ST_GLOSSY_REFL};
        return true;
    }
    bool sample_mf_transmission(V3 wo, V3 wm, SS tr, float prob, float etap, BsdfSample* out) const {   // :316-344
        V3 wmr = entering ? wm : -wm;
        V3 wi;
        if (!refract(wo, wmr, etap, &wi)) return false;
        if (same_hemisphere(wo, wi) || std::fabs(wi.z) == 0.0f) return false;
        float s = dot(wi, wm) + dot(wo, wm) / etap;
        float denom = s * s;
        float dwm_dwi = std::fabs(dot(wi, wm)) / denom;
        float pdf = Dw(wo, wm) * dwm_dwi * prob;
        float d = D(wm), g = G(wo, wi);
        SS ft = tr * d * g * std::fabs(dot(wi, wm)) * std::fabs(dot(wo, wm)) / (denom * abs_cos_theta(wo) * etap * etap);
        *out = BsdfSample{ft, wi, pdf, ST_GLOSSY_TRANS};
        return true;
    }
    bool sample(V3 wo, V2 uv, float uc, Wavelengths& wl, BsdfSample* out) const {            // :164-182
        if (wo.z == 0.0f) return false;
        if (effectively_smooth()) return sample_specular(wo, uc, wl, out);
        return sample_microfacet(wo, uv, uc, wl, out);
    }
    SS evaluate(V3 wo, V3 wi) const {                                       // :468-537
        if (effectively_smooth()) return SS::zero();
        SS er = eta_rel(); float es = er.v[0];
        V3 wm;
        if (!half_vector(wo, wi, es, &wm)) return SS::zero();
        SS fr = fresnel_dielectric(std::fabs(dot(wo, wm)), er);
        bool refl = wi.z * wo.z > 0.0f;
        float d = D(wm), g = G(wo, wi);
        if (refl) return fr * d * g / (4.0f * abs_cos_theta(wo));
        float s = dot(wi, wm) + dot(wo, wm) / es;
        float denom = s * s;
        return (SS::one() - fr) * d * g * std::fabs(dot(wi, wm)) * std::fabs(dot(wo, wm)) / (denom * abs_cos_theta(wo) * es * es);
    }
    float pdf(V3 wo, V3 wi) const {                                         // :483-645
        if (effectively_smooth()) return 0.0f;
        SS er = eta_rel(); float es = er.v[0];
        V3 wm;
        if (!half_vector(wo, wi, es, &wm)) return 0.0f;
        SS fr = fresnel_dielectric(std::fabs(dot(wo, wm)), er);
        float pr = fr.average(), pt = 1.0f - pr;
        bool refl = wi.z * wo.z > 0.0f;
        if (refl) return Dw(wo, wm) / (4.0f * std::fabs(dot(wo, wm))) * pr / (pr + pt);
        if (thin) return pt / (pr + pt);
        float s = dot(wi, wm) + dot(wo, wm) / es;
        float denom = s * s;
        float dwm_dwi = std::fabs(dot(wi, wm)) / denom;
        return Dw(wo, wm) * dwm_dwi * pt / (pr + pt);
    }
};


// ---------------- ConductorBsdf (bsdf/conductor.rs) ----------------
struct Cplx { float re, im; };                                                    // :14-84
static inline Cplx cadd(Cplx a, Cplx b) { return {a.re + b.re, a.im + b.im}; }
static inline Cplx csub(Cplx a, Cplx b) { return {a.re - b.re, a.im - b.im}; }
static inline Cplx cmul(Cplx a, Cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
static inline Cplx cscale(Cplx a, float s) { return {a.re * s, a.im * s}; }
static inline Cplx cdiv(Cplx a, Cplx b) {
    float den = b.re * b.re + b.im * b.im;
    if (den == 0.0f) return {0.0f, 0.0f};
    return {(a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den};
}
static inline float cnorm(Cplx a) { return a.re * a.re + a.im * a.im; }           // |z|^2
static inline Cplx csqrt(Cplx a) {                                                // polar form, :29-36
    float r = std::sqrt(a.re * a.re + a.im * a.im);
    float theta = std::atan2(a.im, a.re);
    float sr = std::sqrt(r), ht = theta * 0.5f;
    return {sr * std::cos(ht), sr * std::sin(ht)};
}
static inline SS fresnel_complex(float cos_i, SS eta, SS k) {                      // :92-124
    cos_i = std::fmin(std::fmax(cos_i, 0.0f), 1.0f);
    SS out;
    for (int i = 0; i < 4; ++i) {
        Cplx ce{eta.v[i], k.v[i]};
        float sin2_i = 1.0f - cos_i * cos_i;
        Cplx sin2_t = cdiv(Cplx{sin2_i, 0.0f}, cmul(ce, ce));
        Cplx cos_t = csqrt(csub(Cplx{1.0f, 0.0f}, sin2_t));
        Cplx r_parl = cdiv(csub(cscale(ce, cos_i), cos_t), cadd(cscale(ce, cos_i), cos_t));
        Cplx r_perp = cdiv(csub(Cplx{cos_i, 0.0f}, cmul(ce, cos_t)), cadd(Cplx{cos_i, 0.0f}, cmul(ce, cos_t)));
        out.v[i] = (cnorm(r_parl) + cnorm(r_perp)) * 0.5f;
    }
    return out;
}
struct ConductorBsdf {
    SS eta, k; float ax, ay;
    DielectricBsdf ggx() const { return DielectricBsdf(SS::one(), true, false, ax, ay); }   // same Trowbridge-Reitz code (:151-250)
    bool effectively_smooth() const { return std::fmax(ax, ay) < 1e-3f; }
    static bool half_vec(V3 wo, V3 wi, V3* wm) {                                  // common.rs:47-57
        V3 h = wo + wi;
        if (length_squared(h) == 0.0f) return false;
        *wm = normalize(h); return true;
    }
    SS torrance_sparrow(V3 wo, V3 wi, V3 wm) const {                              // :331-354
        float co = std::fabs(wo.z), ci = std::fabs(wi.z);
        if (co == 0.0f || ci == 0.0f) return SS::zero();
        SS fr = fresnel_complex(std::fabs(dot(wo, wm)), eta, k);
        DielectricBsdf g = ggx();
        return fr * g.D(wm) * g.G(wo, wi) / (4.0f * co);
    }
    float pdf_microfacet(V3 wo, V3 wi) const {                                    // :414-439
        if (!same_hemisphere(wo, wi)) return 0.0f;
        V3 wm;
        if (!half_vec(wo, wi, &wm)) return 0.0f;
        float vis = ggx().Dw(wo, wm);
        float jac = 4.0f * std::fabs(dot(wo, wm));
        if (jac == 0.0f) return 0.0f;
        return vis / jac;
    }
    bool sample(V3 wo, V2 uv, BsdfSample* out) const {                            // :257-329
        if (wo.z == 0.0f) return false;
        if (effectively_smooth()) {
            V3 wi{-wo.x, -wo.y, wo.z};
            if (wi.z == 0.0f) return false;
            *out = BsdfSample{fresnel_complex(std::fabs(wi.z), eta, k), wi, 1.0f, ST_SPEC_REFL};
            return true;
        }
        V3 wm = ggx().sample_wm(wo, uv);
        V3 wi = reflect(wo, wm);
        if (!same_hemisphere(wo, wi)) return false;
        *out = BsdfSample{torrance_sparrow(wo, wi, wm), wi, pdf_microfacet(wo, wi), ST_GLOSSY_REFL};
        return true;
    }
    SS evaluate(V3 wo, V3 wi) const {                                             // :356-395
        if (effectively_smooth()) return SS::zero();
        if (std::fabs(wo.z) == 0.0f || std::fabs(wi.z) == 0.0f) return SS::zero();
        if (!same_hemisphere(wo, wi)) return SS::zero();
        V3 wm;
        if (!half_vec(wo, wi, &wm)) return SS::zero();
        return torrance_sparrow(wo, wi, wm);
    }
    float pdf(V3 wo, V3 wi) const { return effectively_smooth() ? 0.0f : pdf_microfacet(wo, wi); }   // :397-412
};

// ---------------- GeneralizedSchlickBsdf, ScatterMode::R (bsdf/generalized_schlick.rs) ----------------
// The clearcoat material only ever instantiates it with ScatterMode::R, entering = true, thin = false
// (simple_pbr_clearcoat_material.rs:121-133,445-456,...), so only the reflection arms are restated.
struct McRng {              // stands in for rand::rng() inside directional_albedo (generalized_schlick.rs:901): a counter
    uint64_t key; uint32_t n = 0;   // stream keyed per shading vertex, shared bit for bit with the HIP kernel
    float next() { uint64_t h = mix_bits(key + 0x632be59bd9b4e019ull * (uint64_t)(++n)); return (float)(uint32_t)(h >> 40) * 5.9604644775390625e-8f; }
};
struct GenSchlickBsdf {
    SS r0, r90; float exponent; SS tint; float ax, ay;
    bool effectively_smooth() const { return std::fmax(ax, ay) < 1e-3f; }
    SS fresnel(float cos_theta) const {                                       // :92-116
        cos_theta = clampf(cos_theta, 0.0f, 1.0f);
        float omc = 1.0f - cos_theta;
        const float CMAX = 1.0f / 7.0f, OMCMAX = 1.0f - CMAX;
        SS base = r0 + (r90 - r0) * std::pow(omc, exponent);
        SS at_max = r0 + (r90 - r0) * std::pow(OMCMAX, exponent);
        float om6 = OMCMAX * OMCMAX * OMCMAX * OMCMAX * OMCMAX * OMCMAX;     // powi(6)
        SS a = at_max * (SS::one() - tint) / (CMAX * om6);
        float o6 = omc * omc * omc * omc * omc * omc;
        SS laz = a * cos_theta * o6;
        return base - laz;
    }
    // same GGX helpers as DielectricBsdf (:119-210)
    DielectricBsdf ggx() const { return DielectricBsdf(SS::one(), true, false, ax, ay); }
    bool sample_R(V3 wo, V2 uv, BsdfSample* out) const {                       // sample(.., ScatterMode::R) :212-229,232-251,322-333
        if (wo.z == 0.0f) return false;
        if (effectively_smooth()) {
            SS f = fresnel(std::fabs(wo.z));
            V3 wi{-wo.x, -wo.y, wo.z};
            if (wi.z == 0.0f) return false;
            *out = BsdfSample{f, wi, 1.0f, ST_SPEC_REFL};
            return true;
        }
        DielectricBsdf g = ggx();
        V3 wm = g.sample_wm(wo, uv);
        SS fr = fresnel(std::fabs(dot(wo, wm)));
        V3 wi = reflect(wo, wm);                                              // sample_microfacet_reflection :342-372
        if (!same_hemisphere(wo, wi)) return false;
        float cd = std::fabs(dot(wo, wm));
        if (cd < 1e-6f) return false;
        float pdf = g.Dw(wo, wm) / (4.0f * cd) * 1.0f;
        float d = g.D(wm), gg = g.G(wo, wi);
        float ci = std::fabs(wi.z), co = std::fabs(wo.z);
        if (ci == 0.0f || co == 0.0f) return false;
        *out = BsdfSample{fr * d * gg / (4.0f * co), wi, pdf, ST_GLOSSY_REFL};
        return true;
    }
    SS evaluate_R(V3 wo, V3 wi) const {                                        // :438-505
        if (effectively_smooth()) return SS::zero();
        float co = std::fabs(wo.z), ci = std::fabs(wi.z);
        if (co == 0.0f || ci == 0.0f) return SS::zero();
        if (!same_hemisphere(wo, wi)) return SS::zero();
        V3 wm = wo + wi;                                                      // common.rs:47-57 half_vector
        if (length_squared(wm) == 0.0f) return SS::zero();
        wm = normalize(wm);
        DielectricBsdf g = ggx();
        SS fr = fresnel(std::fabs(dot(wo, wm)));
        return fr * g.D(wm) * g.G(wo, wi) / (4.0f * co);
    }
    float pdf_R(V3 wo, V3 wi) const {                                          // :640-700,769-785
        if (effectively_smooth()) return 0.0f;
        if (!same_hemisphere(wo, wi)) return 0.0f;
        V3 wm = wo + wi;
        if (length_squared(wm) == 0.0f) return 0.0f;
        wm = normalize(wm);
        DielectricBsdf g = ggx();
        float jac = 4.0f * std::fabs(dot(wo, wm));
        if (jac == 0.0f) return 0.0f;
        return g.Dw(wo, wm) / jac;
    }
    // directional_albedo (:893-918): 64-sample Monte Carlo of f * |cos_i| / pdf
    // The 64 terms are summed as a balanced pairwise tree over the sample index (t[i] += t[i ^ 1], ^2, ^4, ...): the order in
    // which the HIP kernel's 64 lanes combine them with an xor butterfly, so both sides agree bit for bit.
    SS directional_albedo(V3 wo, uint64_t key) const {
        SS term[64];
        McRng rng{key};
        for (int k = 0; k < 64; ++k) {
            float uc = rng.next(); (void)uc;
            V2 uv{0, 0}; uv.x = rng.next(); uv.y = rng.next();
            BsdfSample s;
            term[k] = SS::zero();
            if (sample_R(wo, uv, &s)) {
                float ci = std::fabs(s.wi.z);
                if (ci > 0.0f && s.pdf > 0.0f) term[k] = s.f * ci / s.pdf;
            }
        }
        for (int m = 1; m < 64; m <<= 1) {
            SS next[64];
            for (int i = 0; i < 64; ++i) next[i] = term[i] + term[i ^ m];
            for (int i = 0; i < 64; ++i) term[i] = next[i];
        }
        return term[0] / 64.0f;
    }
};

// ---------------- material dispatch (BsdfSurfaceMaterial impls) ----------------
struct ShadingPoint {   // SurfaceInteraction<VertexNormalTangent>
    V3 normal;          // geometric normal in the vertex-normal tangent frame
    V2 uv;
};

struct MaterialEval {
    const Scene& scene; Counters* ctr;
    uint64_t mc_key = 0;   // per-vertex key of the clearcoat's inner Monte-Carlo stream (see McRng)

    M4 normal_map_transform(const Material& m, V2 uv) const {
        V3 nm = m.normal_tex >= 0 ? sample_normal_map(scene.textures[m.normal_tex], m.normal_flip_y, uv) : V3{0, 0, 1};
        return from_normal_map(nm);
    }
    SS eta_of(const Material& m, const Wavelengths& wl) const { return m.eta.sample(wl); }

    MaterialSample sample(const Material& m, float uc, V2 uv, Wavelengths& wl, V3 wo, const ShadingPoint& sp) const;
    SS evaluate(const Material& m, const Wavelengths& wl, V3 wo, V3 wi, const ShadingPoint& sp) const;
    float pdf(const Material& m, const Wavelengths& wl, V3 wo, V3 wi, const ShadingPoint& sp) const;
    // EmissiveMaterial::radiance (emissive_material.rs:48-60)
    SS emissive_radiance(const Material& m, const Wavelengths& wl, V2 uv) const {
        SS rad = scene.sample_spectrum_param(m.color, uv, wl, ctr);
        return rad * scene.sample_float_param(m.intensity, m.intensity_tex, uv);        // FloatParameter intensity (:55-56)
    }
};

// ---------------- SimpleClearcoatPbrMaterial (impls/simple_pbr_clearcoat_material.rs) ----------------
struct Clearcoat {
    const Material& m; SS base_color, tint; uint64_t key;
    float metallic, roughness;     // FloatParameter values at the shading point
    float thickness;
    int draws_mode = 0;            // Scene::cc_draws: 0 one estimate per vertex shared by sample / evaluate / pdf, 1 three independent ones, 2 table
    static float r2a(float r) { return r * r; }                                            // :76-78
    static float diel_r0(float ior) { float r = (ior - 1.0f) / (ior + 1.0f); return r * r; }   // :81-84
    static SS attenuation(SS tint, float thickness, float cos_theta) {                     // :88-107
        SS log_tint = ss_log(tint);
        SS sigma = (-1.0f * log_tint) / 0.001f;
        float thickness_m = thickness * 0.001f;
        float l = thickness_m / std::fmax(cos_theta, 1e-4f);
        return ss_exp((-1.0f * sigma) * l);
    }
    GenSchlickBsdf coat() const { return GenSchlickBsdf{SS::constant(diel_r0(m.cc_ior)), SS::one(), 5.0f, SS::one(), r2a(m.cc_roughness), r2a(m.cc_roughness)}; }
    GenSchlickBsdf metal(SS r0) const { float a = r2a(roughness); return GenSchlickBsdf{r0, SS::one(), 5.0f, SS::one(), a, a}; }
    GenSchlickBsdf diel() const { float a = r2a(roughness); return GenSchlickBsdf{SS::constant(diel_r0(m.cc_base_ior)), SS::one(), 5.0f, SS::one(), a, a}; }

    // sample_base_material (:336-383) in the normal-map frame; returns BsdfSample with wi in that frame
    bool sample_metallic(V3 wo, V2 uv, BsdfSample* out) const { return metal(base_color).sample_R(wo, uv, out); }     // :455-493
    bool sample_dielectric(V3 wo, float uc, V2 uv, BsdfSample* out) const {                                            // :494-551
        GenSchlickBsdf g = diel();
        float fr = g.fresnel(std::fabs(wo.z)).average();
        if (uc < fr) {
            BsdfSample b;
            if (!g.sample_R(wo, uv, &b)) return false;
            b.pdf = b.pdf * fr; *out = b; return true;
        }
        BsdfSample b;
        if (!lambert_sample(base_color, wo, uv, &b)) return false;
        b.f = b.f * (1.0f - fr); b.pdf = b.pdf * (1.0f - fr); *out = b; return true;
    }
    bool sample_base(V3 wo, float uc, V2 uv, BsdfSample* out) const {
        if (metallic >= 1.0f) return sample_metallic(wo, uv, out);
        if (metallic <= 0.0f) return sample_dielectric(wo, uc, uv, out);
        if (uc <= metallic) return sample_metallic(wo, uv, out);                                                  // sample_mixed :552-578
        return sample_dielectric(wo, (uc - metallic) / (1.0f - metallic), uv, out);
    }
    SS eval_dielectric(V3 wo, V3 wi) const {                                                                            // :603-633
        GenSchlickBsdf g = diel();
        SS direct = g.evaluate_R(wo, wi);
        float fr = g.fresnel(std::fabs(wo.z)).average();
        return direct + (1.0f - fr) * lambert_evaluate(base_color, wo, wi);
    }
    SS eval_base(V3 wo, V3 wi) const {                                                                                  // :384-417
        if (metallic >= 1.0f) return metal(base_color).evaluate_R(wo, wi);
        if (metallic <= 0.0f) return eval_dielectric(wo, wi);
        return metal(base_color).evaluate_R(wo, wi) * metallic + eval_dielectric(wo, wi) * (1.0f - metallic);
    }
    float pdf_dielectric(V3 wo, V3 wi) const {                                                                          // :646-675
        GenSchlickBsdf g = diel();
        float direct = g.pdf_R(wo, wi);
        float fr = g.fresnel(std::fabs(wo.z)).average();
        return fr * direct + (1.0f - fr) * lambert_pdf(wo, wi);
    }
    float pdf_base(V3 wo, V3 wi) const {                                                                                // :418-454
        if (metallic >= 1.0f) return metal(SS::one()).pdf_R(wo, wi);
        if (metallic <= 0.0f) return pdf_dielectric(wo, wi);
        return metal(SS::one()).pdf_R(wo, wi) * metallic + pdf_dielectric(wo, wi) * (1.0f - metallic);
    }
    // The coat weight (:190-192, 318-320, 416-418).  `which`: 0 sample, 1 evaluate, 2 pdf.  The reference draws a fresh 64-sample estimate from the
    // thread RNG in each of the three; the product (and this oracle by default) share ONE estimate per vertex.  draws_mode 1 keys
    // the three calls differently — three independent estimates, the reference's structure — so that the deviation can be measured
    // (tools/clearcoat_modes.py); draws_mode 2 reads the estimate's expectation from the material's table (mi355pt_params.albedo_lut).
    float coat_weight(V3 wo, int which) const {
        if (draws_mode == 2 && !m.cc_albedo_lut.empty()) {
            const float* tab = m.cc_albedo_lut.data();
            const float x = std::fmin(std::fmax(std::fabs(wo.z) * 64.0f - 0.5f, 0.0f), 63.0f);
            const int i0 = std::min((int)x, 62);
            const float tt = std::fmin(x - (float)i0, 1.0f);
            return tab[i0] + (tab[i0 + 1] - tab[i0]) * tt;
        }
        const uint64_t k = draws_mode == 1 ? key ^ (0x9E3779B97F4A7C15ull * (uint64_t)which) : key;
        return coat().directional_albedo(wo, k).average();
    }
};

inline MaterialSample MaterialEval::sample(const Material& m, float uc, V2 uv, Wavelengths& wl, V3 wo, const ShadingPoint& sp) const {
    MaterialSample ms;
    if (m.type == MAT_CLEARCOAT) {                                           // simple_pbr_clearcoat_material.rs:137-260
        Clearcoat cc{m, scene.sample_spectrum_param(m.color, sp.uv, wl, ctr), scene.sample_spectrum_param(m.cc_tint, sp.uv, wl, ctr), mc_key,
                     scene.sample_float_param(m.cc_metallic, m.metallic_tex, sp.uv), scene.sample_float_param(m.roughness, m.roughness_tex, sp.uv),
                     scene.sample_float_param(m.cc_thickness, m.cc_thickness_tex, sp.uv), scene.cc_draws};
        M4 tf = normal_map_transform(m, sp.uv);
        M4 tf_inv = inverse(tf);
        V3 wo_nm = transform_vector3(tf, wo);
        BsdfSample bs;
        if (cc.thickness <= 0.0f) {
            if (!cc.sample_base(wo_nm, uc, uv, &bs)) return ms;
            ms.f = bs.f; ms.wi = transform_vector3(tf_inv, bs.wi); ms.pdf = bs.pdf; ms.sample_type = bs.type; ms.is_sampled = true;
            return ms;
        }
        float fc = cc.coat_weight(wo_nm, 0);
        if (uc < fc) {
            if (!cc.coat().sample_R(wo_nm, uv, &bs)) return ms;
            ms.f = bs.f; ms.wi = transform_vector3(tf_inv, bs.wi); ms.pdf = bs.pdf * fc; ms.sample_type = bs.type; ms.is_sampled = true;
            return ms;
        }
        float uc_adj = (uc - fc) / (1.0f - fc);
        if (!cc.sample_base(wo_nm, uc_adj, uv, &bs)) return ms;
        V3 wi_sh = transform_vector3(tf_inv, bs.wi);
        SS att = Clearcoat::attenuation(cc.tint, cc.thickness, wo_nm.z) * Clearcoat::attenuation(cc.tint, cc.thickness, wi_sh.z);   // Q14
        ms.f = bs.f * att; ms.wi = wi_sh; ms.pdf = bs.pdf * (1.0f - fc); ms.sample_type = bs.type; ms.is_sampled = true;
        return ms;
    }
    if (m.type == MAT_METAL) {                                               // metal_material.rs:96-148
        SS eta = m.eta.sample(wl), k = m.k.sample(wl);
        M4 tf = normal_map_transform(m, sp.uv);
        M4 tf_inv = inverse(tf);
        V3 wo_nm = transform_vector3(tf, wo);
        float rough = scene.sample_float_param(m.roughness, m.roughness_tex, sp.uv);
        float alpha = rough * rough;
        ConductorBsdf bsdf{eta, k, alpha, alpha};
        BsdfSample bs;
        if (!bsdf.sample(wo_nm, uv, &bs)) return ms;
        V3 wi_sh = transform_vector3(tf_inv, bs.wi);
        if (signum(dot(sp.normal, wi_sh)) != signum(dot(sp.normal, wo))) return ms;
        ms.f = bs.f; ms.wi = wi_sh; ms.pdf = bs.pdf; ms.sample_type = bs.type; ms.is_sampled = true;
        return ms;
    }
    if (m.type == MAT_LAMBERT) {                                             // lambert_material.rs:42-97
        SS albedo = scene.sample_spectrum_param(m.color, sp.uv, wl, ctr);
        M4 tf = normal_map_transform(m, sp.uv);
        M4 tf_inv = inverse(tf);
        V3 wo_nm = transform_vector3(tf, wo);
        BsdfSample bs;
        if (!lambert_sample(albedo, wo_nm, uv, &bs)) return ms;
        V3 wi_sh = transform_vector3(tf_inv, bs.wi);
        if (signum(dot(sp.normal, wi_sh)) != signum(dot(sp.normal, wo))) return ms;
        ms.f = bs.f; ms.wi = wi_sh; ms.pdf = bs.pdf; ms.sample_type = bs.type; ms.is_sampled = true;
        return ms;
    }
    if (m.type == MAT_GLASS || m.type == MAT_PLASTIC) {                      // glass_material.rs:69-118, plastic_material.rs:88-139
        SS eta = m.type == MAT_GLASS ? eta_of(m, wl) : SS::constant(m.eta.c[0]);
        M4 tf = normal_map_transform(m, sp.uv);
        M4 tf_inv = inverse(tf);
        V3 wo_nm = transform_vector3(tf, wo);
        bool entering = dot(sp.normal, wo) > 0.0f;
        const float d_rough = scene.sample_float_param(m.roughness, m.roughness_tex, sp.uv);   // FloatParameter::sample (glass_material.rs:116)
        DielectricBsdf bsdf(eta, entering, m.thin, d_rough, d_rough);
        BsdfSample bs;
        if (!bsdf.sample(wo_nm, uv, uc, wl, &bs)) return ms;
        if (m.type == MAT_PLASTIC && dot(bs.wi, wo_nm) < 0.0f) {
            // reference quirk Q15: colour indexed with the *random* uv (plastic_material.rs:123-126)
            bs.f = bs.f * scene.sample_spectrum_param(m.color, uv, wl, ctr);
        }
        ms.f = bs.f; ms.wi = transform_vector3(tf_inv, bs.wi); ms.pdf = bs.pdf; ms.sample_type = bs.type; ms.is_sampled = true;
        return ms;
    }
    return ms;
}

inline SS MaterialEval::evaluate(const Material& m, const Wavelengths& wl, V3 wo, V3 wi, const ShadingPoint& sp) const {
    if (m.type == MAT_CLEARCOAT) {                                           // :261-341
        Clearcoat cc{m, scene.sample_spectrum_param(m.color, sp.uv, wl, ctr), scene.sample_spectrum_param(m.cc_tint, sp.uv, wl, ctr), mc_key,
                     scene.sample_float_param(m.cc_metallic, m.metallic_tex, sp.uv), scene.sample_float_param(m.roughness, m.roughness_tex, sp.uv),
                     scene.sample_float_param(m.cc_thickness, m.cc_thickness_tex, sp.uv), scene.cc_draws};
        M4 tf = normal_map_transform(m, sp.uv);
        V3 wo_nm = transform_vector3(tf, wo), wi_nm = transform_vector3(tf, wi);
        if (cc.thickness <= 0.0f) return cc.eval_base(wo_nm, wi_nm);
        float fc = cc.coat_weight(wo_nm, 1);
        SS cf = cc.coat().evaluate_R(wo_nm, wi_nm);
        SS sf = cc.eval_base(wo_nm, wi_nm);
        SS att = Clearcoat::attenuation(cc.tint, cc.thickness, wo_nm.z) * Clearcoat::attenuation(cc.tint, cc.thickness, wi_nm.z);
        return cf * fc + sf * att * (1.0f - fc);
    }
    if (m.type == MAT_METAL) {                                               // metal_material.rs:150-192
        SS eta = m.eta.sample(wl), k = m.k.sample(wl);
        M4 tf = normal_map_transform(m, sp.uv);
        V3 wo_nm = transform_vector3(tf, wo), wi_nm = transform_vector3(tf, wi);
        if (signum(dot(sp.normal, wi)) != signum(dot(sp.normal, wo))) return SS::zero();
        float rough = scene.sample_float_param(m.roughness, m.roughness_tex, sp.uv);
        float alpha = rough * rough;
        return ConductorBsdf{eta, k, alpha, alpha}.evaluate(wo_nm, wi_nm);
    }
    if (m.type == MAT_LAMBERT) {                                             // lambert_material.rs:99-131
        SS albedo = scene.sample_spectrum_param(m.color, sp.uv, wl, ctr);
        M4 tf = normal_map_transform(m, sp.uv);
        V3 wo_nm = transform_vector3(tf, wo), wi_nm = transform_vector3(tf, wi);
        if (signum(dot(sp.normal, wi)) != signum(dot(sp.normal, wo))) return SS::zero();
        return lambert_evaluate(albedo, wo_nm, wi_nm);
    }
    if (m.type == MAT_GLASS || m.type == MAT_PLASTIC) {
        SS eta = m.type == MAT_GLASS ? eta_of(m, wl) : SS::constant(m.eta.c[0]);
        M4 tf = normal_map_transform(m, sp.uv);
        V3 wo_nm = transform_vector3(tf, wo), wi_nm = transform_vector3(tf, wi);
        bool entering = dot(sp.normal, wo) > 0.0f;
        const float d_rough = scene.sample_float_param(m.roughness, m.roughness_tex, sp.uv);   // FloatParameter::sample (glass_material.rs:116)
        DielectricBsdf bsdf(eta, entering, m.thin, d_rough, d_rough);
        SS f = bsdf.evaluate(wo_nm, wi_nm);
        if (m.type == MAT_PLASTIC && dot(wi_nm, wo_nm) < 0.0f) f = f * scene.sample_spectrum_param(m.color, sp.uv, wl, ctr);
        return f;
    }
    return SS::zero();
}

inline float MaterialEval::pdf(const Material& m, const Wavelengths& wl, V3 wo, V3 wi, const ShadingPoint& sp) const {
    if (m.type == MAT_CLEARCOAT) {                                           // :342-433
        Clearcoat cc{m, scene.sample_spectrum_param(m.color, sp.uv, wl, ctr), scene.sample_spectrum_param(m.cc_tint, sp.uv, wl, ctr), mc_key,
                     scene.sample_float_param(m.cc_metallic, m.metallic_tex, sp.uv), scene.sample_float_param(m.roughness, m.roughness_tex, sp.uv),
                     scene.sample_float_param(m.cc_thickness, m.cc_thickness_tex, sp.uv), scene.cc_draws};
        M4 tf = normal_map_transform(m, sp.uv);
        V3 wo_nm = transform_vector3(tf, wo), wi_nm = transform_vector3(tf, wi);
        if (cc.thickness <= 0.0f) return cc.pdf_base(wo_nm, wi_nm);
        float fc = cc.coat_weight(wo_nm, 2);
        return cc.coat().pdf_R(wo_nm, wi_nm) * fc + cc.pdf_base(wo_nm, wi_nm) * (1.0f - fc);
    }
    if (m.type == MAT_METAL) {                                               // metal_material.rs:194-229
        M4 tf = normal_map_transform(m, sp.uv);
        if (signum(dot(sp.normal, wi)) != signum(dot(sp.normal, wo))) return 0.0f;
        V3 wo_nm = transform_vector3(tf, wo), wi_nm = transform_vector3(tf, wi);
        float rough = scene.sample_float_param(m.roughness, m.roughness_tex, sp.uv);
        float alpha = rough * rough;
        return ConductorBsdf{m.eta.sample(wl), m.k.sample(wl), alpha, alpha}.pdf(wo_nm, wi_nm);
    }
    if (m.type == MAT_LAMBERT) {                                             // lambert_material.rs:133-159
        M4 tf = normal_map_transform(m, sp.uv);
        if (signum(dot(sp.normal, wi)) != signum(dot(sp.normal, wo))) return 0.0f;
        V3 wo_nm = transform_vector3(tf, wo), wi_nm = transform_vector3(tf, wi);
        return lambert_pdf(wo_nm, wi_nm);
    }
    if (m.type == MAT_GLASS || m.type == MAT_PLASTIC) {
        SS eta = m.type == MAT_GLASS ? eta_of(m, wl) : SS::constant(m.eta.c[0]);
        M4 tf = normal_map_transform(m, sp.uv);
        V3 wo_nm = transform_vector3(tf, wo), wi_nm = transform_vector3(tf, wi);
        bool entering = dot(sp.normal, wo) > 0.0f;
        const float d_rough = scene.sample_float_param(m.roughness, m.roughness_tex, sp.uv);   // FloatParameter::sample (glass_material.rs:116)
        DielectricBsdf bsdf(eta, entering, m.thin, d_rough, d_rough);
        return bsdf.pdf(wo_nm, wi_nm);
    }
    return 0.0f;
}

}  // namespace oracle
