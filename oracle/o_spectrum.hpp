// ORACLE — TEST INFRASTRUCTURE ONLY.
// Restatement of spectrum/src/{sampled_spectrum,spectrum,rgb_sigmoid_polynomial}.rs,
// spectrum/src/spectrum/{densely_sampled,constant,rgb_albedo}_spectrum.rs,
// color/src/{gamut,eotf}.rs and scene/src/texture/{sampler,normal_texture,rgb_texture}.rs.
#pragma once
#include <vector>
#include "o_math.hpp"

namespace oracle {

constexpr int NS = 4;                    // N_SPECTRUM_SAMPLES  sampled_spectrum.rs:11
constexpr float LAMBDA_MIN = 360.0f;     // spectrum.rs:24-25
constexpr float LAMBDA_MAX = 830.0f;
constexpr int NLUT = 470;                // N_SPECTRUM_DENSELY_SAMPLES

struct SS {                              // SampledSpectrum
    float v[NS];
    static SS constant(float c) { return SS{{c, c, c, c}}; }
    static SS zero() { return constant(0.0f); }
    static SS one() { return constant(1.0f); }
    float max_value() const {            // fold(NEG_INFINITY, f32::max)   :229-234
        float m = -std::numeric_limits<float>::infinity();
        for (int i = 0; i < NS; ++i) m = std::fmax(m, v[i]);
        return m;
    }
    float average() const {              // :236-239
        float s = 0.0f;
        for (int i = 0; i < NS; ++i) s += v[i];
        return s / (float)NS;
    }
    bool is_constant() const {           // :241-244
        for (int i = 0; i < NS; ++i) if (!(v[i] == v[0])) return false;
        return true;
    }
};
static inline SS operator+(SS a, SS b) { SS r; for (int i = 0; i < NS; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
static inline SS operator-(SS a, SS b) { SS r; for (int i = 0; i < NS; ++i) r.v[i] = a.v[i] - b.v[i]; return r; }
static inline SS operator*(SS a, SS b) { SS r; for (int i = 0; i < NS; ++i) r.v[i] = a.v[i] * b.v[i]; return r; }
static inline SS operator*(SS a, float s) { SS r; for (int i = 0; i < NS; ++i) r.v[i] = a.v[i] * s; return r; }
static inline SS operator*(float s, SS a) { SS r; for (int i = 0; i < NS; ++i) r.v[i] = s * a.v[i]; return r; }
// division by zero yields zero (sampled_spectrum.rs:58-81)
static inline SS operator/(SS a, float s) {
    SS r; for (int i = 0; i < NS; ++i) r.v[i] = (s == 0.0f) ? 0.0f : a.v[i] / s; return r;
}
static inline SS operator/(SS a, SS b) {
    SS r; for (int i = 0; i < NS; ++i) r.v[i] = (b.v[i] == 0.0f) ? 0.0f : a.v[i] / b.v[i]; return r;
}
// `/= 0.0` is a no-op (sampled_spectrum.rs:106-115)
static inline void div_assign(SS& a, float s) {
    if (s == 0.0f) return;
    for (int i = 0; i < NS; ++i) a.v[i] /= s;
}
static inline SS ss_clamp(SS a, float lo, float hi) { SS r; for (int i = 0; i < NS; ++i) r.v[i] = clampf(a.v[i], lo, hi); return r; }
static inline SS ss_sqrt(SS a) { SS r; for (int i = 0; i < NS; ++i) r.v[i] = std::sqrt(a.v[i]); return r; }
static inline SS ss_exp(SS a) { SS r; for (int i = 0; i < NS; ++i) r.v[i] = std::exp(a.v[i]); return r; }
static inline SS ss_log(SS a) { SS r; for (int i = 0; i < NS; ++i) r.v[i] = std::log(std::fmax(a.v[i], 1e-10f)); return r; }

struct Wavelengths {                     // SampledWavelengths  sampled_spectrum.rs:304-366
    float lambda[NS];
    float pdf[NS];
    static Wavelengths new_uniform(float u) {
        Wavelengths w;
        for (int i = 0; i < NS; ++i) w.pdf[i] = 1.0f / (LAMBDA_MAX - LAMBDA_MIN);
        w.lambda[0] = LAMBDA_MIN + u * (LAMBDA_MAX - LAMBDA_MIN);
        float delta = (LAMBDA_MAX - LAMBDA_MIN) / (float)NS;
        for (int i = 1; i < NS; ++i) {
            w.lambda[i] = w.lambda[i - 1] + delta;
            if (w.lambda[i] >= LAMBDA_MAX) w.lambda[i] = LAMBDA_MIN + (w.lambda[i] - LAMBDA_MAX);
        }
        return w;
    }
    bool is_secondary_terminated() const {
        for (int i = 1; i < NS; ++i) if (!(pdf[i] == 0.0f)) return false;
        return true;
    }
    void terminate_secondary() {
        if (is_secondary_terminated()) return;
        for (int i = 1; i < NS; ++i) pdf[i] = 0.0f;
        pdf[0] /= (float)NS;
    }
};

// -------- colour helpers (color/src/eotf.rs:51-73, gamut.rs:29-71) --------
static inline float srgb_eotf_inverse(float c) {      // GammaSrgb::inverse_transform (encoded -> linear)
    return c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f);
}
static inline float srgb_oetf(float c) {               // GammaSrgb::transform (linear -> encoded)
    return c <= 0.0031308f ? 12.92f * c : 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f;
}
struct M3 { float m[3][3]; };   // m[col][row]
static inline M3 m3_inverse(const M3& a) {
    // glam Mat3::inverse: cross-product form
    V3 x{a.m[0][0], a.m[0][1], a.m[0][2]}, y{a.m[1][0], a.m[1][1], a.m[1][2]}, z{a.m[2][0], a.m[2][1], a.m[2][2]};
    V3 t0 = cross(y, z), t1 = cross(z, x), t2 = cross(x, y);
    float det = dot(z, t2);
    float inv = 1.0f / det;
    V3 r0 = t0 * inv, r1 = t1 * inv, r2 = t2 * inv;
    // transpose of [r0 r1 r2] as columns
    M3 o;
    o.m[0][0] = r0.x; o.m[0][1] = r1.x; o.m[0][2] = r2.x;
    o.m[1][0] = r0.y; o.m[1][1] = r1.y; o.m[1][2] = r2.y;
    o.m[2][0] = r0.z; o.m[2][1] = r1.z; o.m[2][2] = r2.z;
    return o;
}
static inline V3 m3_mul(const M3& a, V3 v) {
    return V3{a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z,
              a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z,
              a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z};
}
static inline M3 srgb_xyz_to_rgb() {                   // GamutSrgb::new (gamut.rs:50-63)
    auto xy_to_xyz = [](float x, float y) {
        if (y == 0.0f) return V3{0, 0, 0};
        return V3{x * 1.0f / y, 1.0f, (1.0f - x - y) * 1.0f / y};
    };
    V3 r = xy_to_xyz(0.64f, 0.33f), g = xy_to_xyz(0.30f, 0.60f), b = xy_to_xyz(0.15f, 0.06f);
    V3 w = xy_to_xyz(0.3127f, 0.3290f);
    M3 rgb{{{r.x, r.y, r.z}, {g.x, g.y, g.z}, {b.x, b.y, b.z}}};
    V3 c = m3_mul(m3_inverse(rgb), w);
    M3 rgb_to_xyz;
    for (int row = 0; row < 3; ++row) {
        rgb_to_xyz.m[0][row] = rgb.m[0][row] * c.x;
        rgb_to_xyz.m[1][row] = rgb.m[1][row] * c.y;
        rgb_to_xyz.m[2][row] = rgb.m[2][row] * c.z;
    }
    return m3_inverse(rgb_to_xyz);
}

// -------- RGB -> sigmoid-polynomial table (rgb_sigmoid_polynomial.rs:17-155) --------
constexpr int TBL = 64;
struct Rgb2SpecTable {
    std::vector<float> data;             // [64 z_nodes][3][64][64][64][3]
    bool valid() const { return data.size() == (size_t)(TBL + 3 * TBL * TBL * TBL * 3); }
    float z_node(int i) const { return data[i]; }
    float coef(int m, int zi, int yi, int xi, int k) const {
        return data[TBL + ((((size_t)m * TBL + zi) * TBL + yi) * TBL + xi) * 3 + k];
    }
    // RgbToSpectrumTable::get for ColorSrgb (gamma-encoded input): invert EOTF then look up.
    void get_srgb_encoded(const float rgb_enc[3], float c[3], bool linear = false) const {
        // color.invert_eotf(): ColorSrgb inverts the sRGB EOTF, ColorSrgbLinear's is the identity (:96-97)
        V3 rgb{std::fmax(linear ? rgb_enc[0] : srgb_eotf_inverse(rgb_enc[0]), 0.0f), std::fmax(linear ? rgb_enc[1] : srgb_eotf_inverse(rgb_enc[1]), 0.0f),
               std::fmax(linear ? rgb_enc[2] : srgb_eotf_inverse(rgb_enc[2]), 0.0f)};
        if (rgb.x == rgb.y && rgb.y == rgb.z) {
            c[0] = 0.0f; c[1] = 0.0f; c[2] = std::log(rgb.x / (1.0f - rgb.x));
            return;
        }
        int mc = max_position(rgb);
        float z = rgb[mc];
        float x = rgb[(mc + 1) % 3] * ((float)TBL - 1.0f) / z;
        float y = rgb[(mc + 2) % 3] * ((float)TBL - 1.0f) / z;
        int xi = std::min((int)x, TBL - 2), yi = std::min((int)y, TBL - 2);
        int zi = TBL - 2;
        for (int i = 0; i <= TBL - 2; ++i) if (z_node(i + 1) > z) { zi = i; break; }
        float dx = x - (float)xi, dy = y - (float)yi;
        float dz = (z - z_node(zi)) / (z_node(zi + 1) - z_node(zi));
        auto lerp = [](float a, float b, float t) { return a + (b - a) * t; };
        for (int i = 0; i < 3; ++i) {
            auto co = [&](int ddx, int ddy, int ddz) { return coef(mc, zi + ddz, yi + ddy, xi + ddx, i); };
            c[i] = lerp(lerp(lerp(co(0, 0, 0), co(1, 0, 0), dx), lerp(co(0, 1, 0), co(1, 1, 0), dx), dy),
                        lerp(lerp(co(0, 0, 1), co(1, 0, 1), dx), lerp(co(0, 1, 1), co(1, 1, 1), dx), dy), dz);
        }
    }
};
static inline float sigmoid_poly_value(const float c[3], float lambda) {   // :179-182
    float t = (lambda - LAMBDA_MIN) / (LAMBDA_MAX - LAMBDA_MIN);
    float x = t * t * c[0] + t * c[1] + c[2];
    return 1.0f / (1.0f + std::exp(-x));
}

// -------- Spectrum (dyn SpectrumTrait) --------
enum SpectrumKind : uint32_t { SPEC_CONSTANT = 0, SPEC_SIGMOID = 1, SPEC_LUT470 = 2, SPEC_RGB_ILLUMINANT = 3 };
struct Spectrum {
    uint32_t kind = SPEC_CONSTANT;
    float c[3] = {0, 0, 0};              // constant value in c[0], or sigmoid coefficients
    const float* lut = nullptr;          // 470 entries (SPEC_RGB_ILLUMINANT: the illuminant, presets::cie_illum_d6500())
    float scale = 1.0f;                  // SPEC_RGB_ILLUMINANT: 2 * max(rgb)
    static float lut_value(const float* lut, float lambda) {
        if (!(lambda >= LAMBDA_MIN && lambda <= LAMBDA_MAX)) return 0.0f;
        int idx = (int)std::floor(lambda - LAMBDA_MIN);
        return idx < NLUT ? lut[idx] : 0.0f;
    }
    float value(float lambda) const {
        switch (kind) {
            case SPEC_CONSTANT: return c[0];                                    // constant_spectrum.rs:17-19
            case SPEC_SIGMOID: return sigmoid_poly_value(c, lambda);
            case SPEC_RGB_ILLUMINANT: return scale * sigmoid_poly_value(c, lambda) * lut_value(lut, lambda);   // rgb_illuminant_spectrum.rs:44-46
            default: {                                                          // densely_sampled_spectrum.rs:57-67
                if (!(lambda >= LAMBDA_MIN && lambda <= LAMBDA_MAX)) return 0.0f;
                int idx = (int)std::floor(lambda - LAMBDA_MIN);
                return idx < NLUT ? lut[idx] : 0.0f;
            }
        }
    }
    // SpectrumTrait::sample (spectrum.rs:32-46)
    SS sample(const Wavelengths& w) const {
        SS r = SS::zero();
        if (w.is_secondary_terminated()) { r.v[0] = value(w.lambda[0]); return r; }
        for (int i = 0; i < NS; ++i) r.v[i] = value(w.lambda[i]);
        return r;
    }
};

// -------- textures (texture/sampler.rs:6-143) --------
struct TextureRgb8 { std::vector<uint8_t> data; uint32_t w = 0, h = 0; };
static inline float rust_fract(float x) { return x - std::trunc(x); }
static inline void bilinear_sample_rgb(const TextureRgb8& t, V2 uv, float out[3]) {
    float u = std::fabs(rust_fract(uv.x));
    float v = 1.0f - std::fabs(rust_fract(uv.y));
    float x = u * ((float)t.w - 1.0f), y = v * ((float)t.h - 1.0f);
    uint32_t x0 = (uint32_t)std::floor(x), y0 = (uint32_t)std::floor(y);
    uint32_t x1 = std::min(x0 + 1, t.w - 1), y1 = std::min(y0 + 1, t.h - 1);
    float fx = x - (float)x0, fy = y - (float)y0;
    auto px = [&](uint32_t xx, uint32_t yy, int c) { return (float)t.data[((size_t)yy * t.w + xx) * 3 + c] / 255.0f; };
    for (int c = 0; c < 3; ++c) {
        float top = px(x0, y0, c) * (1.0f - fx) + px(x1, y0, c) * fx;
        float bottom = px(x0, y1, c) * (1.0f - fx) + px(x1, y1, c) * fx;
        out[c] = top * (1.0f - fy) + bottom * fy;
    }
}
// NormalTexture::sample_normal (normal_texture.rs:39-66)
static inline V3 sample_normal_map(const TextureRgb8& t, bool flip_y, V2 uv) {
    float rgb[3];
    bilinear_sample_rgb(t, uv, rgb);
    float x = rgb[0] * 2.0f - 1.0f, y = rgb[1] * 2.0f - 1.0f, z = rgb[2] * 2.0f - 1.0f;
    if (flip_y) y = -y;
    float len = std::sqrt(x * x + y * y + z * z);
    if (len > 0.0f) {
        x /= len; y /= len; z /= len;
        return normalize(normalize(V3{x, y, z}));      // Normal::new normalises, Normal::from again
    }
    return V3{0, 0, 1};
}

}  // namespace oracle
