// Scene lowering: C-ABI description -> flat HBM layout (layout.hpp).  Host-only C++ (compiled by hipcc).
#include "scene.hpp"

#include <hip/hip_runtime.h>

#include <array>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pt {

namespace {

struct V3 { float x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { float r = 1.0f / length(a); return {a.x * r, a.y * r, a.z * r}; }

// column-major 4x4 * point (same operation order as glam::Mat4::transform_point3 so that light triangles
// and BVH triangles land on the very floats the reference computes in EmissiveTriangleMesh::sample_radiance)
inline V3 xform_point(const float* m, V3 p) {
    float r[4];
    for (int i = 0; i < 4; ++i) r[i] = m[i] * p.x;
    for (int i = 0; i < 4; ++i) r[i] = r[i] + m[4 + i] * p.y;
    for (int i = 0; i < 4; ++i) r[i] = r[i] + m[8 + i] * p.z;
    for (int i = 0; i < 4; ++i) r[i] = r[i] + m[12 + i];
    return {r[0], r[1], r[2]};
}
inline void mat4_mul(const float* a, const float* b, float* o) {   // o = a * b, column-major
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = a[r] * b[4 * c];
            s = s + a[4 + r] * b[4 * c + 1];
            s = s + a[8 + r] * b[4 * c + 2];
            s = s + a[12 + r] * b[4 * c + 3];
            o[4 * c + r] = s;
        }
}
inline bool mat3_inverse_transpose(const float* m /*col-major 3x3*/, float* o, float* det_out) {
    double a = m[0], b = m[3], c = m[6], d = m[1], e = m[4], f = m[7], g = m[2], h = m[5], i = m[8];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    *det_out = (float)det;
    if (det == 0.0) return false;
    double inv[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                     (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                     (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};   // row-major inverse
    // inverse-transpose, column-major: o[col*3+row] = inv[col][row] (row-major inv) => element (row,col) of inv^T = inv[col][row]
    for (int col = 0; col < 3; ++col) for (int row = 0; row < 3; ++row) o[col * 3 + row] = (float)inv[col * 3 + row];
    return true;
}

// glam::Mat4::inverse (the cofactor form glam inherits from GLM), operation for operation: the device multiplies rays and hits with THESE
// floats where the reference inverts local_to_render in every intersect call (primitive/impls/triangle_mesh.rs:97,  math/src/transform.rs:159-162).
inline void mat4_inverse_glam(const float* s, float* o) {   // column-major 4x4
    const float m00 = s[0], m01 = s[1], m02 = s[2], m03 = s[3], m10 = s[4], m11 = s[5], m12 = s[6], m13 = s[7];
    const float m20 = s[8], m21 = s[9], m22 = s[10], m23 = s[11], m30 = s[12], m31 = s[13], m32 = s[14], m33 = s[15];
    const float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    const float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    const float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    const float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    const float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    const float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    const float fac0[4] = {coef00, coef00, coef02, coef03}, fac1[4] = {coef04, coef04, coef06, coef07}, fac2[4] = {coef08, coef08, coef10, coef11};
    const float fac3[4] = {coef12, coef12, coef14, coef15}, fac4[4] = {coef16, coef16, coef18, coef19}, fac5[4] = {coef20, coef20, coef22, coef23};
    const float vec0[4] = {m10, m00, m00, m00}, vec1[4] = {m11, m01, m01, m01}, vec2[4] = {m12, m02, m02, m02}, vec3[4] = {m13, m03, m03, m03};
    const float sa[4] = {1, -1, 1, -1}, sb[4] = {-1, 1, -1, 1};
    float inv[16];
    for (int i = 0; i < 4; ++i) {
        inv[i] = ((vec1[i] * fac0[i] - vec2[i] * fac1[i]) + vec3[i] * fac2[i]) * sa[i];
        inv[4 + i] = ((vec0[i] * fac0[i] - vec2[i] * fac3[i]) + vec3[i] * fac4[i]) * sb[i];
        inv[8 + i] = ((vec0[i] * fac1[i] - vec1[i] * fac3[i]) + vec3[i] * fac5[i]) * sa[i];
        inv[12 + i] = ((vec0[i] * fac2[i] - vec1[i] * fac4[i]) + vec2[i] * fac5[i]) * sb[i];
    }
    const float d0 = s[0] * inv[0], d1 = s[1] * inv[4], d2 = s[2] * inv[8], d3 = s[3] * inv[12];
    const float det = ((d0 + d1) + d2) + d3;
    const float rcp = 1.0f / det;
    for (int i = 0; i < 16; ++i) o[i] = inv[i] * rcp;
}

template <typename T>
int upload(SceneImpl* s, const std::vector<T>& v, const T** out, std::string* err) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { *err = std::string("hipMalloc: ") + hipGetErrorString(e); return MI355PT_E_DEVICE; }
    s->allocs.push_back(p);
    if (!v.empty()) {
        e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) { *err = std::string("hipMemcpy: ") + hipGetErrorString(e); return MI355PT_E_DEVICE; }
    }
    *out = (const T*)p;
    return MI355PT_OK;
}

}  // namespace

SceneImpl::~SceneImpl() { release(); }
void SceneImpl::release() {
    for (void* p : allocs) (void)hipFree(p);
    allocs.clear();
    built = false;
}

static inline float srgb_eotf_inverse(float c) { return c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f); }

bool SceneImpl::table_lookup_srgb(const float enc[3], float c[3], bool linear) const {
    const int TBL = 64;
    if (table.size() != (size_t)(TBL + 3 * TBL * TBL * TBL * 3)) return false;
    float rgb[3];
    for (int i = 0; i < 3; ++i) rgb[i] = std::fmax(linear ? enc[i] : srgb_eotf_inverse(enc[i]), 0.0f);   // color.invert_eotf() (:96-97)
    if (rgb[0] == rgb[1] && rgb[1] == rgb[2]) { c[0] = 0; c[1] = 0; c[2] = std::log(rgb[0] / (1.0f - rgb[0])); return true; }
    int mc = 0; float mx = rgb[0];
    if (rgb[1] > mx) { mx = rgb[1]; mc = 1; }
    if (rgb[2] > mx) { mc = 2; }
    float z = rgb[mc];
    float x = rgb[(mc + 1) % 3] * 63.0f / z, y = rgb[(mc + 2) % 3] * 63.0f / z;
    int xi = std::min((int)x, TBL - 2), yi = std::min((int)y, TBL - 2), zi = TBL - 2;
    for (int i = 0; i <= TBL - 2; ++i) if (table[i + 1] > z) { zi = i; break; }
    float dx = x - (float)xi, dy = y - (float)yi, dz = (z - table[zi]) / (table[zi + 1] - table[zi]);
    auto co = [&](int ddx, int ddy, int ddz, int k) {
        return table[TBL + ((((size_t)mc * TBL + zi + ddz) * TBL + yi + ddy) * TBL + xi + ddx) * 3 + k];
    };
    auto lerp = [](float a, float b, float t) { return a + (b - a) * t; };
    for (int k = 0; k < 3; ++k)
        c[k] = lerp(lerp(lerp(co(0, 0, 0, k), co(1, 0, 0, k), dx), lerp(co(0, 1, 0, k), co(1, 1, 0, k), dx), dy),
                    lerp(lerp(co(0, 0, 1, k), co(1, 0, 1, k), dx), lerp(co(0, 1, 1, k), co(1, 1, 1, k), dx), dy), dz);
    return true;
}

// Expectation of GeneralizedSchlickBsdf::directional_albedo's estimator (generalized_schlick.rs:893-918) for ScatterMode::R, a scalar r0,
// r90 = 1, exponent 5, tint 1 and alpha_x = alpha_y = alpha: mean over (u, v) in [0,1)^2 of f |cos i| / pdf with wi drawn by the GGX
// visible-normal sampler (:165-199) — the quantity the reference estimates with 64 random points per call.  Double precision, 256 x 256
// midpoints: deterministic, error ~1e-5 (the 64-point estimate it replaces has a standard deviation of 1e-2 .. 1e-1).
void coat_albedo_table(float alpha_f, float r0_f, float out[64]) {
    const double a = alpha_f, r0 = r0_f, PI = 3.14159265358979323846;
    auto lambda = [&](double x, double y, double z) { double c2 = z * z; if (c2 == 0.0) return 0.0; return (std::sqrt(1.0 + a * a * (x * x + y * y) / c2) - 1.0) / 2.0; };
    auto D = [&](double x, double y, double z) { double c2 = z * z; if (c2 == 0.0) return 0.0; double e = (x * x + y * y) / c2 / (a * a); return 1.0 / (PI * a * a * c2 * c2 * (1.0 + e) * (1.0 + e)); };
    for (int k = 0; k < 64; ++k) {
        const double cz = (k + 0.5) / 64.0, sx = std::sqrt(std::max(1.0 - cz * cz, 0.0));     // wo = (sin, 0, cos): the estimate is isotropic in wo
        const double wo[3] = {sx, 0.0, cz};
        double sum = 0.0;
        const int N = 256;
        for (int iu = 0; iu < N; ++iu) for (int iv = 0; iv < N; ++iv) {
            const double u = (iu + 0.5) / N, v = (iv + 0.5) / N;
            double term = 0.0;
            if (a < 1e-3) {                                                     // effectively smooth: wi = mirror, f = F, pdf = 1 (:232-251)
                double o = 1.0 - std::min(std::max(cz, 0.0), 1.0);
                term = (r0 + (1.0 - r0) * o * o * o * o * o) * cz;
            } else {
                double wh[3] = {a * wo[0], a * wo[1], wo[2]};
                double l = std::sqrt(wh[0] * wh[0] + wh[1] * wh[1] + wh[2] * wh[2]); wh[0] /= l; wh[1] /= l; wh[2] /= l;
                double t1[3] = {1, 0, 0};
                if (wh[2] < 0.99999) { double tl = std::sqrt(wh[0] * wh[0] + wh[1] * wh[1]); t1[0] = -wh[1] / tl; t1[1] = wh[0] / tl; t1[2] = 0.0; }
                const double t2[3] = {wh[1] * t1[2] - wh[2] * t1[1], wh[2] * t1[0] - wh[0] * t1[2], wh[0] * t1[1] - wh[1] * t1[0]};
                const double r = std::sqrt(u), th = 2.0 * PI * v;
                const double px = r * std::cos(th), pyy = r * std::sin(th);
                const double h = std::sqrt(std::max(1.0 - px * px, 0.0)), lf = (1.0 + wh[2]) / 2.0;
                const double py = h * (1.0 - lf) + pyy * lf, pz = std::sqrt(std::max(1.0 - px * px - py * py, 0.0));
                double nh[3] = {t1[0] * px + t2[0] * py + wh[0] * pz, t1[1] * px + t2[1] * py + wh[1] * pz, t1[2] * px + t2[2] * py + wh[2] * pz};
                double wm[3] = {a * nh[0], a * nh[1], std::max(1e-6, nh[2])};
                l = std::sqrt(wm[0] * wm[0] + wm[1] * wm[1] + wm[2] * wm[2]); wm[0] /= l; wm[1] /= l; wm[2] /= l;
                const double wodm = wo[0] * wm[0] + wo[1] * wm[1] + wo[2] * wm[2];
                const double wi[3] = {2.0 * wodm * wm[0] - wo[0], 2.0 * wodm * wm[1] - wo[1], 2.0 * wodm * wm[2] - wo[2]};
                const double cd = std::fabs(wodm), ci = std::fabs(wi[2]);
                if (wo[2] * wi[2] > 0.0 && cd >= 1e-6 && ci > 0.0) {
                    const double d = D(wm[0], wm[1], wm[2]);
                    const double pdf = (1.0 / (1.0 + lambda(wo[0], wo[1], wo[2]))) / cz * d * cd / (4.0 * cd);
                    const double g = 1.0 / (1.0 + lambda(wo[0], wo[1], wo[2]) + lambda(wi[0], wi[1], wi[2]));
                    const double o = 1.0 - std::min(std::max(cd, 0.0), 1.0);
                    const double f = (r0 + (1.0 - r0) * o * o * o * o * o) * d * g / (4.0 * cz);
                    if (pdf > 0.0) term = f * ci / pdf;
                }
            }
            sum += term;
        }
        out[k] = (float)(sum / ((double)N * N));
    }
}

int SceneImpl::lower_spectrum(const mi355pt_spectrum& in, DevSpectrum* out, int allow_texture, std::string* err) const {
    std::memset(out, 0, sizeof(*out));
    switch (in.kind) {
        case MI355PT_SPEC_CONSTANT: out->kind = SPK_CONSTANT; out->c[0] = in.c[0]; return MI355PT_OK;
        case MI355PT_SPEC_SIGMOID: out->kind = SPK_SIGMOID; std::memcpy(out->c, in.c, 12); return MI355PT_OK;
        case MI355PT_SPEC_RGB_ALBEDO_SRGB:
            if (!table_lookup_srgb(in.c, out->c)) { *err = "RGB spectrum needs mi355pt_scene_set_rgb2spec first"; return MI355PT_E_INVALID; }
            out->kind = SPK_SIGMOID; return MI355PT_OK;
        case MI355PT_SPEC_RGB_ALBEDO_SRGB_LINEAR:
            if (!table_lookup_srgb(in.c, out->c, true)) { *err = "RGB spectrum needs mi355pt_scene_set_rgb2spec first"; return MI355PT_E_INVALID; }
            out->kind = SPK_SIGMOID; return MI355PT_OK;
        case MI355PT_SPEC_LUT470:
            if (in.id >= luts.size()) { *err = "bad LUT id"; return MI355PT_E_INVALID; }
            out->kind = SPK_LUT; out->id = in.id; return MI355PT_OK;
        case MI355PT_SPEC_TEXTURE_ALBEDO_SRGB:
            if (!allow_texture) { *err = "texture spectrum not supported for this parameter"; return MI355PT_E_INVALID; }
            if (in.id >= textures.size() || table.empty()) { *err = "bad texture id or missing rgb2spec table"; return MI355PT_E_INVALID; }
            out->kind = SPK_TEXTURE; out->id = in.id; return MI355PT_OK;
        case MI355PT_SPEC_TEXTURE_ILLUMINANT_SRGB:
        case MI355PT_SPEC_TEXTURE_UNBOUNDED_SRGB: {
            // SpectrumType::{Illuminant, Unbounded} (rgb_texture.rs:56-64): the emitters' types
            if (allow_texture < 2) { *err = "Illuminant / Unbounded texture spectra are accepted for emitter radiance only"; return MI355PT_E_INVALID; }
            if (in.id >= textures.size() || table.empty()) { *err = "bad texture id or missing rgb2spec table"; return MI355PT_E_INVALID; }
            const uint32_t sub = in.kind == MI355PT_SPEC_TEXTURE_ILLUMINANT_SRGB ? 1u : 2u, lut = (uint32_t)in.c[0];
            if (sub == 1u && (!(in.c[0] >= 0.0f) || lut >= luts.size())) { *err = "Illuminant texture: c[0] must hold the LUT470 id of the illuminant"; return MI355PT_E_INVALID; }
            out->kind = SPK_TEXTURE; out->id = in.id;
            std::memcpy(&out->c[0], &sub, 4); std::memcpy(&out->c[1], &lut, 4);
            return MI355PT_OK;
        }
        default: *err = "unknown spectrum kind"; return MI355PT_E_INVALID;
    }
}

int SceneImpl::build(const mi355pt_camera* cam, const float* cmf4 /*470*4*/, std::string* err) {
    release();
    if (instances.empty()) { *err = "scene has no instances"; return MI355PT_E_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { *err = "no HIP device: the product path requires a gfx950 GPU"; return MI355PT_E_NO_DEVICE; }
    (void)hipGetDevice(&device);
    std::memcpy(build_cam_pos, cam->position, sizeof(build_cam_pos));

    // world -> render = translate(-camera position)  (camera.rs:84-86, scene.rs:65-66)
    float w2r[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -cam->position[0], -cam->position[1], -cam->position[2], 1};

    std::vector<DevInstance> dinst(instances.size());
    std::vector<DevTri> tris_unordered;
    std::vector<DevTriLocal> local_unordered;
    std::vector<uint8_t> deg_render, deg_local;      // cross product exactly zero (render-space / local vertices)
    bool all_shared = true, have_shared = false;
    float shared_iw[3] = {0, 0, 0}, shared_mw[3] = {0, 0, 0};
    std::vector<DevTriShade> shade_unordered;
    std::vector<BuildTri> btris;
    std::vector<DevLight> lights;
    std::vector<DevLightTri> light_tris;
    std::vector<float> light_uvs;            // 6 per light triangle (original vertex order), zeros without texcoords

    // delta lights enter the light list in creation order, interleaved with the emissive instances
    size_t next_delta = 0;
    std::vector<uint32_t> env_light_index(envs.size(), 0u);                  // position of environment light k in the light list
    std::vector<std::array<float, 16>> env_l2r(envs.size());
    float sb_lo[3] = {INFINITY, INFINITY, INFINITY}, sb_hi[3] = {-INFINITY, -INFINITY, -INFINITY};   // scene bounds (render space)
    auto push_delta = [&](const HostDeltaLight& hl) {
        DevLight dl{};
        dl.first_tri = 0; dl.n_tris = 0; dl.material = hl.material; dl.kind = hl.d.kind;   // LK_* == MI355PT_LIGHT_*
        dl.intensity = hl.d.intensity; dl.angle_inner = hl.d.angle_inner; dl.angle_outer = hl.d.angle_outer;
        float l2r[16];
        mat4_mul(w2r, hl.d.local_to_world, l2r);
        if (hl.d.kind == LK_ENV) {                                              // EnvironmentLight: phi = intensity * integrated spectrum
            dl.area_sum = hl.d.intensity;
            dl.first_tri = hl.env_index;                                        // DevScene::envs index
            env_light_index[hl.env_index] = (uint32_t)lights.size();
            std::memcpy(env_l2r[hl.env_index].data(), l2r, sizeof(float) * 16);
            lights.push_back(dl);
            return;
        }
        if (hl.d.kind == MI355PT_LIGHT_DIRECTIONAL) {
            V3 d = normalize(V3{l2r[8], l2r[9], l2r[10]});                        // local_to_render * (0,0,1), normalised
            dl.pos[0] = d.x; dl.pos[1] = d.y; dl.pos[2] = d.z;
        } else {
            dl.pos[0] = l2r[12]; dl.pos[1] = l2r[13]; dl.pos[2] = l2r[14];       // local_to_render * Point3::ZERO
            double a[9] = {l2r[0], l2r[1], l2r[2], l2r[4], l2r[5], l2r[6], l2r[8], l2r[9], l2r[10]};   // column-major linear part
            double det = a[0] * (a[4] * a[8] - a[7] * a[5]) - a[3] * (a[1] * a[8] - a[7] * a[2]) + a[6] * (a[1] * a[5] - a[4] * a[2]);
            // third row of the inverse: (inv * w).z (spot_light.rs:110)
            dl.axis[0] = (float)((a[1] * a[5] - a[4] * a[2]) / det);
            dl.axis[1] = (float)(-(a[0] * a[5] - a[3] * a[2]) / det);
            dl.axis[2] = (float)((a[0] * a[4] - a[3] * a[1]) / det);
        }
        // phi's scalar factor ({point,spot,directional}_light.rs: phi()); the directional area is filled once the bounds are known
        const float PI_F = 3.14159265358979323846f;
        if (hl.d.kind == MI355PT_LIGHT_POINT) dl.area_sum = 4.0f * PI_F * hl.d.intensity;
        else if (hl.d.kind == MI355PT_LIGHT_SPOT)   // ((I*s)*2*pi)*bracket in the reference; here s*(I*2*pi*bracket): same value up to rounding
            dl.area_sum = hl.d.intensity * 2.0f * PI_F * ((1.0f - std::cos(hl.d.angle_inner)) + (std::cos(hl.d.angle_inner) - std::cos(hl.d.angle_outer)) / 2.0f);
        lights.push_back(dl);
    };

    for (size_t ii = 0; ii < instances.size(); ++ii) {
        while (next_delta < delta_lights.size() && delta_lights[next_delta].after_instances <= ii) push_delta(delta_lights[next_delta++]);
        const HostInstance& inst = instances[ii];
        const HostMesh& mesh = meshes[inst.geom];
        const DevMaterial& mat = materials[inst.mat];
        float l2r[16];
        mat4_mul(w2r, inst.l2w, l2r);                                          // triangle_mesh.rs:38-40
        {   // primitive bounds = the mesh's local AABB carried through local_to_render (primitive/impls/triangle_mesh.rs:62-70)
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t v = 0; v < mesh.n_vert; ++v) for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], mesh.pos[3 * v + a]); hi[a] = std::fmax(hi[a], mesh.pos[3 * v + a]); }
            for (int k = 0; k < 8; ++k) {
                V3 q = xform_point(l2r, V3{(k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2]});
                sb_lo[0] = std::fmin(sb_lo[0], q.x); sb_lo[1] = std::fmin(sb_lo[1], q.y); sb_lo[2] = std::fmin(sb_lo[2], q.z);
                sb_hi[0] = std::fmax(sb_hi[0], q.x); sb_hi[1] = std::fmax(sb_hi[1], q.y); sb_hi[2] = std::fmax(sb_hi[2], q.z);
            }
        }
        DevInstance& di = dinst[ii];
        float lin[9] = {l2r[0], l2r[1], l2r[2], l2r[4], l2r[5], l2r[6], l2r[8], l2r[9], l2r[10]}, nrm_unused[9], det;
        if (!mat3_inverse_transpose(lin, nrm_unused, &det)) { *err = "singular instance transform"; return MI355PT_E_INVALID; }
        float inv[16];
        mat4_inverse_glam(l2r, inv);
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) { di.m[3 * c + r] = l2r[4 * c + r]; di.inv[3 * c + r] = inv[4 * c + r]; }
        for (int k = 0; k < 12; ++k) if (!std::isfinite(di.inv[k])) { *err = "singular instance transform"; return MI355PT_E_INVALID; }
        // a pure translation: both 3x3 parts equal the identity NUMERICALLY (glam's inverse of a translation holds -0.0 in some off-diagonal
        // entries; a * 1 + b * (-0) + c * 0 is still a, up to the sign of a zero, which no later operation can see)
        di.identity = 1u;
        for (int k = 0; k < 9; ++k) { const float e = (k % 4 == 0) ? 1.0f : 0.0f; if (!(di.m[k] == e) || !(di.inv[k] == e)) di.identity = 0u; }
        if (lowering >= 2) di.identity = 0u;                                 // (mi355pt_scene_debug_set_lowering: every instance through the general matrix path)
        di.pad[0] = di.pad[1] = di.pad[2] = 0;
        // do all instances share ONE pure translation?  (DevScene::tris_are_local)
        if (!di.identity) all_shared = false;
        else if (!have_shared) { have_shared = true; std::memcpy(shared_iw, di.inv + 9, 12); std::memcpy(shared_mw, di.m + 9, 12); }
        else if (std::memcmp(shared_iw, di.inv + 9, 12) != 0 || std::memcmp(shared_mw, di.m + 9, 12) != 0) all_shared = false;
        bool emissive = mat.type == MT_EMISSIVE;
        uint32_t light_index = ~0u;
        std::vector<float> area_list, area_table;
        float area_sum = 0.0f;
        if (emissive) {
            // EmissiveTriangleMesh::new: areas in WORLD space (emissive_triangle_mesh.rs:28-68)
            for (uint32_t t = 0; t < mesh.n_tri; ++t) {
                V3 p[3];
                for (int k = 0; k < 3; ++k) { uint32_t v = mesh.idx[3 * t + k]; p[k] = xform_point(inst.l2w, V3{mesh.pos[3 * v], mesh.pos[3 * v + 1], mesh.pos[3 * v + 2]}); }
                V3 e0 = p[0] - p[1], e1 = p[0] - p[2];
                area_list.push_back(length(cross(e0, e1)) * 0.5f);
            }
            for (float a : area_list) { area_sum += a; area_table.push_back(area_sum); }
            for (float& a : area_table) a /= area_sum;
            light_index = (uint32_t)lights.size();
            { DevLight al{}; al.first_tri = (uint32_t)light_tris.size(); al.n_tris = mesh.n_tri; al.material = inst.mat; al.area_sum = area_sum; al.kind = LK_AREA;
              lights.push_back(al); }
        }
        for (uint32_t t = 0; t < mesh.n_tri; ++t) {
            V3 p[3];
            uint32_t vi[3] = {mesh.idx[3 * t], mesh.idx[3 * t + 1], mesh.idx[3 * t + 2]};
            for (int k = 0; k < 3; ++k) p[k] = xform_point(l2r, V3{mesh.pos[3 * vi[k]], mesh.pos[3 * vi[k] + 1], mesh.pos[3 * vi[k] + 2]});
            DevTri dt{};
            dt.p0[0] = p[0].x; dt.p0[1] = p[0].y; dt.p0[2] = p[0].z; dt.p1x = p[1].x;
            dt.p1yz[0] = p[1].y; dt.p1yz[1] = p[1].z; dt.p2xy[0] = p[2].x; dt.p2xy[1] = p[2].y; dt.p2z = p[2].z;
            {   // sort class of the deferral queue (pt_kernel.hpp): material type, + 8 if the material has a SPECTRUM texture (texel fetches +
                // the rgb2spec lookup: the long branch).  Normal / roughness maps alone do not make a class: measured -4 % on scene 5.
                const DevMaterial& dm = materials[inst.mat];
                const bool tex = dm.color.kind == SPK_TEXTURE || dm.cc_tint.kind == SPK_TEXTURE || dm.eta.kind == SPK_TEXTURE;
                dt.mclass = dm.type | (tex ? 8u : 0u);
            }
            dt.instance = (uint32_t)ii; dt.flags = di.identity;
            tris_unordered.push_back(dt);
            BuildTri bt;
            for (int a = 0; a < 3; ++a) {
                float v0 = (&p[0].x)[a], v1 = (&p[1].x)[a], v2 = (&p[2].x)[a];
                bt.lo[a] = std::fmin(v0, std::fmin(v1, v2)); bt.hi[a] = std::fmax(v0, std::fmax(v1, v2));
                bt.c[a] = 0.5f * (bt.lo[a] + bt.hi[a]);
            }
            btris.push_back(bt);
            // math::intersect_triangle rejects a triangle whose cross product is exactly zero (ray.rs:49-56) before anything else: such a
            // triangle can never be hit, so it stays out of the tree and the traversals' triangle test does not repeat the check for every
            // candidate (pt_device.hpp intersect_triangle<false>: +1.5 % on the Cornell scenes, +5 % on the 20 k-triangle hero of scene 17).
            // Decided on the vertices the traversal will test, with the device's arithmetic.
            {
                const V3 cr = cross(p[1] - p[0], p[2] - p[0]);
                deg_render.push_back(dot(cr, cr) == 0.0f ? 1 : 0);
            }
            DevTriShade sh{};
            V3 pl[3];
            for (int k = 0; k < 3; ++k) pl[k] = V3{mesh.pos[3 * vi[k]], mesh.pos[3 * vi[k] + 1], mesh.pos[3 * vi[k] + 2]};
            {   // the hit's geometric normal is a function of the triangle alone: ray.rs:167-174 in LOCAL space, then Transform * Normal
                // (samples.rs:135, transform.rs:45-51: transpose(inverse) * n, renormalised) - computed here once with the arithmetic of the
                // device code it replaces (xf_normal, pt_device.hpp; a translation leaves normalize(n))
                V3 g = normalize(normalize(cross(pl[1] - pl[0], pl[2] - pl[0])));
                if (!di.identity)
                    g = V3{(di.inv[0] * g.x + di.inv[1] * g.y) + di.inv[2] * g.z, (di.inv[3] * g.x + di.inv[4] * g.y) + di.inv[5] * g.z, (di.inv[6] * g.x + di.inv[7] * g.y) + di.inv[8] * g.z};
                g = normalize(g);
                sh.ng[0] = g.x; sh.ng[1] = g.y; sh.ng[2] = g.z; sh.pad_ng = 0;
            }
            DevTriLocal tl{};
            tl.p0[0] = pl[0].x; tl.p0[1] = pl[0].y; tl.p0[2] = pl[0].z; tl.p1x = pl[1].x;
            tl.p1yz[0] = pl[1].y; tl.p1yz[1] = pl[1].z; tl.p2xy[0] = pl[2].x; tl.p2xy[1] = pl[2].y; tl.p2z = pl[2].z;
            tl.instance = (uint32_t)ii; tl.flags = di.identity; tl.mclass = dt.mclass;
            local_unordered.push_back(tl);
            { const V3 cl = cross(pl[1] - pl[0], pl[2] - pl[0]); deg_local.push_back(dot(cl, cl) == 0.0f ? 1 : 0); }
            const float* n0 = &mesh.nrm[3 * vi[0]]; const float* n1 = &mesh.nrm[3 * vi[1]]; const float* n2 = &mesh.nrm[3 * vi[2]];
            sh.n0[0] = n0[0]; sh.n0[1] = n0[1]; sh.n0[2] = n0[2]; sh.n1x = n1[0];
            sh.n1yz[0] = n1[1]; sh.n1yz[1] = n1[2]; sh.n2xy[0] = n2[0]; sh.n2xy[1] = n2[1]; sh.n2z = n2[2];
            sh.flags = di.identity ? 4u : 0u;                                      // bit 2: the instance's linear part is the identity (load_surface)
            if (!mesh.uv.empty()) {
                sh.flags |= 1u;
                sh.tangent[0] = mesh.tangent[3 * t]; sh.tangent[1] = mesh.tangent[3 * t + 1]; sh.tangent[2] = mesh.tangent[3 * t + 2];
                sh.uv0[0] = mesh.uv[2 * vi[0]]; sh.uv0[1] = mesh.uv[2 * vi[0] + 1];
                sh.uv1[0] = mesh.uv[2 * vi[1]]; sh.uv1[1] = mesh.uv[2 * vi[1] + 1];
                sh.uv2[0] = mesh.uv[2 * vi[2]]; sh.uv2[1] = mesh.uv[2 * vi[2] + 1];
            }
            sh.material = inst.mat; sh.instance = (uint32_t)ii; sh.local_tri = t; sh.light = light_index;
            sh.light_pdf_area = 0.0f;
            if (emissive) {
                sh.flags |= 2u;
                float probability = t == 0 ? area_table[0] : area_table[t] - area_table[t - 1];
                sh.light_pdf_area = 1.0f / area_list[t] * probability;                       // :334-353
                DevLightTri lt{};
                // sample_radiance transforms the ORIGINAL vertex order with local_to_render (:200-206)
                V3 q[3];
                for (int k = 0; k < 3; ++k) { uint32_t v = mesh.idx[3 * t + k]; q[k] = xform_point(l2r, V3{mesh.pos[3 * v], mesh.pos[3 * v + 1], mesh.pos[3 * v + 2]}); }
                lt.p0[0] = q[0].x; lt.p0[1] = q[0].y; lt.p0[2] = q[0].z; lt.p1x = q[1].x;
                lt.p1yz[0] = q[1].y; lt.p1yz[1] = q[1].z; lt.p2xy[0] = q[2].x; lt.p2xy[1] = q[2].y; lt.p2z = q[2].z;
                lt.cdf = area_table[t];
                { V3 g = normalize(normalize(cross(q[1] - q[0], q[2] - q[0]))); lt.n[0] = g.x; lt.n[1] = g.y; lt.n[2] = g.z; }
                light_tris.push_back(lt);
                for (int k = 0; k < 3; ++k) {
                    uint32_t v = mesh.idx[3 * t + k];
                    light_uvs.push_back(mesh.uv.empty() ? 0.0f : mesh.uv[2 * v]); light_uvs.push_back(mesh.uv.empty() ? 0.0f : mesh.uv[2 * v + 1]);
                }
            }
            shade_unordered.push_back(sh);
        }
    }
    while (next_delta < delta_lights.size()) push_delta(delta_lights[next_delta++]);
    {   // DirectionalLight::preprocess (directional_light.rs:46-54): area = pi r^2 of the scene's bounding sphere (bounds.rs:59-77)
        V3 c{(sb_lo[0] + sb_hi[0]) * 0.5f, (sb_lo[1] + sb_hi[1]) * 0.5f, (sb_lo[2] + sb_hi[2]) * 0.5f};
        float radius = length(V3{c.x - sb_hi[0], c.y - sb_hi[1], c.z - sb_hi[2]});
        const float PI_F = 3.14159265358979323846f;
        for (DevLight& dl : lights) {
            if (dl.kind == LK_DIRECTIONAL) dl.area_sum = dl.intensity * (PI_F * radius * radius);
        }
    }
    // which vertices will the traversal test?  (decided here, before the tree is built: mi355pt_scene_debug_set_lowering included)
    const bool tris_local_mode = all_shared && have_shared && lowering < 1;
    std::vector<uint32_t> kept;                       // build index -> triangle
    {
        const std::vector<uint8_t>& deg = tris_local_mode ? deg_local : deg_render;
        std::vector<BuildTri> keep_b;
        for (size_t i = 0; i < btris.size(); ++i) if (!deg[i]) { kept.push_back((uint32_t)i); keep_b.push_back(btris[i]); }
        n_degenerate = btris.size() - kept.size();
        btris.swap(keep_b);
    }
    if (btris.empty()) { *err = "scene has no triangles"; return MI355PT_E_INVALID; }
    if (btris.size() > ((size_t)MAX_LEAF_TRIS << MAX_BUILD_DEPTH)) { *err = "too many triangles"; return MI355PT_E_INVALID; }   // 16.7 M: depth bound of the traversal stack

    // BVH: host sweep SAH, or the GPU binned-SAH builder for large triangle counts
    BvhOut bvh;
    {
        int mode = bvh_builder;
        if (const char* e = getenv("MI355PT_BVH_BUILDER")) {
            if (!strcmp(e, "host")) mode = MI355PT_BVH_HOST; else if (!strcmp(e, "gpu")) mode = MI355PT_BVH_GPU; else if (!strcmp(e, "auto")) mode = MI355PT_BVH_AUTO;
        }
        const bool want_gpu = mode == MI355PT_BVH_GPU || (mode == MI355PT_BVH_AUTO && btris.size() >= BVH_GPU_AUTO_TRIS);
        auto t0 = std::chrono::steady_clock::now();
        bvh_builder_used = MI355PT_BVH_HOST; bvh_device_ms = 0.0;
        bool done = false;
        if (want_gpu) {
            std::string why;
            done = build_bvh_gpu(btris, &bvh, &bvh_device_ms, &why);
            if (done) bvh_builder_used = MI355PT_BVH_GPU;
            else if (mode == MI355PT_BVH_GPU) { *err = why; return MI355PT_E_DEVICE; }   // asked for explicitly: no silent substitute
        }
        if (!done) build_bvh(btris, &bvh);
        bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    bvh_depth = bvh.max_depth;
    if (bvh.max_depth >= STACK_DEPTH) { *err = "BVH deeper than the traversal stack"; return MI355PT_E_INVALID; }
    std::vector<DevTri> tris(bvh.order.size());
    std::vector<DevTriShade> shade(bvh.order.size());
    std::vector<DevTriLocal> tris_local(bvh.order.size());
    for (size_t i = 0; i < bvh.order.size(); ++i) { const uint32_t t = kept[bvh.order[i]]; tris[i] = tris_unordered[t]; shade[i] = shade_unordered[t]; tris_local[i] = local_unordered[t]; }

    // LUT pool, CMF, table (repacked to float4 cells), textures (RGBA8)
    std::vector<float> lut_pool;
    for (auto& l : luts) lut_pool.insert(lut_pool.end(), l.begin(), l.end());
    std::vector<float> cmf(cmf4, cmf4 + 470 * 4);
    std::vector<float> tab4, znodes;
    if (!table.empty()) {
        const size_t cells = (size_t)3 * 64 * 64 * 64;
        tab4.resize(cells * 4);
        for (size_t c = 0; c < cells; ++c) { tab4[4 * c] = table[64 + 3 * c]; tab4[4 * c + 1] = table[64 + 3 * c + 1]; tab4[4 * c + 2] = table[64 + 3 * c + 2]; tab4[4 * c + 3] = 0.0f; }
        znodes.assign(table.begin(), table.begin() + 64);
    }
    std::vector<uint32_t> texels;
    std::vector<DevTexture> dtex;
    for (auto& t : textures) {
        dtex.push_back(DevTexture{(uint32_t)texels.size(), t.w, t.h, 0});
        size_t n = (size_t)t.w * t.h;
        for (size_t i = 0; i < n; ++i) texels.push_back((uint32_t)t.rgb[3 * i] | ((uint32_t)t.rgb[3 * i + 1] << 8) | ((uint32_t)t.rgb[3 * i + 2] << 16));
    }

    // EnvironmentLight::new (environment_light.rs:28-75) + build_2d_cdf (:153-199)
    std::vector<std::vector<float>> all_texels(envs.size()), all_marginal(envs.size()), all_conditional(envs.size());
    std::vector<DevEnv> denvs(envs.size());
    for (size_t ek = 0; ek < envs.size(); ++ek) {
        const HostEnv& env = envs[ek];
        std::vector<float>&env_texels = all_texels[ek], &env_marginal = all_marginal[ek], &env_conditional = all_conditional[ek];
        DevEnv& denv = denvs[ek];
        const float* env_l2r_k = env_l2r[ek].data();
        if (table.empty()) { *err = "environment light needs the rgb2spec table"; return MI355PT_E_INVALID; }
        const uint32_t w = env.w, h = env.h;
        float tot[3] = {0, 0, 0};
        env_texels.resize((size_t)w * h * 4);
        for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) {
            const float* p = &env.rgb[((size_t)y * w + x) * 3];
            for (int c = 0; c < 3; ++c) { tot[c] += p[c]; env_texels[((size_t)y * w + x) * 4 + c] = p[c]; }
            env_texels[((size_t)y * w + x) * 4 + 3] = 0.0f;
        }
        float n = (float)(w * h);
        for (int c = 0; c < 3; ++c) tot[c] /= n;
        DevMaterial& hm = materials[lights[env_light_index[ek]].material];     // integrated RgbIlluminantSpectrum (rgb_illuminant_spectrum.rs:26-41)
        float scale = 2.0f * std::fmax(tot[0], std::fmax(tot[1], tot[2]));
        if (scale == 0.0f) { hm.color.kind = SPK_CONSTANT; hm.color.c[0] = 0.0f; }
        else {
            float enc[3] = {tot[0] / scale, tot[1] / scale, tot[2] / scale};
            hm.color.kind = SPK_ILLUM; hm.color.id = env.illuminant_lut;
            if (!table_lookup_srgb(enc, hm.color.c)) { *err = "rgb2spec lookup failed"; return MI355PT_E_INVALID; }
            std::memcpy(&hm.color.pad[0], &scale, sizeof(float));
        }
        const float PI_F = 3.14159265358979323846f;
        std::vector<float> row_w(h, 0.0f);
        env_conditional.assign((size_t)w * h, 0.0f); env_marginal.assign(h, 0.0f);
        for (uint32_t y = 0; y < h; ++y) {
            float row_sum = 0.0f;
            for (uint32_t x = 0; x < w; ++x) {
                float v = ((float)y + 0.5f) / (float)h;
                float theta = v * PI_F;
                const float* p = &env.rgb[((size_t)y * w + x) * 3];
                float lum = 0.299f * p[0] + 0.587f * p[1] + 0.114f * p[2];
                row_sum += lum * std::fmax(std::sin(theta), 1e-8f);
                env_conditional[(size_t)y * w + x] = row_sum;
            }
            row_w[y] = row_sum;
            if (row_sum > 0.0f) for (uint32_t x = 0; x < w; ++x) env_conditional[(size_t)y * w + x] /= row_sum;
        }
        float total = 0.0f;
        for (float r : row_w) total += r;
        float cum = 0.0f;
        for (uint32_t y = 0; y < h; ++y) { cum += row_w[y]; env_marginal[y] = total > 0.0f ? cum / total : (float)(y + 1) / (float)h; }
        denv.w = w; denv.h = h; denv.total_weight = total; denv.intensity = env.intensity; denv.illuminant_lut = env.illuminant_lut;
        denv.light_index = env_light_index[ek];
        double a[9] = {env_l2r_k[0], env_l2r_k[1], env_l2r_k[2], env_l2r_k[4], env_l2r_k[5], env_l2r_k[6], env_l2r_k[8], env_l2r_k[9], env_l2r_k[10]};
        for (int i = 0; i < 9; ++i) denv.l2r[i] = (float)a[i];
        double det = a[0] * (a[4] * a[8] - a[7] * a[5]) - a[3] * (a[1] * a[8] - a[7] * a[2]) + a[6] * (a[1] * a[5] - a[4] * a[2]);
        double inv[9] = {(a[4] * a[8] - a[7] * a[5]) / det, -(a[1] * a[8] - a[7] * a[2]) / det, (a[1] * a[5] - a[4] * a[2]) / det,
                         -(a[3] * a[8] - a[6] * a[5]) / det, (a[0] * a[8] - a[6] * a[2]) / det, -(a[0] * a[5] - a[3] * a[2]) / det,
                         (a[3] * a[7] - a[6] * a[4]) / det, -(a[0] * a[7] - a[6] * a[1]) / det, (a[0] * a[4] - a[3] * a[1]) / det};
        for (int i = 0; i < 9; ++i) denv.r2l[i] = (float)inv[i];
    }

    int rc;
    std::memset(&dev, 0, sizeof(dev));
    if ((rc = upload(this, bvh.nodes, &dev.nodes, err))) return rc;
    {
        std::vector<DevNode4> nodes4;
        int max_stack = 0;
        auto tc = std::chrono::steady_clock::now();
        if (!collapse_bvh4(bvh.nodes, bvh.root, tris.size(), &nodes4, &dev.root4, &max_stack, err, &collapse_method)) return MI355PT_E_INVALID;
        bvh4_stack_need = max_stack;
        collapse_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc).count();
        dev.n_nodes4 = (uint32_t)nodes4.size();
#if PT_NODE_FMA
        {   // pad the boxes the fma slab test sees (layout.hpp PT_NODE_FMA); unused slots (point boxes at FLT_MAX) stay as they are
            float r = 0.0f;
            auto used = [](const DevNode4& n, int c) { return n.lox[c] <= n.hix[c] && n.lox[c] < FLT_MAX; };
            for (const DevNode4& n : nodes4) for (int c = 0; c < 4; ++c) if (used(n, c))
                for (float v : {n.lox[c], n.loy[c], n.loz[c], n.hix[c], n.hiy[c], n.hiz[c]}) r = std::max(r, std::fabs(v));
            const float pad = r * NODE4_PAD_REL;
            for (DevNode4& n : nodes4) for (int c = 0; c < 4; ++c) if (used(n, c)) {
                n.lox[c] -= pad; n.loy[c] -= pad; n.loz[c] -= pad; n.hix[c] += pad; n.hiy[c] += pad; n.hiz[c] += pad;
            }
        }
#endif
#if PT_NODE_Q16
        {   // the boxes on the 16-bit scene grid (layout.hpp PT_NODE_Q16): lo planes down, hi planes up, checked against the float boxes
            auto used = [](const DevNode4& n, int c) { return n.lox[c] <= n.hix[c] && n.lox[c] < FLT_MAX; };
            float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (const DevNode4& n : nodes4) for (int c = 0; c < 4; ++c) if (used(n, c)) {
                const float l[3] = {n.lox[c], n.loy[c], n.loz[c]}, h[3] = {n.hix[c], n.hiy[c], n.hiz[c]};
                for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], l[a]); hi[a] = std::fmax(hi[a], h[a]); }
            }
            float rmax = 0.0f;
            for (int a = 0; a < 3; ++a) rmax = std::fmax(rmax, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
            // the fma slab test is off by up to 2 u max(|o|, |plane|) in space (layout.hpp PT_NODE_FMA): every plane moves out by `pad` more
            const float pad = rmax * NODE4_PAD_REL;
            for (int a = 0; a < 3; ++a) {
                const float ext = std::fmax(hi[a] - lo[a], 1e-20f) + 4.0f * pad;
                dev.grid_org[a] = lo[a] - 2.0f * pad;
                dev.grid_cell[a] = ext / 65533.0f;
            }
            std::vector<DevNode4Q> q4(nodes4.size());
            for (size_t i = 0; i < nodes4.size(); ++i) {
                const DevNode4& n = nodes4[i]; DevNode4Q& q = q4[i];
                for (int c = 0; c < 4; ++c) {
                    q.child[c] = n.child[c];
                    const float l[3] = {n.lox[c], n.loy[c], n.loz[c]}, h[3] = {n.hix[c], n.hiy[c], n.hiz[c]};
                    for (int a = 0; a < 3; ++a) {
                        if (!used(n, c)) { q.q[a][0][c] = 65535; q.q[a][1][c] = 0; continue; }
                        const float org = dev.grid_org[a], cell = dev.grid_cell[a];
                        long ql = (long)std::floor((l[a] - pad - org) / cell), qh = (long)std::ceil((h[a] + pad - org) / cell);
                        ql = std::min<long>(std::max<long>(ql, 0), 65535); qh = std::min<long>(std::max<long>(qh, 0), 65535);
                        // in the arithmetic the kernel sees (origin + q * cell in f32): the quantised planes enclose the padded box
                        while (ql > 0 && org + (float)ql * cell > l[a] - pad) --ql;
                        while (qh < 65535 && org + (float)qh * cell < h[a] + pad) ++qh;
                        if (org + (float)ql * cell > l[a] || org + (float)qh * cell < h[a]) { *err = "internal error: quantised BVH box does not enclose its box"; return MI355PT_E_INVALID; }
                        q.q[a][0][c] = (uint16_t)ql; q.q[a][1][c] = (uint16_t)qh;
                    }
                }
            }
            if ((rc = upload(this, q4, &dev.nodes4q, err))) return rc;
        }
#else
        if ((rc = upload(this, nodes4, &dev.nodes4, err))) return rc;
#endif
        bvh4_nodes = nodes4.size();
    }
    if ((rc = upload(this, tris, &dev.tris_render, err))) return rc;
    if ((rc = upload(this, shade, &dev.shade, err))) return rc;
    if ((rc = upload(this, dinst, &dev.instances, err))) return rc;
    if ((rc = upload(this, tris_local, &dev.tris_local, err))) return rc;
    dev.tris_are_local = tris_local_mode ? 1u : 0u;                           // (lowering >= 1, mi355pt_scene_debug_set_lowering: never)
    dev.tris = dev.tris_are_local ? dev.tris_local : dev.tris_render;
    for (int k = 0; k < 3; ++k) { dev.tri_shift[k] = dev.tris_are_local ? shared_iw[k] : 0.0f; dev.shared_mw[k] = dev.tris_are_local ? shared_mw[k] : 0.0f; }
    dev.pad_mw = 0;
    {   // per clearcoat material: the coat's directional-albedo table (mi355pt_params.albedo_lut); the device copy of the material names its offset
        std::vector<float> cc_tab;
        std::vector<DevMaterial> mats = materials;
        for (DevMaterial& m : mats) {
            // texture descriptors ride in the records that name the texture (layout.hpp)
            m.normal_desc = m.normal_tex != 0xffffffffu ? dtex[m.normal_tex] : DevTexture{0, 0, 0, 0};
            for (DevSpectrum* sp : {&m.color, &m.eta, &m.cc_tint})
                if (sp->kind == SPK_TEXTURE) { sp->pad[0] = dtex[sp->id].offset; sp->pad[1] = dtex[sp->id].w; sp->pad[2] = dtex[sp->id].h; }
            m.cc_albedo_lut = 0;
            if (m.type != MT_CLEARCOAT) continue;
            float r = (m.cc_ior - 1.0f) / (m.cc_ior + 1.0f);
            m.cc_albedo_lut = (uint32_t)cc_tab.size();
            cc_tab.resize(cc_tab.size() + 64);
            coat_albedo_table(m.cc_roughness * m.cc_roughness, r * r, cc_tab.data() + m.cc_albedo_lut);
        }
        if (cc_tab.empty()) cc_tab.assign(64, 0.0f);
        if ((rc = upload(this, cc_tab, &dev.cc_albedo, err))) return rc;
        if ((rc = upload(this, mats, &dev.materials, err))) return rc;
    }
    if ((rc = upload(this, lights, &dev.lights, err))) return rc;
    if ((rc = upload(this, light_tris, &dev.light_tris, err))) return rc;
    if ((rc = upload(this, light_uvs, &dev.light_uvs, err))) return rc;
    if ((rc = upload(this, lut_pool, &dev.luts, err))) return rc;
    if ((rc = upload(this, cmf, &dev.cmf, err))) return rc;
    if ((rc = upload(this, tab4, &dev.rgb2spec, err))) return rc;
    if ((rc = upload(this, znodes, &dev.z_nodes, err))) return rc;
    if ((rc = upload(this, texels, &dev.texels, err))) return rc;
    if ((rc = upload(this, dtex, &dev.textures, err))) return rc;
    for (size_t ek = 0; ek < envs.size(); ++ek) {
        if ((rc = upload(this, all_texels[ek], &denvs[ek].texels, err))) return rc;
        if ((rc = upload(this, all_marginal[ek], &denvs[ek].marginal, err))) return rc;
        if ((rc = upload(this, all_conditional[ek], &denvs[ek].conditional, err))) return rc;
    }
    if ((rc = upload(this, denvs, &dev.envs, err))) return rc;
    dev.n_envs = (uint32_t)envs.size();
    dev.n_nodes = (uint32_t)bvh.nodes.size(); dev.n_tris = (uint32_t)tris.size();
    dev.n_lights = (uint32_t)lights.size(); dev.n_materials = (uint32_t)materials.size();
    dev.root = bvh.root;
    features = lights.size() == 1 ? 0u : FEAT_MLIGHT;
    if (!delta_lights.empty()) features |= FEAT_DELTA;
    if (!envs.empty()) features |= FEAT_ENV;
    for (const HostInstance& inst : instances) {
        const DevMaterial& m = materials[inst.mat];
        if (m.type == MT_GLASS || m.type == MT_PLASTIC) features |= FEAT_DIEL | ((m.roughness >= 1e-3f || m.roughness_tex != 0xffffffffu) ? FEAT_ROUGH : 0u);
        if (m.type == MT_CLEARCOAT) features |= FEAT_CC;
        if (m.type == MT_METAL) features |= FEAT_METAL;
        if (m.type == MT_EMISSIVE && (m.color.kind == SPK_TEXTURE || m.metallic_tex != 0xffffffffu)) features |= FEAT_EMTEX;   // textured radiance or intensity
        if (m.normal_tex != 0xffffffffu || m.color.kind == SPK_TEXTURE || m.cc_tint.kind == SPK_TEXTURE || m.metallic_tex != 0xffffffffu ||
            m.roughness_tex != 0xffffffffu || m.cc_thickness_tex != 0xffffffffu) features |= FEAT_TEX;
    }
    {
        char tail[224];
        std::snprintf(tail, sizeof(tail), " builder=%s bvh_ms=%.2f bvh_device_ms=%.2f collapse=%s collapse_ms=%.2f stack_need=%d/%d degenerate=%zu tri_space=%s", bvh_builder_used == MI355PT_BVH_GPU ? "gpu" : "host",
                      bvh_build_ms, bvh_device_ms, collapse_method, collapse_ms, bvh4_stack_need, STACK_DEPTH, n_degenerate, dev.tris_are_local ? "local" : "render");
        info = "nodes4=" + std::to_string(bvh4_nodes) + " nodes=" + std::to_string(bvh.nodes.size()) + " tris=" + std::to_string(tris.size()) + " depth=" + std::to_string(bvh.max_depth) + tail;
    }
    built = true;
    return MI355PT_OK;
}

}  // namespace pt
