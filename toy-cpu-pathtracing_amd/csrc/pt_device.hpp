// gfx950 device functions of the spectral path tracer: vector math, Sobol/Owen sampling, spectrum
// evaluation, BVH traversal with per-lane LDS stacks, watertight triangle test.
// Hand-written for CDNA4 wave64; no host fallbacks.  Reference citations are file:line under
// /root/reference (the algorithms' semantics come from there; the structure does not).
#pragma once
#include <hip/hip_runtime.h>

#include "layout.hpp"
#include "pt_libm.hpp"

namespace pt {

#define PT_DEV __device__ __forceinline__

struct f3 { float x, y, z; };
struct f2 { float x, y; };
// sin / cos as the reference's host computes them (pt_libm.hpp: glibc's algorithm in double, bit for bit); PT_LIBM_EXACT 0 = the device libm
#ifndef PT_LIBM_EXACT
#define PT_LIBM_EXACT 1
#endif
PT_DEV void ref_sincosf(float x, float* sn, float* cs) {
#if PT_LIBM_EXACT
    if (ptlibm::sincosf_glibc(x, sn, cs)) return;
#endif
    sincosf(x, sn, cs);
}
PT_DEV float ref_sinf(float x) { float s, c; ref_sincosf(x, &s, &c); return s; }
PT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
PT_DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
PT_DEV f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PT_DEV f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
PT_DEV float dot(f3 a, f3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
PT_DEV f3 cross(f3 a, f3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
PT_DEV float length(f3 a) { return sqrtf(dot(a, a)); }
PT_DEV f3 normalize(f3 a) { return a * (1.0f / length(a)); }   // glam: v * (1/len)
PT_DEV float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
PT_DEV float sgn1(float x) { return copysignf(1.0f, x); }       // f32::signum for non-NaN input
PT_DEV float max3f(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }

constexpr float PI_F = 3.14159265358979323846f;
// The light-connection EVALUATION of the Lambert f and pdf multiplies by 1/pi instead of dividing by pi (the reference divides):
// <= 1 ulp apart, on contribution / MIS-weight values only.  The SAMPLED f and pdf keep the division: they form the throughput, and
// for an albedo of exactly 1 the product f * (1 / pdf) decides whether the next vertex plays Russian roulette at all (p >= 1)
constexpr float INV_PI_F = 0.31830988618379067154f;
constexpr float LAMBDA_MIN = 360.0f, LAMBDA_MAX = 830.0f;

// ---------------------------------------------------------------------------------------------
// ZSobolSampler (renderer/src/sampler/z_sobol_sampler.rs) — integer-exact.
// ---------------------------------------------------------------------------------------------
PT_DEV uint64_t mix_bits(uint64_t v) {                                   // :68-75
    v ^= v >> 31; v *= 0x7fb5d329728ea185ull;
    v ^= v >> 27; v *= 0x81dadef4bc2dd44dull;
    v ^= v >> 33;
    return v;
}
__host__ __device__ inline uint64_t murmur_dim_seed(uint32_t dimension, uint32_t seed) {   // :77-99
    const uint64_t M = 0xc6a4a7935bd1e995ull;
    uint64_t h = 8ull * M;
    uint64_t k = (uint64_t)dimension | ((uint64_t)seed << 32);
    k *= M; k ^= k >> 47; k *= M;
    h ^= k; h *= M;
    h ^= h >> 47; h *= M; h ^= h >> 47;
    return h;
}
PT_DEV uint32_t fast_owen(uint32_t v, uint32_t seed) {                  // :3-28
    v = __brev(v);
    v ^= v * 0x3d20adeau;
    v += seed;
    v *= (seed >> 16) | 1u;
    v ^= v * 0x05526c56u;
    v ^= v * 0x53a22864u;
    return __brev(v);
}
PT_DEV uint32_t part1by1(uint32_t x) {   // 16 low bits -> even bit positions
    x &= 0x0000ffffu;
    x = (x ^ (x << 8)) & 0x00ff00ffu;
    x = (x ^ (x << 4)) & 0x0f0f0f0fu;
    x = (x ^ (x << 2)) & 0x33333333u;
    x = (x ^ (x << 1)) & 0x55555555u;
    return x;
}
// encode_morton2 truncated to u32 exactly like the reference (:53-66): bits >= 16 of x/y fall off.
PT_DEV uint32_t encode_morton2_u32(uint32_t x, uint32_t y) { return (part1by1(y) << 1) | part1by1(x); }

// The 24 base-4 digit permutations (:102-127), 2 bits per digit packed into a byte.
// Kept in three 64-bit immediates and selected arithmetically: a memory table would put 16 dependent loads on the
// critical path of every Sobol draw.
//   bytes: E4 B4 D8 78 6C 9C E1 B1 | C9 39 2D 8D C6 36 D2 72 | 4E 1E 27 87 1B 4B 63 93
PT_DEV uint32_t perm_packed(uint32_t p) {
    const uint64_t A = 0xB1E19C6C78D8B4E4ull, B = 0x72D236C68D2D39C9ull, C = 0x93634B1B87271E4Eull;
    uint64_t w = p < 8u ? A : (p < 16u ? B : C);
    return (uint32_t)(w >> ((p & 7u) * 8u)) & 0xffu;
}

// `hi_digits`/`hi_first`: the digits with index >= hi_first were already permuted and OR-ed into `hi_digits`
// (tile-uniform part, see sobol_tile_hi_digits); 0 / n_base4_digits computes everything here.
// `perm_lds`: the 24 packed permutations as a byte table in LDS (the render kernel is VALU-issue bound; a ds_read_u8 replaces
// the ~8 VALU instructions of perm_packed); nullptr: arithmetic selection
// (the table is indexed by permutation * 4 + digit and holds the permuted digit: one v_lshl_add + ds_read_u8 per digit)
PT_DEV uint32_t perm_digit(const uint8_t* perm_lds, uint32_t p, uint32_t digit) {
    return perm_lds ? (uint32_t)perm_lds[p * 4u + digit] : ((perm_packed(p) >> (2u * digit)) & 3u);
}
PT_DEV uint64_t sobol_sample_index(uint32_t morton, uint32_t dimension, uint32_t log2_spp, uint32_t n_base4_digits, uint64_t hi_digits = 0ull,
                                   uint32_t hi_first = 0xffffu, const uint8_t* perm_lds = nullptr) {   // :101-156
    uint64_t sample_index = hi_digits;
    if (hi_first < n_base4_digits) n_base4_digits = hi_first;
    const bool pow2 = (log2_spp & 1u) != 0;
    const int last = pow2 ? 1 : 0;
    const uint64_t dmix = 0x55555555ull * (uint64_t)dimension;
    for (int i = (int)n_base4_digits - 1; i >= last; --i) {
        int shift = 2 * i - (pow2 ? 1 : 0);
        uint32_t digit = (uint32_t)((uint64_t)morton >> shift) & 3u;
        uint64_t higher = (uint64_t)morton >> (shift + 2);
        // (mix >> 24) % 24 on a 40-bit value with 32-bit ops: 2^32 mod 24 == 16
        uint64_t mx = mix_bits(higher ^ dmix) >> 24;
        uint32_t p = (((uint32_t)(mx >> 32) * 16u) + ((uint32_t)mx % 24u)) % 24u;
        digit = perm_digit(perm_lds, p, digit);
        sample_index |= (uint64_t)digit << shift;
    }
    if (pow2) {
        // reference quirk: `morton & i` with i == 0 after the loop, so only the hashed bit survives (:147-153)
        sample_index |= mix_bits(((uint64_t)morton >> 1) ^ dmix) & 1ull;
    }
    return sample_index;
}
// Generator matrices, dimension 0: identity (van der Corput) => bit reversal of the low 32 index bits;
// columns 32..51 are zero (sobol_matrices.rs:7, first 52 words).
PT_DEV uint32_t sobol_dim0(uint64_t a) { return __brev((uint32_t)a); }
// dimension 1: column j has row i set iff (j & i) == i, periodic in j mod 32 (words 52..103).  The product is a
// GF(2) superset-sum (zeta) transform = 5 butterfly stages, then bit reversal.
PT_DEV uint32_t sobol_dim1(uint64_t a) {
    uint32_t b = (uint32_t)a ^ (uint32_t)(a >> 32);
    b ^= (b >> 1) & 0x55555555u;
    b ^= (b >> 2) & 0x33333333u;
    b ^= (b >> 4) & 0x0f0f0f0fu;
    b ^= (b >> 8) & 0x00ff00ffu;
    b ^= (b >> 16) & 0x0000ffffu;
    return __brev(b);
}
PT_DEV float sobol_bits_to_float(uint32_t v) {                           // :174-176
    return fminf((float)v * 2.3283064365386963e-10f, 0.99999994f);
}

struct Sampler {
    uint32_t morton, dimension;
    uint32_t rkey_lo, rkey_hi;   // random-mode stream key
};

constexpr int SOBOL_HI_DIMS = HASH_TABLE_DIMS;   // every dimension a 16-bounce path can reach has a cached prefix: a wave
                                                  // must never need the full-length loop for a few deep lanes
struct SamplerCtx {
    uint32_t mode, seed, log2_spp, n_base4_digits, width;
    const uint64_t* hash_lds;     // murmur(dimension, seed) for dimension < HASH_TABLE_DIMS
    const uint32_t* hi_lds;       // per-dimension permuted high digits of this wave's 8x8 tile, >> hi_shift (nullptr: none)
    uint32_t hi_first;            // first digit index covered by hi_lds
    uint32_t hi_shift;            // bit position of that digit
    const uint32_t* p6_lds;       // per dimension: the permutation indices of digit hi_first-2 for the four values of digit hi_first-1
    const uint8_t* perm_lds;      // the 24 digit permutations as a [permutation][digit] byte table (96 B), or nullptr
};
// The pixels of an aligned 8x8 tile share every Morton digit above the lowest three (2^b x 2^b block: the lowest b), and a digit's permutation
// only depends on the digits above it and on the dimension (:134-145): for those digits the permuted prefix of the
// sample index is a function of (tile, dimension) alone.  It is computed once per tile and dimension by one lane.
// `block_log2`: the work item's pixels form an aligned 2^b x 2^b block, so they share every Morton digit above the lowest b.
PT_DEV uint32_t sobol_hi_first(uint32_t log2_spp, uint32_t block_log2) { return (log2_spp + 1u) / 2u + block_log2; }
PT_DEV uint64_t sobol_tile_hi_digits(uint32_t tile_morton_shifted, uint32_t dimension, uint32_t log2_spp, uint32_t n_base4_digits, uint32_t first) {
    uint64_t out = 0;
    const bool pow2 = (log2_spp & 1u) != 0;
    const uint64_t dmix = 0x55555555ull * (uint64_t)dimension;
    for (int i = (int)n_base4_digits - 1; i >= (int)first; --i) {
        int shift = 2 * i - (pow2 ? 1 : 0);
        uint32_t digit = (uint32_t)((uint64_t)tile_morton_shifted >> shift) & 3u;
        uint64_t higher = (uint64_t)tile_morton_shifted >> (shift + 2);
        uint64_t mx = mix_bits(higher ^ dmix) >> 24;
        uint32_t p = (((uint32_t)(mx >> 32) * 16u) + ((uint32_t)mx % 24u)) % 24u;
        digit = (perm_packed(p) >> (2u * digit)) & 3u;
        out |= (uint64_t)digit << shift;
    }
#ifdef PT_HI_BREAK
    out ^= 0x300000000ull;
#endif
    return out;
}
// permutation index (0..23) of the digit whose higher digits are `higher` (z_sobol_sampler.rs:134-145)
PT_DEV uint32_t sobol_perm_index(uint64_t higher, uint32_t dimension) {
    uint64_t mx = mix_bits(higher ^ (0x55555555ull * (uint64_t)dimension)) >> 24;
    return (((uint32_t)(mx >> 32) * 16u) + ((uint32_t)mx % 24u)) % 24u;
}
PT_DEV uint64_t sampler_index(const Sampler& s, const SamplerCtx& c) {
#ifdef PT_SOBOL_ABLATE   // timing experiment only: skips the digit permutation (wrong sequence)
    return (uint64_t)s.morton;
#endif
    if (c.hi_lds != nullptr && s.dimension < (uint32_t)SOBOL_HI_DIMS) {
        // tile-uniform prefix, plus the two digits below it through tile-uniform permutation indices: the permutation of digit
        // hi_first-1 depends on the prefix only, the one of digit hi_first-2 on the prefix and the 4 values of digit hi_first-1
        const uint32_t e = c.hi_lds[s.dimension], e6 = c.p6_lds[s.dimension];
        const uint32_t sh7 = c.hi_shift - 2u, sh6 = c.hi_shift - 4u;
        const uint32_t d7 = (s.morton >> sh7) & 3u, d6 = (s.morton >> sh6) & 3u;
        const uint32_t q7 = perm_digit(c.perm_lds, e >> 27, d7);
        const uint32_t q6 = perm_digit(c.perm_lds, (e6 >> (5u * d7)) & 31u, d6);
        uint64_t hi = ((uint64_t)(e & 0x07ffffffu) << c.hi_shift) | ((uint64_t)q7 << sh7) | ((uint64_t)q6 << sh6);
        return sobol_sample_index(s.morton, s.dimension, c.log2_spp, c.n_base4_digits, hi, c.hi_first - 2u, c.perm_lds);
    }
    return sobol_sample_index(s.morton, s.dimension, c.log2_spp, c.n_base4_digits, 0ull, 0xffffu, c.perm_lds);
}

PT_DEV void sampler_start(Sampler& s, const SamplerCtx& c, uint32_t px, uint32_t py, uint32_t sample_index) {   // :198-201
    s.dimension = 0;
    s.morton = (encode_morton2_u32(px, py) << c.log2_spp) | sample_index;
    uint64_t k = mix_bits(((uint64_t)(py * c.width + px) << 32) ^ (uint64_t)sample_index ^ ((uint64_t)c.seed << 20) ^ 0x9e3779b97f4a7c15ull);
    s.rkey_lo = (uint32_t)k; s.rkey_hi = (uint32_t)(k >> 32);
}
PT_DEV uint64_t dim_hash(const SamplerCtx& c, uint32_t dimension) {
    return dimension < (uint32_t)HASH_TABLE_DIMS ? c.hash_lds[dimension] : murmur_dim_seed(dimension, c.seed);
}
PT_DEV float random_next(Sampler& s) {
    uint64_t key = ((uint64_t)s.rkey_hi << 32) | s.rkey_lo;
    uint64_t h = mix_bits(key + 0x632be59bd9b4e019ull * (uint64_t)(++s.dimension));
    return (float)(uint32_t)(h >> 40) * 5.9604644775390625e-8f;
}
// (the murmur(dimension, seed) word is fetched BEFORE the digit loop of sampler_index: it does not depend on it, and its L1 round trip
// then overlaps the ~250 instructions of the loop instead of following them)
PT_DEV uint32_t get_1d_bits(Sampler& s, const SamplerCtx& c) {           // :203-213
    const uint64_t h = dim_hash(c, s.dimension + 1u);
    uint64_t si = sampler_index(s, c);
    s.dimension += 1;
    return fast_owen(sobol_dim0(si), (uint32_t)h);
}
PT_DEV void get_2d_bits(Sampler& s, const SamplerCtx& c, uint32_t& b0, uint32_t& b1) {   // :215-230
    const uint64_t h = dim_hash(c, s.dimension + 2u);
    uint64_t si = sampler_index(s, c);
    s.dimension += 2;
    b0 = fast_owen(sobol_dim0(si), (uint32_t)h);
    b1 = fast_owen(sobol_dim1(si), (uint32_t)(h >> 32));
}
PT_DEV float get_1d(Sampler& s, const SamplerCtx& c) {
    if (c.mode == 0) return random_next(s);
    return sobol_bits_to_float(get_1d_bits(s, c));
}
PT_DEV f2 get_2d(Sampler& s, const SamplerCtx& c) {
    if (c.mode == 0) { float a = random_next(s); float b = random_next(s); return f2{a, b}; }
    uint32_t b0, b1; get_2d_bits(s, c, b0, b1);
    return f2{sobol_bits_to_float(b0), sobol_bits_to_float(b1)};
}

// ---------------------------------------------------------------------------------------------
// Spectra
// ---------------------------------------------------------------------------------------------
struct Wl {                 // SampledWavelengths (sampled_spectrum.rs:304-366)
    float lam0;             // the hero wavelength; the other three follow from it (wl_lams): one register of path state instead of four
    bool term;              // secondary wavelengths terminated: pdf = {(1/470)/4, 0, 0, 0}
};
PT_DEV void wl_init(Wl& w, float u) {
    w.lam0 = LAMBDA_MIN + u * (LAMBDA_MAX - LAMBDA_MIN);
    w.term = false;
}
// new_uniform_range (sampled_spectrum.rs:318-336): lambda_i = lambda_{i-1} + 470/4, wrapped into [360, 830) — the same three additions and
// wraps every time they are needed, so the values are bit-identical to the stored ones
PT_DEV void wl_lams(const Wl& w, float lam[4]) {
    lam[0] = w.lam0;
    const float delta = (LAMBDA_MAX - LAMBDA_MIN) / 4.0f;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        float l = lam[i - 1] + delta;
        if (l >= LAMBDA_MAX) l = LAMBDA_MIN + (l - LAMBDA_MAX);
        lam[i] = l;
    }
}
// sRGB decode of a texture colour (eotf.rs:40-52 has c / 12.92 and ((c + 0.055) / 1.055).powf(2.4)).  The colour only selects
// rgb2spec coefficients, a continuous map, so the hardware log2 / exp2 (relative error < 1e-6 on this range) replace the ~240 VALU
// instructions of a correctly rounded powf, three times per textured lookup; the two constant divisions become multiplies.
PT_DEV float srgb_eotf_inverse(float c) {
    return c <= 0.04045f ? c * (1.0f / 12.92f) : __builtin_amdgcn_exp2f(2.4f * __builtin_amdgcn_logf((c + 0.055f) * (1.0f / 1.055f)));
}
PT_DEV float lut_value(const float* lut, float lambda) {                 // densely_sampled_spectrum.rs:57-67
    if (!(lambda >= LAMBDA_MIN && lambda <= LAMBDA_MAX)) return 0.0f;
    int idx = (int)floorf(lambda - LAMBDA_MIN);
    return idx < 470 ? lut[idx] : 0.0f;
}
// PT_SIGMOID_EXACT 1: the reference's expression with the host libm's expf restated in double (and the division by 470): every albedo / emission
// value bit-equal to the reference's.  Default 0 - see DESIGN.md 2.1 for the measured share of bit-equal samples and the cost.
#ifndef PT_SIGMOID_EXACT
#define PT_SIGMOID_EXACT 0
#endif
PT_DEV float sigmoid_value(float c0, float c1, float c2, float lambda) {   // rgb_sigmoid_polynomial.rs:17-23,179-182
    float t = PT_SIGMOID_EXACT ? (lambda - LAMBDA_MIN) / (LAMBDA_MAX - LAMBDA_MIN) : (lambda - LAMBDA_MIN) * (1.0f / (LAMBDA_MAX - LAMBDA_MIN));   // the reference divides by 470: <= 1 ulp apart, albedo values only
    float x = t * t * c0 + t * c1 + c2;
    // 1 / (1 + exp(-x)) (rgb_sigmoid_polynomial.rs:18-20) through the hardware exp2 and reciprocal: a reflectance / emission
    // value, relative error < 1e-6 for the |x| < 50 the table produces; the correctly rounded form costs ~26 VALU instructions
    // more, 16 times per sample
#if PT_SIGMOID_EXACT
    float e;
    if (ptlibm::expf_glibc(-x, &e)) return 1.0f / (1.0f + e);          // the reference's own 1.0 / (1.0 + (-x).exp()) (f32::exp = the host's expf, pt_libm.hpp)
#endif
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
// (float)c / 255.0f for c = 0..255, correctly rounded at compile time: a load (the memory pipes are idle) instead of the ~10 VALU
// instructions of an IEEE division, twelve times per bilinear lookup
struct U8UnitTable { float v[256]; };
constexpr U8UnitTable make_u8_unit() { U8UnitTable t{}; for (int i = 0; i < 256; ++i) t.v[i] = (float)i / 255.0f; return t; }
__device__ constexpr U8UnitTable U8_UNIT = make_u8_unit();
#ifndef PT_U8_VALU
#define PT_U8_VALU 1
#endif
// (float)c / 255.0f in four VALU instructions and no memory round trip: q = c * fl(1/255) is off by an ulp for 126 of the 256 values; one
// residual step q + (c - 255 q) * fl(1/255), both in fma, is the correctly rounded quotient for every c in 0..255 (checked exhaustively on
// the host with glibc's fmaf, tests/test_abi.py).  The table variant put a dependent gather between the texel fetch and the colour maths.
PT_DEV float u8_unit(uint32_t c) {
#if PT_U8_VALU
    const float x = (float)c, r = 1.0f / 255.0f;
    const float q = x * r;
    return fmaf(fmaf(-255.0f, q, x), r, q);
#else
    return U8_UNIT.v[c];
#endif
}
PT_DEV void fetch_texel(const DevScene& sc, const DevTexture& t, uint32_t x, uint32_t y, float out[3]) {
    uint32_t v = sc.texels[t.offset + y * t.w + x];
    out[0] = u8_unit(v & 255u); out[1] = u8_unit((v >> 8) & 255u); out[2] = u8_unit((v >> 16) & 255u);   // = (float)c / 255.0f, exactly
}
PT_DEV void bilinear_rgb(const DevScene& sc, const DevTexture& t, f2 uv, float out[3]) {   // texture/sampler.rs:6-45
    float u = fabsf(uv.x - truncf(uv.x));
    float v = 1.0f - fabsf(uv.y - truncf(uv.y));
    float x = u * ((float)t.w - 1.0f), y = v * ((float)t.h - 1.0f);
    uint32_t x0 = (uint32_t)floorf(x), y0 = (uint32_t)floorf(y);
    uint32_t x1 = min(x0 + 1u, t.w - 1u), y1 = min(y0 + 1u, t.h - 1u);
    float fx = x - (float)x0, fy = y - (float)y0;
    float p00[3], p10[3], p01[3], p11[3];
    fetch_texel(sc, t, x0, y0, p00); fetch_texel(sc, t, x1, y0, p10);
    fetch_texel(sc, t, x0, y1, p01); fetch_texel(sc, t, x1, y1, p11);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float top = p00[c] * (1.0f - fx) + p10[c] * fx;
        float bottom = p01[c] * (1.0f - fx) + p11[c] * fx;
        out[c] = top * (1.0f - fy) + bottom * fy;
    }
}
PT_DEV void bilinear_rgb(const DevScene& sc, uint32_t tex, f2 uv, float out[3]) { const DevTexture t = sc.textures[tex]; bilinear_rgb(sc, t, uv, out); }
// The 64 z nodes of the table (256 B) live in LDS: the search for the z cell is a chain of 6 dependent reads, then 2 more for the cell's
// ends — LDS round trips (~100 cycles) instead of L1 ones (several hundred under this kernel's load) on the critical path of every
// textured lookup.  Filled by the kernel's prologue (pt_kernel.hpp); the LDS allocation stays inside the same 512-B granule.
#ifndef PT_ZNODES_LDS
#define PT_ZNODES_LDS 1
#endif
#if PT_ZNODES_LDS
static __shared__ float s_znodes[64];
#define PT_ZNODE(sc, i) s_znodes[i]
#else
#define PT_ZNODE(sc, i) (sc).z_nodes[i]
#endif
// RgbToSpectrumTable::get for gamma-encoded sRGB input (rgb_sigmoid_polynomial.rs:87-155); table repacked to float4 cells.
PT_DEV void rgb2spec_lookup(const DevScene& sc, const float enc[3], float c[3]) {
    float rgb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) rgb[i] = fmaxf(srgb_eotf_inverse(enc[i]), 0.0f);
    if (rgb[0] == rgb[1] && rgb[1] == rgb[2]) { c[0] = 0.0f; c[1] = 0.0f; c[2] = logf(rgb[0] / (1.0f - rgb[0])); return; }
    int mc = 0; float mx = rgb[0];
    if (rgb[1] > mx) { mx = rgb[1]; mc = 1; }
    if (rgb[2] > mx) { mc = 2; }
    float z = mc == 0 ? rgb[0] : (mc == 1 ? rgb[1] : rgb[2]);
    float r1 = mc == 0 ? rgb[1] : (mc == 1 ? rgb[2] : rgb[0]);
    float r2 = mc == 0 ? rgb[2] : (mc == 1 ? rgb[0] : rgb[1]);
    float x = r1 * 63.0f / z, y = r2 * 63.0f / z;
    int xi = min((int)x, 62), yi = min((int)y, 62);
    // first i in [0,62] with z_nodes[i+1] > z (else 62): the nodes increase monotonically -> binary search
#if PT_ZNODES_LDS
    // = the number of nodes 1..63 that are not above z, at most 62 (the nodes increase monotonically; a NaN z counts them all, like the
    // reference's search that finds no node above it): two rounds of independent LDS reads — every eighth node, then the seven inside
    // the octant — instead of a chain of six dependent ones
    int c1 = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) c1 += !(s_znodes[8 * k] > z) ? 1 : 0;
    int zi = 8 * c1;
#pragma unroll
    for (int j = 1; j < 8; ++j) zi += !(s_znodes[8 * c1 + j] > z) ? 1 : 0;
    zi = min(zi, 62);
#else
    int lo = 0, hi = 62;
    if (!(PT_ZNODE(sc, 63) > z)) lo = 62;
    else {
        while (lo < hi) { int mid = (lo + hi) >> 1; if (PT_ZNODE(sc, mid + 1) > z) hi = mid; else lo = mid + 1; }
    }
    int zi = lo;
#endif
    float zn0 = PT_ZNODE(sc, zi), zn1 = PT_ZNODE(sc, zi + 1);
    float dx = x - (float)xi, dy = y - (float)yi, dz = (z - zn0) / (zn1 - zn0);
    const float4* tab = (const float4*)sc.rgb2spec;
    size_t base = (((size_t)mc * 64 + zi) * 64 + yi) * 64 + xi;
    // trilinear in the order of the reference (x, then y, then z: rgb_sigmoid_polynomial.rs:140-152), one x-pair of cells at a time: eight
    // float4 cells live at once were the register peak of the textured kernels
#define PT_LERP(a, b, t) ((a) + ((b) - (a)) * (t))
    float r00[3], r10[3], r01[3], r11[3];
    { const float4 a = tab[base], b = tab[base + 1]; r00[0] = PT_LERP(a.x, b.x, dx); r00[1] = PT_LERP(a.y, b.y, dx); r00[2] = PT_LERP(a.z, b.z, dx); }
    { const float4 a = tab[base + 64], b = tab[base + 65]; r10[0] = PT_LERP(a.x, b.x, dx); r10[1] = PT_LERP(a.y, b.y, dx); r10[2] = PT_LERP(a.z, b.z, dx); }
    { const float4 a = tab[base + 4096], b = tab[base + 4097]; r01[0] = PT_LERP(a.x, b.x, dx); r01[1] = PT_LERP(a.y, b.y, dx); r01[2] = PT_LERP(a.z, b.z, dx); }
    { const float4 a = tab[base + 4160], b = tab[base + 4161]; r11[0] = PT_LERP(a.x, b.x, dx); r11[1] = PT_LERP(a.y, b.y, dx); r11[2] = PT_LERP(a.z, b.z, dx); }
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = PT_LERP(PT_LERP(r00[k], r10[k], dy), PT_LERP(r01[k], r11[k], dy), dz);
#undef PT_LERP
}

struct StatCounters {
    uint32_t closest_rays, shadow_rays, nodes_closest, tris_closest, nodes_shadow, tris_shadow, closest_hits, bounces, spectrum_evals,
        textured_lookups, samples;
    uint32_t w[8];      // wave-level step counts (incremented by the first active lane only), mi355pt_stats.wave_steps
    uint32_t hist[16];  // mi355pt_stats.busy_hist (wave leader only)
    uint32_t dv[4];     // mi355pt_stats.divergence (lane 0 only)
    uint32_t ties;      // closest-hit merges that met an EXACT tie in t with another triangle (mi355pt_stats.phase_cycles[9], low 32 bits)
    uint32_t ties_differ;   // ... between triangles of different material or geometric normal: the ties whose winner can matter (high 32 bits)
};
PT_DEV bool wave_leader() { return __builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0u)) == 0u; }
// number of set bits of a wave mask below this lane (v_mbcnt: no lane-mask registers to keep, two instructions)
PT_DEV uint32_t rank_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// SpectrumParameter::sample(uv).sample(lambda)  (parameter.rs:38-47, spectrum.rs:32-46)
// XT: the texture may be of SpectrumType Illuminant / Unbounded (emitters: the callers that pass TEX = FEAT_EMTEX)
template <bool STATS, bool TEX = true, bool XT = false>
PT_DEV void eval_spectrum(const DevScene& sc, const DevSpectrum& sp, const Wl& w, f2 uv, float out[4], StatCounters& st) {
    if (STATS) st.spectrum_evals++;
    float c0 = sp.c[0], c1 = sp.c[1], c2 = sp.c[2];
    uint32_t kind = sp.kind;
    float tex_scale = 1.0f; uint32_t tex_sub = 0u, tex_lut = 0u;
    if (TEX && kind == SPK_TEXTURE) {
        if (STATS) st.textured_lookups++;
        float rgb[3], c[3];
        bilinear_rgb(sc, DevTexture{sp.pad[0], sp.pad[1], sp.pad[2], 0u}, uv, rgb);
        if (XT) {
            // SpectrumType::{Illuminant, Unbounded} textures (rgb_texture.rs:56-64 -> rgb_illuminant_spectrum.rs:26-46, rgb_unbounded_spectrum.rs:23-42):
            // the sigmoid of rgb / (2 max rgb), times that scale (and the illuminant)
            tex_sub = __float_as_uint(sp.c[0]); tex_lut = __float_as_uint(sp.c[1]);
            if (tex_sub != 0u) {
                tex_scale = 2.0f * fmaxf(rgb[0], fmaxf(rgb[1], rgb[2]));
                if (tex_scale == 0.0f) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }      // black texel: 0 (the reference divides 0 / 0)
                rgb[0] = rgb[0] / tex_scale; rgb[1] = rgb[1] / tex_scale; rgb[2] = rgb[2] / tex_scale;
            }
        }
        rgb2spec_lookup(sc, rgb, c);
        c0 = c[0]; c1 = c[1]; c2 = c[2];
        kind = SPK_SIGMOID;
    }
    const float* lut = sc.luts + (size_t)sp.id * 470;
    float lam[4];
    wl_lams(w, lam);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v;
        if (kind == SPK_CONSTANT) v = c0;
        else if (kind == SPK_SIGMOID) v = sigmoid_value(c0, c1, c2, lam[i]);
        else if (kind == SPK_ILLUM) v = __uint_as_float(sp.pad[0]) * sigmoid_value(c0, c1, c2, lam[i]) * lut_value(lut, lam[i]);   // rgb_illuminant_spectrum.rs:44-46
        else v = lut_value(lut, lam[i]);
        if (XT && tex_sub != 0u) v = tex_sub == 1u ? (tex_scale * v) * lut_value(sc.luts + (size_t)tex_lut * 470, lam[i]) : tex_scale * v;
        out[i] = (i > 0 && w.term) ? 0.0f : v;
    }
}

// ---------------------------------------------------------------------------------------------
// Geometry: watertight ray/triangle (math/src/ray.rs:44-182) and BVH2 traversal with LDS stacks.
// ---------------------------------------------------------------------------------------------
struct TriVerts { f3 p0, p1, p2; };
PT_DEV TriVerts load_tri(const DevTri* tris, uint32_t i, uint32_t* mclass = nullptr) {
    const float4* q = (const float4*)(tris + i);
    float4 a = q[0], b = q[1], c = q[2];
    if (mclass) *mclass = __float_as_uint(c.w);                       // DevTri::mclass: sort class of the triangle's material (layout.hpp)
    return TriVerts{mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x)};
}

// the origin every TRIANGLE test of a traversal uses (boxes use the render-space origin): shifted into the meshes' common local space when the
// scene's triangle array holds local vertices (DevScene::tris_are_local), else + 0 (x + 0 is x)
PT_DEV f3 tri_origin(const DevScene& sc, f3 o) { return o + mk3(sc.tri_shift[0], sc.tri_shift[1], sc.tri_shift[2]); }

// returns true and (t, b0, b1, b2) when the ray hits within (0, t_max]
// DEGENERATE_CHECK false: the caller tests triangles of the tree, and the host left every triangle with a zero cross product out of it
// (scene.cpp: decided on these very vertices with this arithmetic), so the first test of ray.rs:49-56 cannot fire
template <bool DEGENERATE_CHECK = true>
PT_DEV bool intersect_triangle(f3 ro, f3 rd, int kx, int ky, int kz, float sx, float sy, float sz, float t_max, const TriVerts& tv,
                               float& t_out, float& b0o, float& b1o, float& b2o) {
    if (DEGENERATE_CHECK) {
        f3 c = cross(tv.p1 - tv.p0, tv.p2 - tv.p0);
        if (dot(c, c) == 0.0f) return false;                               // degenerate (:49-56)
    }
    f3 a0 = tv.p0 - ro, a1 = tv.p1 - ro, a2 = tv.p2 - ro;
    float p0x = comp(a0, kx), p0y = comp(a0, ky), p0z = comp(a0, kz);
    float p1x = comp(a1, kx), p1y = comp(a1, ky), p1z = comp(a1, kz);
    float p2x = comp(a2, kx), p2y = comp(a2, ky), p2z = comp(a2, kz);
    p0x += sx * p0z; p0y += sy * p0z;
    p1x += sx * p1z; p1y += sy * p1z;
    p2x += sx * p2z; p2y += sy * p2z;
    float e0 = p2x * p1y - p2y * p1x;
    float e1 = p0x * p2y - p0y * p2x;
    float e2 = p1x * p0y - p1y * p0x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {                         // f64 fallback (:91-101)
        e0 = (float)((double)p2x * (double)p1y - (double)p2y * (double)p1x);
        e1 = (float)((double)p0x * (double)p2y - (double)p0y * (double)p2x);
        e2 = (float)((double)p1x * (double)p0y - (double)p1y * (double)p0x);
    }
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
    float det = e0 + e1 + e2;
    if (det == 0.0f) return false;
    p0z *= sz; p1z *= sz; p2z *= sz;
    float t_scaled = e0 * p0z + e1 * p1z + e2 * p2z;
    if (det < 0.0f && (t_scaled >= 0.0f || t_scaled < t_max * det)) return false;
    if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > t_max * det)) return false;
    float inv_det = 1.0f / det;
    float t_hit = t_scaled * inv_det;
    // conservative t > 0 (:137-158); gamma(n) = n*eps/(1-n*eps), eps = 2^-24
    constexpr float EPSH = 1.1920929e-7f * 0.5f;
    constexpr float G2 = (2.0f * EPSH) / (1.0f - 2.0f * EPSH), G3 = (3.0f * EPSH) / (1.0f - 3.0f * EPSH),
                    G5 = (5.0f * EPSH) / (1.0f - 5.0f * EPSH);
    float max_zt = max3f(fabsf(p0z), fabsf(p1z), fabsf(p2z));
    float delta_z = G3 * max_zt;
    float max_xt = max3f(fabsf(p0x), fabsf(p1x), fabsf(p2x));
    float max_yt = max3f(fabsf(p0y), fabsf(p1y), fabsf(p2y));
    float delta_x = G5 * max_xt, delta_y = G5 * max_yt;
    float delta_e = 2.0f * (G2 * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    float max_e = max3f(fabsf(e0), fabsf(e1), fabsf(e2));
    float delta_t = 3.0f * (G3 * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * fabsf(inv_det);
    if (t_hit < delta_t) return false;
    t_out = t_hit; b0o = e0 * inv_det; b1o = e1 * inv_det; b2o = e2 * inv_det;
    return true;
}

struct RaySetup { int kx, ky, kz; float sx, sy, sz; f3 inv; };
PT_DEV RaySetup setup_ray(f3 rd) {
    RaySetup r;
    float ax = fabsf(rd.x), ay = fabsf(rd.y), az = fabsf(rd.z);
    int kz = 0; float m = ax;                                              // glam max_position: first maximum
    if (ay > m) { m = ay; kz = 1; }
    if (az > m) { kz = 2; }
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    float dx = comp(rd, kx), dy = comp(rd, ky), dz = comp(rd, kz);
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.sx = -dx / dz; r.sy = -dy / dz; r.sz = 1.0f / dz;
    r.inv = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    return r;
}

struct Hit { float t, b0, b1, b2; uint32_t tri; uint32_t mclass; };   // mclass: the sort class of the triangle's material (DevTri::pad[0]: MT_* | 8 if it has a spectrum texture), filled by trace_pair_coop


// ---- the reference's per-primitive lowering (primitive/impls/triangle_mesh.rs:89-119) ----
// glam Mat4::transform_point3 / transform_vector3 on the 3x4 part of a matrix held as its four columns' xyz
struct InstXf { f3 mx, my, mz, mw, ix, iy, iz, iw; bool identity; };
PT_DEV InstXf load_instance(const DevInstance* di) {
    const float4* q = (const float4*)di;
    const float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6];
    InstXf x;
    x.mx = mk3(a.x, a.y, a.z); x.my = mk3(a.w, b.x, b.y); x.mz = mk3(b.z, b.w, c.x); x.mw = mk3(c.y, c.z, c.w);
    x.ix = mk3(d.x, d.y, d.z); x.iy = mk3(d.w, e.x, e.y); x.iz = mk3(e.z, e.w, f.x); x.iw = mk3(f.y, f.z, f.w);
    x.identity = __float_as_uint(g.x) != 0u;
    return x;
}
PT_DEV f3 xf_vector(f3 cx, f3 cy, f3 cz, f3 v) { f3 r = cx * v.x; r = r + cy * v.y; r = r + cz * v.z; return r; }
PT_DEV f3 xf_point(f3 cx, f3 cy, f3 cz, f3 cw, f3 p) { f3 r = cx * p.x; r = r + cy * p.y; r = r + cz * p.z; r = r + cw; return r; }
// Transform * Normal (math/src/transform.rs:45-51): transpose(inverse) * n, renormalised by Normal::from
PT_DEV f3 xf_normal(const InstXf& x, f3 n) {
    return normalize(xf_vector(mk3(x.ix.x, x.iy.x, x.iz.x), mk3(x.ix.y, x.iy.y, x.iz.y), mk3(x.ix.z, x.iy.z, x.iz.z), n));
}
PT_DEV TriVerts load_tri_local(const DevTriLocal* tris, uint32_t i, uint32_t* instance, bool* identity, uint32_t* mclass = nullptr) {
    const float4* q = (const float4*)(tris + i);
    const float4 a = q[0], b = q[1], c = q[2];
    *instance = __float_as_uint(c.y);
    *identity = __float_as_uint(c.z) != 0u;                               // DevTri::flags bit 0: the instance is a translation (DevInstance::identity)
    if (mclass) *mclass = __float_as_uint(c.w);                           // DevTri::mclass
    return TriVerts{mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x)};
}
// the translation columns alone (one 16-byte load each): all an identity instance needs
PT_DEV f3 load_instance_mw(const DevInstance* di) { const float4 c = ((const float4*)di)[2]; return mk3(c.y, c.z, c.w); }
PT_DEV f3 load_instance_iw(const DevInstance* di) { const float4 f = ((const float4*)di)[5]; return mk3(f.y, f.z, f.w); }

// PT_EXACT_HIT 1 (default): the triangle the render-space traversal returned is intersected once more the way the reference does it - the
// ray carried into the mesh's local space by the numeric inverse of local_to_render, the watertight test on the mesh's own vertices —
// and the hit's barycentrics are replaced by that test's.  load_surface then builds position, normals and wo from LOCAL data through
// local_to_render, operation for operation (samples.rs:130-143), so the shading point equals the reference's bit for bit (measured: with
// this, exact GGX terms, libm's sin / cos and the numeric frame inverse every scene's frames equal the faithful oracle's to 1e-7 with the
// reference's own Russian-roulette gate).  If the local test rejects the triangle (an edge decided the other way by
// rounding: the reference would have hit a neighbour), the render-space barycentrics stay: one sample, an ulp off.
#ifndef PT_EXACT_HIT
#define PT_EXACT_HIT 1
#endif
// The barycentrics of the triangle a closest-hit traversal found (its last step: every traversal ends here).
//   PT_EXACT_HIT 0: the render-space test once more (same function, same inputs as inside the traversal);
//   PT_EXACT_HIT 1: the reference's test - local ray, local vertices; the render-space test only if that one rejects the triangle.
PT_DEV void winner_hit(const DevScene& sc, f3 ro, f3 rd, const RaySetup& rs, uint32_t tri, Hit& hit) {
    float t = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
    hit.tri = tri;
#if PT_EXACT_HIT
    bool done = false;
    if (!sc.tris_are_local) {                                             // (wave-uniform: a kernel argument)
        uint32_t inst; bool ident;
        const TriVerts tl = load_tri_local(sc.tris_local, tri, &inst, &ident, &hit.mclass);
        f3 ol, dl;
        if (ident) { ol = ro + load_instance_iw(sc.instances + inst); dl = rd; }   // 1 * a + 0 * b + 0 * c is a: the multiplies are exact
        else { const InstXf x = load_instance(sc.instances + inst); ol = xf_point(x.ix, x.iy, x.iz, x.iw, ro); dl = xf_vector(x.ix, x.iy, x.iz, rd); }
        const RaySetup ls = setup_ray(dl);
        done = intersect_triangle(ol, dl, ls.kx, ls.ky, ls.kz, ls.sx, ls.sy, ls.sz, 3.402823466e+38f, tl, t, b0, b1, b2);
    }
    if (!done)
#endif
    {   // the traversal's own test once more (same function, same inputs): in a tris_are_local scene that IS the reference's test
        const TriVerts tv = load_tri(sc.tris, tri, &hit.mclass);
        intersect_triangle<false>(tri_origin(sc, ro), rd, rs.kx, rs.ky, rs.kz, rs.sx, rs.sy, rs.sz, 3.402823466e+38f, tv, t, b0, b1, b2);
    }
    hit.t = t; hit.b0 = b0; hit.b1 = b1; hit.b2 = b2;
}

// Closest hit (Scene::intersect, scene.rs:80-90).  One flat BVH2; the far child goes to this lane's LDS stack
// (stack[depth*64 + lane]: consecutive lanes hit consecutive banks, no conflicts); t_best prunes both
// boxes and triangles, ties keep the first triangle found.
template <bool STATS>
PT_DEV bool trace_closest(const DevScene& sc, f3 ro, f3 rd, float t_max, uint32_t* stack, Hit& hit, StatCounters& st) {
    RaySetup rs = setup_ray(rd);
    float t_best = t_max;
    bool found = false;
    int sp = 0;
    int32_t cur = sc.root;
    if (STATS) st.closest_rays++;
    for (;;) {
        if (cur >= 0) {
            const float4* q = (const float4*)(sc.nodes + cur);
            float4 nx = q[0], ny = q[1], nz = q[2];
            int2 ch = *(const int2*)(q + 3);
            if (STATS) { st.nodes_closest++; if (wave_leader()) st.w[0]++; }
            float l0x = (nx.x - ro.x) * rs.inv.x, h0x = (nx.z - ro.x) * rs.inv.x;
            float l1x = (nx.y - ro.x) * rs.inv.x, h1x = (nx.w - ro.x) * rs.inv.x;
            float l0y = (ny.x - ro.y) * rs.inv.y, h0y = (ny.z - ro.y) * rs.inv.y;
            float l1y = (ny.y - ro.y) * rs.inv.y, h1y = (ny.w - ro.y) * rs.inv.y;
            float l0z = (nz.x - ro.z) * rs.inv.z, h0z = (nz.z - ro.z) * rs.inv.z;
            float l1z = (nz.y - ro.z) * rs.inv.z, h1z = (nz.w - ro.z) * rs.inv.z;
            float n0 = fmaxf(fmaxf(fminf(l0x, h0x), fminf(l0y, h0y)), fmaxf(fminf(l0z, h0z), 0.0f));
            float f0 = fminf(fminf(fmaxf(l0x, h0x), fmaxf(l0y, h0y)), fminf(fmaxf(l0z, h0z), t_best));
            float n1 = fmaxf(fmaxf(fminf(l1x, h1x), fminf(l1y, h1y)), fmaxf(fminf(l1z, h1z), 0.0f));
            float f1 = fminf(fminf(fmaxf(l1x, h1x), fmaxf(l1y, h1y)), fminf(fmaxf(l1z, h1z), t_best));
            bool hit0 = n0 <= f0, hit1 = n1 <= f1;
            if (hit0 && hit1) {
                bool first0 = n0 <= n1;
                int32_t nearc = first0 ? ch.x : ch.y, farc = first0 ? ch.y : ch.x;
                stack[sp * 64] = (uint32_t)farc; ++sp;
                cur = nearc;
                continue;
            } else if (hit0) { cur = ch.x; continue; }
            else if (hit1) { cur = ch.y; continue; }
        } else {
            uint32_t first = leaf_first(cur), cnt = leaf_count(cur);
            for (uint32_t i = 0; i < cnt; ++i) {
                TriVerts tv = load_tri(sc.tris, first + i);
                float t, b0, b1, b2;
                if (STATS) { st.tris_closest++; if (wave_leader()) st.w[1]++; }
                if (intersect_triangle<false>(tri_origin(sc, ro), rd, rs.kx, rs.ky, rs.kz, rs.sx, rs.sy, rs.sz, t_best, tv, t, b0, b1, b2)) {
                    if (!found || t < t_best) { found = true; t_best = t; hit.t = t; hit.b0 = b0; hit.b1 = b1; hit.b2 = b2; hit.tri = first + i; }
                }
            }
        }
        if (sp == 0) break;
        --sp; cur = (int32_t)stack[sp * 64];
    }
    if (PT_EXACT_HIT && found) winner_hit(sc, ro, rd, rs, hit.tri, hit);
    if (STATS && found) st.closest_hits++;
    return found;
}

// Any hit within (0, t_max] (Scene::intersect_p, scene.rs:93-103)
template <bool STATS>
PT_DEV bool trace_any(const DevScene& sc, f3 ro, f3 rd, float t_max, uint32_t* stack, StatCounters& st) {
    RaySetup rs = setup_ray(rd);
    int sp = 0;
    int32_t cur = sc.root;
    if (STATS) st.shadow_rays++;
    for (;;) {
        if (cur >= 0) {
            const float4* q = (const float4*)(sc.nodes + cur);
            float4 nx = q[0], ny = q[1], nz = q[2];
            int2 ch = *(const int2*)(q + 3);
            if (STATS) { st.nodes_shadow++; if (wave_leader()) st.w[2]++; }
            float l0x = (nx.x - ro.x) * rs.inv.x, h0x = (nx.z - ro.x) * rs.inv.x;
            float l1x = (nx.y - ro.x) * rs.inv.x, h1x = (nx.w - ro.x) * rs.inv.x;
            float l0y = (ny.x - ro.y) * rs.inv.y, h0y = (ny.z - ro.y) * rs.inv.y;
            float l1y = (ny.y - ro.y) * rs.inv.y, h1y = (ny.w - ro.y) * rs.inv.y;
            float l0z = (nz.x - ro.z) * rs.inv.z, h0z = (nz.z - ro.z) * rs.inv.z;
            float l1z = (nz.y - ro.z) * rs.inv.z, h1z = (nz.w - ro.z) * rs.inv.z;
            float n0 = fmaxf(fmaxf(fminf(l0x, h0x), fminf(l0y, h0y)), fmaxf(fminf(l0z, h0z), 0.0f));
            float f0 = fminf(fminf(fmaxf(l0x, h0x), fmaxf(l0y, h0y)), fminf(fmaxf(l0z, h0z), t_max));
            float n1 = fmaxf(fmaxf(fminf(l1x, h1x), fminf(l1y, h1y)), fmaxf(fminf(l1z, h1z), 0.0f));
            float f1 = fminf(fminf(fmaxf(l1x, h1x), fmaxf(l1y, h1y)), fminf(fmaxf(l1z, h1z), t_max));
            bool hit0 = n0 <= f0, hit1 = n1 <= f1;
            if (hit0 && hit1) { stack[sp * 64] = (uint32_t)ch.y; ++sp; cur = ch.x; continue; }
            else if (hit0) { cur = ch.x; continue; }
            else if (hit1) { cur = ch.y; continue; }
        } else {
            uint32_t first = leaf_first(cur), cnt = leaf_count(cur);
            for (uint32_t i = 0; i < cnt; ++i) {
                TriVerts tv = load_tri(sc.tris, first + i);
                float t, b0, b1, b2;
                if (STATS) { st.tris_shadow++; if (wave_leader()) st.w[3]++; }
                if (intersect_triangle<false>(tri_origin(sc, ro), rd, rs.kx, rs.ky, rs.kz, rs.sx, rs.sy, rs.sz, t_max, tv, t, b0, b1, b2)) return true;
            }
        }
        if (sp == 0) break;
        --sp; cur = (int32_t)stack[sp * 64];
    }
    return false;
}

// One step through the collapsed tree (layout.hpp DevNode4): slab tests of up to four child boxes.  n[c] = entry distance of child c, or
// +inf if the ray misses it.  Unused slots hold a POINT box at +FLT_MAX (bvh_builder.cpp collapse_bvh4): its slab distances are
// +-FLT_MAX * |1/d| (|1/d| >= 1 for a unit direction), outside [0, t_lim] as long as t_lim <= 1e30: the callers' contract, see below.
struct Node4Hits { float n[4]; int32_t link[4]; };
// PT_NODE_SEL 1: the NEAR and FAR plane of every slab are chosen by the sign of the ray direction through the LOAD ADDRESS (the node keeps
// lo and hi planes of an axis as two consecutive float4: near = the one at +16 B when 1/d is negative) instead of by a min / max pair per
// child and axis after the fact: (lo - o) * inv <= (hi - o) * inv exactly when inv > 0 (rounding is monotonic), so the values are the ones
// min / max produced, for 24 VALU instructions less per node step (of ~125).  0: the min / max form.
#ifndef PT_NODE_SEL
#define PT_NODE_SEL 1
#endif
#if PT_NODE_Q16
// The quantised form (layout.hpp DevNode4Q): `ro` holds (grid origin - o) / d and `inv` cell / d (walk_ray below), so a plane's distance is
// one fma on the converted 16-bit integer.  Four requests: the x, y, z rows (lo and hi planes of the four children in ONE 16-byte row each)
// and the links; near / far are chosen by the sign of the direction among the row's two halves.
PT_DEV Node4Hits node4q_step(const DevNode4Q* nodes, int32_t cur, f3 ro, f3 inv, float t_lim) {
    const uint4* q = (const uint4*)(nodes + cur);
    const uint4 X = q[0], Y = q[1], Z = q[2];
    const int4 ch = *(const int4*)(q + 3);
#if PT_NODE_LOADS_FIRST
    __builtin_amdgcn_sched_barrier(0);
#endif
    const bool sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;        // (cell > 0: the sign of cell / d is the sign of d; -0 counts as +)
    const uint32_t nx01 = sx ? X.z : X.x, nx23 = sx ? X.w : X.y, fx01 = sx ? X.x : X.z, fx23 = sx ? X.y : X.w;
    const uint32_t ny01 = sy ? Y.z : Y.x, ny23 = sy ? Y.w : Y.y, fy01 = sy ? Y.x : Y.z, fy23 = sy ? Y.y : Y.w;
    const uint32_t nz01 = sz ? Z.z : Z.x, nz23 = sz ? Z.w : Z.y, fz01 = sz ? Z.x : Z.z, fz23 = sz ? Z.y : Z.w;
#define PT_QLO(v) ((float)((v) & 0xffffu))
#define PT_QHI(v) ((float)((v) >> 16))
#define PT_SLABQ(NXW, NYW, NZW, FXW, FYW, FZW, H, N, F)                                                                                   \
    N = fmaxf(fmaxf(fmaxf(fmaf(H(NXW), inv.x, ro.x), fmaf(H(NYW), inv.y, ro.y)), fmaf(H(NZW), inv.z, ro.z)), 0.0f);                       \
    F = fminf(fminf(fminf(fmaf(H(FXW), inv.x, ro.x), fmaf(H(FYW), inv.y, ro.y)), fmaf(H(FZW), inv.z, ro.z)), t_lim);
    float n0, n1, n2, n3, f0, f1, f2, f3_;
    PT_SLABQ(nx01, ny01, nz01, fx01, fy01, fz01, PT_QLO, n0, f0)
    PT_SLABQ(nx01, ny01, nz01, fx01, fy01, fz01, PT_QHI, n1, f1)
    PT_SLABQ(nx23, ny23, nz23, fx23, fy23, fz23, PT_QLO, n2, f2)
    PT_SLABQ(nx23, ny23, nz23, fx23, fy23, fz23, PT_QHI, n3, f3_)
#undef PT_SLABQ
#undef PT_QLO
#undef PT_QHI
    Node4Hits h;
    h.n[0] = n0 <= f0 ? n0 : INFINITY; h.n[1] = n1 <= f1 ? n1 : INFINITY; h.n[2] = n2 <= f2 ? n2 : INFINITY; h.n[3] = n3 <= f3_ ? n3 : INFINITY;
    h.link[0] = ch.x; h.link[1] = ch.y; h.link[2] = ch.z; h.link[3] = ch.w;
    return h;
}
#endif
PT_DEV Node4Hits node4_step(const DevNode4* nodes, int32_t cur, f3 ro, f3 inv, float t_lim) {
    // CONTRACT: t_lim <= 1e30 (every caller clamps its limit once per ray, not once per step: trace_any_deferred / trace_closest_coop /
    // trace_pair_coop initialise and only ever shrink w_t / w_tmax / w_tbest from min(t, 1e30)).
    float n0, n1, n2, n3, f0, f1, f2, f3_;
#if PT_NODE_SEL
    const char* base = (const char*)nodes;
    const uint32_t o = (uint32_t)cur << 7;
    const uint32_t ox = o + ((__float_as_uint(inv.x) >> 27) & 16u), oy = o + ((__float_as_uint(inv.y) >> 27) & 16u), oz = o + ((__float_as_uint(inv.z) >> 27) & 16u);
    const float4 ax = *(const float4*)(base + ox), bx = *(const float4*)(base + (ox ^ 16u));
    const float4 ay = *(const float4*)(base + 32 + oy), by = *(const float4*)(base + 32 + (oy ^ 16u));
    const float4 az = *(const float4*)(base + 64 + oz), bz = *(const float4*)(base + 64 + (oz ^ 16u));
    const int4 ch = *(const int4*)(base + 96 + o);
    // all seven loads of the node are issued before the first slab is evaluated: left alone, the scheduler sinks the links' load below the
    // box arithmetic (fewer live registers) and the step then waits for a SECOND L1 round trip at its very end
#ifndef PT_NODE_LOADS_FIRST
#define PT_NODE_LOADS_FIRST 1
#endif
#ifdef PT_ABLATE_EXTRA_NODE_LOADS   // timing experiment: N more 16-B requests per lane and node step (same cache line, results unused) — how
    {                               // much does a node step's time depend on the NUMBER of L1 requests?  
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f extra[PT_ABLATE_EXTRA_NODE_LOADS];
#pragma unroll
        for (int e = 0; e < PT_ABLATE_EXTRA_NODE_LOADS; ++e) extra[e] = *(const volatile v4f*)(base + o + 112 - 16 * (e % 7));
#pragma unroll
        for (int e = 0; e < PT_ABLATE_EXTRA_NODE_LOADS; ++e) asm volatile("" ::"v"(extra[e]));
    }
#endif
#if PT_NODE_LOADS_FIRST
    __builtin_amdgcn_sched_barrier(0);
#endif
#if PT_NODE_FMA     // `ro` holds -(o * inv) (walk_origin below), the boxes are padded by the host: layout.hpp
#define PT_SLAB(C, N, F) N = fmaxf(fmaxf(fmaxf(fmaf(ax.C, inv.x, ro.x), fmaf(ay.C, inv.y, ro.y)), fmaf(az.C, inv.z, ro.z)), 0.0f); \
                         F = fminf(fminf(fminf(fmaf(bx.C, inv.x, ro.x), fmaf(by.C, inv.y, ro.y)), fmaf(bz.C, inv.z, ro.z)), t_lim);
#else
#define PT_SLAB(C, N, F) N = fmaxf(fmaxf(fmaxf((ax.C - ro.x) * inv.x, (ay.C - ro.y) * inv.y), (az.C - ro.z) * inv.z), 0.0f); \
                         F = fminf(fminf(fminf((bx.C - ro.x) * inv.x, (by.C - ro.y) * inv.y), (bz.C - ro.z) * inv.z), t_lim);
#endif
    PT_SLAB(x, n0, f0) PT_SLAB(y, n1, f1) PT_SLAB(z, n2, f2) PT_SLAB(w, n3, f3_)
#undef PT_SLAB
#else
    const float4* q = (const float4*)(nodes + cur);
    // axis by axis: the interval of each child narrows as its x, y, z slabs arrive (max / min are exact, so the order does not change a
    // result); two float4 of planes are live at a time instead of six — the traversal loop sits inside the register budget of the path state
    n0 = 0.0f; n1 = 0.0f; n2 = 0.0f; n3 = 0.0f; f0 = t_lim; f1 = t_lim; f2 = t_lim; f3_ = t_lim;
#define PT_AXIS4(LO, HI, O, I)                                                                                              \
    {                                                                                                                       \
        const float4 lo = q[LO], hi = q[HI];                                                                                \
        float l, h;                                                                                                         \
        l = (lo.x - O) * I; h = (hi.x - O) * I; n0 = fmaxf(n0, fminf(l, h)); f0 = fminf(f0, fmaxf(l, h));                    \
        l = (lo.y - O) * I; h = (hi.y - O) * I; n1 = fmaxf(n1, fminf(l, h)); f1 = fminf(f1, fmaxf(l, h));                    \
        l = (lo.z - O) * I; h = (hi.z - O) * I; n2 = fmaxf(n2, fminf(l, h)); f2 = fminf(f2, fmaxf(l, h));                    \
        l = (lo.w - O) * I; h = (hi.w - O) * I; n3 = fmaxf(n3, fminf(l, h)); f3_ = fminf(f3_, fmaxf(l, h));                  \
    }
    PT_AXIS4(0, 1, ro.x, inv.x) PT_AXIS4(2, 3, ro.y, inv.y) PT_AXIS4(4, 5, ro.z, inv.z)
#undef PT_AXIS4
    const int4 ch = *(const int4*)(q + 6);
#endif
    Node4Hits h;
    h.n[0] = n0 <= f0 ? n0 : INFINITY; h.n[1] = n1 <= f1 ? n1 : INFINITY; h.n[2] = n2 <= f2 ? n2 : INFINITY; h.n[3] = n3 <= f3_ ? n3 : INFINITY;
    h.link[0] = ch.x; h.link[1] = ch.y; h.link[2] = ch.z; h.link[3] = ch.w;
    return h;
}
// what a walking lane keeps as the "origin" of its ray: the origin itself, or with PT_NODE_FMA in the 4-wide tree the slab addend -(o * 1/d)
#if PT_NODE_FMA && !PT_NODE_SEL
#error "PT_NODE_FMA needs PT_NODE_SEL"
#endif
template <bool WIDE>
PT_DEV f3 walk_origin(const DevScene& sc, f3 ro, f3 inv) {
    if (WIDE && PT_NODE_Q16) return mk3((sc.grid_org[0] - ro.x) * inv.x, (sc.grid_org[1] - ro.y) * inv.y, (sc.grid_org[2] - ro.z) * inv.z);
    if (WIDE && PT_NODE_FMA) return mk3(-(ro.x * inv.x), -(ro.y * inv.y), -(ro.z * inv.z));
    return ro;
}
// ... and as its reciprocal direction: 1 / d, or with PT_NODE_Q16 the grid cell over d
template <bool WIDE>
PT_DEV f3 walk_inv(const DevScene& sc, f3 inv) {
    if (WIDE && PT_NODE_Q16) return mk3(sc.grid_cell[0] * inv.x, sc.grid_cell[1] * inv.y, sc.grid_cell[2] * inv.z);
    return inv;
}
PT_DEV Node4Hits wide_step(const DevScene& sc, int32_t cur, f3 wo, f3 wi, float t_lim) {
#if PT_NODE_Q16
    return node4q_step(sc.nodes4q, cur, wo, wi, t_lim);
#else
    return node4_step(sc.nodes4, cur, wo, wi, t_lim);
#endif
}
// nearest child to slot 0 (PT_SORT_MODE 1) or ascending by entry distance (0); misses (+inf) are skipped by the pushes
// stack pushes of the 4-wide step as stores-always / advance-conditionally: a slot written for a miss lies above the top and is never read
// (the collapse bounds the need below STACK_DEPTH, so slot `sp` itself always exists); three LDS stores instead of three exec-mask regions:
// +0.4...+0.6 % once the step was down to three compare-exchanges (0.0 before)
#ifndef PT_PUSH_BRANCHFREE
#define PT_PUSH_BRANCHFREE 1
#endif
#ifndef PT_SORT_MODE
#define PT_SORT_MODE 1      // 0: full sorting network (5 compare-exchanges); 1: nearest child first, the rest in slot order (3): the
                            // later pops are a little less well ordered, the step is 16 instructions shorter: +0.4...+0.7 %
#endif
PT_DEV void sort4(Node4Hits& h) {
#define PT_CSWAP(a, b) { const bool s = h.n[b] < h.n[a]; const float tn = s ? h.n[b] : h.n[a]; const float tx = s ? h.n[a] : h.n[b]; \
                         const int32_t ln = s ? h.link[b] : h.link[a]; const int32_t lx = s ? h.link[a] : h.link[b]; h.n[a] = tn; h.n[b] = tx; h.link[a] = ln; h.link[b] = lx; }
#if PT_SORT_MODE == 1
    PT_CSWAP(0, 1) PT_CSWAP(0, 2) PT_CSWAP(0, 3)          // the minimum to slot 0; misses (+inf) among 1..3 are skipped by the pushes
#else
    PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2) PT_CSWAP(1, 3) PT_CSWAP(1, 2)
#endif
#undef PT_CSWAP
}

// Any hit with DEFERRED, DENSE triangle tests (wave-cooperative; every lane of the wave must call it, `want` = this lane has
// a ray).  In the plain loop above a wave spends most of its any-hit VALU time on triangle steps executed for the one or
// two straggler lanes that happen to sit in a leaf (measured: 24 triangle steps per wave iteration at 3.7 % lane use).
// Here lanes never test triangles themselves: a lane that reaches a leaf appends (triangle, lane) pairs to a per-wave ring
// in LDS and keeps traversing; whenever 64 pairs are queued, the whole wave tests them at once — lane i takes pair i,
// fetches the owner's ray with ds_bpermute and runs the same intersect_triangle().  Any-hit results are order independent
// (exists a hit in (0, t_max]), so the outcome is identical to trace_any; an occluded lane stops at the next flush.
constexpr uint32_t ANY_RING = 256;            // entries; a step appends <= 2 per lane, a flush leaves < 64 behind
struct AnyLds { uint32_t* ring; uint32_t* occl; uint32_t* pair; };   // ring[ANY_RING] (tri | owner << 26), occl[2] (bit per lane), pair[64] (steal rounds)

#ifndef PT_ANY_STEAL
#define PT_ANY_STEAL 1        // idle lanes take whole subtrees off the stacks of the lanes still walking (same ray, same result)
#endif
#ifndef PT_STEAL_EVERY
#define PT_STEAL_EVERY 1u
#endif
#ifndef PT_STEAL_EVERY_CLOSEST
#define PT_STEAL_EVERY_CLOSEST 1u
#endif
#ifndef PT_STEAL_MAX_ACTIVE
#define PT_STEAL_MAX_ACTIVE 52
#endif
template <bool STATS, bool WIDE>
PT_DEV bool trace_any_deferred(const DevScene& sc, f3 ro, f3 rd, float t_max, bool want, uint32_t* stack, uint32_t lane, const AnyLds& L,
                               StatCounters& st) {
    if (!want) { rd = mk3(0.0f, 0.0f, 1.0f); ro = mk3(0.0f, 0.0f, 0.0f); t_max = 0.0f; }
    RaySetup rs = setup_ray(rd);
    const uint32_t kpack = (uint32_t)rs.kx | ((uint32_t)rs.ky << 2) | ((uint32_t)rs.kz << 4);
    if (lane < 2) L.occl[lane] = 0u;
    __syncthreads();
    // the ray this lane is WALKING (its own, or one it is helping with) — the lane's own ray stays in ro/rd/rs for the flushes
    // (box tests only see distances up to 1e30: the unused slots of a DevNode4 are point boxes at FLT_MAX, whose slab distance FLT_MAX / |d|
    // must never fall inside [0, t_lim] — an unbounded shadow ray (t_max = FLT_MAX, directional and environment lights) would let it)
    f3 w_ro = walk_origin<WIDE>(sc, ro, rs.inv), w_inv = walk_inv<WIDE>(sc, rs.inv); float w_tmax = fminf(t_max, 1e30f); uint32_t owner = lane;
    int sp = 0, sb = 0;
    int32_t cur = WIDE ? sc.root4 : sc.root;
    uint32_t leaf_off = 0;                    // triangles of the current leaf already queued
    bool done = !want;
    uint32_t head = 0, tail = 0;              // wave-uniform ring cursors
    uint32_t since_steal = 0;
    if (STATS && want) st.shadow_rays++;

    auto flush = [&](uint32_t n) {            // test ring entries [head, head + n), n <= 64
        const bool valid = lane < n;
        const uint32_t e = valid ? L.ring[(head + lane) & (ANY_RING - 1u)] : (lane << 26);
        const uint32_t own = e >> 26, tri = e & 0x03ffffffu;
        f3 o2 = tri_origin(sc, mk3(__shfl(ro.x, own), __shfl(ro.y, own), __shfl(ro.z, own)));
        f3 d2 = mk3(__shfl(rd.x, own), __shfl(rd.y, own), __shfl(rd.z, own));
        float tm = __shfl(t_max, own);
        uint32_t kp = __shfl(kpack, own);
        float sx = __shfl(rs.sx, own), sy = __shfl(rs.sy, own), sz = __shfl(rs.sz, own);
        if (valid) {
            TriVerts tv = load_tri(sc.tris, tri);
            float t, b0, b1, b2;
            if (STATS) { st.tris_shadow++; if (wave_leader()) st.w[3]++; }
            if (intersect_triangle<false>(o2, d2, (int)(kp & 3u), (int)((kp >> 2) & 3u), (int)(kp >> 4), sx, sy, sz, tm, tv, t, b0, b1, b2))
                atomicOr(&L.occl[own >> 5], 1u << (own & 31u));
        }
        head += n;
        __syncthreads();
        if ((L.occl[owner >> 5] >> (owner & 31u)) & 1u) done = true;  // early out: the ray this lane walks is already occluded
    };

    for (;;) {
        if (!done && cur >= 0) {
            if constexpr (WIDE) {
                if (STATS) { st.nodes_shadow++; const int busy = __popcll(__ballot(true)); if (wave_leader()) { st.w[2]++; st.hist[8 + ((busy - 1) >> 3)]++; } }
                const Node4Hits h = wide_step(sc, cur, w_ro, w_inv, w_tmax);
                // any-hit needs no order: continue into the first child hit, queue the others
                int32_t nxt = 0; bool have = false;
    #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (h.n[c] < INFINITY) {
                        if (!have) { have = true; nxt = h.link[c]; }
                        else { stack[sp * 64] = (uint32_t)h.link[c]; ++sp; }
                    }
                }
                if (have) cur = nxt;
                else if (sp == sb) done = true;
                else { --sp; cur = (int32_t)stack[sp * 64]; }
            } else {
                const float4* q = (const float4*)(sc.nodes + cur);
                float4 nx = q[0], ny = q[1], nz = q[2];
                int2 ch = *(const int2*)(q + 3);
                if (STATS) { st.nodes_shadow++; const int busy = __popcll(__ballot(true)); if (wave_leader()) { st.w[2]++; st.hist[8 + ((busy - 1) >> 3)]++; } }
                float l0x = (nx.x - w_ro.x) * w_inv.x, h0x = (nx.z - w_ro.x) * w_inv.x;
                float l1x = (nx.y - w_ro.x) * w_inv.x, h1x = (nx.w - w_ro.x) * w_inv.x;
                float l0y = (ny.x - w_ro.y) * w_inv.y, h0y = (ny.z - w_ro.y) * w_inv.y;
                float l1y = (ny.y - w_ro.y) * w_inv.y, h1y = (ny.w - w_ro.y) * w_inv.y;
                float l0z = (nz.x - w_ro.z) * w_inv.z, h0z = (nz.z - w_ro.z) * w_inv.z;
                float l1z = (nz.y - w_ro.z) * w_inv.z, h1z = (nz.w - w_ro.z) * w_inv.z;
                float n0 = fmaxf(fmaxf(fminf(l0x, h0x), fminf(l0y, h0y)), fmaxf(fminf(l0z, h0z), 0.0f));
                float f0 = fminf(fminf(fmaxf(l0x, h0x), fmaxf(l0y, h0y)), fminf(fmaxf(l0z, h0z), w_tmax));
                float n1 = fmaxf(fmaxf(fminf(l1x, h1x), fminf(l1y, h1y)), fmaxf(fminf(l1z, h1z), 0.0f));
                float f1 = fminf(fminf(fmaxf(l1x, h1x), fmaxf(l1y, h1y)), fminf(fmaxf(l1z, h1z), w_tmax));
                bool hit0 = n0 <= f0, hit1 = n1 <= f1;
                if (hit0 && hit1) { stack[sp * 64] = (uint32_t)ch.y; ++sp; cur = ch.x; }
                else if (hit0) cur = ch.x;
                else if (hit1) cur = ch.y;
                else if (sp == sb) done = true;
                else { --sp; cur = (int32_t)stack[sp * 64]; }
            }
        }
        // lanes sitting in a leaf queue up to two of its triangles per step, then move on
        const bool at_leaf = !done && cur < 0;
        const unsigned long long m1 = __ballot(at_leaf);
        if (m1 != 0ull) {
            uint32_t first = 0, rem = 0;
            if (at_leaf) { first = leaf_first(cur) + leaf_off; rem = leaf_count(cur) - leaf_off; }
            const uint32_t a = rem < 2u ? rem : 2u;
            const unsigned long long m2 = __ballot(a == 2u);
            const uint32_t pos = tail + rank_below(m1) + rank_below(m2);
            if (a >= 1u) L.ring[pos & (ANY_RING - 1u)] = first | (owner << 26);
            if (a == 2u) L.ring[(pos + 1u) & (ANY_RING - 1u)] = (first + 1u) | (owner << 26);
            tail += (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
            if (at_leaf) {
                if (rem > 2u) leaf_off += 2u;
                else { leaf_off = 0u; if (sp == sb) done = true; else { --sp; cur = (int32_t)stack[sp * 64]; } }
            }
            __syncthreads();
            while (tail - head >= 64u) flush(64u);
        }
        const unsigned long long m_act = __ballot(!done);
        if (m_act == 0ull) break;
#if PT_ANY_STEAL
        // work stealing: when most of the wave idles, idle lanes take the BOTTOM stack entry (the largest pending subtree) of
        // lanes that still have one, together with that lane's working ray.  Any-hit is an OR over all subtrees, so the
        // result is unchanged; the owner's bit in occl[] is shared by everybody working on the ray.
        if (++since_steal >= PT_STEAL_EVERY && __popcll(m_act) <= PT_STEAL_MAX_ACTIVE) {
            const unsigned long long m_donor = __ballot(!done && sp > sb);
            if (m_donor != 0ull) {
                since_steal = 0u;
                const unsigned long long m_idle = ~m_act;
                const uint32_t n_pairs = min((uint32_t)__popcll(m_donor), (uint32_t)__popcll(m_idle));
                const bool donor = !done && sp > sb;
                const uint32_t rank = rank_below(donor ? m_donor : m_idle);
                if (donor && rank < n_pairs) L.pair[rank] = lane;
                __syncthreads();
                const bool taker = done && rank < n_pairs;
                const uint32_t from = taker ? L.pair[rank] : lane;
                // every lane reads its partner's state (shuffles are wave-wide); only takers keep it
                const int d_sb = __shfl(sb, from);
                const float rx = __shfl(w_ro.x, from), ry = __shfl(w_ro.y, from), rz = __shfl(w_ro.z, from);
                const float ix = __shfl(w_inv.x, from), iy = __shfl(w_inv.y, from), iz = __shfl(w_inv.z, from);
                const float tm = __shfl(w_tmax, from);
                const uint32_t ow = __shfl(owner, from);
                if (taker) {
                    cur = (int32_t)(stack - lane)[d_sb * 64 + from];      // the donor's bottom entry
                    w_ro = mk3(rx, ry, rz); w_inv = mk3(ix, iy, iz); w_tmax = tm; owner = ow;
                    sp = sb = 0; leaf_off = 0u; done = false;
                }
                if (donor && rank < n_pairs) ++sb;
                __syncthreads();
            }
        }
#endif
    }
    if (tail != head) flush(tail - head);     // stragglers' last pairs (tail - head < 64 here)
    return ((L.occl[lane >> 5] >> (lane & 31u)) & 1u) != 0u;
}

// Closest hit, wave-cooperative: the same three ideas as trace_any_deferred — lanes only walk nodes (near child first,
// pruned by the best distance known for the ray they walk), queue (triangle, owner) pairs, the wave tests the queue densely,
// and idle lanes steal whole subtrees from the bottom of busy lanes' stacks.  Results are merged per ray with an LDS
// atomicMin on (float bits of t << 32 | triangle): the closest hit is order independent; exact ties go to the lower
// triangle index (the plain loop keeps the first one found — a measure-zero difference).  Pruning lags the plain loop by at
// most one flush, so a few more nodes are visited per ray, at 2-3x the lane utilisation.
#ifndef PT_CLOSEST_FLUSH_MIN
#define PT_CLOSEST_FLUSH_MIN 64u      // measured: partial flushes (16/32/48) for earlier pruning cost more than they save
#endif
struct ClosestLds { uint32_t* ring; unsigned long long* best; uint32_t* pair; };   // ring[ANY_RING], best[64], pair[64]

template <bool STATS, bool WIDE>
PT_DEV bool trace_closest_coop(const DevScene& sc, f3 ro, f3 rd, bool want, uint32_t* stack, uint32_t lane, const ClosestLds& L, Hit& hit,
                               StatCounters& st) {
    if (!want) { rd = mk3(0.0f, 0.0f, 1.0f); ro = mk3(0.0f, 0.0f, 0.0f); }
    RaySetup rs = setup_ray(rd);
    const uint32_t kpack = (uint32_t)rs.kx | ((uint32_t)rs.ky << 2) | ((uint32_t)rs.kz << 4);
    L.best[lane] = (0x7f7fffffull << 32) | 0xffffffffull;                 // (FLT_MAX, no triangle)
    __syncthreads();
    f3 w_ro = walk_origin<WIDE>(sc, ro, rs.inv), w_inv = walk_inv<WIDE>(sc, rs.inv); float w_tbest = 1e30f; uint32_t owner = lane;             // box-test limit, see trace_any_deferred
    int sp = 0, sb = 0;
    int32_t cur = WIDE ? sc.root4 : sc.root;
    uint32_t leaf_off = 0;
    bool done = !want;
    uint32_t head = 0, tail = 0;
    uint32_t since_steal = 0u;
    if (STATS && want) st.closest_rays++;

    auto flush = [&](uint32_t n) {
        const bool valid = lane < n;
        const uint32_t e = valid ? L.ring[(head + lane) & (ANY_RING - 1u)] : (lane << 26);
        const uint32_t own = e >> 26, tri = e & 0x03ffffffu;
        f3 o2 = tri_origin(sc, mk3(__shfl(ro.x, own), __shfl(ro.y, own), __shfl(ro.z, own)));
        f3 d2 = mk3(__shfl(rd.x, own), __shfl(rd.y, own), __shfl(rd.z, own));
        uint32_t kp = __shfl(kpack, own);
        float sx = __shfl(rs.sx, own), sy = __shfl(rs.sy, own), sz = __shfl(rs.sz, own);
        if (valid) {
            const float t_lim = __uint_as_float((uint32_t)(L.best[own] >> 32));
            TriVerts tv = load_tri(sc.tris, tri);
            float t, b0, b1, b2;
            if (STATS) { st.tris_closest++; if (wave_leader()) st.w[1]++; }
            if (intersect_triangle<false>(o2, d2, (int)(kp & 3u), (int)((kp >> 2) & 3u), (int)(kp >> 4), sx, sy, sz, t_lim, tv, t, b0, b1, b2)) {
                if (STATS) {    // exact ties in t (see trace_pair_coop)
                    const unsigned long long cur = L.best[own];
                    if ((uint32_t)(cur >> 32) == __float_as_uint(t) && (uint32_t)cur != tri) {
                        st.ties++;
                        const float4* sa = (const float4*)(sc.shade + tri); const float4* sb = (const float4*)(sc.shade + (uint32_t)cur);
                        const float4 ga = sa[6], gb = sb[6];
                        if (sa[4].z != sb[4].z || ga.x != gb.x || ga.y != gb.y || ga.z != gb.z) st.ties_differ++;
                    }
                }
                atomicMin(&L.best[own], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)tri);
            }
        }
        head += n;
        __syncthreads();
        w_tbest = fminf(w_tbest, __uint_as_float((uint32_t)(L.best[owner] >> 32)));
    };

    for (;;) {
        if (!done && cur >= 0) {
            if constexpr (WIDE) {
                if (STATS) { st.nodes_closest++; const int busy = __popcll(__ballot(true)); if (wave_leader()) { st.w[0]++; st.hist[0 + ((busy - 1) >> 3)]++; } }
                Node4Hits h = wide_step(sc, cur, w_ro, w_inv, w_tbest);
                sort4(h);
                // nearest child next; the others go to the stack farthest first, so that the nearer of them is popped first
#if PT_PUSH_BRANCHFREE
                stack[sp * 64] = (uint32_t)h.link[3]; sp += h.n[3] < INFINITY ? 1 : 0;
                stack[sp * 64] = (uint32_t)h.link[2]; sp += h.n[2] < INFINITY ? 1 : 0;
                stack[sp * 64] = (uint32_t)h.link[1]; sp += h.n[1] < INFINITY ? 1 : 0;
#else
                if (h.n[3] < INFINITY) { stack[sp * 64] = (uint32_t)h.link[3]; ++sp; }
                if (h.n[2] < INFINITY) { stack[sp * 64] = (uint32_t)h.link[2]; ++sp; }
                if (h.n[1] < INFINITY) { stack[sp * 64] = (uint32_t)h.link[1]; ++sp; }
#endif
                if (h.n[0] < INFINITY) cur = h.link[0];
                else if (sp == sb) done = true;
                else { --sp; cur = (int32_t)stack[sp * 64]; }
            } else {
                const float4* q = (const float4*)(sc.nodes + cur);
                float4 nx = q[0], ny = q[1], nz = q[2];
                int2 ch = *(const int2*)(q + 3);
                if (STATS) { st.nodes_closest++; const int busy = __popcll(__ballot(true)); if (wave_leader()) { st.w[0]++; st.hist[0 + ((busy - 1) >> 3)]++; } }
                float l0x = (nx.x - w_ro.x) * w_inv.x, h0x = (nx.z - w_ro.x) * w_inv.x;
                float l1x = (nx.y - w_ro.x) * w_inv.x, h1x = (nx.w - w_ro.x) * w_inv.x;
                float l0y = (ny.x - w_ro.y) * w_inv.y, h0y = (ny.z - w_ro.y) * w_inv.y;
                float l1y = (ny.y - w_ro.y) * w_inv.y, h1y = (ny.w - w_ro.y) * w_inv.y;
                float l0z = (nz.x - w_ro.z) * w_inv.z, h0z = (nz.z - w_ro.z) * w_inv.z;
                float l1z = (nz.y - w_ro.z) * w_inv.z, h1z = (nz.w - w_ro.z) * w_inv.z;
                float n0 = fmaxf(fmaxf(fminf(l0x, h0x), fminf(l0y, h0y)), fmaxf(fminf(l0z, h0z), 0.0f));
                float f0 = fminf(fminf(fmaxf(l0x, h0x), fmaxf(l0y, h0y)), fminf(fmaxf(l0z, h0z), w_tbest));
                float n1 = fmaxf(fmaxf(fminf(l1x, h1x), fminf(l1y, h1y)), fmaxf(fminf(l1z, h1z), 0.0f));
                float f1 = fminf(fminf(fmaxf(l1x, h1x), fmaxf(l1y, h1y)), fminf(fmaxf(l1z, h1z), w_tbest));
                bool hit0 = n0 <= f0, hit1 = n1 <= f1;
                if (hit0 && hit1) {
                    bool first0 = n0 <= n1;
                    stack[sp * 64] = (uint32_t)(first0 ? ch.y : ch.x); ++sp;
                    cur = first0 ? ch.x : ch.y;
                } else if (hit0) cur = ch.x;
                else if (hit1) cur = ch.y;
                else if (sp == sb) done = true;
                else { --sp; cur = (int32_t)stack[sp * 64]; }
            }
        }
        const bool at_leaf = !done && cur < 0;
        const unsigned long long m1 = __ballot(at_leaf);
        if (m1 != 0ull) {
            uint32_t first = 0, rem = 0;
            if (at_leaf) { first = leaf_first(cur) + leaf_off; rem = leaf_count(cur) - leaf_off; }
            const uint32_t a = rem < 2u ? rem : 2u;
            const unsigned long long m2 = __ballot(a == 2u);
            const uint32_t pos = tail + rank_below(m1) + rank_below(m2);
            if (a >= 1u) L.ring[pos & (ANY_RING - 1u)] = first | (owner << 26);
            if (a == 2u) L.ring[(pos + 1u) & (ANY_RING - 1u)] = (first + 1u) | (owner << 26);
            tail += (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
            if (at_leaf) {
                if (rem > 2u) leaf_off += 2u;
                else { leaf_off = 0u; if (sp == sb) done = true; else { --sp; cur = (int32_t)stack[sp * 64]; } }
            }
            __syncthreads();
            while (tail - head >= 64u) flush(64u);
            if (tail - head >= PT_CLOSEST_FLUSH_MIN) flush(tail - head);      // early feedback of t_best
        }
        const unsigned long long m_act = __ballot(!done);
        if (m_act == 0ull) break;
#if PT_ANY_STEAL
        if (++since_steal >= PT_STEAL_EVERY_CLOSEST && __popcll(m_act) <= PT_STEAL_MAX_ACTIVE) {
            const unsigned long long m_donor = __ballot(!done && sp > sb);
            if (m_donor != 0ull) {
                since_steal = 0u;
                const unsigned long long m_idle = ~m_act;
                const uint32_t n_pairs = min((uint32_t)__popcll(m_donor), (uint32_t)__popcll(m_idle));
                const bool donor = !done && sp > sb;
                const uint32_t rank = rank_below(donor ? m_donor : m_idle);
                if (donor && rank < n_pairs) L.pair[rank] = lane;
                __syncthreads();
                const bool taker = done && rank < n_pairs;
                const uint32_t from = taker ? L.pair[rank] : lane;
                const int d_sb = __shfl(sb, from);
                const float rx = __shfl(w_ro.x, from), ry = __shfl(w_ro.y, from), rz = __shfl(w_ro.z, from);
                const float ix = __shfl(w_inv.x, from), iy = __shfl(w_inv.y, from), iz = __shfl(w_inv.z, from);
                const float tb = __shfl(w_tbest, from);
                const uint32_t ow = __shfl(owner, from);
                if (taker) {
                    cur = (int32_t)(stack - lane)[d_sb * 64 + from];
                    w_ro = mk3(rx, ry, rz); w_inv = mk3(ix, iy, iz); w_tbest = tb; owner = ow;
                    sp = sb = 0; leaf_off = 0u; done = false;
                }
                if (donor && rank < n_pairs) ++sb;
                __syncthreads();
            }
        }
#endif
    }
    if (tail != head) flush(tail - head);
    const unsigned long long key = L.best[lane];
    const uint32_t tri = (uint32_t)key;
    const bool found = want && tri != 0xffffffffu;
    if (found) {                                                             // the winner's barycentrics
        winner_hit(sc, ro, rd, rs, tri, hit);
        if (STATS) st.closest_hits++;
    }
    return found;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ONE cooperative traversal for both rays a path vertex spawns: the light connection's any-hit ray and the next closest-hit ray.
// They are independent (the connection only decides whether a contribution already computed is added), so the wave walks them as one
// pool of up to 128 rays.  A wave's step count is bounded below by the depth of the deepest path it follows, not by the number of rays
// (stealing spreads breadth, not depth): traced one after the other the two cost ~18 + ~14 node steps per iteration, together ~20.
// Mechanics are those of trace_closest_coop / trace_any_deferred — lanes only walk nodes, queue (triangle, ray) pairs in the LDS ring, the
// wave tests the ring densely, idle lanes steal — plus: a ray id is (owner lane | kind << 6); a lane that has both rays walks its closest-hit
// ray first and keeps the shadow ray PENDING, to start it when it runs dry — or to hand it whole to an idle lane in a steal round, which is
// the largest subtree there is.  Results: closest hits merge through atomicMin on (t bits << 32 | triangle), occlusion is one bit per owner.
// the dense triangle tests of the merged traversal (64 lanes of arithmetic) run above the node steps' priority: +0.2...0.3 %, consistently
#ifndef PT_PRIO_FLUSH
#define PT_PRIO_FLUSH 1
#endif
#if PT_PRIO_FLUSH
#define PT_PRIO_FLUSH_ENTER __builtin_amdgcn_s_setprio(2)
#define PT_PRIO_FLUSH_EXIT __builtin_amdgcn_s_setprio(0)
#else
#define PT_PRIO_FLUSH_ENTER ((void)0)
#define PT_PRIO_FLUSH_EXIT ((void)0)
#endif
struct PairLds { uint32_t* ring; unsigned long long* best; uint32_t* occl; uint32_t* pair; uint32_t* infl; };   // ring[ANY_RING], best[64], occl[2], pair[64], infl[2]
constexpr uint32_t RAY_ANY = 64u;             // kind bit of a ray id

// STRAGGLER CARRY-OVER (CARRY > 0).  A lock-step wave pays for its deepest ray: a fifth of the merged traversal's node steps run with at most 8
// of 64 lanes still walking a chain of dependent fetch -> test round trips that no amount of stealing shortens.  With CARRY = n
// the traversal RETURNS when at most n lanes are still walking and all of them are on closest-hit rays with nothing pending: each of those
// lanes saves its walking context (node link, stack window, working ray, owner: CTX_DWORDS dwords, compacted by rank into the LDS ring, which
// is empty between calls) — the stack entries stay where they are, in the lane's own LDS column — and the OWNERS of the unfinished rays are
// reported in `c_inflight`: such a lane skips its shading stage (its path state is untouched, its best-hit key stays in LDS) and passes
// `c_resume` to the next call, in which the saved contexts walk on beside the next iteration's fresh rays.  The wave's other lanes shade,
// regenerate and trace meanwhile instead of idling through the tail.  Shadow rays are never carried: a connection's contribution would have to
// wait in registers through the shading stage (measured: -2...-6 %), so the exit waits until no lane walks or holds a shadow ray.
// Results do not change: every ray is walked over exactly the same subtrees with the same merges, only in a different call.
struct CarryState { unsigned long long walk; };      // wave-uniform: the lanes that hold a saved walking context (slot = rank within the mask)
constexpr uint32_t CTX_DWORDS = 10u;
#ifndef PT_CARRY_MIN_FRESH
#define PT_CARRY_MIN_FRESH 32    // the early return is allowed only in calls that started at least this many fresh closest-hit rays (a wave
                                 // that is draining its work item must not leave the traversal after every step)
#endif

// PT_SHADOW_FIRST: a lane with both rays walks its SHADOW ray first and keeps the closest-hit ray pending (the other way round by default).
// Shadow rays cannot be carried over, closest-hit rays can: with the shadow rays started first the traversal's tail consists of closest-hit
// rays, which is what the carry-over needs to find there.
#ifndef PT_SHADOW_FIRST
#define PT_SHADOW_FIRST 0
#endif
template <bool STATS, bool WIDE, int CARRY = 0>
PT_DEV void trace_pair_coop(const DevScene& sc, f3 c_ro, f3 c_rd, bool c_want, f3 s_ro, f3 s_rd, float s_tmax, bool s_want, uint32_t* stack,
                            uint32_t lane, const PairLds& L, Hit& hit, bool& c_found, bool& s_occluded, StatCounters& st,
                            CarryState* cs = nullptr, bool c_resume = false, bool* c_inflight = nullptr) {
    if (!c_want) { c_rd = mk3(0.0f, 0.0f, 1.0f); c_ro = mk3(0.0f, 0.0f, 0.0f); }
    if (!s_want) { s_rd = mk3(0.0f, 0.0f, 1.0f); s_ro = mk3(0.0f, 0.0f, 0.0f); s_tmax = 0.0f; }
    const RaySetup crs = setup_ray(c_rd), srs = setup_ray(s_rd);
    const uint32_t c_kpack = (uint32_t)crs.kx | ((uint32_t)crs.ky << 2) | ((uint32_t)crs.kz << 4);
    const uint32_t s_kpack = (uint32_t)srs.kx | ((uint32_t)srs.ky << 2) | ((uint32_t)srs.kz << 4);
    const bool c_fresh = c_want && !(CARRY > 0 && c_resume);             // a resumed owner's ray is already under way: its key in best[] stays
    if (c_fresh || !c_want) L.best[lane] = (0x7f7fffffull << 32) | 0xffffffffull;                 // (FLT_MAX, no triangle)
    if (lane < 2) { L.occl[lane] = 0u; if (CARRY > 0) L.infl[lane] = 0u; }
    const int32_t root = WIDE ? sc.root4 : sc.root;
    // the ray this lane is WALKING (box tests see distances <= 1e30, see trace_any_deferred); `pend`: its own shadow ray is still to be started;
    // `pend_c` (CARRY): so is its own closest-hit ray, because the lane first walks on with the context it saved in the previous call
    constexpr bool SFIRST = PT_SHADOW_FIRST != 0;
    constexpr bool PC = CARRY > 0 || SFIRST;                              // pend_c can be set at all
    const bool first_c = SFIRST ? (c_fresh && !s_want) : c_fresh;       // the ray the lane starts with is its closest-hit ray
    f3 w_ro = walk_origin<WIDE>(sc, first_c ? c_ro : s_ro, first_c ? crs.inv : srs.inv), w_inv = walk_inv<WIDE>(sc, first_c ? crs.inv : srs.inv);
    float w_t = first_c ? 1e30f : fminf(s_tmax, 1e30f);
    uint32_t ow = first_c ? lane : (lane | RAY_ANY);
    bool pend = !SFIRST && c_fresh && s_want, pend_c = SFIRST && c_fresh && s_want;
    bool done = !c_fresh && !s_want;
    int sp = 0, sb = 0;
    int32_t cur = root;
    uint32_t leaf_off = 0;
    uint32_t head = 0, tail = 0;
    bool allow_carry = false, carry_exit = false;
    if constexpr (CARRY > 0) {
        allow_carry = __popcll(__ballot(c_fresh)) >= PT_CARRY_MIN_FRESH;
        const unsigned long long m_ctx = cs->walk;
        if (m_ctx != 0ull) {
            if ((m_ctx >> lane) & 1ull) {
                const uint32_t* c = L.ring + rank_below(m_ctx) * CTX_DWORDS;
                const uint32_t pk = c[1];
                cur = (int32_t)c[0]; sp = (int)(pk & 255u); sb = (int)((pk >> 8) & 255u); leaf_off = pk >> 16;
                w_ro = mk3(__uint_as_float(c[2]), __uint_as_float(c[3]), __uint_as_float(c[4]));
                w_inv = mk3(__uint_as_float(c[5]), __uint_as_float(c[6]), __uint_as_float(c[7]));
                w_t = __uint_as_float(c[8]); ow = c[9];
                pend_c = c_fresh; pend = s_want; done = false;
            }
            cs->walk = 0ull;
        }
    }
    __syncthreads();
    if (STATS) { if (c_fresh) st.closest_rays++; if (s_want) st.shadow_rays++; }

    auto flush = [&](uint32_t n) {            // test ring entries [head, head + n), n <= 64
        PT_PRIO_FLUSH_ENTER;
        const bool valid = lane < n;
        const uint32_t e = valid ? L.ring[(head + lane) & (ANY_RING - 1u)] : (lane << 25);
        const uint32_t id = e >> 25, own = id & 63u, tri = e & 0x01ffffffu;
        const bool any = (id & RAY_ANY) != 0u;
        // the owner's ray of the entry's kind: both kinds are shuffled (a lane cannot know which of its two rays the asker wants)
        const f3 co = mk3(__shfl(c_ro.x, own), __shfl(c_ro.y, own), __shfl(c_ro.z, own)), so = mk3(__shfl(s_ro.x, own), __shfl(s_ro.y, own), __shfl(s_ro.z, own));
        const f3 cd = mk3(__shfl(c_rd.x, own), __shfl(c_rd.y, own), __shfl(c_rd.z, own)), sd = mk3(__shfl(s_rd.x, own), __shfl(s_rd.y, own), __shfl(s_rd.z, own));
        const uint32_t ck = __shfl(c_kpack, own), sk = __shfl(s_kpack, own);
        const float csx = __shfl(crs.sx, own), csy = __shfl(crs.sy, own), csz = __shfl(crs.sz, own);
        const float ssx = __shfl(srs.sx, own), ssy = __shfl(srs.sy, own), ssz = __shfl(srs.sz, own);
        const float stm = __shfl(s_tmax, own);
        if (valid) {
            const f3 o2 = tri_origin(sc, any ? so : co), d2 = any ? sd : cd;
            const uint32_t kp = any ? sk : ck;
            const float sx = any ? ssx : csx, sy = any ? ssy : csy, sz = any ? ssz : csz;
            const float t_lim = any ? stm : __uint_as_float((uint32_t)(L.best[own] >> 32));
            TriVerts tv = load_tri(sc.tris, tri);
            float t, b0, b1, b2;
            if (STATS) { if (any) { st.tris_shadow++; } else { st.tris_closest++; } if (wave_leader()) st.w[1]++; }
            if (intersect_triangle<false>(o2, d2, (int)(kp & 3u), (int)((kp >> 2) & 3u), (int)(kp >> 4), sx, sy, sz, t_lim, tv, t, b0, b1, b2)) {
                if (any) atomicOr(&L.occl[own >> 5], 1u << (own & 31u));
                else {
                    if (STATS) {    // the merge keeps the LOWER triangle index on an exact tie, the reference the second child / earlier leaf item (bvh.rs:381-388): how often does it matter?
                        const unsigned long long cur = L.best[own];
                        if ((uint32_t)(cur >> 32) == __float_as_uint(t) && (uint32_t)cur != tri) {
                            st.ties++;
                            const float4* sa = (const float4*)(sc.shade + tri); const float4* sb = (const float4*)(sc.shade + (uint32_t)cur);
                            const float4 ga = sa[6], gb = sb[6];
                            if (sa[4].z != sb[4].z || ga.x != gb.x || ga.y != gb.y || ga.z != gb.z) st.ties_differ++;
                        }
                    }
                    atomicMin(&L.best[own], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)tri);
                }
            }
        }
        PT_PRIO_FLUSH_EXIT;
        head += n;
        __syncthreads();
        // feedback for the ray this lane walks: a shadow ray that is occluded is finished, a closest-hit ray prunes with the best distance so far
        const uint32_t wl_ = ow & 63u;
        if (ow & RAY_ANY) { if ((L.occl[wl_ >> 5] >> (wl_ & 31u)) & 1u) done = true; }
        else w_t = fminf(w_t, __uint_as_float((uint32_t)(L.best[wl_] >> 32)));
    };

    for (;;) {
        if (!done && cur >= 0) {
            if (STATS) { if (ow & RAY_ANY) st.nodes_shadow++; else st.nodes_closest++; const int busy = __popcll(__ballot(true)); if (wave_leader()) { st.w[0]++; st.hist[(busy - 1) >> 3]++; } }   // (the pool's steps are booked as closest-hit steps)
            if constexpr (WIDE) {
                Node4Hits h = wide_step(sc, cur, w_ro, w_inv, w_t);
                sort4(h);                 // nearest first (any-hit does not need the order, and does not mind it)
#if PT_PUSH_BRANCHFREE
                stack[sp * 64] = (uint32_t)h.link[3]; sp += h.n[3] < INFINITY ? 1 : 0;
                stack[sp * 64] = (uint32_t)h.link[2]; sp += h.n[2] < INFINITY ? 1 : 0;
                stack[sp * 64] = (uint32_t)h.link[1]; sp += h.n[1] < INFINITY ? 1 : 0;
#else
                if (h.n[3] < INFINITY) { stack[sp * 64] = (uint32_t)h.link[3]; ++sp; }
                if (h.n[2] < INFINITY) { stack[sp * 64] = (uint32_t)h.link[2]; ++sp; }
                if (h.n[1] < INFINITY) { stack[sp * 64] = (uint32_t)h.link[1]; ++sp; }
#endif
                if (h.n[0] < INFINITY) cur = h.link[0];
                else if (sp == sb) done = true;
                else { --sp; cur = (int32_t)stack[sp * 64]; }
            } else {
                const float4* q = (const float4*)(sc.nodes + cur);
                float4 nx = q[0], ny = q[1], nz = q[2];
                int2 ch = *(const int2*)(q + 3);
                float l0x = (nx.x - w_ro.x) * w_inv.x, h0x = (nx.z - w_ro.x) * w_inv.x;
                float l1x = (nx.y - w_ro.x) * w_inv.x, h1x = (nx.w - w_ro.x) * w_inv.x;
                float l0y = (ny.x - w_ro.y) * w_inv.y, h0y = (ny.z - w_ro.y) * w_inv.y;
                float l1y = (ny.y - w_ro.y) * w_inv.y, h1y = (ny.w - w_ro.y) * w_inv.y;
                float l0z = (nz.x - w_ro.z) * w_inv.z, h0z = (nz.z - w_ro.z) * w_inv.z;
                float l1z = (nz.y - w_ro.z) * w_inv.z, h1z = (nz.w - w_ro.z) * w_inv.z;
                float n0 = fmaxf(fmaxf(fminf(l0x, h0x), fminf(l0y, h0y)), fmaxf(fminf(l0z, h0z), 0.0f));
                float f0 = fminf(fminf(fmaxf(l0x, h0x), fmaxf(l0y, h0y)), fminf(fmaxf(l0z, h0z), w_t));
                float n1 = fmaxf(fmaxf(fminf(l1x, h1x), fminf(l1y, h1y)), fmaxf(fminf(l1z, h1z), 0.0f));
                float f1 = fminf(fminf(fmaxf(l1x, h1x), fmaxf(l1y, h1y)), fminf(fmaxf(l1z, h1z), w_t));
                bool hit0 = n0 <= f0, hit1 = n1 <= f1;
                if (hit0 && hit1) {
                    bool first0 = n0 <= n1;
                    stack[sp * 64] = (uint32_t)(first0 ? ch.y : ch.x); ++sp;
                    cur = first0 ? ch.x : ch.y;
                } else if (hit0) cur = ch.x;
                else if (hit1) cur = ch.y;
                else if (sp == sb) done = true;
                else { --sp; cur = (int32_t)stack[sp * 64]; }
            }
        }
        // lanes sitting in a leaf queue up to two of its triangles per step, then move on
        const bool at_leaf = !done && cur < 0;
        const unsigned long long m1 = __ballot(at_leaf);
        if (m1 != 0ull) {
            uint32_t first = 0, rem = 0;
            if (at_leaf) { first = leaf_first(cur) + leaf_off; rem = leaf_count(cur) - leaf_off; }
            const uint32_t a = rem < 2u ? rem : 2u;
            const unsigned long long m2 = __ballot(a == 2u);
            const uint32_t pos = tail + rank_below(m1) + rank_below(m2);
            if (a >= 1u) L.ring[pos & (ANY_RING - 1u)] = first | (ow << 25);
            if (a == 2u) L.ring[(pos + 1u) & (ANY_RING - 1u)] = (first + 1u) | (ow << 25);
            tail += (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
            if (at_leaf) {
                if (rem > 2u) leaf_off += 2u;
                else { leaf_off = 0u; if (sp == sb) done = true; else { --sp; cur = (int32_t)stack[sp * 64]; } }
            }
            __syncthreads();
            while (tail - head >= 64u) flush(64u);
        }
        // a lane that ran dry starts its own pending ray (CARRY: the closest-hit ray it has not started yet, then the shadow ray)
        if (PC && done && pend_c && !(SFIRST && pend)) {
            pend_c = false; done = false;
            w_ro = walk_origin<WIDE>(sc, c_ro, crs.inv); w_inv = walk_inv<WIDE>(sc, crs.inv); w_t = 1e30f; ow = lane;
            cur = root; sp = sb = 0; leaf_off = 0u;
        } else
        if (done && pend) {
            pend = false; done = false;
            w_ro = walk_origin<WIDE>(sc, s_ro, srs.inv); w_inv = walk_inv<WIDE>(sc, srs.inv); w_t = fminf(s_tmax, 1e30f); ow = lane | RAY_ANY;
            cur = root; sp = sb = 0; leaf_off = 0u;
        }
        const unsigned long long m_act = __ballot(!done);
        if (m_act == 0ull) break;
        if constexpr (CARRY > 0) {
            // the tail: few lanes left, every one of them on a closest-hit ray, nothing pending -> leave, the stragglers walk on next call
            if (allow_carry && __popcll(m_act) <= CARRY && !__any((!done && (ow & RAY_ANY) != 0u) || pend || pend_c)) { carry_exit = true; break; }
        }
        // work stealing: idle lanes take, from lanes that still have something to give, either the PENDING shadow ray as a whole or the
        // bottom stack entry (the largest pending subtree) together with the working ray it belongs to
        if (__popcll(m_act) <= PT_STEAL_MAX_ACTIVE) {
            const bool donor = !done && (pend || (PC && pend_c) || sp > sb);
            const unsigned long long m_donor = __ballot(donor);
            if (m_donor != 0ull) {
                if (STATS && lane == 0) st.w[7]++;       // steal rounds (mi355pt_stats.wave_steps[7])
                const unsigned long long m_idle = ~m_act;
                const uint32_t n_pairs = min((uint32_t)__popcll(m_donor), (uint32_t)__popcll(m_idle));
                const uint32_t rank = rank_below(donor ? m_donor : m_idle);
                if (donor && rank < n_pairs) L.pair[rank] = lane;
                __syncthreads();
                const bool taker = done && rank < n_pairs;
                const uint32_t from = taker ? L.pair[rank] : lane;
                // what this lane would give: its pending shadow ray from the root, else the bottom of its stack with its working ray
                const bool give_c = PC && pend_c && !(SFIRST && pend);   // the own closest-hit ray not yet started (walking a carried context, or the shadow ray first)
                const bool give_ray = pend || give_c;
                const f3 p_inv = give_c ? crs.inv : srs.inv;                      // the pending ray a lane would give away, in walking form
                const f3 g_inv = give_ray ? walk_inv<WIDE>(sc, p_inv) : w_inv;
                const f3 g_ro = give_ray ? walk_origin<WIDE>(sc, give_c ? c_ro : s_ro, p_inv) : w_ro;
                const float g_t = give_c ? 1e30f : (give_ray ? fminf(s_tmax, 1e30f) : w_t);
                const uint32_t g_ow = give_c ? lane : (give_ray ? (lane | RAY_ANY) : ow);
                const int g_sb = give_ray ? -1 : sb;
                const int d_sb = __shfl(g_sb, from);
                const float rx = __shfl(g_ro.x, from), ry = __shfl(g_ro.y, from), rz = __shfl(g_ro.z, from);
                const float ix = __shfl(g_inv.x, from), iy = __shfl(g_inv.y, from), iz = __shfl(g_inv.z, from);
                const float tb = __shfl(g_t, from);
                const uint32_t gow = __shfl(g_ow, from);
                if (taker) {
                    cur = d_sb < 0 ? root : (int32_t)(stack - lane)[d_sb * 64 + from];
                    w_ro = mk3(rx, ry, rz); w_inv = mk3(ix, iy, iz); w_t = tb; ow = gow;
                    sp = sb = 0; leaf_off = 0u; done = false;
                }
                if (donor && rank < n_pairs) { if (give_c) pend_c = false; else if (give_ray) pend = false; else ++sb; }
                __syncthreads();
            }
        }
    }
    if (tail != head) flush(tail - head);     // stragglers' last pairs (tail - head < 64 here)
    bool inflight = false;
    if constexpr (CARRY > 0) {
        if (carry_exit) {
            // save the walkers' contexts (the ring is empty now) and mark the owners of the rays they are on
            const unsigned long long m_walk = __ballot(!done);
            cs->walk = m_walk;
            if (!done) {
                uint32_t* c = L.ring + rank_below(m_walk) * CTX_DWORDS;
                c[0] = (uint32_t)cur; c[1] = (uint32_t)sp | ((uint32_t)sb << 8) | (leaf_off << 16);
                c[2] = __float_as_uint(w_ro.x); c[3] = __float_as_uint(w_ro.y); c[4] = __float_as_uint(w_ro.z);
                c[5] = __float_as_uint(w_inv.x); c[6] = __float_as_uint(w_inv.y); c[7] = __float_as_uint(w_inv.z);
                c[8] = __float_as_uint(w_t); c[9] = ow;
                atomicOr(&L.infl[(ow & 63u) >> 5], 1u << (ow & 31u));
            }
            __syncthreads();
            inflight = c_want && (((L.infl[lane >> 5] >> (lane & 31u)) & 1u) != 0u);
        }
        *c_inflight = inflight;
    }
    s_occluded = s_want && (((L.occl[lane >> 5] >> (lane & 31u)) & 1u) != 0u);
    const unsigned long long key = L.best[lane];
    const uint32_t tri = (uint32_t)key;
    c_found = c_want && !inflight && tri != 0xffffffffu;
    if (c_found) {                                                           // the winner's barycentrics
        winner_hit(sc, c_ro, c_rd, crs, tri, hit);
        if (STATS) st.closest_hits++;
    }
}

}  // namespace pt
