// ORACLE — TEST INFRASTRUCTURE ONLY.
// Restatement of renderer/src/sampler/{z_sobol_sampler,random_sampler}.rs.
// The Sobol generator matrices are the first two dimensions (2 x 52 words) of the
// PBRT-v4 table the reference embeds (sobol_matrices.rs:7); they are regenerated
// analytically here and pinned against the reference's words in
// tests/golden/sobol_matrices_dim01.json.
#pragma once
#include "o_math.hpp"

namespace oracle {

constexpr int SOBOL_MATRIX_SIZE = 52;

struct SobolMatrices {
    uint32_t m[2 * SOBOL_MATRIX_SIZE];
    SobolMatrices() {
        for (int j = 0; j < SOBOL_MATRIX_SIZE; ++j) {
            m[j] = j < 32 ? (1u << (31 - j)) : 0u;            // dimension 0: van der Corput
            uint32_t v = 0;                                   // dimension 1: Pascal triangle mod 2
            for (int i = 0; i < 32; ++i) if ((j & i) == i) v |= 1u << (31 - i);
            m[SOBOL_MATRIX_SIZE + j] = v;
        }
    }
};
static inline const SobolMatrices& sobol_matrices() { static SobolMatrices s; return s; }

// FastOwenScrambler (z_sobol_sampler.rs:3-28)
static inline uint32_t reverse_bits_32(uint32_t n) {
    n = (n >> 16) | (n << 16);
    n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
    n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
    n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
    n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
    return n;
}
static inline uint32_t fast_owen(uint32_t v, uint32_t seed) {
    v = reverse_bits_32(v);
    v ^= v * 0x3d20adeau;
    v += seed;
    v *= (seed >> 16) | 1u;
    v ^= v * 0x05526c56u;
    v ^= v * 0x53a22864u;
    return reverse_bits_32(v);
}
static inline uint64_t mix_bits(uint64_t v) {            // :68-75
    v ^= v >> 31; v *= 0x7fb5d329728ea185ull;
    v ^= v >> 27; v *= 0x81dadef4bc2dd44dull;
    v ^= v >> 33;
    return v;
}
static inline uint64_t murmur_hash_dim_seed(uint32_t dimension, uint32_t seed) {   // :77-99
    const uint64_t M = 0xc6a4a7935bd1e995ull; const int R = 47;
    uint64_t h = 8ull * M;
    uint64_t k = (uint64_t)dimension | ((uint64_t)seed << 32);
    k *= M; k ^= k >> R; k *= M;
    h ^= k; h *= M;
    h ^= h >> R; h *= M; h ^= h >> R;
    return h;
}
static inline uint32_t encode_morton2(uint32_t x, uint32_t y) {                    // :53-66 (u32 result)
    auto ls2 = [](uint64_t v) {
        v &= 0xffffffffull;
        v = (v ^ (v << 16)) & 0x0000ffff0000ffffull;
        v = (v ^ (v << 8)) & 0x00ff00ff00ff00ffull;
        v = (v ^ (v << 4)) & 0x0f0f0f0f0f0f0f0full;
        v = (v ^ (v << 2)) & 0x3333333333333333ull;
        v = (v ^ (v << 1)) & 0x5555555555555555ull;
        return v;
    };
    return ((uint32_t)ls2(y) << 1) | (uint32_t)ls2(x);
}
static inline uint32_t log2_int(uint32_t v) { return v == 0 ? 0 : 31 - (uint32_t)__builtin_clz(v); }
static inline uint32_t round_up_pow2(uint32_t v) { return v <= 1 ? 1 : 1u << (32 - __builtin_clz(v - 1)); }

static const uint8_t PERMUTATIONS[24][4] = {
    {0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 1, 3}, {0, 2, 3, 1}, {0, 3, 2, 1}, {0, 3, 1, 2}, {1, 0, 2, 3}, {1, 0, 3, 2},
    {1, 2, 0, 3}, {1, 2, 3, 0}, {1, 3, 2, 0}, {1, 3, 0, 2}, {2, 1, 0, 3}, {2, 1, 3, 0}, {2, 0, 1, 3}, {2, 0, 3, 1},
    {2, 3, 0, 1}, {2, 3, 1, 0}, {3, 1, 2, 0}, {3, 1, 0, 2}, {3, 2, 1, 0}, {3, 2, 0, 1}, {3, 0, 2, 1}, {3, 0, 1, 2}};

struct Sampler {
    // mode 0 = random (counter hash; the reference uses ThreadRng, statistical parity only), 1 = ZSobol
    int mode = 1;
    uint32_t dimension = 0, seed = 0, log2_spp = 0, n_base4_digits = 0, morton_index = 0;
    uint64_t rkey = 0;   // random-mode stream key
    uint64_t draws = 0;  // instrumentation

    static Sampler create(int mode, uint32_t spp, uint32_t w, uint32_t h, uint32_t seed) {   // :179-196
        Sampler s; s.mode = mode; s.seed = seed;
        s.log2_spp = log2_int(spp);
        uint32_t res = round_up_pow2(std::max(w, h));
        uint32_t log4_spp = (s.log2_spp + 1) / 2;
        s.n_base4_digits = log2_int(res) + log4_spp;
        return s;
    }
    void start_pixel_sample(uint32_t px, uint32_t py, uint32_t sample_index, uint32_t width) {   // :198-201
        dimension = 0;
        morton_index = (encode_morton2(px, py) << log2_spp) | sample_index;
        rkey = mix_bits(((uint64_t)(py * width + px) << 32) ^ (uint64_t)sample_index ^ ((uint64_t)seed << 20) ^ 0x9e3779b97f4a7c15ull);
    }
    uint64_t get_sample_index() const {                                                          // :101-156
        uint64_t sample_index = 0;
        bool pow2_samples = (log2_spp & 1) == 1;
        int last_digit = pow2_samples ? 1 : 0;
        int i = (int)n_base4_digits - 1;
        while (i >= last_digit) {
            int digit_shift = 2 * i - (pow2_samples ? 1 : 0);
            uint64_t digit = ((uint64_t)morton_index >> digit_shift) & 3;
            // u64 >> 64 cannot occur: digit_shift + 2 <= 2*18 = 36
            uint64_t higher_digits = (uint64_t)morton_index >> (digit_shift + 2);
            uint64_t p = (mix_bits(higher_digits ^ (0x55555555ull * (uint64_t)dimension)) >> 24) % 24;
            digit = PERMUTATIONS[p][digit];
            sample_index |= digit << digit_shift;
            i -= 1;
        }
        if (pow2_samples) {
            // reference quirk: `& i` with i == 0 after the loop (PBRT: & 1)   :147-153
            uint64_t digit = (uint64_t)morton_index & (uint64_t)(int64_t)i;
            sample_index |= digit ^ ((mix_bits(((uint64_t)morton_index >> 1) ^ (0x55555555ull * (uint64_t)dimension))) & 1);
        }
        return sample_index;
    }
    static uint32_t sobol_bits(uint64_t a, int dim, uint32_t scramble_seed) {                   // :158-177
        const uint32_t* mat = sobol_matrices().m + dim * SOBOL_MATRIX_SIZE;
        uint32_t v = 0; int i = 0;
        while (a != 0) { if (a & 1) v ^= mat[i]; a >>= 1; ++i; }
        return fast_owen(v, scramble_seed);
    }
    static float bits_to_float(uint32_t v) {
        float f = (float)v * 2.3283064365386963e-10f;   // 0x1p-32
        const float one_minus_eps = 0.99999994f;        // 0x3f7fffff
        return f < one_minus_eps ? f : one_minus_eps;
    }
    float random_next() {
        uint64_t h = mix_bits(rkey + 0x632be59bd9b4e019ull * (uint64_t)(++dimension));
        return (float)(uint32_t)(h >> 40) * 5.9604644775390625e-8f;   // 24 bits * 2^-24
    }
    // raw u32 access for the bit-exactness tests
    uint32_t get_1d_bits() {
        uint64_t si = get_sample_index();
        dimension += 1;
        uint64_t h = murmur_hash_dim_seed(dimension, seed);
        return sobol_bits(si, 0, (uint32_t)h);
    }
    void get_2d_bits(uint32_t out[2]) {
        uint64_t si = get_sample_index();
        dimension += 2;
        uint64_t h = murmur_hash_dim_seed(dimension, seed);
        out[0] = sobol_bits(si, 0, (uint32_t)h);
        out[1] = sobol_bits(si, 1, (uint32_t)(h >> 32));
    }
    float get_1d() {                                                                              // :203-213
        ++draws;
        if (mode == 0) return random_next();
        return bits_to_float(get_1d_bits());
    }
    V2 get_2d() {                                                                                 // :215-230
        draws += 2;
        if (mode == 0) { float a = random_next(); float b = random_next(); return V2{a, b}; }
        uint32_t b[2]; get_2d_bits(b);
        return V2{bits_to_float(b[0]), bits_to_float(b[1])};
    }
};

}  // namespace oracle
