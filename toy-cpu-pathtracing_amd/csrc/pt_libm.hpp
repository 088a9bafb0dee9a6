// The host libm's sinf / cosf, restated operation for operation so that device and CPU compute the SAME float.
//
// The reference calls f32::sin / f32::cos (Rust std -> the platform libm: glibc on the Linux hosts it is built for) in the GGX
// visible-normal sample (scene/src/material/bsdf/generalized_schlick.rs:165-199 via common.rs), the cosine-weighted hemisphere
// sample, the conductor Fresnel and the environment light.  In GGX sampling a 1-ulp difference of sin or cos near the rim of the disk
// (pz = sqrt(1 - px^2 - py^2) cancels) moves the sampled normal visibly, so a device libm that is merely "accurate to 1-2 ulp" traces
// other paths than the reference on rough dielectrics and metals (scenes 7, 11, 12, 27).
//
// glibc >= 2.28 computes both functions in double precision (sysdeps/ieee754/flt-32/s_sincosf.h, after ARM's optimized routines):
// |x| < pi/4: a degree-7 odd / degree-8 even polynomial in double; |x| < 120: n = round(x * 2/pi) by a scaled float-to-int conversion,
// x - n * pi/2 in double, then the polynomial the quadrant selects.  The x86-64 build that CPUs with FMA dispatch to contracts every
// a * b + c of that source into one fma.  The functions below do exactly that (double arithmetic is IEEE on gfx950, the kernels are
// compiled with -ffp-contract=off and the fmas are spelled out), and tools/libm_check.cpp compares them on the host, compiled from this
// same header, with the libm of the box for EVERY float in [-120, 120]: 0 mismatches of 2 246 049 792 x 2 on glibc 2.35.
// Arguments outside that range (never produced by the call sites: angles of at most 2 pi) take the device libm.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>

#ifdef __HIPCC__
#define PT_LIBM_FN __device__ __forceinline__
#else
#define PT_LIBM_FN static inline
#endif

namespace ptlibm {

PT_LIBM_FN uint32_t abstop12(float x) {
    uint32_t u;
#ifdef __HIPCC__
    u = __float_as_uint(x);
#else
    std::memcpy(&u, &x, 4);
#endif
    return (u >> 20) & 0x7ffu;
}

// sin and cos of y, both at once (glibc's sincosf returns the same two floats as sinf and cosf: checked by tools/libm_check.cpp too).
// Returns false when |y| >= 120 or y is not finite: the caller then uses its platform function.
PT_LIBM_FN bool sincosf_glibc(float y, float* sn, float* cs) {
    constexpr double HPI_INV = 0x1.45F306DC9C883p+23;   // 2/pi * 2^24
    constexpr double HPI = 0x1.921FB54442D18p0;
    constexpr double C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    constexpr double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const uint32_t top = abstop12(y);
    if (top >= 0x42fu) return false;                    // abstop12(120.0f)
    double x = (double)y;
    int32_t n = 0;
    if (top >= 0x3f4u) {                                // abstop12(pi/4)
        const double r = x * HPI_INV;
        n = ((int32_t)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, HPI, x);
    } else if (top < 0x398u) {                          // abstop12(2^-12): sinf returns y, cosf returns 1
        *sn = y; *cs = 1.0f;
        return true;
    }
    const double x2 = x * x;
    // sine polynomial of the reduced argument (odd: the quadrant's sign is applied to the result, which commutes with every rounding)
    const double x3 = x * x2, s1 = __builtin_fma(x2, S3, S2), x7 = x3 * x2, s = __builtin_fma(x3, S1, x);
    const float sp = (float)__builtin_fma(x7, s1, s);
    // cosine polynomial (the second table of the source holds the negated coefficients: same result negated)
    const double x4 = x2 * x2, c2 = __builtin_fma(x2, C4, C3), c1 = __builtin_fma(x2, C1, 1.0), x6 = x4 * x2, c = __builtin_fma(x4, C2, c1);
    const float cp = (float)__builtin_fma(x6, c2, c);
    // quadrant n & 3:  0: (sp, cp)   1: (cp, -sp)   2: (-sp, -cp)   3: (-cp, sp)
    const float a = (n & 1) ? cp : sp, b = (n & 1) ? sp : cp;
    *sn = (n & 2) ? -a : a;
    *cs = (((n + 1) & 2) != 0) ? -b : b;
    return true;
}


// expf, the same way: glibc >= 2.27's algorithm (sysdeps/ieee754/flt-32/e_expf.c after ARM's optimized routines) - x * 32 / ln 2 = k + r by the
// add-and-subtract-2^52 * 1.5 rounding, 2^(k / 32) from a 32-entry table of doubles, a cubic in r, one multiply, all in double.  Compared on the
// host (tools/libm_check.cpp) with the libm of the box for every float of (-88, 88): 2 of 2 237 661 184 differ (by one ulp).  Returns false
// outside that range (the caller's platform function: overflow, underflow and subnormal results are not restated).
struct Exp2fTable { uint64_t t[32]; };
#ifdef __HIPCC__
__device__
#endif
static const Exp2fTable EXP2F_TAB = {{
0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull
}};
PT_LIBM_FN bool expf_glibc(float x, float* out) {
    if (abstop12(x) >= 0x42bu) return false;                  // abstop12(88.0f); also inf / NaN
    constexpr double INVLN2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
    constexpr double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    double z = INVLN2N * (double)x;
    double kd = z + SHIFT;
    uint64_t ki;
#ifdef __HIPCC__
    ki = (uint64_t)__double_as_longlong(kd);
#else
    std::memcpy(&ki, &kd, 8);
#endif
    kd -= SHIFT;
    const double r = z - kd;
    const uint64_t t = EXP2F_TAB.t[ki % 32u] + (ki << 47);
    double s;
#ifdef __HIPCC__
    s = __longlong_as_double((long long)t);
#else
    std::memcpy(&s, &t, 8);
#endif
    z = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1.0;
    y = z * r2 + y;
    *out = (float)(y * s);
    return true;
}

}  // namespace ptlibm
