// fp32 vector arithmetic of the tracer, usable from host and device code.
//
// The shader this library replaces is written in GLSL, which leaves contraction and the
// precision of normalize/sqrt/division/sin/cos to the implementation.  This header fixes one
// evaluation so that every build of the tracer (and the CPU checker used by the tests)
// produces the same bits:
//   * +,-,*,/ and sqrt are IEEE-754 binary32 round-to-nearest; translation units that include
//     this header must be compiled with -ffp-contract=off so nothing else is fused;
//   * madd(a,b,c) is ONE fused multiply-add and is used exactly where the formulas below say;
//   * dot(a,b)   = madd(a.z,b.z, madd(a.y,b.y, a.x*b.x))
//   * cross(a,b) = (madd(a.y,b.z,-(b.y*a.z)), madd(a.z,b.x,-(b.z*a.x)), madd(a.x,b.y,-(b.x*a.y)))
//   * normalize(v) = v * (1/sqrt(dot(v,v)))
//   * sincos_2pi_range(): Cody-Waite reduction by pi/2 + degree-7/6 polynomials, valid on
//     [0, 2*pi], within 2 ulp of a correctly rounded result.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtm {

struct vec3 { float x, y, z; };

RT_HD float madd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// 1/x, sqrt(x) and 1/sqrt(x) as IEEE-754 defines them (the last one rounds twice).  On the device the
// compiler expands each division or square root into 11-14 instructions that also cover
// subnormal operands and results.  One Newton step on the hardware estimate gives the same bits
// wherever those cases cannot arise - checked for every fp32 input (rt_selftest_exact_math,
// tools/micro/exact_math.hip): 1/x for 2^-126 <= |x| < 2^126, sqrt(x) for 2^-102 <= x < 2^128
// (and NaN for x < -2^-102, as IEEE).  When any lane of the wave holds an operand outside those
// ranges, the whole wave takes the full expansion; both agree on every other lane.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_IEEE_EXPANSIONS_ONLY)
__device__ __forceinline__ float recip_newton(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float sqrt_newton(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y;
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), 0.5f * y, s);
}
__device__ __forceinline__ bool recip_in_range(float x) {        // biased exponent 1..252
    return (__float_as_uint(x) & 0x7FFFFFFFu) - 0x00800000u < 0x7E800000u - 0x00800000u;
}
__device__ __forceinline__ bool sqrt_in_range(float x) {         // |x| in [2^-102, 2^128)
    return (__float_as_uint(x) & 0x7FFFFFFFu) - 0x0C800000u < 0x7F800000u - 0x0C800000u;
}
__device__ __forceinline__ float recip_exact(float x) {
    float r = recip_newton(x);
    if (__builtin_amdgcn_ballot_w64(!recip_in_range(x)) != 0ull) r = 1.0f / x;
    return r;
}
__device__ __forceinline__ float sqrt_exact(float x) {
    float s = sqrt_newton(x);
    if (__builtin_amdgcn_ballot_w64(!sqrt_in_range(x)) != 0ull) s = __builtin_sqrtf(x);
    return s;
}
__device__ __forceinline__ float rsqrt_exact(float x) {
    // x in the sqrt range puts sqrt(x) in [2^-51, 2^64): inside the reciprocal's range
    float r = recip_newton(sqrt_newton(x));
    if (__builtin_amdgcn_ballot_w64(!sqrt_in_range(x) || x < 0.0f) != 0ull) r = 1.0f / __builtin_sqrtf(x);
    return r;
}
#else
RT_HD float recip_exact(float x) { return 1.0f / x; }
RT_HD float sqrt_exact(float x) { return __builtin_sqrtf(x); }
RT_HD float rsqrt_exact(float x) { return 1.0f / __builtin_sqrtf(x); }
#endif

RT_HD vec3 mk(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD vec3 add(vec3 a, vec3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD vec3 sub(vec3 a, vec3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD vec3 neg(vec3 a) { return mk(-a.x, -a.y, -a.z); }
RT_HD vec3 mul(vec3 a, vec3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD vec3 scale(vec3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RT_HD float dot(vec3 a, vec3 b) { return madd(a.z, b.z, madd(a.y, b.y, a.x * b.x)); }
RT_HD vec3 cross(vec3 a, vec3 b) {
    return mk(madd(a.y, b.z, -(b.y * a.z)), madd(a.z, b.x, -(b.z * a.x)), madd(a.x, b.y, -(b.x * a.y)));
}
RT_HD vec3 normalize(vec3 a) { return scale(a, rsqrt_exact(dot(a, a))); }
// (X + 0.5) / W, the coordinate of pixel X of W (frag:680 gets it interpolated).  On the device: the
// refined reciprocal of W and one residual correction, the IEEE quotient for every X < W <= 16384
// (all pairs checked by rt_selftest_exact_math and tools/micro/exact_div_uniform.hip).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_IEEE_EXPANSIONS_ONLY)
__device__ __forceinline__ float pixel_coord(uint32_t X, float w) {
    const float r = recip_newton(w), a = (float)X + 0.5f, q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-w, q, a), r, q);
}
#else
RT_HD float pixel_coord(uint32_t X, float w) { return ((float)X + 0.5f) / w; }
#endif
// (1/a.x, 1/a.y, 1/a.z), IEEE; on the device one range check for the three (see recip_exact)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_IEEE_EXPANSIONS_ONLY)
__device__ __forceinline__ vec3 recip3_exact(vec3 a) {
    vec3 r = mk(recip_newton(a.x), recip_newton(a.y), recip_newton(a.z));
    const bool out = !recip_in_range(a.x) || !recip_in_range(a.y) || !recip_in_range(a.z);
    if (__builtin_amdgcn_ballot_w64(out) != 0ull) r = mk(1.0f / a.x, 1.0f / a.y, 1.0f / a.z);
    return r;
}
#else
RT_HD vec3 recip3_exact(vec3 a) { return mk(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }
#endif
// o + t*d, one fused multiply-add per component
RT_HD vec3 along(vec3 o, float t, vec3 d) { return mk(madd(t, d.x, o.x), madd(t, d.y, o.y), madd(t, d.z, o.z)); }

RT_HD void sincos_2pi_range(float x, float &s_out, float &c_out) {
    const float TWO_OVER_PI = 0.63661977236758134f;
    const float PIO2_HI = 1.57079625129699707031f;
    const float PIO2_LO = 7.54978941586159635335e-08f;
    const int k = (int)madd(x, TWO_OVER_PI, 0.5f);
    const float fk = (float)k;
    float r = madd(-fk, PIO2_HI, x);
    r = madd(-fk, PIO2_LO, r);
    const float r2 = r * r;
    float ps = madd(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = madd(r2, ps, -1.6666654611e-1f);
    const float s = madd(r * r2, ps, r);
    float pc = madd(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = madd(r2, pc, 4.166664568298827e-2f);
    const float c = madd(r2 * r2, pc, madd(r2, -0.5f, 1.0f));
    const int q = k & 3;
    s_out = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    c_out = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}

// One step of the shader's integer hash (frag:36-41 and frag:50-56 are the same mix).
RT_HD uint32_t wang_step(uint32_t seed) {
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    return seed ^ (seed >> 15);
}

}  // namespace rtm
