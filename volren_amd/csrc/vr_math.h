// vr_math.h -- deterministic fp32 math for the HIP path tracer (device + host).
//
// The reference shaders (shader/common.glsl) use GLSL log/sin/cos/acos/atan/tan, whose precision is
// driver-defined.  To make a (pixel, sample) reproducible bit for bit on any gfx950 device -- and checkable
// against the CPU oracle at equal seed -- every elementary function is spelled out here as a fixed
// sequence of IEEE binary32 add/mul/div/sqrt/fma (Cephes single-precision algorithms).  The hardware
// transcendental units (v_log_f32, v_sin_f32, ...) are deliberately not used: they are fast but not
// reproducible on the host.
//
// Build requirements: -ffp-contract=off, no fast-math, correctly rounded fp32 divide/sqrt (hipcc default).
// Spec (also DESIGN.md "Math"): Horner steps are fma; everything else is one rounding per operator.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VR_HD __host__ __device__ __forceinline__
#else
#define VR_HD inline
#endif

// VR_FAST_MATH (device code of the opt-in tolerance-mode kernels only; vr_pathtrace.hip): the elementary functions map to
// the gfx950 transcendental unit (v_log_f32, v_sin_f32, v_cos_f32, v_rcp_f32) and the compiler may contract and use
// reciprocal-based division.  Results are no longer reproducible bit for bit on a host.  Measured against the reference's kernels
// at 64x48x1024 spp (tests/test_gpu_parity.py): within the north star's 1e-3 relative L2 without a transfer function, 1.9e-3
// with one -- which is why RendererHIP::launch refuses the mode while a transfer function is bound.
#if defined(VR_FAST_MATH) && defined(__HIP_DEVICE_COMPILE__)
#define VR_FAST_DEVICE 1
#else
#define VR_FAST_DEVICE 0
#endif

namespace vr {

constexpr float kPi = 3.14159265358979323846f;   // common.glsl:4
constexpr float kPiO2 = 1.5707963267948966192f;
constexpr float kPiO4 = 0.7853981633974483096f;
constexpr float kInv4Pi = 1.0f / (4.0f * kPi);   // common.glsl:7
constexpr int32_t kIntMin = INT32_MIN;

// a * b for a, b < 2^24 whose product fits 32 bits (full-rate v_mul_u32_u24 on the device)
VR_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}
VR_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
VR_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

VR_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
VR_HD float floor_(float x) { return __builtin_floorf(x); }
VR_HD float sqrt_(float x) { return __builtin_sqrtf(x); }
VR_HD float abs_(float x) { return __builtin_fabsf(x); }
// 1.0f / x, correctly rounded.  On the device: v_rcp_f32 (1 ulp) + one Newton step + v_div_fixup_f32 -- 4 instructions instead of the 10 of
// the IEEE division sequence -- which equals the correctly rounded quotient for EVERY x whose exponent field lies in [2, 252]
// (tests/tools_rcp_exact.hip compares all 2^32 inputs on the GPU: 0 mismatches in that range); the others -- zeros, denormals, the two
// smallest and the two largest binades, infinities, NaN -- take the division.  Up to three reciprocals share one range test.
VR_HD bool rcp_in_fast_range(float x) { return ((f2u(x) >> 23) & 255u) - 2u <= 250u; }
VR_HD float rcp_newton(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !VR_FAST_DEVICE
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
    return __builtin_amdgcn_div_fixupf(r1, x, 1.0f);
#else
    return 1.0f / x;
#endif
}
VR_HD float rcp_exact(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !VR_FAST_DEVICE
    float r = rcp_newton(x);
    if (!rcp_in_fast_range(x)) r = 1.0f / x;
    return r;
#else
    return 1.0f / x;
#endif
}
// The IEEE division sequence the compiler emits for n / d, WITHOUT its three guard instructions (v_div_scale_f32 x 2, v_div_fixup_f32; v_div_fmas_f32 becomes an fma):
// the same eight operations on the same values whenever no operand needs rescaling and no special case applies (round 5; 19 instead of 30 issue cycles per
// quotient, c2 +1.6 %, c4 +1.4 %: profiles/r5x_*).  Used only where the call site has established div_core_domain for its operands -- or operands for which both forms
// give NaN (0 / 0, a NaN operand) and only the NaN-ness matters downstream.  VR_DIV_CORE=0: n / d everywhere.
#ifndef VR_DIV_CORE
#define VR_DIV_CORE 1
#endif
// The operands div_core is used on (each call site establishes this before it runs; tests/tools_div_core.hip checks the identity over the whole set on the hardware):
// the denominator normal with 2^-100 <= |d| <= 2^100, the numerator +0 or 2^-100 <= |n| <= 2^100, the quotient's magnitude in [2^-100, 2^100] -- far inside the
// conditions under which v_div_scale_f32 leaves both operands alone (denominator and its reciprocal normal, numerator's exponent field above 23, quotient neither
// denormal nor within 2^-32 of overflow), where v_div_fmas_f32 is an fma and v_div_fixup_f32 returns its first operand.
VR_HD bool div_core_domain(float n, float d) {
    const int32_t en = (int32_t)((f2u(n) >> 23) & 255u), ed = (int32_t)((f2u(d) >> 23) & 255u);
    const bool d_ok = ed >= 27 && ed <= 227;
    const bool n_ok = f2u(n) == 0u || (en >= 27 && en <= 227 && en - ed >= -100 && en - ed <= 100);
    return d_ok && n_ok;
}
VR_HD float div_core(float n, float d) {
#if defined(__HIP_DEVICE_COMPILE__) && !VR_FAST_DEVICE && VR_DIV_CORE
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    const float q0 = n * r1;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), r1, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, n), r1, q1);
#else
    return n / d;
#endif
}
VR_HD float inf_() { return u2f(0x7F800000u); }
VR_HD float nan_() { return u2f(0x7FC00000u); }

// GLSL min/max/clamp/mix semantics (NaN behaviour follows the comparison, like the spec text)
VR_HD float min_(float x, float y) { return y < x ? y : x; }
VR_HD float max_(float x, float y) { return x < y ? y : x; }
VR_HD float clamp_(float x, float lo, float hi) { return min_(max_(x, lo), hi); }
VR_HD float mix_(float x, float y, float a) { return x * (1.0f - a) + y * a; }
VR_HD float sqr(float x) { return x * x; }
VR_HD float saturate(float x) { return clamp_(x, 0.0f, 1.0f); }
VR_HD float sanitize(float x) { return (x != x || abs_(x) == inf_()) ? 0.0f : x; }

// floor()/trunc to int; NaN, inf and out-of-range give kIntMin ("outside every grid")
VR_HD int32_t floor2i(float x) {
    const float f = floor_(x);
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return kIntMin;
    return (int32_t)f;
}
// Voxel index floor(x) + o for the grid fetches, which read 0 for every index outside [0, extent): there only "inside or
// not" matters, so out-of-range values may land on ANY index that is negative or >= 2^30.  The device converts with
// the saturating v_cvt_i32_f32 and lets the addition wrap; a NaN coordinate (converts to 0) is handled by the caller.
VR_HD int32_t voxel_index(float floored, int32_t o) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int32_t)((uint32_t)(int32_t)floored + (uint32_t)o);       // |o| <= 2: INT_MAX + o wraps negative, INT_MIN + o lands >= 2^31 - 2
#else
    const int32_t b = (!(floored >= -2147483648.0f && floored < 2147483648.0f)) ? kIntMin : (int32_t)floored;
    return b == kIntMin ? kIntMin : b + o;
#endif
}

// GLSL round() with halves to even (only used for the DDA mip level in [0,3])
VR_HD int32_t round_half_even(float x) {
    float r = floor_(x + 0.5f);
    if (r - x == 0.5f && (((int32_t)r) & 1)) r -= 1.0f;
    return (int32_t)r;
}

// round_half_even for the DDA mip, which is always a multiple of 1/4 in [0, 3]: 2-bit table lookup on q = 4*mip
// (q: 0 1 2 3 4 5 6 7 8 9 10 11 12 -> 0 0 0 1 1 1 2 2 2 2 2 3 3; 0.5 and 2.5 round to the even neighbour)
VR_HD int32_t round_mip(float mip) {
    const uint32_t q = (uint32_t)(int32_t)(mip * 4.0f);
    return (int32_t)((0x3EAA540u >> (2u * q)) & 3u);
}
// the same with the mip carried as the integer q = 4*mip (what the path state stores)
VR_HD int32_t round_mip_q(int32_t q) { return (int32_t)((0x3EAA540u >> (2u * (uint32_t)q)) & 3u); }

VR_HD float scale2(float z, int n) {
    if (n > 254) n = 254;
    if (n < -252) n = -252;
    if (n > 127) { z *= u2f(0x7F000000u); n -= 127; }
    if (n < -126) { z *= u2f(0x00800000u); n += 126; }
    return z * u2f((uint32_t)(n + 127) << 23);
}

VR_HD float log_(float x) {
    if (!(x > 0.0f)) return x == 0.0f ? -inf_() : nan_();
    if (x == inf_()) return x;
    uint32_t u = f2u(x);
    int e = 0;
    if ((u & 0x7F800000u) == 0u) { x *= 8388608.0f; u = f2u(x); e = -23; }
    e += (int)((u >> 23) & 0xFFu) - 126;
    float m = u2f((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else { m = m - 1.0f; }
    const float z = m * m;
    float y = 7.0376836292E-2f;
    y = fma_(y, m, -1.1514610310E-1f);
    y = fma_(y, m, 1.1676998740E-1f);
    y = fma_(y, m, -1.2420140846E-1f);
    y = fma_(y, m, 1.4249322787E-1f);
    y = fma_(y, m, -1.6668057665E-1f);
    y = fma_(y, m, 2.0000714765E-1f);
    y = fma_(y, m, -2.4999993993E-1f);
    y = fma_(y, m, 3.3333331174E-1f);
    y = y * m * z;
    const float fe = (float)e;
    y = fma_(-2.12194440e-4f, fe, y);
    y = fma_(-0.5f, z, y);
    float r = m + y;
    r = fma_(0.693359375f, fe, r);
    return r;
}

// log_(x) restricted to normal, finite x in (0, 1]: same arithmetic, none of the special-case tests
VR_HD float log_unit_(float x) {
    const uint32_t u = f2u(x);
    int e = (int)((u >> 23) & 0xFFu) - 126;
    float m = u2f((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else { m = m - 1.0f; }
    const float z = m * m;
    float y = 7.0376836292E-2f;
    y = fma_(y, m, -1.1514610310E-1f);
    y = fma_(y, m, 1.1676998740E-1f);
    y = fma_(y, m, -1.2420140846E-1f);
    y = fma_(y, m, 1.4249322787E-1f);
    y = fma_(y, m, -1.6668057665E-1f);
    y = fma_(y, m, 2.0000714765E-1f);
    y = fma_(y, m, -2.4999993993E-1f);
    y = fma_(y, m, 3.3333331174E-1f);
    y = y * m * z;
    const float fe = (float)e;
    y = fma_(-2.12194440e-4f, fe, y);
    y = fma_(-0.5f, z, y);
    float r = m + y;
    r = fma_(0.693359375f, fe, r);
    return r;
}
// -log(1 - xi) for xi = k * 2^-24 in [0,1): the free-flight optical depth draw (common.glsl:421,451,468,497).
// 1 - xi is exact and lies in [2^-24, 1], so the restricted log applies.
VR_HD float neg_log_1m(float xi) {
#if VR_FAST_DEVICE
    return __builtin_amdgcn_logf(1.0f - xi) * -0.69314718055994530942f;       // v_log_f32 = log2
#else
    return -log_unit_(1.0f - xi);
#endif
}

// r < w / s decided without the full IEEE division whenever the answer is not within rounding distance:
// q = w * rcp(s) is within 2.5 ulp of w/s, so outside a +-8 ulp band the comparison with q equals the comparison with
// the correctly rounded quotient; inside the band (probability ~1e-6) the exact division decides.  Host: always exact.
VR_HD bool lt_quot(float r, float w, float s) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float q = w * __builtin_amdgcn_rcpf(s);
    const float m = abs_(q) * 9.5367431640625e-07f;      // 2^-20
    if (r < q - m) return true;
    if (r > q + m) return false;
#endif
    return r < w / s;
}

// float(b) / 255.f for b in 0..255 (GL unorm8), correctly rounded in 3 operations instead of a division
// (y = RN(1/255), q = b*y, q + fma(-q,255,b)*y; verified exhaustively against b / 255.f on host and device)
VR_HD float unorm8(uint32_t b) {
#if VR_FAST_DEVICE
    return (float)b * (1.0f / 255.0f);
#endif
    const float a = (float)b;
    const float y = 1.0f / 255.0f;
    const float q = a * y;
    return fma_(fma_(-q, 255.0f, a), y, q);
}

struct SinCosArg { float r; int j; };
VR_HD SinCosArg sincos_reduce(float ax) {
    int j = (int)(1.27323954473516f * ax);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    float r = fma_(-y, 0.78515625f, ax);
    r = fma_(-y, 2.4187564849853515625e-4f, r);
    r = fma_(-y, 3.77489497744594108e-8f, r);
    return SinCosArg{ r, j & 7 };
}
VR_HD float sin_poly(float r) {
    const float z = r * r;
    float y = -1.9515295891E-4f;
    y = fma_(y, z, 8.3321608736E-3f);
    y = fma_(y, z, -1.6666654611E-1f);
    return fma_(y * z, r, r);
}
VR_HD float cos_poly(float r) {
    const float z = r * r;
    float y = 2.443315711809948E-005f;
    y = fma_(y, z, -1.388731625493765E-003f);
    y = fma_(y, z, 4.166664568298827E-002f);
    y = y * z * z;
    y = fma_(-0.5f, z, y);
    return y + 1.0f;
}
VR_HD float sin_(float x) {
    if (!(abs_(x) < 8192.0f)) return nan_();
    bool neg = x < 0.0f;
    SinCosArg a = sincos_reduce(abs_(x));
    int j = a.j;
    if (j > 3) { neg = !neg; j -= 4; }
    const float y = (j == 1 || j == 2) ? cos_poly(a.r) : sin_poly(a.r);
    return neg ? -y : y;
}
VR_HD float cos_(float x) {
    if (!(abs_(x) < 8192.0f)) return nan_();
    bool neg = false;
    SinCosArg a = sincos_reduce(abs_(x));
    int j = a.j;
    if (j > 3) { j -= 4; neg = !neg; }
    if (j > 1) neg = !neg;
    const float y = (j == 1 || j == 2) ? sin_poly(a.r) : cos_poly(a.r);
    return neg ? -y : y;
}
// sin and cos of one angle sharing the range reduction (same results as sin_/cos_)
VR_HD void sincos_(float x, float& s, float& c) {
#if VR_FAST_DEVICE
    const float rev = x * 0.15915494309189533577f;       // v_sin_f32 / v_cos_f32 take revolutions
    s = __builtin_amdgcn_sinf(rev); c = __builtin_amdgcn_cosf(rev);
    return;
#endif
    if (!(abs_(x) < 8192.0f)) { s = nan_(); c = nan_(); return; }
    SinCosArg a = sincos_reduce(abs_(x));
    const float sp = sin_poly(a.r), cp = cos_poly(a.r);
    int j = a.j;
    bool sneg = x < 0.0f, cneg = false;
    if (j > 3) { j -= 4; sneg = !sneg; cneg = !cneg; }
    if (j > 1) cneg = !cneg;
    const bool swap = (j == 1 || j == 2);
    const float sv = swap ? cp : sp, cv = swap ? sp : cp;
    s = sneg ? -sv : sv;
    c = cneg ? -cv : cv;
}
VR_HD float tan_(float x) { return sin_(x) / cos_(x); }

VR_HD float asin_(float x) {
    float a = abs_(x);
    const bool neg = x < 0.0f;
    if (a > 1.0f) a = 1.0f;
    if (a < 1.0e-4f) return x;
    float z, r; bool flag = false;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); r = sqrt_(z); flag = true; }
    else { r = a; z = r * r; }
    float p = 4.2163199048E-2f;
    p = fma_(p, z, 2.4181311049E-2f);
    p = fma_(p, z, 4.5470025998E-2f);
    p = fma_(p, z, 7.4953002686E-2f);
    p = fma_(p, z, 1.6666752422E-1f);
    float res = fma_(p * z, r, r);
    if (flag) { res = res + res; res = kPiO2 - res; }
    return neg ? -res : res;
}
VR_HD float acos_(float x) {
    if (x != x) return nan_();
    if (x < -1.0f) x = -1.0f;
    if (x > 1.0f) x = 1.0f;
    // three cases, ONE asin_ evaluation: |x| > 0.5 uses asin(sqrt(0.5 * (1 - |x|))) (1 + x == 1 - |x| for x < 0, bit for bit)
    const bool lo = x < -0.5f, hi = x > 0.5f;
    const float a = asin_((lo | hi) ? sqrt_(0.5f * (1.0f - abs_(x))) : x);
    return lo ? kPi - 2.0f * a : (hi ? 2.0f * a : kPiO2 - a);
}
VR_HD float atan_(float x) {
    const bool neg = x < 0.0f;
    float a = abs_(x), y;
    // -(1/a), (a-1)/(a+1) or a: operands selected first, one division (a / 1 and (-1) / a are exact restatements)
    const bool big = a > 2.414213562373095f, mid = !big && a > 0.4142135623730950f;
    y = big ? kPiO2 : (mid ? kPiO4 : 0.0f);
    a = (big ? -1.0f : (mid ? a - 1.0f : a)) / (big ? a : (mid ? a + 1.0f : 1.0f));
    const float z = a * a;
    float p = 8.05374449538e-2f;
    p = fma_(p, z, -1.38776856032E-1f);
    p = fma_(p, z, 1.99777106478E-1f);
    p = fma_(p, z, -3.33329491539E-1f);
    y += fma_(p * z, a, a);
    return neg ? -y : y;
}
VR_HD float atan2_(float y, float x) {
    if (x != x || y != y) return nan_();
    if (x == 0.0f) {
        if (y > 0.0f) return kPiO2;
        if (y < 0.0f) return -kPiO2;
        return 0.0f;
    }
    float z = atan_(y / x);
    if (x < 0.0f) z += (y >= 0.0f) ? kPi : -kPi;
    return z;
}
VR_HD float exp_(float x) {
    if (x != x) return nan_();
    if (x > 88.72283905206835f) return inf_();
    if (x < -103.278929903431851103f) return 0.0f;
    const float n = floor_(fma_(1.44269504088896341f, x, 0.5f));
    float r = fma_(-n, 0.693359375f, x);
    r = fma_(-n, -2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500E-4f;
    p = fma_(p, r, 1.3981999507E-3f);
    p = fma_(p, r, 8.3334519073E-3f);
    p = fma_(p, r, 4.1665795894E-2f);
    p = fma_(p, r, 1.6666665459E-1f);
    p = fma_(p, r, 5.0000001201E-1f);
    const float res = fma_(p, z, r) + 1.0f;
    return scale2(res, (int)n);
}
VR_HD float pow_(float x, float y) {
    if (x != x || y != y) return nan_();
    if (!(x > 0.0f)) return 0.0f;
    return exp_(y * log_(x));
}

VR_HD float half2float(uint32_t h) {   // low 16 bits
#if defined(__HIP_DEVICE_COMPILE__)
    // v_cvt_f32_f16: exact for every binary16 value (gfx9 keeps f16 denormals; every half is representable in binary32)
    return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
#endif
    const uint32_t s = (h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0u) {
        if (m == 0u) return u2f(s);
        const float f = (float)m * u2f(0x33800000u);
        return s ? -f : f;
    }
    if (e == 31u) return u2f(s | 0x7F800000u | (m << 13));
    return u2f(s | ((e + 112u) << 23) | (m << 13));
}

// float -> binary16 conversions used by the brick encoders (range texture is RG16F; ranges are rounded outwards)
VR_HD uint32_t float_to_half_rne(float f) {
    const uint32_t x = f2u(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u : 0u);
    if (ax >= 0x477FF000u) return sign | 0x7C00u;              // overflow -> inf
    if (ax < 0x33000001u) return sign;                          // underflow -> 0
    const int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007FFFFFu) | 0x00800000u;
    int shift;
    uint32_t h;
    if (e < -14) { shift = 13 + (-14 - e); h = 0u; }            // subnormal half
    else { shift = 13; h = (uint32_t)(e + 15) << 10; m &= 0x007FFFFFu; }
    const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    uint32_t r = h + q;
    if (rem > half || (rem == half && (r & 1u))) r += 1u;
    return sign | r;
}
VR_HD uint32_t half_next_up(uint32_t h) {       // next representable towards +inf
    if ((h & 0x7FFFu) == 0u) return 0x0001u;
    return (h & 0x8000u) ? (h - 1u) & 0xFFFFu : (h + 1u) & 0xFFFFu;
}
VR_HD uint32_t half_next_down(uint32_t h) {
    if ((h & 0x7FFFu) == 0u) return 0x8001u;
    return (h & 0x8000u) ? (h + 1u) & 0xFFFFu : (h - 1u) & 0xFFFFu;
}
VR_HD uint32_t float_to_half_down(float f) {     // largest fp16 <= f
    uint32_t h = float_to_half_rne(f);
    if (half2float(h) > f) h = half_next_down(h);
    return h;
}
VR_HD uint32_t float_to_half_up(float f) {       // smallest fp16 >= f
    uint32_t h = float_to_half_rne(f);
    if (half2float(h) < f) h = half_next_up(h);
    return h;
}

// ---- small vectors ----
struct v3 { float x, y, z; };
VR_HD v3 V3(float x, float y, float z) { return v3{ x, y, z }; }
VR_HD v3 operator+(v3 a, v3 b) { return v3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
VR_HD v3 operator-(v3 a, v3 b) { return v3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
VR_HD v3 operator*(v3 a, v3 b) { return v3{ a.x * b.x, a.y * b.y, a.z * b.z }; }
VR_HD v3 operator*(v3 a, float s) { return v3{ a.x * s, a.y * s, a.z * s }; }
VR_HD v3 operator/(v3 a, float s) { return v3{ a.x / s, a.y / s, a.z / s }; }
VR_HD v3 operator-(v3 a) { return v3{ -a.x, -a.y, -a.z }; }
VR_HD float dot(v3 a, v3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
VR_HD v3 axpy(v3 a, float t, v3 b) { return v3{ fma_(t, b.x, a.x), fma_(t, b.y, a.y), fma_(t, b.z, a.z) }; }
VR_HD v3 cross(v3 a, v3 b) { return v3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
VR_HD v3 normalize(v3 v) { const float inv = 1.0f / sqrt_(dot(v, v)); return v * inv; }
VR_HD float luma(v3 c) { return dot(c, v3{ 0.212671f, 0.715160f, 0.072169f }); }
// column-major 3x3 / 4x4 (glm layout)
VR_HD v3 mat3_mul(const float* m, v3 v) {
    return v3{ fma_(m[6], v.z, fma_(m[3], v.y, m[0] * v.x)),
               fma_(m[7], v.z, fma_(m[4], v.y, m[1] * v.x)),
               fma_(m[8], v.z, fma_(m[5], v.y, m[2] * v.x)) };
}
VR_HD v3 mat4_point(const float* m, v3 v) {
    return v3{ fma_(m[8], v.z, fma_(m[4], v.y, fma_(m[0], v.x, m[12]))),
               fma_(m[9], v.z, fma_(m[5], v.y, fma_(m[1], v.x, m[13]))),
               fma_(m[10], v.z, fma_(m[6], v.y, fma_(m[2], v.x, m[14]))) };
}
VR_HD v3 mat4_dir(const float* m, v3 v) {
    return v3{ fma_(m[8], v.z, fma_(m[4], v.y, m[0] * v.x)),
               fma_(m[9], v.z, fma_(m[5], v.y, m[1] * v.x)),
               fma_(m[10], v.z, fma_(m[6], v.y, m[2] * v.x)) };
}

}  // namespace vr
