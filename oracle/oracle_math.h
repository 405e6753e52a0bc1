/*
 * oracle/oracle_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Deterministic fp32 elementary functions for the CPU oracle.
 *
 * Why this exists: the reference (GLSL, /root/reference/shader/common.glsl) leaves
 * the precision of log/sin/cos/acos/atan/tan/exp/pow/inversesqrt to the GL driver.
 * Same-seed parity between two implementations is only possible when every
 * floating-point operation is specified.  This header fixes them as sequences of
 * IEEE-754 binary32 +,-,*,/,sqrt and fused multiply-add (fmaf) -- all of which are
 * correctly rounded on x86-64 (SSE/FMA3) and on gfx950 -- following the published
 * Cephes single-precision algorithms (S. Moshier, netlib cephes/single: logf.c,
 * sinf.c, asinf.c, atanf.c, expf.c).  The HIP product has its OWN statement of the
 * same spec (volren_amd/csrc/vr_math.h); the two must agree bit for bit and
 * tests/test_math_parity.py checks that they do.
 *
 * Build flags that this header relies on: -ffp-contract=off (no implicit fusing),
 * no -ffast-math, -mfma so that fmaf() is one instruction.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

/* -DORC_UNFUSED builds the "unfused" variant of the oracle (oracle/_ref/liboracle_unfused.so): every fused multiply-add of
 * the specification becomes a multiply followed by an add, which is how Mesa llvmpipe evaluates the reference's GLSL
 * (fma() and a*b+c are both unfused there).  Only tests/golden/make_golden_glsl.py uses it, to show that the oracle's
 * LOGIC reproduces the reference's kernels exactly once the contraction convention is the same. */
#ifdef ORC_UNFUSED
#define fmaf(a, b, c) ((a) * (b) + (c))
#endif

#define OM_PI      3.14159265358979323846f   /* common.glsl:4 M_PI as float */
#define OM_PIO2    1.5707963267948966192f
#define OM_PIO4    0.7853981633974483096f

static inline uint32_t om_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    om_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* GLSL min/max (spec 8.3): min(x,y) = y < x ? y : x ; max(x,y) = x < y ? y : x */
static inline float om_min(float x, float y) { return y < x ? y : x; }
static inline float om_max(float x, float y) { return x < y ? y : x; }
static inline float om_clamp(float x, float lo, float hi) { return om_min(om_max(x, lo), hi); }
/* GLSL mix(x,y,a) = x*(1-a) + y*a (spec 8.3), two roundings for the products, no fma */
static inline float om_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }

/* floor to int with a defined result for NaN/inf/out-of-range (C leaves that UB,
 * GLSL leaves it undefined): such inputs map to INT32_MIN, which every caller
 * treats as "outside the grid". */
static inline int32_t om_floor2i(float x) {
    float f = floorf(x);
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}
/* truncating conversion (GLSL int(x)/ivec3(x)) with the same guard */
static inline int32_t om_trunc2i(float x) {
    if (!(x > -2147483648.0f && x < 2147483648.0f)) return INT32_MIN;
    return (int32_t)x;
}

/* GLSL round(): direction of exact halves is implementation-defined; this build
 * fixes round-half-to-even. Only used on mip in [0,3]. */
static inline int32_t om_round_half_even(float x) {
    float r = floorf(x + 0.5f);
    if (r - x == 0.5f && (((int32_t)r) & 1)) r -= 1.0f;
    return (int32_t)r;
}

/* z * 2^n by exact power-of-two multiplies */
static inline float om_scale2(float z, int n) {
    if (n > 254) n = 254;
    if (n < -252) n = -252;
    if (n > 127) { z *= om_u2f(0x7F000000u); n -= 127; }         /* 2^127 */
    if (n < -126) { z *= om_u2f(0x00800000u); n += 126; }        /* 2^-126 */
    return z * om_u2f((uint32_t)(n + 127) << 23);
}

/* natural log, Cephes logf.c structure; domain x > 0 (x == 0 -> -inf, x < 0 -> NaN) */
static inline float om_log(float x) {
    if (!(x > 0.0f)) {
        if (x == 0.0f) return -INFINITY;
        return NAN;                         /* negative or NaN */
    }
    if (x == INFINITY) return INFINITY;
    uint32_t u = om_f2u(x);
    int e = 0;
    if ((u & 0x7F800000u) == 0) {           /* subnormal: scale by 2^23 */
        x *= 8388608.0f; u = om_f2u(x); e = -23;
    }
    e += (int)((u >> 23) & 0xFF) - 126;     /* frexp: x = m * 2^e, m in [0.5,1) */
    float m = om_u2f((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else                           { m = m - 1.0f; }
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = fmaf(y, m, -1.1514610310E-1f);
    y = fmaf(y, m,  1.1676998740E-1f);
    y = fmaf(y, m, -1.2420140846E-1f);
    y = fmaf(y, m,  1.4249322787E-1f);
    y = fmaf(y, m, -1.6668057665E-1f);
    y = fmaf(y, m,  2.0000714765E-1f);
    y = fmaf(y, m, -2.4999993993E-1f);
    y = fmaf(y, m,  3.3333331174E-1f);
    y = y * m * z;
    float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

/* shared range reduction for sin/cos: returns octant j (0..7 after folding to even) and r */
static inline float om_sincos_reduce(float ax, int* jout) {
    /* domain |x| < 8192 (all arguments in this code base are within [-2pi, 2pi]) */
    int j = (int)(1.27323954473516f * ax);          /* 4/pi */
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    *jout = j & 7;
    float r = fmaf(-y, 0.78515625f, ax);
    r = fmaf(-y, 2.4187564849853515625e-4f, r);
    r = fmaf(-y, 3.77489497744594108e-8f, r);
    return r;
}
static inline float om_sin_poly(float r) {
    float z = r * r;
    float y = -1.9515295891E-4f;
    y = fmaf(y, z,  8.3321608736E-3f);
    y = fmaf(y, z, -1.6666654611E-1f);
    return fmaf(y * z, r, r);
}
static inline float om_cos_poly(float r) {
    float z = r * r;
    float y = 2.443315711809948E-005f;
    y = fmaf(y, z, -1.388731625493765E-003f);
    y = fmaf(y, z,  4.166664568298827E-002f);
    y = y * z * z;
    y = fmaf(-0.5f, z, y);
    return y + 1.0f;
}
static inline float om_sin(float x) {
    if (!(fabsf(x) < 8192.0f)) return NAN;
    int sign = x < 0.0f;
    int j; float r = om_sincos_reduce(fabsf(x), &j);
    if (j > 3) { sign = !sign; j -= 4; }
    float y = (j == 1 || j == 2) ? om_cos_poly(r) : om_sin_poly(r);
    return sign ? -y : y;
}
static inline float om_cos(float x) {
    if (!(fabsf(x) < 8192.0f)) return NAN;
    int sign = 0;
    int j; float r = om_sincos_reduce(fabsf(x), &j);
    if (j > 3) { j -= 4; sign = !sign; }
    if (j > 1) sign = !sign;
    float y = (j == 1 || j == 2) ? om_sin_poly(r) : om_cos_poly(r);
    return sign ? -y : y;
}
static inline float om_tan(float x) { return om_sin(x) / om_cos(x); }

/* asin on [-1,1] (input is clamped by the callers), Cephes asinf.c */
static inline float om_asin(float x) {
    float a = fabsf(x);
    int sign = x < 0.0f;
    if (a > 1.0f) a = 1.0f;
    if (a < 1.0e-4f) return x;
    float z, r; int flag = 0;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); r = sqrtf(z); flag = 1; }
    else          { r = a; z = r * r; }
    float p = 4.2163199048E-2f;
    p = fmaf(p, z, 2.4181311049E-2f);
    p = fmaf(p, z, 4.5470025998E-2f);
    p = fmaf(p, z, 7.4953002686E-2f);
    p = fmaf(p, z, 1.6666752422E-1f);
    float res = fmaf(p * z, r, r);
    if (flag) { res = res + res; res = OM_PIO2 - res; }
    return sign ? -res : res;
}
/* acos with the argument clamped to [-1,1] (GLSL: undefined outside) */
static inline float om_acos(float x) {
    if (x != x) return NAN;
    if (x < -1.0f) x = -1.0f;
    if (x > 1.0f) x = 1.0f;
    if (x < -0.5f) return OM_PI - 2.0f * om_asin(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f)  return 2.0f * om_asin(sqrtf(0.5f * (1.0f - x)));
    return OM_PIO2 - om_asin(x);
}

static inline float om_atan(float x) {
    int sign = x < 0.0f;
    float a = fabsf(x), y;
    if (a > 2.414213562373095f)       { y = OM_PIO2; a = -(1.0f / a); }
    else if (a > 0.4142135623730950f) { y = OM_PIO4; a = (a - 1.0f) / (a + 1.0f); }
    else                              { y = 0.0f; }
    float z = a * a;
    float p = 8.05374449538e-2f;
    p = fmaf(p, z, -1.38776856032E-1f);
    p = fmaf(p, z,  1.99777106478E-1f);
    p = fmaf(p, z, -3.33329491539E-1f);
    y += fmaf(p * z, a, a);
    return sign ? -y : y;
}
/* GLSL atan(y,x); (0,0) -> 0 (GLSL: undefined) */
static inline float om_atan2(float y, float x) {
    if (x != x || y != y) return NAN;
    if (x == 0.0f) {
        if (y > 0.0f) return OM_PIO2;
        if (y < 0.0f) return -OM_PIO2;
        return 0.0f;
    }
    float z = om_atan(y / x);
    if (x < 0.0f) z += (y >= 0.0f) ? OM_PI : -OM_PI;
    return z;
}

/* e^x, Cephes expf.c */
static inline float om_exp(float x) {
    if (x != x) return NAN;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.278929903431851103f) return 0.0f;
    float n = floorf(fmaf(1.44269504088896341f, x, 0.5f));
    float r = fmaf(-n, 0.693359375f, x);
    r = fmaf(-n, -2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float res = fmaf(p, z, r) + 1.0f;
    return om_scale2(res, (int)n);
}
/* GLSL pow(x,y) for x >= 0 (undefined for x < 0: returns 0 here) */
static inline float om_pow(float x, float y) {
    if (x != x || y != y) return NAN;
    if (!(x > 0.0f)) return 0.0f;
    return om_exp(y * om_log(x));
}

/* IEEE binary16 -> binary32, exact */
static inline float om_half2float(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return om_u2f(s);
        float f = (float)m * om_u2f(0x33800000u);   /* m * 2^-24, exact */
        return s ? -f : f;
    }
    if (e == 31) return om_u2f(s | 0x7F800000u | (m << 13));
    return om_u2f(s | ((e + 112u) << 23) | (m << 13));
}

#endif /* ORACLE_MATH_H */
