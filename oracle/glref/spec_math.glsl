// oracle/glref/spec_math.glsl -- TEST INFRASTRUCTURE.  GLSL restatement of the transcendental functions of the arithmetic
// specification (oracle/oracle_math.h: log, asin/acos, atan/atan2).  GLSL leaves the precision of these built-ins to the
// implementation (llvmpipe: log 86 ulp, acos 2e-4, atan 2e-5 relative; sin, cos, pow, exp, sqrt, division agree with the
// specification), so the reference's kernels can be run two ways: with the driver's built-ins, and with these spliced in
// through #define -- which isolates "the reference's algorithm" from "this driver's log()".  fma() is not fused on
// llvmpipe, so Horner steps written with fma() here may differ from the oracle's by one rounding.
float spec_log(float x) {
    if (!(x > 0.f)) return x == 0.f ? -uintBitsToFloat(0x7F800000u) : uintBitsToFloat(0x7FC00000u);
    if (x == uintBitsToFloat(0x7F800000u)) return x;
    uint u = floatBitsToUint(x);
    int e = 0;
    if ((u & 0x7F800000u) == 0u) { x *= 8388608.0f; u = floatBitsToUint(x); e = -23; }
    e += int((u >> 23) & 0xFFu) - 126;
    float m = uintBitsToFloat((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else { m = m - 1.0f; }
    const float z = m * m;
    float y = 7.0376836292E-2f;
    y = fma(y, m, -1.1514610310E-1f);
    y = fma(y, m, 1.1676998740E-1f);
    y = fma(y, m, -1.2420140846E-1f);
    y = fma(y, m, 1.4249322787E-1f);
    y = fma(y, m, -1.6668057665E-1f);
    y = fma(y, m, 2.0000714765E-1f);
    y = fma(y, m, -2.4999993993E-1f);
    y = fma(y, m, 3.3333331174E-1f);
    y = y * m * z;
    const float fe = float(e);
    y = fma(-2.12194440e-4f, fe, y);
    y = fma(-0.5f, z, y);
    float r = m + y;
    r = fma(0.693359375f, fe, r);
    return r;
}
float spec_asin(float x) {
    float a = abs(x);
    const bool neg = x < 0.f;
    if (a > 1.f) a = 1.f;
    if (a < 1.0e-4f) return x;
    float z, r; bool flag = false;
    if (a > 0.5f) { z = 0.5f * (1.f - a); r = sqrt(z); flag = true; }
    else { r = a; z = r * r; }
    float p = 4.2163199048E-2f;
    p = fma(p, z, 2.4181311049E-2f);
    p = fma(p, z, 4.5470025998E-2f);
    p = fma(p, z, 7.4953002686E-2f);
    p = fma(p, z, 1.6666752422E-1f);
    float res = fma(p * z, r, r);
    if (flag) { res = res + res; res = 1.57079632679489661923f - res; }
    return neg ? -res : res;
}
float spec_acos(float x) {
    if (isnan(x)) return x;
    if (x < -1.f) x = -1.f;
    if (x > 1.f) x = 1.f;
    if (x < -0.5f) return 3.14159265358979323846f - 2.f * spec_asin(sqrt(0.5f * (1.f + x)));
    if (x > 0.5f) return 2.f * spec_asin(sqrt(0.5f * (1.f - x)));
    return 1.57079632679489661923f - spec_asin(x);
}
float spec_atan1(float x) {
    const bool neg = x < 0.f;
    float a = abs(x), y;
    if (a > 2.414213562373095f) { y = 1.57079632679489661923f; a = -(1.f / a); }
    else if (a > 0.4142135623730950f) { y = 0.78539816339744830962f; a = (a - 1.f) / (a + 1.f); }
    else { y = 0.f; }
    const float z = a * a;
    float p = 8.05374449538e-2f;
    p = fma(p, z, -1.38776856032E-1f);
    p = fma(p, z, 1.99777106478E-1f);
    p = fma(p, z, -3.33329491539E-1f);
    y += fma(p * z, a, a);
    return neg ? -y : y;
}
float spec_atan2(float y, float x) {
    if (isnan(x) || isnan(y)) return uintBitsToFloat(0x7FC00000u);
    if (x == 0.f) {
        if (y > 0.f) return 1.57079632679489661923f;
        if (y < 0.f) return -1.57079632679489661923f;
        return 0.f;
    }
    float z = spec_atan1(y / x);
    if (x < 0.f) z += (y >= 0.f) ? 3.14159265358979323846f : -3.14159265358979323846f;
    return z;
}
#define log(x) spec_log(x)
#define acos(x) spec_acos(x)
#define atan(y, x) spec_atan2(y, x)
