// Verifier for the fast decision of the stochastic-tricubic filter tests (vr_trace.h: tricubic_axis_fast / tricubic_fast_test).
// TEST TOOL -- compiles the product's lane code for the host, like tests/hostkernel.
//
// (Round 5: the product's test is x = RN(k s' - W') against -G / +G with W' = 2^24 w' and an absolute band G, VR_TAP_ABS_BAND in vr_trace.h; this tool
// searches the product's own tricubic_fast_test, so it checks whichever form the header compiles.  The comparisons named below are round 2's.)
// Claim checked: for every fractional coordinate t a float can take in [0, 1] and EVERY one of the 2^24 values a draw can take,
// a test that the fast path decides ("yes" / "no") is decided the same way by the reference's code
//     r < w / s,   r = k * 2^-24,   w, s from tricubic_axis_weights (common.glsl:221-244 restated operation by operation),
// and only draws inside the guard band are left to the exact code.  Both decision sets are intervals of k (k * s' is monotone
// in k), so per (t, test) it is enough to compare three numbers:
//     K_ref  = number of k with k * 2^-24 < w / s          (reference says yes)
//     K_yes  = number of k with RN(k * s') < RN(w' * lo - 1e-20)   (fast path says yes)      must be <= K_ref
//     K_no   = first k with RN(k * s') > RN(w' * hi + 1e-20)       (fast path says no)       must be >= K_ref
// usage: tools_tricubic_band [stride]      stride 1 = all 1 065 353 217 floats in [0, 1] (about a minute on 8 cores)
// build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -mfma -fopenmp tests/tools_tricubic_band.cpp -o build/tricubic_band
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../volren_amd/csrc/vr_trace.h"

using namespace vr;
#if VR_TAP_ABS_BAND
#define VR_BAND_TEXT "absolute, 160 units of k s, round 5's form; draws and weights scaled by kTapDrawScale"
#else
#define VR_BAND_TEXT "2^-18 relative, round 2's form"
#endif

static const double TWO24 = 16777216.0;

static inline int64_t count_ref(float w, float s) {          // number of k in [0, 2^24) with (float)k * 2^-24 < w / s
    const float T = w / s;
    if (!(T > 0.0f)) return 0;                              // NaN, zero and negative thresholds: never
    const double td = (double)T * TWO24;                     // exact
    if (td >= TWO24) return (int64_t)TWO24;
    const double c = std::ceil(td);
    return (int64_t)c;                                       // k < td  <=>  k < ceil(td) (k integer), also when td is an integer
}
// first k in [0, 2^24] with pred(k) true, for a predicate that is monotone in k (false ... false true ... true); guess from k0
template <class Pred> static inline int64_t first_k(double k0, Pred pred) {
    int64_t k = (int64_t)std::floor(k0) - 4;
    if (!(k0 == k0) || k < 0) k = 0;
    if (k > (int64_t)TWO24) k = (int64_t)TWO24;
    while (k > 0 && pred((float)k)) --k;                     // guess too high: walk down to a false
    while (k < (int64_t)TWO24 && !pred((float)k)) ++k;       // then up to the first true
    return k;
}

#if VR_TAP_ABS_BAND
static const float KS = kTapDrawScale;                       // a draw enters the product's test as KS x k (vr_trace.h VR_TAP_PRESHIFT): exact, k < 2^24
#else
static const float KS = 1.0f;
#endif
static inline int decide(float k, float w, float s) {         // the product's test: 1 yes, 0 no, 2 inside the band
    bool yes, no;
    tricubic_fast_test(k * KS, w, s, yes, no);
    if (yes && no) return -1;
    return yes ? 1 : (no ? 0 : 2);
}

int main(int argc, char** argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)std::strtoul(argv[1], nullptr, 10) : 1u;
    const uint32_t last = 0x3F800000u;                       // 1.0f
    long long violations = 0, decided_wrong_yes = 0, decided_wrong_no = 0, checked = 0;
    double max_disc = 0.0, max_band = 0.0;
    float worst_t = 0.0f;
#pragma omp parallel for schedule(dynamic, 65536) reduction(+ : violations, decided_wrong_yes, decided_wrong_no, checked) reduction(max : max_disc, max_band)
    for (int64_t bi = 0; bi <= (int64_t)last + 1; bi += stride) {
        // bit pattern -> t; the extra last index stands for t = 1.0f reached from below (q = -2^-26: floor -1, q + 1 rounds to 1)
        float q;
        if (bi <= (int64_t)last - 1) { const uint32_t b = (uint32_t)bi; std::memcpy(&q, &b, 4); }
        else if (bi == (int64_t)last) q = -0x1p-26f;
        else q = -0x1p-30f;
        const AxisWeights R = tricubic_axis_weights(q);
        const AxisFast F = tricubic_axis_fast(q);
        const float rw[3] = { R.w2, R.w3, R.w4 }, rs[3] = { R.s2, R.s3, R.s4 };
        const float fw[3] = { F.w2, F.w3, F.w4 }, fs[3] = { F.s2, F.s3, 6.0f };
        for (int j = 0; j < 3; ++j) {
            const int64_t K_ref = count_ref(rw[j], rs[j]);
            const float s = fs[j], w = fw[j];
            // both decision sets are intervals of k: the fast test's x(k) (RN(k s) against two thresholds in round 2's form, RN(k s - W) against -G / +G in
            // round 5's) is monotone in k.  The product's own test function is what is searched: first k that is NOT a yes, first k that IS a no
#if VR_TAP_ABS_BAND
            const double g_yes = ((double)w - (double)kTapBand) / (double)s / (double)KS, g_no = ((double)w + (double)kTapBand) / (double)s / (double)KS;
#else
            const double g_yes = (double)fma_(w, kTapLo, -1e-20f) / (double)s, g_no = (double)fma_(w, kTapHi, 1e-20f) / (double)s;
#endif
            const int64_t K_yes = first_k(g_yes, [w, s](float k) { bool y, n; tricubic_fast_test(k * KS, w, s, y, n); return !y; });
            const int64_t K_no = first_k(g_no, [w, s](float k) { bool y, n; tricubic_fast_test(k * KS, w, s, y, n); return n; });
            // cross-check the interval picture with the product's own test function at the boundaries
            if (K_yes > 0 && decide((float)(K_yes - 1), fw[j], s) != 1) ++violations;
            if (K_yes < (int64_t)TWO24 && decide((float)K_yes, fw[j], s) == 1) ++violations;
            if (K_no < (int64_t)TWO24 && decide((float)K_no, fw[j], s) != 0) ++violations;
            if (K_no > 0 && decide((float)(K_no - 1), fw[j], s) == 0) ++violations;
            if (K_yes > K_ref) { ++violations; ++decided_wrong_yes; if (std::getenv("TB_VERBOSE")) std::printf("yes>ref: q %.9g test %d K_yes %lld K_ref %lld K_no %lld  rw %.9g rs %.9g fw %.9g fs %.9g\n", q, j, (long long)K_yes, (long long)K_ref, (long long)K_no, rw[j], rs[j], fw[j], s); }
            if (K_no < K_ref) { ++violations; ++decided_wrong_no; }
            ++checked;
#if VR_TAP_ABS_BAND
            const double T = (double)rw[j] / (double)rs[j], Tf = (double)fw[j] / (TWO24 * (double)KS) / (double)s;
#else
            const double T = (double)rw[j] / (double)rs[j], Tf = (double)fw[j] / (double)s;
#endif
            if (T > 1e-30) { const double d = std::fabs(Tf / T - 1.0); if (d > max_disc) max_disc = d; }
            const double band = (double)(K_no - K_yes) / TWO24;
            if (band > max_band) max_band = band;
        }
    }
    // non-finite coordinates must end in the band (-> exact code)
    const float bad[3] = { INFINITY, -INFINITY, NAN };
    for (float q : bad) {
        const AxisFast F = tricubic_axis_fast(q);
        if (decide(12345.0f, F.w2, F.s2) != 2 || decide(12345.0f, F.w3, F.s3) != 2 || decide(12345.0f, F.w4, 6.0f) != 2) ++violations;
    }
    // the whole call, fast form against the reference's code (vr_trace.h tricubic_tap_t<true> / <false>): taps and RNG end state must agree for finite coordinates
    // (random points and seeds) and for coordinates that are not finite on one, two or all three axes -- a test whose x is NaN drops out of the call's min |x| and
    // decides "no", which is what the reference's comparison against a NaN quotient decides (VR_TAP_PRESHIFT)
    long long calls = 0, call_mismatches = 0;
    {
        uint32_t st = 0x2545F491u;
        auto nextu = [&st]() { st ^= st << 13; st ^= st >> 17; st ^= st << 5; return st; };
        const float special[5] = { INFINITY, -INFINITY, NAN, -NAN, 3.0e38f };
        for (int it = 0; it < 4000000; ++it) {
            v3 p = v3{ (float)(nextu() & 0xFFFFFFu) * (1.0f / 65536.0f) - 8.0f, (float)(nextu() & 0xFFFFFFu) * (1.0f / 65536.0f) - 8.0f, (float)(nextu() & 0xFFFFFFu) * (1.0f / 65536.0f) - 8.0f };
            if (it % 4 == 0) {                                  // every fourth call: 1-3 axes not finite
                const uint32_t m = 1u + nextu() % 7u;
                if (m & 1u) p.x = special[nextu() % 5u];
                if (m & 2u) p.y = special[nextu() % 5u];
                if (m & 4u) p.z = special[nextu() % 5u];
            }
            const uint32_t seed = nextu();
            uint32_t s_fast = seed, s_ref = seed;
            int32_t fx, fy, fz, rx, ry, rz;
            tricubic_tap_t<true>(p, s_fast, fx, fy, fz);
            tricubic_tap_t<false>(p, s_ref, rx, ry, rz);
            ++calls;
            if (fx != rx || fy != ry || fz != rz || s_fast != s_ref) ++call_mismatches;
        }
    }
    violations += call_mismatches;
    std::printf("whole calls, fast form against the reference's code: %lld calls (a quarter with non-finite coordinates), mismatches %lld\n", calls, call_mismatches);
    (void)worst_t;
    std::printf("stride %u: %lld (t, test) pairs checked, violations %lld (fast yes where the reference says no: %lld, fast no where it says yes: %lld)\n",
                stride, checked, violations, decided_wrong_yes, decided_wrong_no);
    std::printf("largest relative difference between the two thresholds: %.3e = 2^%.2f (guard band: " VR_BAND_TEXT "); widest band: %.3e of the draws\n",
                max_disc, std::log2(max_disc > 0 ? max_disc : 1e-300), max_band);
    return violations == 0 ? 0 : 1;
}
