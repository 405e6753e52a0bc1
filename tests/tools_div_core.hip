// Diagnostic / verifier: is the compiler's IEEE fp32 division sequence WITHOUT its three guard instructions (v_div_scale_f32 x 2, v_div_fixup_f32; v_div_fmas_f32
// becomes an fma) -- vr_math.h div_core -- bit-identical to n / d whenever the operands satisfy the predicate the product establishes before it uses it
// (vr_math.h div_core_domain)?  By the ISA's definition the dropped instructions are the identity there (no rescaling: denominator normal and its reciprocal normal,
// numerator's exponent field above 23, quotient neither near overflow nor denormal; no special case); this tool checks the claim on the hardware:
//   (a) every pair of exponent fields of the domain x 2^12 random mantissa pairs each,
//   (b) 2^36 random pairs drawn from the domain,
//   (c) numerator +0 (quotient +0), and for a few denominators every numerator of the domain.
// It also counts, for information, mismatches OUTSIDE the domain (expected: some).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tests/tools_div_core.hip -o build_tools/div_core ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../volren_amd/csrc/vr_math.h"
using namespace vr;

__device__ __forceinline__ uint32_t mix32(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return (uint32_t)x; }
__device__ __forceinline__ bool same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b); }

// (a): exponent fields en, ed over the whole range 1..254; the predicate decides which pairs count
__global__ void grid_check(unsigned long long* out) {      // out[0] pairs in domain, [1] mismatches in domain, [2] pairs outside, [3] mismatches outside, [4] first bad n, [5] first bad d
    const uint32_t en = blockIdx.x % 254u + 1u, ed = blockIdx.x / 254u + 1u;
    unsigned long long in = 0, bad_in = 0, outc = 0, bad_out = 0;
    for (uint32_t k = threadIdx.x; k < 4096u; k += blockDim.x) {
        const uint32_t r0 = mix32(((uint64_t)blockIdx.x << 20) | k), r1 = mix32(((uint64_t)blockIdx.x << 20) | k | (1ull << 40));
        const float n = __uint_as_float((en << 23) | (r0 & 0x7FFFFFu) | (r1 & 0x80000000u));
        const float d = __uint_as_float((ed << 23) | (r1 & 0x7FFFFFu) | (r0 & 0x80000000u));
        const bool ok = same_bits(div_core(n, d), n / d);
        if (div_core_domain(n, d)) { ++in; if (!ok) { ++bad_in; out[4] = __float_as_uint(n); out[5] = __float_as_uint(d); } }
        else { ++outc; if (!ok) ++bad_out; }
    }
    atomicAdd(&out[0], in); atomicAdd(&out[1], bad_in); atomicAdd(&out[2], outc); atomicAdd(&out[3], bad_out);
}
// (b): random pairs inside the domain (exponent fields drawn from the domain's range, rejected when the quotient's condition fails)
__global__ void random_check(unsigned long long* out, uint64_t per_thread, uint64_t salt) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long in = 0, bad = 0;
    for (uint64_t i = 0; i < per_thread; ++i) {
        const uint64_t key = (tid * per_thread + i) ^ salt;
        const uint32_t r0 = mix32(key), r1 = mix32(key ^ 0x9e3779b97f4a7c15ull), r2 = mix32(key ^ 0xc2b2ae3d27d4eb4full);
        const uint32_t en = 27u + r2 % 200u, ed = 27u + (r2 >> 8) % 200u;
        const float n = __uint_as_float((en << 23) | (r0 & 0x807FFFFFu)), d = __uint_as_float((ed << 23) | (r1 & 0x807FFFFFu));
        if (!div_core_domain(n, d)) continue;
        ++in;
        if (!same_bits(div_core(n, d), n / d)) { ++bad; out[4] = __float_as_uint(n); out[5] = __float_as_uint(d); }
    }
    atomicAdd(&out[0], in); atomicAdd(&out[1], bad);
}
// (c): for one denominator, every float numerator (all 2^32 patterns; only those of the domain count, +0 among them)
__global__ void sweep_check(unsigned long long* out, float d) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long in = 0, bad = 0;
    for (uint64_t i = tid; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float n = __uint_as_float((uint32_t)i);
        if (!div_core_domain(n, d)) continue;
        ++in;
        if (!same_bits(div_core(n, d), n / d)) { ++bad; out[4] = (uint32_t)i; out[5] = __float_as_uint(d); }
    }
    atomicAdd(&out[0], in); atomicAdd(&out[1], bad);
}
int main() {
    unsigned long long* out; hipMalloc(&out, 64);
    unsigned long long h[6];
    unsigned long long total_bad = 0;
    hipMemset(out, 0, 64);
    grid_check<<<254 * 254, 256>>>(out);
    hipMemcpy(h, out, 48, hipMemcpyDeviceToHost);
    printf("(a) exponent grid 254 x 254 x 4096 mantissa pairs: %llu pairs in the domain, mismatches %llu (first: n 0x%08llx d 0x%08llx); outside the domain %llu pairs, mismatches %llu\n", h[0], h[1], h[4], h[5], h[2], h[3]);
    total_bad += h[1];
    hipMemset(out, 0, 64);
    random_check<<<4096, 256>>>(out, 1ull << 16, 0x1234567ull);
    hipMemcpy(h, out, 48, hipMemcpyDeviceToHost);
    printf("(b) random pairs: %llu in the domain, mismatches %llu (first: n 0x%08llx d 0x%08llx)\n", h[0], h[1], h[4], h[5]);
    total_bad += h[1];
    const float ds[8] = { 1.0f, 0.99999994f, 1.0000001f, 3.0f, 1.1754944e-38f * 16777216.0f, 7.0e-13f, 1.5e12f, -0.33333334f };
    for (float d : ds) {
        hipMemset(out, 0, 64);
        sweep_check<<<4096, 256>>>(out, d);
        hipMemcpy(h, out, 48, hipMemcpyDeviceToHost);
        printf("(c) denominator %.9g: %llu numerators in the domain, mismatches %llu\n", d, h[0], h[1]);
        total_bad += h[1];
    }
    printf("total mismatches inside the domain: %llu\n", total_bad);
    return total_bad == 0 ? 0 : 1;
}
