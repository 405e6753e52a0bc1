// Diagnostic (not part of the product): how many cycles does one wave64 fp32 VALU instruction occupy a SIMD on gfx950?
// Every wave runs N dependent-free v_fma_f32 on 8 independent accumulators; with W waves per SIMD resident the SIMD is the
// bottleneck, so cycles per instruction per SIMD = elapsed shader cycles * (SIMDs busy) / instructions issued.
//   hipcc --offload-arch=gfx950 -O2 -o build/valu_rate tests/tools_valu_rate.hip && ./build/valu_rate
// Also times the same loop with v_cndmask / v_add_u32 / v_mul_lo_u32 / v_rcp_f32 / v_readlane to price them against fma.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0000001f, c = 1e-9f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define R8(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
        if (OP == 0) {
#define F(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            R8(F) R8(F) R8(F) R8(F)
#undef F
        } else if (OP == 1) {
#define F(a) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
            R8(F) R8(F) R8(F) R8(F)
#undef F
        } else if (OP == 2) {
#define F(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
            R8(F) R8(F) R8(F) R8(F)
#undef F
        } else if (OP == 3) {
#define F(a) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
            R8(F) R8(F) R8(F) R8(F)
#undef F
        } else if (OP == 4) {
#define F(a) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##a) : "v"(q));
            double pa0 = a0, pa1 = a1, pa2 = a2, pa3 = a3, pa4 = a4, pa5 = a5, pa6 = a6, pa7 = a7, q = b;
            R8(F) R8(F) R8(F) R8(F)
            a0 += (float)pa0; a1 += (float)pa1; a2 += (float)pa2; a3 += (float)pa3; a4 += (float)pa4; a5 += (float)pa5; a6 += (float)pa6; a7 += (float)pa7;
#undef F
        } else if (OP == 5) {
#define F(a) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));
            R8(F) R8(F) R8(F) R8(F)
#undef F
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name, int blocks_per_cu) {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * blocks_per_cu, iters = 20000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long* h = new unsigned long long[blocks];
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
    // one block = 4 waves = one wave per SIMD; blocks_per_cu waves share each SIMD
    const double inst_per_simd = (double)iters * 32.0 * blocks_per_cu;
    printf("%-14s waves/SIMD %d: %.3f ms, %.0f shader cycles per wave, %.2f cycles per wave-instruction per SIMD, %.2f T wave-inst-lanes/s\n",
           name, blocks_per_cu, ms, mean, mean / inst_per_simd, inst_per_simd * 4 * cus * 64 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc); delete[] h;
}

int main() {
    for (int w : { 1, 2, 4 }) {
        run<0>("v_fma_f32", w);
        run<3>("v_add_u32", w);
        run<2>("v_mul_lo_u32", w);
        run<5>("v_mul_u32_u24", w);
        run<1>("v_rcp_f32", w);
        run<4>("v_pk_fma_f32", w);
    }
    return 0;
}
