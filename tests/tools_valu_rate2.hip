// Diagnostic (not part of the product): SIMD cycles per wave64 instruction on gfx950 for the instruction kinds the path tracer
// uses, with 4 waves per SIMD resident (the kernel's occupancy).  See tools_valu_rate.hip for the method.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#define DEF_KERNEL(NAME, ASM) \
__global__ void __launch_bounds__(256) k_##NAME(float* out, int iters, unsigned long long* cyc) { \
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b = 1.0000001f, c = 3.0f; unsigned long long sm = 0x5555555555555555ull; \
    const unsigned long long t0 = __builtin_readcyclecounter(); \
    for (int i = 0; i < iters; ++i) { R8(ASM) R8(ASM) R8(ASM) R8(ASM) } \
    const unsigned long long t1 = __builtin_readcyclecounter(); \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b; \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; }
#define A_fma(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_mul(a) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_add(a) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_mov(a) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));
#define A_and(a) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_lshl(a) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a));
#define A_or3(a) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_mad24(a) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_mullo(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_cvtfu(a) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a));
#define A_cvtif(a) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a));
#define A_floor(a) asm volatile("v_floor_f32 %0, %0" : "+v"(a));
#define A_cmp(a) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
#define A_cmpe64(a) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(sm) : "v"(a), "v"(b));
#define A_cnd(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
#define A_cnde64(a) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(sm));
#define A_rcp(a) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
#define A_sqrt(a) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a));
#define A_log(a) asm volatile("v_log_f32 %0, %0" : "+v"(a));
#define A_divscale(a) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a) : "v"(b) : "vcc");
#define A_divfmas(a) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
#define A_divfixup(a) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_min3(a) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_pkmul(a) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##a) : "v"(q));
#define A_pkadd(a) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##a) : "v"(q));
#define A_fmac(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_fmaak(a) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3e2aaaab" : "+v"(a) : "v"(b));
#define A_xor(a) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_addu(a) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_lshladd64(a) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(p##a) : "v"(q));
#define A_mulsgpr(a) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "s"(sb));
#define A_readlane(a) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(si) : "v"(a));
#define A_snop(a) asm volatile("s_nop 0");
#define A_sadd(a) asm volatile("s_add_u32 %0, %0, 1" : "+s"(si));
DEF_KERNEL(fma, A_fma) DEF_KERNEL(mul, A_mul) DEF_KERNEL(add, A_add) DEF_KERNEL(mov, A_mov) DEF_KERNEL(and_, A_and) DEF_KERNEL(lshl, A_lshl)
DEF_KERNEL(or3, A_or3) DEF_KERNEL(mad24, A_mad24) DEF_KERNEL(mullo, A_mullo) DEF_KERNEL(cvtfu, A_cvtfu) DEF_KERNEL(cvtif, A_cvtif) DEF_KERNEL(floor_, A_floor)
DEF_KERNEL(cmp, A_cmp) DEF_KERNEL(cmpe64, A_cmpe64) DEF_KERNEL(cnd, A_cnd) DEF_KERNEL(cnde64, A_cnde64) DEF_KERNEL(rcp, A_rcp) DEF_KERNEL(sqrt_, A_sqrt) DEF_KERNEL(log_, A_log)
DEF_KERNEL(divscale, A_divscale) DEF_KERNEL(divfmas, A_divfmas) DEF_KERNEL(divfixup, A_divfixup) DEF_KERNEL(min3, A_min3) DEF_KERNEL(fmac, A_fmac) DEF_KERNEL(fmaak, A_fmaak)
DEF_KERNEL(xor_, A_xor) DEF_KERNEL(addu, A_addu)
#undef DEF_KERNEL
#define DEF_KERNEL2(NAME, ASM) \
__global__ void __launch_bounds__(256) k_##NAME(float* out, int iters, unsigned long long* cyc) { \
    double pa0 = threadIdx.x, pa1 = pa0 + 1, pa2 = pa0 + 2, pa3 = pa0 + 3, pa4 = pa0 + 4, pa5 = pa0 + 5, pa6 = pa0 + 6, pa7 = pa0 + 7, q = 1.000001; \
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; float sb = 1.0001f; unsigned si = 0; \
    asm volatile("s_mov_b32 %0, 0x3f800347" : "=s"(sb)); \
    const unsigned long long t0 = __builtin_readcyclecounter(); \
    for (int i = 0; i < iters; ++i) { R8(ASM) R8(ASM) R8(ASM) R8(ASM) } \
    const unsigned long long t1 = __builtin_readcyclecounter(); \
    out[blockIdx.x * 256 + threadIdx.x] = (float)(pa0 + pa1 + pa2 + pa3 + pa4 + pa5 + pa6 + pa7) + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)si; \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; }
DEF_KERNEL2(pkmul, A_pkmul) DEF_KERNEL2(pkadd, A_pkadd) DEF_KERNEL2(lshladd64, A_lshladd64) DEF_KERNEL2(mulsgpr, A_mulsgpr) DEF_KERNEL2(readlane, A_readlane) DEF_KERNEL2(snop, A_snop) DEF_KERNEL2(sadd, A_sadd)

typedef void (*KF)(float*, int, unsigned long long*);
static void run(const char* name, KF k, int blocks_per_cu) {
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * blocks_per_cu, iters = 8000;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100, cyc);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long* h = new unsigned long long[blocks];
    (void)hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
    printf("%-12s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD\n", name, blocks_per_cu, mean / ((double)iters * 32.0 * blocks_per_cu));
    (void)hipFree(out); (void)hipFree(cyc); delete[] h;
}
int main() {
#define RUN(N) run(#N, k_##N, 4);
    RUN(fma) RUN(mul) RUN(add) RUN(mov) RUN(and_) RUN(lshl) RUN(or3) RUN(mad24) RUN(mullo) RUN(cvtfu) RUN(cvtif) RUN(floor_) RUN(cmp) RUN(cmpe64) RUN(cnd) RUN(cnde64)
    RUN(rcp) RUN(sqrt_) RUN(log_) RUN(divscale) RUN(divfmas) RUN(divfixup) RUN(min3) RUN(fmac) RUN(fmaak) RUN(xor_) RUN(addu) RUN(pkmul) RUN(pkadd) RUN(lshladd64) RUN(mulsgpr) RUN(readlane) RUN(snop) RUN(sadd)
    run("fma", k_fma, 1); run("fma", k_fma, 2); run("snop", k_snop, 1); run("sadd", k_sadd, 1);
    return 0;
}
