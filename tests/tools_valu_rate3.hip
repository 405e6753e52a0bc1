// Diagnostic (not part of the product; round 5): what makes a vector instruction dear on gfx950?  SIMD cycles per wave64 instruction with 4 wavefronts
// per SIMD, for encodings (VOP2 / VOP3), operand kinds (VGPR / SGPR / inline constant / literal), modifiers, and the integer / select / convert
// opcodes the path tracer's index arithmetic is made of.  Method as tools_valu_rate2.hip: 8 independent chains x 32 instructions per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#define DEF_KERNEL(NAME, ASM) \
__global__ void __launch_bounds__(256) k_##NAME(float* out, int iters, unsigned long long* cyc) { \
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b = 1.0000001f, c = 3.0f; unsigned long long sm = 0x5555555555555555ull; float sb; unsigned si = 3u; \
    asm volatile("s_mov_b32 %0, 0x3f800347" : "=s"(sb)); asm volatile("s_mov_b32 %0, 3" : "=s"(si)); \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a0), "v"(a3) : "vcc"); \
    const unsigned long long t0 = __builtin_readcyclecounter(); \
    for (int i = 0; i < iters; ++i) { R8(ASM) R8(ASM) R8(ASM) R8(ASM) } \
    const unsigned long long t1 = __builtin_readcyclecounter(); \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b + (float)si + (float)(sm & 1); \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; }
#define A_mul(a) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_mul_e64(a) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_mul_neg(a) asm volatile("v_mul_f32_e64 %0, -%0, %1" : "+v"(a) : "v"(b));
#define A_mul_sgpr(a) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "s"(sb));
#define A_mul_inl(a) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a));
#define A_mul_lit(a) asm volatile("v_mul_f32 %0, 0x3f800347, %0" : "+v"(a));
#define A_add_sgpr(a) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "s"(sb));
#define A_fma_sgpr(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "s"(sb), "v"(c));
#define A_fma_neg(a) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_sub(a) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_min(a) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_max(a) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_med3(a) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_minu(a) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_maxi(a) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_cnd_vcc(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));
#define A_cnd_sgpr(a) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(sm));
#define A_cmp_vcc(a) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
#define A_cmp_sgpr(a) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(sm) : "v"(a), "v"(b));
#define A_cmp_u32(a) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
#define A_cmpx(a) asm volatile("v_cmp_class_f32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
#define A_cmpcnd(a) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
#define A_mul24(a) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_mul24_inl(a) asm volatile("v_mul_u32_u24 %0, 9, %0" : "+v"(a));
#define A_mad24(a) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_lshl(a) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a));
#define A_lshl_v(a) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(c));
#define A_lshr(a) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a));
#define A_ashr(a) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a));
#define A_lshladd(a) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a) : "v"(b));
#define A_addlshl(a) asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(a) : "v"(b));
#define A_lshlor(a) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a) : "v"(b));
#define A_andor(a) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_bfe(a) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a));
#define A_and_inl(a) asm volatile("v_and_b32 %0, 7, %0" : "+v"(a));
#define A_and_lit(a) asm volatile("v_and_b32 %0, 0xffffff, %0" : "+v"(a));
#define A_or(a) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_add3(a) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_addu_sgpr(a) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "s"(si));
#define A_subu(a) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define A_cvt_flr(a) asm volatile("v_cvt_flr_i32_f32 %0, %0" : "+v"(a));
#define A_cvt_u32(a) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a));
#define A_cvt_f32_i32(a) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a));
#define A_cvt_f32_ub0(a) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a));
#define A_cvt_f16(a) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a));
#define A_fract(a) asm volatile("v_fract_f32 %0, %0" : "+v"(a));
#define A_trunc(a) asm volatile("v_trunc_f32 %0, %0" : "+v"(a));
#define A_floor(a) asm volatile("v_floor_f32 %0, %0" : "+v"(a));
#define A_mov_dpp(a) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));
#define A_add_dpp(a) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));
#define A_add_sdwa(a) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(a) : "v"(b));
#define A_bperm(a) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(b));
#define A_rdfl(a) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(si) : "v"(a));
#define A_sand(a) asm volatile("s_and_b64 %0, %0, exec" : "+s"(sm));
#define A_sbcnt(a) asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(si) : "s"(sm) : "scc");
#define A_smov(a) asm volatile("s_mov_b32 %0, %0" : "+s"(si));
#define A_ldexp(a) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a) : "v"(c));
#define A_mbcnt(a) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a) : "s"(si));
#define A_xor3(a) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define A_mulhi(a) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define LIST(X) X(mul) X(mul_e64) X(mul_neg) X(mul_sgpr) X(mul_inl) X(mul_lit) X(add_sgpr) X(fma_sgpr) X(fma_neg) X(sub) X(min) X(max) X(med3) X(minu) X(maxi) \
    X(cnd_vcc) X(cnd_sgpr) X(cmp_vcc) X(cmp_sgpr) X(cmp_u32) X(cmpx) X(cmpcnd) X(mul24) X(mul24_inl) X(mad24) X(lshl) X(lshl_v) X(lshr) X(ashr) X(lshladd) X(addlshl) X(lshlor) X(andor) X(bfe) \
    X(and_inl) X(and_lit) X(or) X(add3) X(addu_sgpr) X(subu) X(cvt_flr) X(cvt_u32) X(cvt_f32_i32) X(cvt_f32_ub0) X(cvt_f16) X(fract) X(trunc) X(floor) X(mov_dpp) X(add_dpp) X(add_sdwa) X(bperm) X(rdfl) \
    X(sand) X(sbcnt) X(smov) X(ldexp) X(mbcnt) X(xor3) X(mulhi)
#define DK(N) DEF_KERNEL(N, A_##N)
LIST(DK)
typedef void (*KF)(float*, int, unsigned long long*);
static void run(const char* name, KF k, int blocks_per_cu, int per_macro) {
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * blocks_per_cu, iters = 4000;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100, cyc);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long* h = new unsigned long long[blocks];
    (void)hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
    printf("%-12s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD\n", name, blocks_per_cu, mean / ((double)iters * 32.0 * blocks_per_cu * per_macro));
    (void)hipFree(out); (void)hipFree(cyc); delete[] h;
}
int main() {
#define RUN(N) run(#N, k_##N, 4, 1);
    LIST(RUN)
    return 0;
}
