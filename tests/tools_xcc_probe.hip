// Diagnostic (not part of the product): checks on the device that workgroups are dealt round-robin to the 8 XCDs, i.e. that
// blockIdx.x & 7 == HW_REG_XCC_ID, the assumption behind the XCD-aware work queue of vr_kernels.hip.
//   hipcc --offload-arch=gfx950 -O2 -o build/xcc tests/tools_xcc_probe.hip && ./build/xcc      (MI355X: 4096 of 4096 blocks match)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID, bits [3:0]
}
int main() {
    const int n = 4096;
    unsigned* d; hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
    unsigned h[n]; hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int match = 0; for (int i = 0; i < n; ++i) match += ((h[i] & 15u) == (unsigned)(i & 7));
    printf("first 24 blocks -> XCC_ID:"); for (int i = 0; i < 24; ++i) printf(" %u", h[i] & 15u); printf("\nblocks with XCC_ID == blockIdx & 7: %d of %d\n", match, n);
    return 0;
}
