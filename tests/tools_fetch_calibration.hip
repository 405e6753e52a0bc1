// Diagnostic: calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on the access patterns of the path tracer, as MI355X_MICROARCH.md ("HBM") asks
// before an absolute is trusted ("other access widths are uncalibrated").  Every kernel touches a KNOWN number of distinct 128-byte lines of a
// 1 GiB table (far beyond L2 and Infinity Cache) exactly once, so bytes-per-access = counter / accesses:
//   stream16   coalesced 16 B per lane (the guide's reference pattern: FETCH_SIZE reports half)
//   gather4    fully divergent dword gathers, one per line                      (voxel taps, majorant cells, environment records)
//   gather16   fully divergent dwordx4 gathers, one per line
//   slot64     four dwordx4 loads of one 64-byte slot per lane, slots in distinct lines   (an event reading a path's cold slot)
//   line2halves one dword from each 64-byte half of a line per lane: tells whether a gather miss fills 64 or 128 bytes
//   sector32w  two dwordx4 stores into one 32-byte sector per lane, sectors in distinct lines (an event writing one sector of a cold slot)
//   sample16w  one dwordx4 store per lane to consecutive 16-byte slots            (write_sample: coalesced)
// usage: rocprofv3 --pmc FETCH_SIZE -d out -o out --output-format csv -- ./fetch_calibration ; same with WRITE_SIZE; tests/tools_fetch_calibration.sh sums per kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr uint32_t kLines = 1u << 23;            // 1 GiB / 128 B
constexpr uint32_t kThreads = 1u << 22;          // every thread touches its own line(s): 2 lines apart so that no two accesses share a line
__device__ __forceinline__ uint32_t my_line(uint32_t i) { return (i * 2654435761u) & (kLines - 1u) & ~1u | (i >> 22); }     // a permutation-ish scatter; distinct for i < 2^22
__global__ void stream16(const uint4* __restrict__ t, uint32_t* out) { const uint32_t i = blockIdx.x * 256u + threadIdx.x; const uint4 v = t[i]; if (v.x == 0xDEADBEEFu) out[0] = v.y; }
__global__ void gather4(const uint32_t* __restrict__ t, uint32_t* out) { const uint32_t i = blockIdx.x * 256u + threadIdx.x; const uint32_t v = t[(size_t)((i * 2654435761u) & (kLines - 1u)) * 32u + (i & 31u)]; if (v == 0xDEADBEEFu) out[0] = v; }
__global__ void gather16(const uint4* __restrict__ t, uint32_t* out) { const uint32_t i = blockIdx.x * 256u + threadIdx.x; const uint4 v = t[(size_t)((i * 2654435761u) & (kLines - 1u)) * 8u + (i & 7u)]; if (v.x == 0xDEADBEEFu) out[0] = v.y; }
__global__ void slot64(const uint4* __restrict__ t, uint32_t* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint4* p = t + (size_t)((i * 2654435761u) & (kLines - 1u)) * 8u + 4u * (i & 1u);
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    if ((a.x ^ b.x ^ c.x ^ d.x) == 0xDEADBEEFu) out[0] = a.y;
}
// both 64-byte halves of one line per lane: one request (128-byte fills) or two (64-byte fills)?
__global__ void line2halves(const uint32_t* __restrict__ t, uint32_t* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t* p = t + (size_t)((i * 2654435761u) & (kLines - 1u)) * 32u;
    const uint32_t a = p[1], b = p[25];
    if ((a ^ b) == 0xDEADBEEFu) out[0] = a;
}
__global__ void sector32w(uint4* __restrict__ t) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint4* p = t + (size_t)((i * 2654435761u) & (kLines - 1u)) * 8u + 2u * (i & 3u);
    p[0] = make_uint4(i, 1, 2, 3); p[1] = make_uint4(i, 4, 5, 6);
}
__global__ void sample16w(uint4* __restrict__ t) { const uint32_t i = blockIdx.x * 256u + threadIdx.x; t[i] = make_uint4(i, 1, 2, 3); }
int main() {
    void* t; if (hipMalloc(&t, (size_t)kLines * 128u) != hipSuccess) return 1;
    (void)hipMemset(t, 1, (size_t)kLines * 128u);
    uint32_t* out; (void)hipMalloc(&out, 64);
    const dim3 g(kThreads / 256u), b(256);
    // (i * 2654435761) & (2^23 - 1) is injective for i < 2^23 (odd multiplier): 2^22 threads touch 2^22 distinct lines
    hipLaunchKernelGGL(stream16, g, b, 0, 0, (const uint4*)t, out);
    hipLaunchKernelGGL(gather4, g, b, 0, 0, (const uint32_t*)t, out);
    hipLaunchKernelGGL(gather16, g, b, 0, 0, (const uint4*)t, out);
    hipLaunchKernelGGL(slot64, g, b, 0, 0, (const uint4*)t, out);
    hipLaunchKernelGGL(line2halves, g, b, 0, 0, (const uint32_t*)t, out);
    hipLaunchKernelGGL(sector32w, g, b, 0, 0, (uint4*)t);
    hipLaunchKernelGGL(sample16w, g, b, 0, 0, (uint4*)t);
    (void)hipDeviceSynchronize();
    printf("accesses per kernel: %u (one distinct 128-byte line each; stream16 / sample16w: 16 bytes per access, consecutive)\n", kThreads);
    return 0;
}
