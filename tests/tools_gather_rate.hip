// Diagnostic (not part of the product): how many per-lane gather accesses a gfx950 CU's vector memory path (TA -> TCP) sustains
// per cycle, as a function of how the 64 addresses of one wave-level load spread over cache lines and of where the lines
// live (L1 / L2 / memory).  The path tracer issues ~110 per-lane loads per sample, almost all of them fully divergent; this
// tool gives the ceiling that count has to be compared with.
//   hipcc --offload-arch=gfx950 -O3 -o build/gather_rate tests/tools_gather_rate.hip && build/gather_rate
// 16 wavefronts per CU (the kernel's occupancy), 8 independent loads in flight per wavefront, addresses from an LCG (not from
// the loaded data: throughput, not latency).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// SPREAD = lanes that share one 128-byte line within a wave-level load: 64 (one line per wave), 16, 4, 1 (fully divergent)
template <int SPREAD, int BYTES>
__global__ void __launch_bounds__(256, 4) gather(const uint32_t* __restrict__ table, uint32_t line_mask, int iters, uint32_t* out, unsigned long long* cyc) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t group = (blockIdx.x * 256u + threadIdx.x) / (uint32_t)SPREAD;       // lanes of one group draw the same line
    uint32_t s = group * 2654435761u + 12345u;
    uint32_t acc = 0u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s = s * 1664525u + 1013904223u;
            const uint32_t line = (s >> 8) & line_mask;
            const uint32_t dw = (lane * (BYTES / 4)) & 31u & ~(uint32_t)(BYTES / 4 - 1);        // position inside the line
            const uint32_t* p = table + (size_t)line * 32u + dw;
            if (BYTES == 4) v[k] = p[0];
            else if (BYTES == 8) { const uint2 q = *reinterpret_cast<const uint2*>(p); v[k] = q.x ^ q.y; }
            else { const uint4 q = *reinterpret_cast<const uint4*>(p); v[k] = q.x ^ q.y ^ q.z ^ q.w; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256u + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*KF)(const uint32_t*, uint32_t, int, uint32_t*, unsigned long long*);
static void run(const char* name, KF k, const uint32_t* table, size_t table_bytes, int spread, int bytes) {
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 4;
    const int iters = table_bytes > (64u << 20) ? 400 : 2000;
    uint32_t* out; unsigned long long* cyc;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&cyc, (size_t)blocks * 8);
    const uint32_t line_mask = (uint32_t)(table_bytes / 128u) - 1u;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, table, line_mask, 50, out, cyc);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, table, line_mask, iters, out, cyc);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
    const double lane_loads_per_cu = 4.0 * 256.0 * iters * 8.0;
    const double wave_loads_per_cu = lane_loads_per_cu / 64.0;
    printf("%-22s lanes/line %2d  %2d B/lane  table %8.2f MiB : %6.1f counter ticks per wave-level load per CU, %7.3f lane-accesses per tick per CU;  wall %.3f ms -> %6.1f G lane-accesses/s chip\n",
           name, spread, bytes, table_bytes / 1048576.0, mean / wave_loads_per_cu, lane_loads_per_cu / mean, ms, lane_loads_per_cu * cus / (ms * 1e6));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    const size_t big = 1024u << 20;
    uint32_t* table; (void)hipMalloc(&table, big); (void)hipMemset(table, 1, big);
    const size_t sizes[4] = { 16u << 10, 2u << 20, 128u << 20, big };
    const char* names[4] = { "L1-resident", "L2-resident", "MALL-sized", "HBM" };
    for (int t = 0; t < 4; ++t) {
        run(names[t], gather<64, 4>, table, sizes[t], 64, 4);
        run(names[t], gather<16, 4>, table, sizes[t], 16, 4);
        run(names[t], gather<4, 4>, table, sizes[t], 4, 4);
        run(names[t], gather<2, 4>, table, sizes[t], 2, 4);
        run(names[t], gather<1, 4>, table, sizes[t], 1, 4);
        run(names[t], gather<1, 8>, table, sizes[t], 1, 8);
        run(names[t], gather<1, 16>, table, sizes[t], 1, 16);
    }
    // fully divergent dword gathers against the table size: where the cache levels end
    for (size_t kb = 8; kb <= (1u << 20); kb *= 2) run("size sweep", gather<1, 4>, table, kb << 10, 1, 4);
    (void)hipFree(table);
    return 0;
}
