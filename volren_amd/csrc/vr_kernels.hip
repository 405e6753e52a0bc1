// vr_kernels.hip -- gfx950 kernels of the volume path tracer.
//
// pathtrace_kernel (the hot path): persistent wavefronts pull (8x8 pixel tile x 8 samples) work units from an XCD-aware
// counter; each wavefront keeps a private pool of more path slots than it has lanes, so that the frequent march/collide
// code always finds lanes to fill and the rare, expensive events (new sample with the 32-round TEA hash, next-event
// estimation, scatter, escape) run as near-full-width batches of parked paths.  The per-path code -- the reference's
// trace_path and everything it calls, restated as a state machine -- is in vr_trace.h; this file is scheduling and launch.
// Radiance per (pixel, sample) goes to a sample pool; accumulate_kernel folds it into the RGBA32F running mean in sample
// order, which makes the result bit-identical to the reference's one-dispatch-per-sample loop (renderer.cpp:138-140,
// pathtracer_brick.glsl:36).  MFMA is not used: there is no dense contraction on this path.
//
// Also here: the environment importance pyramid + warp table (env_setup.glsl, environment.cpp), the dense->brick encoder
// (voldata to_brick_grid at commit()), majorant remap, tonemap.glsl, direct volume rendering (common.glsl:571-591),
// tile pack/unpack for the multi-GPU gather, and a math probe for the tests.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

#include "vr_device.h"
#include "vr_pathtrace.h"

namespace vr {

// integrator = 2: direct volume rendering, one thread per (pixel, sample) item, same sample-buffer layout
__global__ void __launch_bounds__(256)
dvr_kernel(const SceneParams P, float* __restrict__ sbuf, const LaunchDesc D) {
    const uint32_t per_unit = (uint32_t)(D.spu * 64);
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t u = g / per_unit, item = g - u * per_unit;      // dvr: queue position order, one thread per item
    if (u >= D.n_units) return;
    const WorkUnit wu = make_unit(D, P.u.resolution[0], u, sbuf);
    if ((int32_t)item >= wu.n_items) return;
    const int32_t px = wu.px0 + (int32_t)(item & 7u), py = wu.py0 + (int32_t)((item >> 3) & 7u);
    if (px >= P.u.resolution[0] || py >= P.u.resolution[1]) return;
    float L[4];
    dvr_sample(P, px, py, wu.first_sample + (int32_t)(item >> 6), L);
    reinterpret_cast<float4*>(sbuf)[wu.base + item] = make_float4(L[0], L[1], L[2], L[3]);
}

// integrator = 3: trace_path with the ray-marching trackers, one thread per (pixel, sample) item like dvr_kernel
__global__ void __launch_bounds__(256)
raymarch_kernel(const SceneParams P, float* __restrict__ sbuf, const LaunchDesc D) {
    const uint32_t per_unit = (uint32_t)(D.spu * 64);
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t u = g / per_unit, item = g - u * per_unit;
    if (u >= D.n_units) return;
    const WorkUnit wu = make_unit(D, P.u.resolution[0], u, sbuf);
    if ((int32_t)item >= wu.n_items) return;
    const int32_t px = wu.px0 + (int32_t)(item & 7u), py = wu.py0 + (int32_t)((item >> 3) & 7u);
    if (px >= P.u.resolution[0] || py >= P.u.resolution[1]) return;
    float L[4];
    raymarch_path_sample(P, px, py, wu.first_sample + (int32_t)(item >> 6), L);
    reinterpret_cast<float4*>(sbuf)[wu.base + item] = make_float4(L[0], L[1], L[2], L[3]);
}

// Running mean over the samples of one launch, in sample order (pathtracer_brick.glsl:36): one thread per pixel.
__global__ void __launch_bounds__(256)
accumulate_kernel(const float* __restrict__ sbuf, float* __restrict__ fb, const int32_t* __restrict__ tiles, int32_t n_tiles,
                  int32_t W, int32_t H, int32_t first_sample, int32_t n_samples, int32_t spu) {
    const int32_t tiles_x = (W + 15) >> 4;
    const int32_t tile = tiles ? tiles[blockIdx.x] : (int32_t)blockIdx.x;
    const int32_t wave = threadIdx.x >> 6, p = threadIdx.x & 63;
    const int32_t px = (tile % tiles_x) * 16 + ((wave & 1) << 3) + (p & 7);
    const int32_t py = (tile / tiles_x) * 16 + ((wave >> 1) << 3) + (p >> 3);
    if (px >= W || py >= H) return;
    float4* texel = reinterpret_cast<float4*>(fb) + (size_t)py * W + px;
    float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (first_sample > 1) { const float4 c = *texel; acc[0] = c.x; acc[1] = c.y; acc[2] = c.z; acc[3] = c.w; }
    const float4* sb = reinterpret_cast<const float4*>(sbuf);
    for (int32_t k = 0; k < n_samples; ++k) {
        const int32_t chunk = k / spu, sl = k - chunk * spu;
        const size_t unit = ((size_t)chunk * n_tiles + blockIdx.x) * 4u + (uint32_t)wave;
        const float4 v = sb[unit * (size_t)(spu * 64) + (size_t)sl * 64u + (uint32_t)p];
        const float L[4] = { v.x, v.y, v.z, v.w };
        accumulate_sample(acc, L, first_sample + k);
    }
    *texel = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// Units of 8 samples x 64 pixels, except on the dense-grid kernel, whose long paths (128 bounces, every camera ray scatters) fill
// the pools better from units of 4: c4 +1.3 %, c2 +-0, c3 -1 % (profiles/r2x_occupancy_recheck.txt) -- and, round 4, on the emission kernel:
// c5cloud +1 %, c5full +0.5 % (profiles/r4d_*)
constexpr int32_t kMaxSamplesPerUnit = 8;
static int32_t samples_per_unit(const PathtraceTuning& T, int variant) { return T.samples_per_unit > 0 ? T.samples_per_unit : ((variant == 1 || variant == 2 || variant == 4) ? 4 : kMaxSamplesPerUnit); }
constexpr int kPtVariants = 5;

PathtraceTuning default_tuning() {
    static std::once_flag once;
    static int32_t env_spu = 0, env_blocks = 0;       // diagnostics only
    std::call_once(once, [] {
        if (const char* e = getenv("VR_SPU")) env_spu = std::max(1, atoi(e));
        if (const char* e = getenv("VR_BLOCKS_PER_CU")) env_blocks = std::max(0, atoi(e));
    });
    PathtraceTuning T;
    T.samples_per_unit = env_spu;
    T.blocks_per_cu = env_blocks;
    return T;
}

size_t pathtrace_pool_floats(const PathtraceTuning& T, int32_t n_tiles, int32_t n_samples) {
    // the samples are rounded up to whole units: sized for whichever unit size a variant may use (samples_per_unit)
    size_t padded = 0;
    for (int variant = 0; variant < kPtVariants; ++variant) {
        const int32_t spu = std::min(n_samples, samples_per_unit(T, variant));
        padded = std::max(padded, (size_t)((n_samples + spu - 1) / spu) * (size_t)spu);
    }
    return padded * (size_t)n_tiles * 4u * 64u * 4u;
}

// the kernel variants live in vr_pathtrace.hip, one compilation each, in two arithmetic modes (bit-exact / tolerance)
extern "C" {
#define VR_PT_DECL(N) \
    int vr_pt_occupancy_##N(int tf, int stats); \
    void vr_pt_launch_##N(int tf, int stats, unsigned grid, hipStream_t stream, const void* P, float* sbuf, float* cold_ws, const void* D, const void* S, uint32_t* status, unsigned long long* stats_buf);
VR_PT_DECL(0) VR_PT_DECL(1) VR_PT_DECL(2) VR_PT_DECL(3) VR_PT_DECL(4) VR_PT_DECL(0_fast) VR_PT_DECL(1_fast) VR_PT_DECL(2_fast) VR_PT_DECL(3_fast) VR_PT_DECL(4_fast)
#undef VR_PT_DECL
}
typedef int (*PtOccupancy)(int, int);
typedef void (*PtLaunch)(int, int, unsigned, hipStream_t, const void*, float*, float*, const void*, const void*, uint32_t*, unsigned long long*);
static const PtOccupancy kPtOccupancy[2][kPtVariants] = { { vr_pt_occupancy_0, vr_pt_occupancy_1, vr_pt_occupancy_2, vr_pt_occupancy_3, vr_pt_occupancy_4 },
                                                          { vr_pt_occupancy_0_fast, vr_pt_occupancy_1_fast, vr_pt_occupancy_2_fast, vr_pt_occupancy_3_fast, vr_pt_occupancy_4_fast } };
static const PtLaunch kPtLaunch[2][kPtVariants] = { { vr_pt_launch_0, vr_pt_launch_1, vr_pt_launch_2, vr_pt_launch_3, vr_pt_launch_4 },
                                                    { vr_pt_launch_0_fast, vr_pt_launch_1_fast, vr_pt_launch_2_fast, vr_pt_launch_3_fast, vr_pt_launch_4_fast } };

// which compiled variant serves a scene (see vr_pathtrace.hip), and why -- `why` is a mask of PathtraceVariantReason (vr_device.h): the run-time variant (3) is an
// order of magnitude slower than the kernels of one scene kind on some scenes (one copy of the hot pair, general forms only, everything decided per wavefront), so a
// scene that lands there for a reason its caller did not ask for is told so (launch_pathtrace, vr_get_int "kernel_variant_reason")
int pathtrace_variant_of(const SceneParams& P, int* why) {
    // (an environment whose warp table failed the check of env_cdf_kernel -- thresholds below 2^-76: the kernels of one scene kind divide by vr_math.h div_core there --
    // goes to the run-time variant, which divides in full; RendererHIP::fill_params pairs no atlases for it)
    // -- and so does a density scale outside [2^-16, 2^24]: the CLEAN form of the march divides by majorants = density_scale x an fp16 number (vr_trace.h march_finish)
    const bool scale_ok = P.u.vol_density_scale >= 1.0f / 65536.0f && P.u.vol_density_scale <= 16777216.0f;
    int reason = 0;
    if (P.u.integrator != 0) reason |= VR_VARIANT_INTEGRATOR;
    if (!P.env_div_safe) reason |= VR_VARIANT_ENV_DIVISION;
    if (!scale_ok) reason |= VR_VARIANT_DENSITY_SCALE;
    int variant;
    if (reason) variant = 3;
    // variant 2: both grids in brick form -- and, when the kernels are built for the paired atlas, sharing one (same brick layout: RendererHIP::commit)
    // (4 = 2 compiled for majorant levels 0-1 in 4x4x4-cell blocks; every other kernel of a fixed layout reads linear tables: RendererHIP::fill_params only
    // sets maj_blocked on the views of frames this variant -- or the run-time variant -- serves)
    else if (P.u.has_emission) {
        const bool mixed = P.density.dense || P.emission.dense || (VR_PAIRED_ATLAS && !P.paired);
        if (mixed) reason |= VR_VARIANT_GRID_FORMS;
        variant = mixed ? 3 : (P.density.maj_blocked ? 4 : 2);
    } else variant = P.density.dense ? 1 : 0;
    if (why) *why = reason;
    return variant;
}
static int pathtrace_variant(const SceneParams& P) {
    int why = 0;
    const int variant = pathtrace_variant_of(P, &why);
    // the two reasons a caller cannot see coming: said once per process, on stderr
    if (why & (VR_VARIANT_ENV_DIVISION | VR_VARIANT_DENSITY_SCALE)) {
        static std::once_flag once;
        std::call_once(once, [&] {
            fprintf(stderr, "volren_amd: note: this scene is rendered by the run-time kernel variant (slower; results unchanged):%s%s\n",
                    (why & VR_VARIANT_ENV_DIVISION) ? " the environment's warp table has thresholds below 2^-76 (vr_get_int env_div_safe = 0);" : "",
                    (why & VR_VARIANT_DENSITY_SCALE) ? " the density scale lies outside [2^-16, 2^24];" : "");
        });
    }
    return variant;
}

// resident workgroups of a kernel instance on the CURRENT device: occupancy query x CU count, cached per (device, instance)
static int resident_blocks(const PathtraceTuning& T, int mode, int variant, bool tf, bool stats) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024;
    static std::mutex mu;
    static std::map<uint32_t, std::pair<int, int>> cache;            // key -> (CUs, workgroups per CU)
    const uint32_t key = ((uint32_t)dev << 8) | ((uint32_t)mode << 5) | ((uint32_t)variant << 2) | (tf ? 2u : 0u) | (stats ? 1u : 0u);
    std::pair<int, int> v;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find(key);
        if (it == cache.end()) {
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            it = cache.emplace(key, std::make_pair(cus, kPtOccupancy[mode][variant](tf, stats))).first;
        }
        v = it->second;
    }
    const int per_cu = T.blocks_per_cu > 0 ? T.blocks_per_cu : v.second;
    return std::min(v.first * per_cu, kMaxWorkgroups * 4 / kWgWaves);
}

size_t pathtrace_workspace_floats() { return kColdMainFloats + (size_t)kMaxWorkgroups * 4u * (size_t)kColdSideWaveFloats; }      // cold state of 4 wavefronts per resident workgroup: main slots, then the side array

void launch_pathtrace(const PathtraceTuning& T, const SceneParams& P, float* fb, float* sample_pool, float* workspace, uint32_t* unit_counter, const int32_t* tiles, int32_t n_tiles,
                      int32_t first_sample, int32_t n_samples, uint32_t* status, hipStream_t stream, bool fast_math, hipEvent_t ev_kernel_begin, hipEvent_t ev_kernel_end) {
    if (n_tiles <= 0 || n_samples <= 0) return;
    SchedParams S;
    for (int i = 0; i < ST_COUNT; ++i) S.thr[i] = T.thr[i];
    LaunchDesc D;
    D.tiles = tiles; D.n_tiles = n_tiles; D.first_sample = first_sample; D.n_samples = n_samples;
    const int variant = pathtrace_variant(P);
    D.spu = std::min(n_samples, samples_per_unit(T, variant));
    const int32_t chunks = (n_samples + D.spu - 1) / D.spu;
    // item indices are 32-bit (WorkUnit::base, the sample pool's slots): RendererHIP::samples_per_launch sizes sub-launches below that; anything else is a caller's bug
    if ((uint64_t)chunks * (uint64_t)n_tiles * 4u * (uint64_t)(D.spu * 64) > 0xFFFFFFFFull)
        throw std::runtime_error("launch_pathtrace: " + std::to_string(n_tiles) + " tiles x " + std::to_string(n_samples) + " samples do not fit the 32-bit item index of one launch");
    D.n_units = (uint32_t)chunks * (uint32_t)n_tiles * 4u;
    D.chunks = (uint32_t)chunks;
    D.seg_len = (D.n_units + kQueueSegments - 1u) / kQueueSegments;
    D.unit_counter = unit_counter;
    S.max_iters = kMaxIdleIters;
    // lanes that must stand at a tentative collision before the collision code runs while others still march (vr_pathtrace.h):
    // measured optimum 24 (smoke.brick +0.5 %, dense +0.7 %, sparse + emission +2...3.5 %), 32 with a transfer function, whose
    // collision code (8 corner taps + LUT) is the dearest (+4.4 %); profiles/r2ab_collide_threshold.txt
    // (round 4: 32 also with an emission grid, whose collision code carries two stochastic taps: c5cloud +0.8 %, c5full +1 %, profiles/r4a_*)
    if (S.thr[ST_COLLIDE] <= 0) S.thr[ST_COLLIDE] = (P.u.use_tf || P.u.has_emission) ? 32 : 24;
    const bool tf = P.u.use_tf != 0, stats = T.stats != nullptr;
    const int mode = fast_math ? 1 : 0;
    const int blocks = resident_blocks(T, mode, variant, tf, stats);
    const uint32_t groups_needed = (D.n_units + (uint32_t)kWgWaves - 1u) / (uint32_t)kWgWaves;      // a wavefront per unit at least
    const dim3 grid((unsigned)std::min<uint32_t>((uint32_t)blocks, groups_needed > 0 ? groups_needed : 1u)), block(256);
    if (P.u.integrator == 2 && P.u.use_tf) {
        const uint64_t items = (uint64_t)D.n_units * (uint64_t)(D.spu * 64);
        hipLaunchKernelGGL(dvr_kernel, dim3((unsigned)((items + 255) / 256)), block, 0, stream, P, sample_pool, D);
    } else if (P.u.integrator == 3) {
        const uint64_t items = (uint64_t)D.n_units * (uint64_t)(D.spu * 64);
        hipLaunchKernelGGL(raymarch_kernel, dim3((unsigned)((items + 255) / 256)), block, 0, stream, P, sample_pool, D);
    } else {
        (void)hipMemsetAsync(unit_counter, 0, kQueueSegments * sizeof(uint32_t), stream);
        if (ev_kernel_begin) (void)hipEventRecord(ev_kernel_begin, stream);
        kPtLaunch[mode][variant](tf, stats, grid.x, stream, &P, sample_pool, workspace, &D, &S, status, T.stats);
        if (ev_kernel_end) (void)hipEventRecord(ev_kernel_end, stream);
    }
    hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)n_tiles), block, 0, stream, sample_pool, fb, tiles, n_tiles,
                       P.u.resolution[0], P.u.resolution[1], first_sample, n_samples, D.spu);
}

// ---------------------------------------------------------------------------------------------------
// environment importance pyramid (env_setup.glsl:18-34; DIMENSION 512, SAMPLES 64: environment.cpp:6-7)
__global__ void __launch_bounds__(256)
impmap_base_kernel(const float* __restrict__ envmap, int32_t env_w, int32_t env_h, int32_t dim, float* __restrict__ out) {
    const int32_t px = blockIdx.x * 16 + (threadIdx.x & 15), py = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (px >= dim || py >= dim) return;
    SceneParams P;                    // only the envmap view is used by env_texture
    P.envmap = envmap; P.env_rgbe = nullptr; P.env_w = env_w; P.env_h = env_h;      // (the pyramid is built from the float map: every other field stays unset)
    const int32_t ns = 8;
    const float inv_samples = 1.0f / (float)(ns * ns);
    const float oss = (float)(dim * ns);
    float importance = 0.0f;
    for (int32_t y = 0; y < ns; ++y)
        for (int32_t x = 0; x < ns; ++x) {
            const float u = ((float)(px * ns) + ((float)x + 0.5f)) / oss;
            const float v = ((float)(py * ns) + ((float)y + 0.5f)) / oss;
            importance += luma(env_texture(P, u, v));
        }
    out[(size_t)py * dim + px] = importance * inv_samples;
}
// glGenerateMipmap on R32F: 2x2 box, ((t00 + t10) + (t01 + t11)) * 0.25
__global__ void __launch_bounds__(256)
impmap_mip_kernel(const float* __restrict__ src, int32_t d, float* __restrict__ dst) {
    const int32_t hd = d >> 1;
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hd * hd) return;
    const int32_t x = i % hd, y = i / hd;
    const float a = src[(size_t)(2 * y) * d + 2 * x], b = src[(size_t)(2 * y) * d + 2 * x + 1];
    const float c = src[(size_t)(2 * y + 1) * d + 2 * x], e = src[(size_t)(2 * y + 1) * d + 2 * x + 1];
    dst[i] = ((a + b) + (c + e)) * 0.25f;
}
// warp table of sample_environment (see vr_trace.h; layout: vr_scene.h env_cdf_index): one thread per 2x2 block of pyramid
// level `mip` = one record of table level k = top - mip
__global__ void __launch_bounds__(256)
env_cdf_kernel(const float* __restrict__ level, int32_t d, int32_t top, int32_t k, float* __restrict__ table, uint32_t* __restrict__ unsafe) {
    const int32_t hd = d >> 1;
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hd * hd) return;
    const int32_t x = i % hd, y = i / hd;
    const float w0 = level[(size_t)(2 * y) * d + 2 * x], w1 = level[(size_t)(2 * y) * d + 2 * x + 1];
    const float w2 = level[(size_t)(2 * y + 1) * d + 2 * x], w3 = level[(size_t)(2 * y + 1) * d + 2 * x + 1];
    const float q0 = w0 + w2, q1 = w1 + w3;
    float* o = table + env_cdf_index(top, k, (uint32_t)x, (uint32_t)y);
    o[0] = q0 / max_(1e-8f, q0 + q1); o[1] = w0 / q0; o[2] = w1 / q1;
    // may sample_environment's quotients use div_core (vr_math.h)?  Every threshold NaN (0 / 0 of an empty block: NaN either way), 0, or in [2^-76, 1]
    bool ok = true;
    for (int j = 0; j < 3; ++j) { const float v = o[j]; ok = ok && (v != v || v == 0.0f || (v >= 1.3234890e-23f && v <= 1.0f)); }
    if (!ok) atomicOr(unsafe, 1u);
    if (k == top) { o[3] = w0; o[4] = w1; o[5] = w2; o[6] = w3; }      // finest level: the texels themselves (pdf of the sampled direction)
}
void launch_build_env_cdf(const float* pyramid, int32_t dim, float* table, uint32_t* unsafe_flag, hipStream_t stream) {
    // levels base-1 .. 0; level m lives at pyramid offset imp_level_offset(dim, m) and has (dim >> m)^2 texels
    int32_t base = 0;
    while ((1 << base) < dim) ++base;
    (void)hipMemsetAsync(table, 0, env_cdf_table_floats(base - 1) * sizeof(float), stream);      // padding words and unused child records
    (void)hipMemsetAsync(unsafe_flag, 0, sizeof(uint32_t), stream);
    for (int32_t mip = base - 1; mip >= 0; --mip) {
        const int32_t d = dim >> mip, n = (d >> 1) * (d >> 1);
        hipLaunchKernelGGL(env_cdf_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pyramid + imp_level_offset(dim, mip), d, base - 1, base - 1 - mip, table, unsafe_flag);
    }
}

void launch_build_impmap(const float* envmap_rgba, int32_t env_w, int32_t env_h, int32_t dim, float* pyramid, hipStream_t stream) {
    const dim3 grid((dim + 15) / 16, (dim + 15) / 16), block(256);
    hipLaunchKernelGGL(impmap_base_kernel, grid, block, 0, stream, envmap_rgba, env_w, env_h, dim, pyramid);
    float* src = pyramid;
    for (int32_t d = dim; d > 1; d >>= 1) {
        float* dst = src + (size_t)d * d;
        const int32_t n = (d >> 1) * (d >> 1);
        hipLaunchKernelGGL(impmap_mip_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, d, dst);
        src = dst;
    }
}

// ---------------------------------------------------------------------------------------------------
// Dense -> brick encoder on the device (voldata's Volume::to_brick_grid, commit() step of the reference:
// src/renderer.cpp:63).  Same rules, same arithmetic and same slot order as the host encoder in grids.cpp, so both
// produce identical device arrays (tests compare checksums):
//   1. encode_range_kernel : per brick, (min, max) over the brick dilated by 2 voxels, rounded outwards to fp16;
//                            flag = the brick's voxels matter (max != min)
//   2. encode_brick_kernel : per brick, BrickRec + 512 quantised voxels straight into its block of the brick-linear atlas (5 lines of range + 120 voxels: vr_scene.h)
//   3. range_mip_kernel    : (min of mins, max of maxes) over 2x2x2 children, three levels
__global__ void __launch_bounds__(256)
encode_range_kernel(const float* __restrict__ dense, int32_t nx, int32_t ny, int32_t nz, int32_t nbx, int32_t nby, int32_t nbz,
                    uint32_t* __restrict__ range, uint32_t* __restrict__ flag) {
    // one wavefront per brick: 12^3 = 1728 taps, 27 per lane
    const int32_t brick = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (brick >= nbx * nby * nbz) return;
    const int32_t bx = brick % nbx, by = (brick / nbx) % nby, bz = brick / (nbx * nby);
    const int32_t x0 = bx * 8 - 2, y0 = by * 8 - 2, z0 = bz * 8 - 2;
    float lo = inf_(), hi = -inf_();
    if (x0 >= nx || y0 >= ny || z0 >= nz) { lo = hi = 0.0f; }
    else
        for (int32_t i = lane; i < 1728; i += 64) {
            const int32_t x = x0 + i % 12, y = y0 + (i / 12) % 12, z = z0 + i / 144;
            float v = 0.0f;
            if (x >= 0 && y >= 0 && z >= 0 && x < nx && y < ny && z < nz) v = dense[((size_t)z * ny + y) * nx + x];
            lo = v < lo ? v : lo; hi = v > hi ? v : hi;
        }
    for (int32_t o = 32; o > 0; o >>= 1) {
        const float l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if (lane == 0) {
        const uint32_t hlo = float_to_half_down(lo), hhi = float_to_half_up(hi);
        range[brick] = hlo | (hhi << 16);
        flag[brick] = half2float(hhi) != half2float(hlo) ? 1u : 0u;
    }
}
__global__ void __launch_bounds__(64)
encode_brick_kernel(const float* __restrict__ dense, int32_t nx, int32_t ny, int32_t nz, int32_t nbx, int32_t nby,
                    const uint32_t* __restrict__ range, const uint32_t* __restrict__ flag,
                    BrickRec* __restrict__ recs, float* __restrict__ rng, uint8_t* __restrict__ atlas) {
    const int32_t brick = blockIdx.x, lane = threadIdx.x;
    const int32_t bx = brick % nbx, by = (brick / nbx) % nby, bz = brick / (nbx * nby);
    const uint32_t rg = range[brick];
    const float lo = half2float(rg & 0xFFFFu), hi = half2float(rg >> 16);
    const bool alloc = flag[brick] != 0u;                    // a brick whose range is one value keeps its zeroed block
    const size_t idx = ((size_t)bz * nby + by) * nbx + bx;            // brick-linear atlas: block index = record index = linear brick index
    if (lane == 0) { BrickRec r; r.slot = (uint32_t)idx; r.rmin = lo; r.rdiff = hi - lo; r.range = rg; recs[idx] = r; rng[2 * idx] = r.rmin; rng[2 * idx + 1] = r.rdiff; }
    uint8_t* dst = atlas + idx * (size_t)kBrickBlockBytes;
    if (VR_BRICK_HEADERS && lane < 5) { float* h = reinterpret_cast<float*>(dst + lane * 128); h[0] = lo; h[1] = hi - lo; }      // every line of every brick carries the range
    if (!alloc) return;
    const float inv = 255.0f / (hi - lo);
    for (int32_t i = lane; i < 512; i += 64) {
        const int32_t x = bx * 8 + (i & 7), y = by * 8 + ((i >> 3) & 7), z = bz * 8 + (i >> 6);
        float v = 0.0f;
        if (x < nx && y < ny && z < nz) v = dense[((size_t)z * ny + y) * nx + x];
        float qv = floor_((v - lo) * inv + 0.5f);
        qv = qv < 0.0f ? 0.0f : (qv > 255.0f ? 255.0f : qv);
        dst[brick_voxel_byte((uint32_t)i)] = (uint8_t)qv;
    }
}
__global__ void __launch_bounds__(256)
range_mip_kernel(const uint32_t* __restrict__ src, int32_t sx, int32_t sy, int32_t sz, uint32_t* __restrict__ dst, int32_t dx, int32_t dy, int32_t dz) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dx * dy * dz) return;
    const int32_t x = i % dx, y = (i / dx) % dy, z = i / (dx * dy);
    float lo = inf_(), hi = -inf_(); uint32_t hlo = 0u, hhi = 0u;
    for (int32_t c = 0; c < 8; ++c) {
        const int32_t cx = 2 * x + (c & 1), cy = 2 * y + ((c >> 1) & 1), cz = 2 * z + (c >> 2);
        if (cx >= sx || cy >= sy || cz >= sz) continue;
        const uint32_t rg = src[((size_t)cz * sy + cy) * sx + cx];
        const float l = half2float(rg & 0xFFFFu), h = half2float(rg >> 16);
        if (l < lo) { lo = l; hlo = rg & 0xFFFFu; }
        if (h > hi) { hi = h; hhi = rg >> 16; }
    }
    dst[i] = hlo | (hhi << 16);
}

void launch_encode_ranges(const float* dense, const int32_t dim[3], const int32_t nb[3], uint32_t* range, uint32_t* flag, hipStream_t stream) {
    const int32_t n = nb[0] * nb[1] * nb[2];
    hipLaunchKernelGGL(encode_range_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, dense, dim[0], dim[1], dim[2], nb[0], nb[1], nb[2], range, flag);
}
void launch_encode_bricks(const float* dense, const int32_t dim[3], const int32_t nb[3], const uint32_t* range, const uint32_t* flag,
                          BrickRec* recs, float* rng, uint8_t* atlas, hipStream_t stream) {
    const int32_t n = nb[0] * nb[1] * nb[2];
    hipLaunchKernelGGL(encode_brick_kernel, dim3(n), dim3(64), 0, stream, dense, dim[0], dim[1], dim[2], nb[0], nb[1], range, flag, recs, rng, atlas);
}
void launch_range_mip(const uint32_t* src, const int32_t sdim[3], uint32_t* dst, const int32_t ddim[3], hipStream_t stream) {
    const int32_t n = ddim[0] * ddim[1] * ddim[2];
    hipLaunchKernelGGL(range_mip_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, sdim[0], sdim[1], sdim[2], dst, ddim[0], ddim[1], ddim[2]);
}

// paired atlas (vr_scene.h): the voxels of a density brick and of the emission brick at the same index, interleaved, with both decode ranges at the head of every line
__global__ void __launch_bounds__(64)
pair_atlas_kernel(const uint8_t* __restrict__ atlas_d, const uint8_t* __restrict__ atlas_e, uint8_t* __restrict__ out) {
    const size_t rec = blockIdx.x;
    const uint8_t* bd = atlas_d + rec * (size_t)kBrickBlockBytes;
    const uint8_t* be = atlas_e + rec * (size_t)kBrickBlockBytes;
    uint8_t* dst = out + rec * (size_t)kPairBlockBytes;
    const int32_t lane = threadIdx.x;
    if (lane < 10) {          // every line's header: (rmin, rdiff) of both bricks = the first 8 bytes of any line of their own blocks
        const float* hd = reinterpret_cast<const float*>(bd);
        const float* he = reinterpret_cast<const float*>(be);
        float* h = reinterpret_cast<float*>(dst + lane * 128);
        h[0] = hd[0]; h[1] = hd[1]; h[2] = he[0]; h[3] = he[1];
    }
    for (int32_t i = lane; i < 512; i += 64) {
        dst[pair_voxel_byte((uint32_t)i, 0u)] = bd[brick_voxel_byte((uint32_t)i)];
        dst[pair_voxel_byte((uint32_t)i, 1u)] = be[brick_voxel_byte((uint32_t)i)];
    }
}
void launch_pair_atlas(const uint8_t* atlas_d, const uint8_t* atlas_e, uint8_t* out, size_t n_records, hipStream_t stream) {
    if (n_records == 0) return;
    hipLaunchKernelGGL(pair_atlas_kernel, dim3((unsigned)n_records), dim3(64), 0, stream, atlas_d, atlas_e, out);
}

// decoded float atlas for transfer-function renders: out[i*512 + v] = rmin_i + unorm8(atlas[i*512 + v]) * rdiff_i (common.glsl:268-275)
__global__ void __launch_bounds__(256)
decode_atlas_kernel(const float* __restrict__ rng, const uint8_t* __restrict__ atlas, float* __restrict__ out, size_t n_voxels) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_voxels) return;
    const size_t cell = i >> 9;
    out[i] = rng[2 * cell] + unorm8(atlas[cell * (size_t)kBrickBlockBytes + brick_voxel_byte((uint32_t)(i & 511u))]) * rng[2 * cell + 1];
}
void launch_decode_atlas(const float* rng, const uint8_t* atlas, float* out, size_t n_records, hipStream_t stream) {
    const size_t n = n_records * 512u;
    if (n == 0) return;
    hipLaunchKernelGGL(decode_atlas_kernel, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, stream, rng, atlas, out, n);
}

// ---------------------------------------------------------------------------------------------------
// effective majorant of every cell of every level, written in the padded power-of-two layout that majorant_at indexes
// (vr_scene.h); cells beyond a level's real extent -- and levels the grid does not have -- hold 0
struct MajorantLayout { int32_t nb[3], mip_off[4], n_mips, mshift[3], blocked; };
__global__ void __launch_bounds__(256)
majorant_kernel(const SceneParams P, const uint32_t* __restrict__ range_words, const MajorantLayout L, uint32_t n_padded, float* __restrict__ out, uint16_t* __restrict__ out16) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i > n_padded) return;                                 // cell n_padded: "outside the grid" (vr_scene.h majorant_table_cells)
    const uint32_t k = (uint32_t)(L.mshift[0] + L.mshift[1] + L.mshift[2]);
    uint32_t mip = 0u;
    while (mip < 3u && i >= majorant_level_offset(k, mip + 1u)) ++mip;
    const uint32_t j = i - majorant_level_offset(k, mip);
    const uint32_t sx = (uint32_t)L.mshift[0] - mip, sy = (uint32_t)L.mshift[1] - mip;
    uint32_t cx, cy, cz;                                   // invert majorant_cell_index: which cell lives at position j of this level
    if (L.blocked && mip <= 1u) {
        const uint32_t blk = j >> 6, in = j & 63u;
        cx = ((blk & ((1u << (sx - 2u)) - 1u)) << 2) | (in & 3u);
        cy = (((blk >> (sx - 2u)) & ((1u << (sy - 2u)) - 1u)) << 2) | ((in >> 2) & 3u);
        cz = ((blk >> (sx + sy - 4u)) << 2) | (in >> 4);
    } else { cx = j & ((1u << sx) - 1u); cy = (j >> sx) & ((1u << sy) - 1u); cz = j >> (sx + sy); }
    const uint32_t rnd = (1u << mip) - 1u;
    const uint32_t dx = ((uint32_t)L.nb[0] + rnd) >> mip, dy = ((uint32_t)L.nb[1] + rnd) >> mip, dz = ((uint32_t)L.nb[2] + rnd) >> mip;
    // a cell beyond the level's real extent, a level the grid does not have and the table's last cell read what the reference's out-of-range texelFetch
    // returns, 0, and go through the same arithmetic: density_scale * 0, TF-remapped when a LUT is bound (common.glsl:278-281, 425)
    uint32_t h = 0u;
    if (i < n_padded && (int32_t)mip <= L.n_mips && cx < dx && cy < dy && cz < dz)
        h = range_words[(uint32_t)L.mip_off[mip] + (cz * dy + cy) * dx + cx] >> 16;
    float m = P.u.vol_density_scale * half2float(h);
    if (P.u.use_tf) {
        float rgba[4];
        tf_lookup(P, m * P.u.vol_inv_majorant, rgba);
        m = P.u.vol_majorant * rgba[3];
    }
    out[i] = m;
    out16[i] = (uint16_t)h;
}
void launch_majorants(const SceneParams& P, const uint32_t* range_words_all_mips, const int32_t nb[3], const int32_t mip_off[4], int32_t n_mips,
                      const int32_t mshift[3], float* out_padded, uint16_t* out16_padded, hipStream_t stream) {
    MajorantLayout L;
    L.blocked = P.density.maj_blocked;
    for (int i = 0; i < 3; ++i) { L.nb[i] = nb[i]; L.mshift[i] = mshift[i]; }
    for (int i = 0; i < 4; ++i) L.mip_off[i] = mip_off[i];
    L.n_mips = n_mips;
    const uint32_t n = (uint32_t)majorant_padded_cells((uint32_t)(mshift[0] + mshift[1] + mshift[2]));
    hipLaunchKernelGGL(majorant_kernel, dim3((n + 256u) / 256u), dim3(256), 0, stream, P, range_words_all_mips, L, n, out_padded, out16_padded);      // n + 1 cells
}

// ---------------------------------------------------------------------------------------------------
// tonemap.glsl:13-36
__device__ __forceinline__ float hable(float x) {
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
__global__ void __launch_bounds__(256)
tonemap_kernel(float* __restrict__ fb, int32_t n, float exposure, float inv_gamma) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4* p = reinterpret_cast<float4*>(fb) + i;
    float4 c = *p;
    const float hw = hable(11.2f);
    c.x = sanitize(pow_(hable(exposure * c.x) / hw, inv_gamma));
    c.y = sanitize(pow_(hable(exposure * c.y) / hw, inv_gamma));
    c.z = sanitize(pow_(hable(exposure * c.z) / hw, inv_gamma));
    c.w = sanitize(c.w);
    *p = c;
}
void launch_tonemap(float* fb, int32_t w, int32_t h, float exposure, float gamma, hipStream_t stream) {
    const int32_t n = w * h;
    if (n <= 0) return;
    hipLaunchKernelGGL(tonemap_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fb, n, exposure, 1.0f / gamma);
}

// ---------------------------------------------------------------------------------------------------
// tile <-> frame copies for the sharded framebuffer
__global__ void __launch_bounds__(256)
pack_tiles_kernel(const float* __restrict__ fb, int32_t w, int32_t h, const int32_t* __restrict__ tiles, float* __restrict__ packed) {
    const int32_t tiles_x = (w + 15) >> 4;
    const int32_t tile = tiles[blockIdx.x];
    const int32_t px = (tile % tiles_x) * 16 + (threadIdx.x & 15), py = (tile / tiles_x) * 16 + (threadIdx.x >> 4);
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (px < w && py < h) c = reinterpret_cast<const float4*>(fb)[(size_t)py * w + px];
    reinterpret_cast<float4*>(packed)[(size_t)blockIdx.x * 256 + threadIdx.x] = c;
}
__global__ void __launch_bounds__(256)
unpack_tiles_kernel(const float* __restrict__ packed, const int32_t* __restrict__ tiles, float* __restrict__ fb, int32_t w, int32_t h) {
    const int32_t tiles_x = (w + 15) >> 4;
    const int32_t tile = tiles[blockIdx.x];
    if (tile < 0) return;             // padding entry
    const int32_t px = (tile % tiles_x) * 16 + (threadIdx.x & 15), py = (tile / tiles_x) * 16 + (threadIdx.x >> 4);
    if (px < w && py < h)
        reinterpret_cast<float4*>(fb)[(size_t)py * w + px] = reinterpret_cast<const float4*>(packed)[(size_t)blockIdx.x * 256 + threadIdx.x];
}
void launch_pack_tiles(const float* fb, int32_t w, int32_t h, const int32_t* tiles, int32_t n_tiles, float* packed, hipStream_t stream) {
    if (n_tiles <= 0) return;
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(n_tiles), dim3(256), 0, stream, fb, w, h, tiles, packed);
}
void launch_unpack_tiles(const float* packed, const int32_t* tiles, int32_t n_tiles, float* fb, int32_t w, int32_t h, hipStream_t stream) {
    if (n_tiles <= 0) return;
    hipLaunchKernelGGL(unpack_tiles_kernel, dim3(n_tiles), dim3(256), 0, stream, packed, tiles, fb, w, h);
}

// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
math_probe_kernel(int32_t fn, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int32_t n) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    float r;
    switch (fn) {
    case 0: r = log_(x); break;
    case 1: r = sin_(x); break;
    case 2: r = cos_(x); break;
    case 3: r = tan_(x); break;
    case 4: r = acos_(x); break;
    case 5: r = atan2_(x, y); break;
    case 6: r = exp_(x); break;
    case 7: r = pow_(x, y); break;
    case 8: r = asin_(x); break;
    case 9: r = x / y; break;
    case 10: r = sqrt_(x); break;
    case 11: r = fma_(x, y, x); break;
    case 12: r = (float)((uint32_t)x & 255u) / 255.0f; break;
    case 13: { float s, c; sincos_(x, s, c); r = s * y + c; break; }
    case 14: r = x * y + x; break;     // must stay two roundings (-ffp-contract=off)
    case 15: r = half2float(f2u(x)); break;     // bit pattern of x: low 16 bits = binary16
    case 16: r = rcp_exact(x); break;
    case 17: { const v3 q = rcp3_exact(v3{ x, y, x }); r = q.y; break; }      // the three-at-once form: y's reciprocal, range test shared with x
    default: r = nan_(); break;
    }
    out[i] = r;
}
void launch_math_probe(int32_t fn, const float* a, const float* b, float* out, int32_t n, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fn, a, b, out, n);
}

}  // namespace vr
