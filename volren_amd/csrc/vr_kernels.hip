// vr_kernels.hip -- gfx950 kernels of the volume path tracer.
//
// pathtrace_kernel: one wavefront = one 8x8 pixel tile, one workgroup = 4 wavefronts = the reference's 16x16
// work group (pathtracer_brick.glsl:3).  Each lane owns one pixel for the whole launch and runs ALL requested
// samples for it (the reference issues one dispatch per sample, renderer.cpp:138-140).  The wavefront is driven
// by a small scheduler: every iteration it counts lanes per state with ballots (scalar registers), then
// executes the code of those states that enough lanes are waiting in -- see vr_trace.h for the state bodies.
// No LDS, no cross-lane data exchange: lanes only vote.  MFMA is not used (there is no dense contraction).
#include <hip/hip_runtime.h>

#include "vr_device.h"
#include "vr_trace.h"

namespace vr {

struct SchedParams {
    int32_t thr[ST_COUNT];     // minimum number of lanes that must wait in a state before its code runs
    uint32_t max_iters;        // watchdog: scheduler iterations per wavefront
};

template <bool USE_TF>
__global__ void __launch_bounds__(256)
pathtrace_kernel(const SceneParams P, float* __restrict__ fb, const int32_t* __restrict__ tiles,
                 int32_t first_sample, int32_t n_samples, const SchedParams S, uint32_t* __restrict__ status) {
    const int32_t W = P.u.resolution[0], H = P.u.resolution[1];
    const int32_t tiles_x = (W + 15) >> 4;
    const int32_t tile = tiles ? tiles[blockIdx.x] : (int32_t)blockIdx.x;
    const int32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const int32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int32_t px = tx * 16 + ((wave & 1) << 3) + (lane & 7);
    const int32_t py = ty * 16 + ((wave >> 1) << 3) + (lane >> 3);
    const bool valid = px < W && py < H;

    Lane l;
    float texel[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    float4* fb4 = reinterpret_cast<float4*>(fb);
    if (valid && first_sample > 1) {
        const float4 c = fb4[(size_t)py * W + px];
        texel[0] = c.x; texel[1] = c.y; texel[2] = c.z; texel[3] = c.w;
    }
    lane_init(l, px, py, first_sample, n_samples, texel);
    if (!valid) l.state = ST_DONE;

    uint32_t iters = 0u;
    for (;;) {
        const int32_t st = l.state;
        const int32_t n_new = __popcll(__ballot(st == ST_NEW));
        const int32_t n_begin = __popcll(__ballot(st == ST_BEGIN));
        const int32_t n_march = __popcll(__ballot(st == ST_MARCH));
        const int32_t n_collide = __popcll(__ballot(st == ST_COLLIDE));
        const int32_t n_nee = __popcll(__ballot(st == ST_NEE));
        const int32_t n_post = __popcll(__ballot(st == ST_POSTNEE));
        const int32_t n_escape = __popcll(__ballot(st == ST_ESCAPE));
        if ((n_new | n_begin | n_march | n_collide | n_nee | n_post | n_escape) == 0) break;
        if (++iters > S.max_iters) {
            if (lane == 0) atomicOr(status, 1u);
            break;
        }
        // most populated state always runs (progress guarantee); the others when they pass their threshold
        int32_t best = ST_NEW, nbest = n_new;
        if (n_begin > nbest) { best = ST_BEGIN; nbest = n_begin; }
        if (n_march > nbest) { best = ST_MARCH; nbest = n_march; }
        if (n_collide > nbest) { best = ST_COLLIDE; nbest = n_collide; }
        if (n_nee > nbest) { best = ST_NEE; nbest = n_nee; }
        if (n_post > nbest) { best = ST_POSTNEE; nbest = n_post; }
        if (n_escape > nbest) { best = ST_ESCAPE; nbest = n_escape; }

        if (n_new >= S.thr[ST_NEW] || best == ST_NEW) { if (l.state == ST_NEW) do_new(l, P); }
        if (n_begin >= S.thr[ST_BEGIN] || best == ST_BEGIN) { if (l.state == ST_BEGIN) do_begin(l, P); }
        if (n_march >= S.thr[ST_MARCH] || best == ST_MARCH) { if (l.state == ST_MARCH) do_march(l, P); }
        if (n_collide >= S.thr[ST_COLLIDE] || best == ST_COLLIDE) { if (l.state == ST_COLLIDE) do_collide<USE_TF>(l, P); }
        if (n_nee >= S.thr[ST_NEE] || best == ST_NEE) { if (l.state == ST_NEE) do_nee(l, P); }
        if (n_post >= S.thr[ST_POSTNEE] || best == ST_POSTNEE) { if (l.state == ST_POSTNEE) do_postnee(l, P); }
        if (n_escape >= S.thr[ST_ESCAPE] || best == ST_ESCAPE) { if (l.state == ST_ESCAPE) do_escape(l, P); }
    }
    if (valid) fb4[(size_t)py * W + px] = make_float4(l.acc[0], l.acc[1], l.acc[2], l.acc[3]);
}

static SchedParams g_sched = { { 24, 16, 8, 16, 16, 16, 24, 0 }, 0u };

void set_sched_thresholds(const int32_t thr[ST_COUNT]) {
    for (int i = 0; i < ST_COUNT; ++i) g_sched.thr[i] = thr[i];
}

void launch_pathtrace(const SceneParams& P, float* fb, const int32_t* tiles, int32_t n_tiles,
                      int32_t first_sample, int32_t n_samples, uint32_t* status, hipStream_t stream) {
    if (n_tiles <= 0 || n_samples <= 0) return;
    SchedParams S = g_sched;
    // a sample needs a few hundred scheduler iterations at the very worst (bounces x steps); generous cap
    const uint64_t cap = (uint64_t)n_samples * 200000ull + 1000000ull;
    S.max_iters = cap > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)cap;
    const dim3 grid((unsigned)n_tiles), block(256);
    if (P.u.use_tf)
        hipLaunchKernelGGL(pathtrace_kernel<true>, grid, block, 0, stream, P, fb, tiles, first_sample, n_samples, S, status);
    else
        hipLaunchKernelGGL(pathtrace_kernel<false>, grid, block, 0, stream, P, fb, tiles, first_sample, n_samples, S, status);
}

// ---------------------------------------------------------------------------------------------------
// environment importance pyramid (env_setup.glsl:18-34; DIMENSION 512, SAMPLES 64: environment.cpp:6-7)
__global__ void __launch_bounds__(256)
impmap_base_kernel(const float* __restrict__ envmap, int32_t env_w, int32_t env_h, int32_t dim, float* __restrict__ out) {
    const int32_t px = blockIdx.x * 16 + (threadIdx.x & 15), py = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (px >= dim || py >= dim) return;
    SceneParams P;                    // only the envmap view is used by env_texture
    P.envmap = envmap; P.env_w = env_w; P.env_h = env_h;
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
__global__ void __launch_bounds__(256)
majorant_kernel(const SceneParams P, const uint32_t* __restrict__ range_words, int32_t n, float* __restrict__ out) {
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float m = P.u.vol_density_scale * half2float(range_words[i] >> 16);
    if (P.u.use_tf) {
        float rgba[4];
        tf_lookup(P, m * P.u.vol_inv_majorant, rgba);
        m = P.u.vol_majorant * rgba[3];
    }
    out[i] = m;
}
void launch_majorants(const SceneParams& P, const uint32_t* range_words_all_mips, int32_t n_cells, float* out, hipStream_t stream) {
    if (n_cells <= 0) return;
    hipLaunchKernelGGL(majorant_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, stream, P, range_words_all_mips, n_cells, out);
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
    default: r = nan_(); break;
    }
    out[i] = r;
}
void launch_math_probe(int32_t fn, const float* a, const float* b, float* out, int32_t n, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fn, a, b, out, n);
}

}  // namespace vr
