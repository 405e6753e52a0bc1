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

#include "vr_device.h"
#include "vr_trace.h"

namespace vr {

struct SchedParams {
    int32_t thr[ST_COUNT];     // minimum number of lanes that must wait in a state before its code runs
    uint32_t max_iters;        // watchdog: scheduler iterations per wavefront
};

// Persistent wavefronts.  The frame's work is cut into units = one 8x8 pixel tile x `spu` consecutive samples
// (64*spu items); unit u = ((chunk * n_tiles + tile_slot) * 4 + sub_tile) and its items occupy slots
// [u*64*spu, (u+1)*64*spu) of the sample buffer.  Every wavefront pulls units from a global queue and refills
// idle lanes item by item, WITHOUT waiting for its other lanes to finish: the only drain is at the end of the launch.
// The queue is XCD-aware: queue position j enumerates the units tile-major (all sample chunks of a sub-tile are adjacent) and
// the positions are cut into 8 contiguous segments, one per XCD (workgroups are dealt round-robin to the 8 XCDs, so
// blockIdx.x & 7 names the XCD): the waves of one XCD -- which share one L2 -- work on one band of tile rows; a wave whose
// segment is empty takes units from the next one.
constexpr uint32_t kQueueSegments = 8u;
struct LaunchDesc {
    const int32_t* tiles;     // 16x16 tile ids (raster, row 0 = bottom) or nullptr = all tiles
    int32_t n_tiles, first_sample, n_samples, spu;
    uint32_t n_units, chunks, seg_len;
    uint32_t* unit_counter;   // kQueueSegments counters, zeroed before the launch
};

__device__ __forceinline__ WorkUnit make_unit(const LaunchDesc& D, int32_t W, uint32_t j, float* sbuf) {
    const uint32_t rem = j / D.chunks, chunk = j - rem * D.chunks;        // queue position -> (tile slot, sub-tile), sample chunk
    const uint32_t u = chunk * ((uint32_t)D.n_tiles * 4u) + rem;
    const uint32_t slot = rem >> 2, sub = rem & 3u;
    const int32_t tiles_x = (W + 15) >> 4;
    const int32_t tile = D.tiles ? D.tiles[slot] : (int32_t)slot;
    const int32_t tx = tile % tiles_x, ty = tile / tiles_x;
    WorkUnit wu;
    wu.px0 = tx * 16 + (int32_t)((sub & 1u) << 3);
    wu.py0 = ty * 16 + (int32_t)((sub >> 1) << 3);
    wu.first_sample = D.first_sample + (int32_t)chunk * D.spu;
    wu.n_items = min(D.spu, D.n_samples - (int32_t)chunk * D.spu) * 64;
    wu.base = u * (uint32_t)(D.spu * 64);
    wu.out = sbuf;
    return wu;
}

#ifndef VR_WAVES_PER_SIMD
#define VR_WAVES_PER_SIMD 4
#endif
#ifndef VR_BATCH_REGS
#define VR_BATCH_REGS 1
#endif

// ---------------------------------------------------------------------------------------------------
// Wave-private path pool.
//
// A wavefront owns NSLOT path slots, more than it has lanes.  The 64 lanes hold, in registers, the hot state of the
// paths that are currently marching; every other path of the pool is parked: its hot state (NHOT dwords) sits in LDS
// and its slot id in one of the wave's LDS stacks -- READY (may march), NEE / POSTNEE / ESCAPE (wait for that event),
// FREE.  Cold path state lives in global memory (one 128-byte line per slot, L2 / Infinity Cache resident); only the
// events touch it.
//   * a lane whose path reaches an event parks it (ds_write2 pairs + a stack push) and immediately resumes a READY path,
//     so the march/collide code runs with most lanes holding a path;
//   * an event's code runs when a (nearly) full-width batch of parked paths has piled up, or -- when the wave runs dry --
//     for its largest batch: lane i loads parked path i, runs the unchanged per-path code of vr_trace.h, stores it and
//     routes the slot to the stack of its new state.  The lanes double as batch workers; the batch path lives in its own
//     register set so the marching path stays put (VR_BATCH_REGS=0 swaps it through its LDS slot instead).
// Everything is wave-synchronous (ballots, mbcnt ranks, scalar counters): no atomics, no barriers, no spinning; the only
// global atomic is the work-queue head.  Which lane runs which path never changes a result.
#ifndef VR_NSLOT
#define VR_NSLOT 152
#endif
constexpr int32_t NSLOT = VR_NSLOT;        // <= 256 (slot ids are bytes)

enum PoolStack : int32_t { Q_READY = 0, Q_NEE = 1, Q_POST = 2, Q_ESC = 3, Q_FREE = 4, Q_COUNT = 5 };

constexpr int32_t HOT_STRIDE = 15;     // dwords per slot in LDS (= the parked fields): odd, so that lanes with different slots spread over the banks
struct HotStore {                      // [slot][field]: a path's 15 parked dwords are adjacent (ds_read2/ds_write2 pairs)
    uint32_t* base;
    // mip (a multiple of 1/4 in [0,3]) rides in the flag word
    __device__ __forceinline__ void save(const Hot& h, int32_t slot) const {
        uint32_t* p = base + slot * HOT_STRIDE;
        p[0] = h.seed;
        p[1] = f2u(h.ipos.x); p[2] = f2u(h.ipos.y); p[3] = f2u(h.ipos.z);
        p[4] = f2u(h.idir.x); p[5] = f2u(h.idir.y); p[6] = f2u(h.idir.z);
        p[7] = f2u(h.t); p[8] = f2u(h.far); p[9] = f2u(h.tau);
        p[10] = f2u(h.Tr);
        p[11] = (uint32_t)h.state | ((uint32_t)h.shadow << 8) | ((uint32_t)h.mipq << 16);
        p[12] = f2u(h.ri.x); p[13] = f2u(h.ri.y); p[14] = f2u(h.ri.z);
    }
    __device__ __forceinline__ void load(Hot& h, int32_t slot) const {
        const uint32_t* p = base + slot * HOT_STRIDE;
        h.seed = p[0];
        h.ipos = v3{ u2f(p[1]), u2f(p[2]), u2f(p[3]) };
        h.idir = v3{ u2f(p[4]), u2f(p[5]), u2f(p[6]) };
        h.ri = v3{ u2f(p[12]), u2f(p[13]), u2f(p[14]) };
        h.t = u2f(p[7]); h.far = u2f(p[8]); h.tau = u2f(p[9]);
        h.Tr = u2f(p[10]);
        const uint32_t f = p[11];
        h.state = (int32_t)(f & 0xFFu); h.shadow = (int32_t)((f >> 8) & 0xFFu);
        h.mipq = (int32_t)(f >> 16);
        h.majorant = 0.0f;
    }
};
struct ColdGlobal {                    // one 128-byte line per path slot in this wavefront's slice of the workspace
    float* base;
    __device__ __forceinline__ float ld(int32_t f) const { return static_cast<const float*>(__builtin_assume_aligned(base, 128))[f]; }
    __device__ __forceinline__ void st(int32_t f, float v) { static_cast<float*>(__builtin_assume_aligned(base, 128))[f] = v; }
};

__device__ __forceinline__ int32_t popc(uint64_t mask) { return (int32_t)__popcll(mask); }     // int: min(long long, int) would go through double
__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <bool USE_TF, bool STATS>
__global__ void __launch_bounds__(256, VR_WAVES_PER_SIMD)
pathtrace_kernel(const SceneParams P, float* __restrict__ sbuf, float* __restrict__ cold_ws, const LaunchDesc D, const SchedParams S,
                 uint32_t* __restrict__ status, unsigned long long* __restrict__ stats) {
    const int32_t W = P.u.resolution[0];
    const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

    __shared__ uint8_t lds_q[4 * Q_COUNT * NSLOT];
    uint8_t* const q = lds_q + wave * (Q_COUNT * NSLOT);
    // per-wavefront slice of the workspace: the cold fields of its NSLOT paths
    float* const cold_base = cold_ws + (size_t)(blockIdx.x * 4u + (uint32_t)wave) * (size_t)(C_STRIDE * NSLOT);
    __shared__ uint32_t lds_hot[4 * HOT_STRIDE * NSLOT];
    const HotStore hs{ lds_hot + wave * (HOT_STRIDE * NSLOT) };

    const int32_t pool = (S.thr[ST_BEGIN] > 0 && S.thr[ST_BEGIN] < NSLOT) ? S.thr[ST_BEGIN] : NSLOT;     // slots in use (diagnostic cap)
    int32_t cnt_ready = 0, cnt_nee = 0, cnt_post = 0, cnt_esc = 0, cnt_free = pool;     // stack heights (wave-uniform)
    for (int32_t i = lane; i < pool; i += 64) q[Q_FREE * NSLOT + i] = (uint8_t)i;
    __builtin_amdgcn_wave_barrier();

    WorkUnit wu;
    wu.px0 = wu.py0 = 0; wu.first_sample = 1; wu.n_items = 0; wu.base = 0u; wu.out = sbuf;
    uint32_t cursor = 0u;             // next item of the current unit (wave-uniform)
    bool exhausted = false;           // the global queue has no more units
    uint32_t seg_tries = 0u;          // queue segments this wavefront has found empty (wave-uniform)

    Hot l;
    hot_init(l);
    int32_t slot = -1;                // path held in this lane's registers (-1: none)

    uint32_t iters = 0u;
    unsigned long long t_last = __builtin_readcyclecounter(), t_elapsed = 0ull;
    uint32_t st_exec[ST_DONE] = { 0, 0, 0, 0, 0, 0, 0 }, st_lanes[ST_DONE] = { 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long st_cyc[ST_DONE] = { 0, 0, 0, 0, 0, 0, 0 }, t_blk = 0ull, t_start = STATS ? __builtin_readcyclecounter() : 0ull;
    unsigned long long occ[6] = { 0, 0, 0, 0, 0, 0 };      // summed per iteration: marching lanes, READY, NEE, POSTNEE, ESCAPE, FREE
#define VR_STAT(ST, N) do { if (STATS) { st_exec[ST] += 1u; st_lanes[ST] += (uint32_t)(N); t_blk = __builtin_readcyclecounter(); } } while (0)
#define VR_STAT_END(ST) do { if (STATS) { st_cyc[ST] += __builtin_readcyclecounter() - t_blk; } } while (0)
// push the slots of all lanes where COND holds onto stack QI (wave-synchronous)
#define VR_PUSH(QI, CNT, COND, SLOTV) do { \
        const uint64_t m_ = __ballot(COND); \
        if (m_) { if (COND) q[(QI) * NSLOT + (CNT) + (int32_t)lane_rank(m_)] = (uint8_t)(SLOTV); (CNT) += popc(m_); } \
    } while (0)

// route the batch paths to the stack of their new state; an impossible state is reported and the slot recycled
#define VR_ROUTE_ST(BS, STV) do { \
        const bool v_ = (BS) >= 0; \
        const int32_t s_ = (STV); \
        VR_PUSH(Q_READY, cnt_ready, v_ && (s_ == ST_MARCH || s_ == ST_COLLIDE), BS); \
        VR_PUSH(Q_NEE, cnt_nee, v_ && s_ == ST_NEE, BS); \
        VR_PUSH(Q_POST, cnt_post, v_ && s_ == ST_POSTNEE, BS); \
        VR_PUSH(Q_ESC, cnt_esc, v_ && s_ == ST_ESCAPE, BS); \
        const bool lost_ = v_ && (s_ < ST_NEW || s_ > ST_ESCAPE || s_ == ST_BEGIN); \
        if (__ballot(lost_)) { if (lost_) atomicOr(status, 2u); } \
        VR_PUSH(Q_FREE, cnt_free, v_ && (s_ == ST_NEW || lost_), BS); \
    } while (0)
#define VR_ROUTE(BS) VR_ROUTE_ST(BS, l.state)
#define VR_ROUTE_B(BS) VR_ROUTE_ST(BS, b.state)

    for (;;) {
        // watchdog: a wavefront's share of a launch is tens of milliseconds; give up (and report) after S.max_iters
        // scheduler iterations or ~8 s of shader clock, whichever comes first -- a kernel must never hang the GPU
        bool give_up = ++iters > S.max_iters;
        if ((iters & 1023u) == 0u) {
            // elapsed shader-clock time, summed over 1024-iteration windows.  A window that appears to take more than 2^34 ticks (or
            // a negative time) is a counter discontinuity -- a wavefront that was saved and restored on another XCD when several
            // processes time-share the GPU reads a different counter -- and is not counted.
            const unsigned long long now = __builtin_readcyclecounter(), d = now - t_last;
            t_last = now;
            if (d < (1ull << 34)) t_elapsed += d;
            give_up = give_up || t_elapsed > 20000000000ull;
        }
        if (give_up) {
            if (lane == 0) atomicOr(status, 1u);
            break;
        }
        // (1) idle lanes resume READY paths
        {
            const uint64_t idle = __ballot(slot < 0);
            const int32_t take = min(popc(idle), cnt_ready);
            if (take > 0) {
                if (slot < 0) {
                    const int32_t r = (int32_t)lane_rank(idle);
                    if (r < take) { slot = q[Q_READY * NSLOT + cnt_ready - 1 - r]; hs.load(l, slot); }
                }
                cnt_ready -= take;
            }
        }
        if (STATS) { occ[0] += (unsigned)popc(__ballot(slot >= 0)); occ[1] += (unsigned)cnt_ready; occ[2] += (unsigned)cnt_nee; occ[3] += (unsigned)cnt_post; occ[4] += (unsigned)cnt_esc; occ[5] += (unsigned)cnt_free; }
        // (2) the hot pair: up to thr[COLLIDE] march steps, then the collision code
        int32_t n;
        for (int32_t k = 0; k < S.thr[ST_COLLIDE]; ++k) {
            n = popc(__ballot(slot >= 0 && l.state == ST_MARCH));
            if (n == 0) break;
            VR_STAT(ST_MARCH, n);
            if (slot >= 0 && l.state == ST_MARCH) do_march(l, P);
            VR_STAT_END(ST_MARCH);
        }
        n = popc(__ballot(slot >= 0 && l.state == ST_COLLIDE));
        if (n > 0) {
            VR_STAT(ST_COLLIDE, n);
            if (slot >= 0 && l.state == ST_COLLIDE) {
                ColdGlobal c{ cold_base + slot * C_STRIDE };
                if (P.u.integrator != 0) do_collide_global<USE_TF>(l, c, P); else do_collide<USE_TF>(l, c, P);
            }
            VR_STAT_END(ST_COLLIDE);
        }
        // (3) park paths that reached an event
        {
            const bool parked = slot >= 0 && l.state != ST_MARCH && l.state != ST_COLLIDE;
            if (__ballot(parked)) {
                if (parked) hs.save(l, slot);
                // the hot pair can only leave a path in NEE, POSTNEE or ESCAPE
                VR_PUSH(Q_NEE, cnt_nee, parked && l.state == ST_NEE, slot);
                VR_PUSH(Q_POST, cnt_post, parked && l.state == ST_POSTNEE, slot);
                VR_PUSH(Q_ESC, cnt_esc, parked && l.state == ST_ESCAPE, slot);
                if (parked) slot = -1;
            }
        }
        // (4) event batches
        const int32_t n_live = popc(__ballot(slot >= 0)) + cnt_ready;
        const bool hungry = n_live < S.thr[ST_MARCH];                    // the hot pair is about to run under-filled
        // a batch runs when it is full enough; a hungry wave additionally runs its LARGEST batch (only that one, so that the
        // others keep filling up)
        const int32_t c_new = exhausted ? 0 : cnt_free;
        int32_t big = c_new;
        if (cnt_nee > big) big = cnt_nee;
        if (cnt_post > big) big = cnt_post;
        if (cnt_esc > big) big = cnt_esc;
        const bool want_new = c_new > 0 && (c_new >= S.thr[ST_NEW] || (hungry && c_new == big));
        const bool want_nee = cnt_nee > 0 && (cnt_nee >= S.thr[ST_NEE] || (hungry && cnt_nee == big));
        const bool want_post = cnt_post > 0 && (cnt_post >= S.thr[ST_POSTNEE] || (hungry && cnt_post == big));
        const bool want_esc = cnt_esc > 0 && (cnt_esc >= S.thr[ST_ESCAPE] || (hungry && cnt_esc == big));
        if (want_new || want_nee || want_post || want_esc) {
            // the lanes double as batch workers.  VR_BATCH_REGS=1: the batch path lives in its own register set `b` and the
            // marching path `l` stays put; =0: the marching path is saved to its LDS slot and `l` is reused (fewer VGPRs)
#if VR_BATCH_REGS
            Hot b;
#else
            const int32_t my_slot = slot;
            if (my_slot >= 0) hs.save(l, my_slot);
            __builtin_amdgcn_wave_barrier();
            Hot& b = l;
#endif
            if (want_esc) {
                n = min(64, cnt_esc);
                VR_STAT(ST_ESCAPE, n);
                int32_t bs = -1;
                if (lane < n) {
                    bs = q[Q_ESC * NSLOT + cnt_esc - 1 - lane];
                    hs.load(b, bs);
                    ColdGlobal c{ cold_base + bs * C_STRIDE };
                    do_escape(b, c, P, wu);                              // writes the sample; the slot becomes free
                }
                cnt_esc -= n;
                VR_ROUTE_B(bs);                                            // ST_NEW: the slot is free again
                VR_STAT_END(ST_ESCAPE);
            }
            if (want_post) {
                n = min(64, cnt_post);
                VR_STAT(ST_POSTNEE, n);
                int32_t bs = -1;
                if (lane < n) {
                    bs = q[Q_POST * NSLOT + cnt_post - 1 - lane];
                    hs.load(b, bs);
                    ColdGlobal c{ cold_base + bs * C_STRIDE };
                    do_postnee(b, c, P, wu);
                    hs.save(b, bs);
                }
                cnt_post -= n;
                VR_ROUTE_B(bs);                                            // ST_NEW = path ended (bounce cap / roulette)
                VR_STAT_END(ST_POSTNEE);
            }
            if (want_new) {
                if (cursor == (uint32_t)wu.n_items) {
                    uint32_t j = 0xFFFFFFFFu;
                    while (seg_tries < kQueueSegments) {                    // own segment first, then the following ones
                        const uint32_t k = ((blockIdx.x & (kQueueSegments - 1u)) + seg_tries) & (kQueueSegments - 1u);
                        const uint32_t lo = k * D.seg_len, hi = min(lo + D.seg_len, D.n_units);
                        uint32_t v = 0xFFFFFFFFu;
                        if (lo < hi) { if (lane == 0) v = atomicAdd(D.unit_counter + k, 1u); v = __builtin_amdgcn_readfirstlane(v); }
                        if (lo < hi && v < hi - lo) { j = lo + v; break; }
                        ++seg_tries;                                        // this segment is used up for good
                    }
                    if (j == 0xFFFFFFFFu) exhausted = true;
                    else { wu = make_unit(D, W, j, sbuf); cursor = 0u; }
                }
                n = min(min(64, cnt_free), (int32_t)((uint32_t)wu.n_items - cursor));
                if (n > 0) {
                    VR_STAT(ST_NEW, n);
                    int32_t bs = -1;
                    if (lane < n) {
                        bs = q[Q_FREE * NSLOT + cnt_free - 1 - lane];
                        hot_init(b);
                        ColdGlobal c{ cold_base + bs * C_STRIDE };
                        do_new(b, c, P, wu, cursor + (uint32_t)lane);
                        hs.save(b, bs);
                    }
                    cnt_free -= n;
                    cursor += (uint32_t)n;
                    VR_ROUTE_B(bs);                                        // ST_NEW = pixel outside a ragged frame
                    VR_STAT_END(ST_NEW);
                }
            }
            if (want_nee) {
                n = min(64, cnt_nee);
                VR_STAT(ST_NEE, n);
                int32_t bs = -1;
                if (lane < n) {
                    bs = q[Q_NEE * NSLOT + cnt_nee - 1 - lane];
                    hs.load(b, bs);
                    ColdGlobal c{ cold_base + bs * C_STRIDE };
                    do_nee(b, c, P);
                    hs.save(b, bs);
                }
                cnt_nee -= n;
                VR_ROUTE_B(bs);
                VR_STAT_END(ST_NEE);
            }
#if !VR_BATCH_REGS
            __builtin_amdgcn_wave_barrier();
            if (my_slot >= 0) hs.load(l, my_slot);
#endif
        }
        if (exhausted && cnt_free == pool) break;                        // every path of the pool has finished
    }
    if (STATS && stats && lane == 0) {
#pragma unroll
        for (int k = 0; k < ST_DONE; ++k) { atomicAdd(&stats[2 * k], (unsigned long long)st_exec[k]); atomicAdd(&stats[2 * k + 1], (unsigned long long)st_lanes[k]); }
        atomicAdd(&stats[16], (unsigned long long)iters);
        atomicAdd(&stats[17], 1ull);
#pragma unroll
        for (int k = 0; k < ST_DONE; ++k) atomicAdd(&stats[18 + k], st_cyc[k]);
        atomicAdd(&stats[25], __builtin_readcyclecounter() - t_start);
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&stats[26 + k], occ[k]);
    }
#undef VR_STAT
#undef VR_STAT_END
#undef VR_PUSH
#undef VR_ROUTE
#undef VR_ROUTE_B
#undef VR_ROUTE_ST
}

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

// thr[]: NEW (free slots that trigger a NEW batch), diagnostic cap on the slots in use (0 = all NSLOT), MARCH (= low-water mark of live paths: below it every
// non-empty batch runs), COLLIDE (= march steps per pass), NEE, POSTNEE, ESCAPE (batch sizes that trigger the event)
static SchedParams g_sched = { { 64, 0, 56, 2, 60, 60, 64, 0 }, 0u };
static unsigned long long* g_stats = nullptr;      // device buffer of 26 counters, or null
static int32_t g_samples_per_unit = 8;
static int32_t g_blocks_per_cu = 0;                 // 0 = from the occupancy query

void set_stats_buffer(unsigned long long* dev) { g_stats = dev; }
void set_samples_per_unit(int32_t n) { g_samples_per_unit = n < 1 ? 1 : n; }

void set_sched_thresholds(const int32_t thr[ST_COUNT]) {
    for (int i = 0; i < ST_COUNT; ++i) g_sched.thr[i] = thr[i];
}

static void tuning_from_env() {
    static bool done = false;
    if (done) return;
    done = true;
    if (const char* e = getenv("VR_SPU")) set_samples_per_unit(atoi(e));      // diagnostics only
    if (const char* e = getenv("VR_BLOCKS_PER_CU")) g_blocks_per_cu = atoi(e);
}

size_t pathtrace_pool_floats(int32_t n_tiles, int32_t n_samples) {
    tuning_from_env();
    const int32_t spu = n_samples < g_samples_per_unit ? n_samples : g_samples_per_unit;
    const int32_t chunks = (n_samples + spu - 1) / spu;
    return (size_t)chunks * (size_t)n_tiles * 4u * (size_t)spu * 64u * 4u;
}

template <typename K>
static int resident_blocks(K kernel) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (g_blocks_per_cu > 0) per_cu = g_blocks_per_cu;
    else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
    return cus * per_cu;
}

size_t pathtrace_workspace_floats() { return (size_t)8192 * C_STRIDE * NSLOT; }      // cold state of up to 8192 resident wavefronts

void launch_pathtrace(const SceneParams& P, float* fb, float* sample_pool, float* workspace, uint32_t* unit_counter, const int32_t* tiles, int32_t n_tiles,
                      int32_t first_sample, int32_t n_samples, uint32_t* status, hipStream_t stream) {
    if (n_tiles <= 0 || n_samples <= 0) return;
    tuning_from_env();
    SchedParams S = g_sched;
    LaunchDesc D;
    D.tiles = tiles; D.n_tiles = n_tiles; D.first_sample = first_sample; D.n_samples = n_samples;
    D.spu = n_samples < g_samples_per_unit ? n_samples : g_samples_per_unit;
    const int32_t chunks = (n_samples + D.spu - 1) / D.spu;
    D.n_units = (uint32_t)chunks * (uint32_t)n_tiles * 4u;
    D.chunks = (uint32_t)chunks;
    D.seg_len = (D.n_units + kQueueSegments - 1u) / kQueueSegments;
    D.unit_counter = unit_counter;
    S.max_iters = 1u << 27;          // watchdog (see the kernel): ~100x the iterations of the heaviest wavefront seen
    auto kernel = P.u.use_tf ? (g_stats ? pathtrace_kernel<true, true> : pathtrace_kernel<true, false>)
                             : (g_stats ? pathtrace_kernel<false, true> : pathtrace_kernel<false, false>);
    static int blocks_cache[4] = { 0, 0, 0, 0 };
    int& blocks = blocks_cache[(P.u.use_tf ? 2 : 0) + (g_stats ? 1 : 0)];
    if (blocks == 0 || g_blocks_per_cu > 0) blocks = std::min(resident_blocks(kernel), 2048);      // workspace holds 2048 workgroups
    const uint32_t waves_needed = (D.n_units + 3u) / 4u;
    const dim3 grid((unsigned)std::min<uint32_t>((uint32_t)blocks, waves_needed > 0 ? waves_needed : 1u)), block(256);
    if (P.u.integrator == 2 && P.u.use_tf) {
        const uint64_t items = (uint64_t)D.n_units * (uint64_t)(D.spu * 64);
        hipLaunchKernelGGL(dvr_kernel, dim3((unsigned)((items + 255) / 256)), block, 0, stream, P, sample_pool, D);
    } else {
        (void)hipMemsetAsync(unit_counter, 0, kQueueSegments * sizeof(uint32_t), stream);
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, P, sample_pool, workspace, D, S, status, g_stats);
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
// warp table of sample_environment (see vr_trace.h): one thread per 2x2 block of pyramid level `mip`
__global__ void __launch_bounds__(256)
env_cdf_kernel(const float* __restrict__ level, int32_t d, float* __restrict__ out) {
    const int32_t hd = d >> 1;
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hd * hd) return;
    const int32_t x = i % hd, y = i / hd;
    const float w0 = level[(size_t)(2 * y) * d + 2 * x], w1 = level[(size_t)(2 * y) * d + 2 * x + 1];
    const float w2 = level[(size_t)(2 * y + 1) * d + 2 * x], w3 = level[(size_t)(2 * y + 1) * d + 2 * x + 1];
    const float q0 = w0 + w2, q1 = w1 + w3;
    reinterpret_cast<float4*>(out)[i] = make_float4(q0 / max_(1e-8f, q0 + q1), w0 / q0, w1 / q1, 0.0f);
}
void launch_build_env_cdf(const float* pyramid, int32_t dim, float* table, hipStream_t stream) {
    // levels base-1 .. 0; level m lives at pyramid offset imp_level_offset(dim, m) and has (dim >> m)^2 texels
    int32_t base = 0;
    while ((1 << base) < dim) ++base;
    for (int32_t mip = base - 1; mip >= 0; --mip) {
        const int32_t d = dim >> mip, n = (d >> 1) * (d >> 1);
        float* dst = table + 4 * (size_t)env_cdf_offset(base - 1 - mip);
        hipLaunchKernelGGL(env_cdf_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pyramid + imp_level_offset(dim, mip), d, dst);
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
//                            flag = the brick needs an atlas block (max != min)
//   2. exclusive scan of the flags (brick index order = slot order)
//   3. encode_brick_kernel : per brick, BrickRec + 512 quantised voxels straight into the brick-major atlas
//   4. range_mip_kernel    : (min of mins, max of maxes) over 2x2x2 children, three levels
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
// single-workgroup exclusive scan (brick counts are modest: 2M for a 1024^3 grid); out[n] = total
__global__ void __launch_bounds__(1024)
exclusive_scan_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int32_t n) {
    __shared__ uint32_t part[1024];
    const int32_t per = (n + 1023) / 1024, b = threadIdx.x * per, e = min(n, b + per);
    uint32_t sum = 0u;
    for (int32_t i = b; i < e; ++i) sum += in[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t acc = 0u; for (int32_t i = 0; i < 1024; ++i) { const uint32_t v = part[i]; part[i] = acc; acc += v; } out[n] = acc; }
    __syncthreads();
    uint32_t acc = part[threadIdx.x];
    for (int32_t i = b; i < e; ++i) { out[i] = acc; acc += in[i]; }
}
__global__ void __launch_bounds__(64)
encode_brick_kernel(const float* __restrict__ dense, int32_t nx, int32_t ny, int32_t nz, int32_t nbx, int32_t nby, int32_t bsx, int32_t bsy,
                    const uint32_t* __restrict__ range, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ slot_of,
                    BrickRec* __restrict__ recs, uint8_t* __restrict__ atlas) {
    const int32_t brick = blockIdx.x, lane = threadIdx.x;
    const int32_t bx = brick % nbx, by = (brick / nbx) % nby, bz = brick / (nbx * nby);
    const uint32_t rg = range[brick];
    const float lo = half2float(rg & 0xFFFFu), hi = half2float(rg >> 16);
    const bool alloc = flag[brick] != 0u;
    const uint32_t slot = alloc ? slot_of[brick] : 0u;       // bricks without a block point at slot 0 (indirection word 0)
    if (lane == 0) { BrickRec r; r.slot = slot; r.rmin = lo; r.rdiff = hi - lo; r.range = rg; recs[((((size_t)bz << bsy) + by) << bsx) + bx] = r; }
    if (!alloc) return;
    const float inv = 255.0f / (hi - lo);
    uint8_t* dst = atlas + (size_t)slot * 512u;
    for (int32_t i = lane; i < 512; i += 64) {
        const int32_t x = bx * 8 + (i & 7), y = by * 8 + ((i >> 3) & 7), z = bz * 8 + (i >> 6);
        float v = 0.0f;
        if (x < nx && y < ny && z < nz) v = dense[((size_t)z * ny + y) * nx + x];
        float qv = floor_((v - lo) * inv + 0.5f);
        qv = qv < 0.0f ? 0.0f : (qv > 255.0f ? 255.0f : qv);
        dst[i] = (uint8_t)qv;
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

void launch_encode_ranges(const float* dense, const int32_t dim[3], const int32_t nb[3], uint32_t* range, uint32_t* flag, uint32_t* slot_of, hipStream_t stream) {
    const int32_t n = nb[0] * nb[1] * nb[2];
    hipLaunchKernelGGL(encode_range_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, dense, dim[0], dim[1], dim[2], nb[0], nb[1], nb[2], range, flag);
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, stream, flag, slot_of, n);
}
void launch_encode_bricks(const float* dense, const int32_t dim[3], const int32_t nb[3], const int32_t bshift[2], const uint32_t* range, const uint32_t* flag, const uint32_t* slot_of,
                          BrickRec* recs, uint8_t* atlas, hipStream_t stream) {
    const int32_t n = nb[0] * nb[1] * nb[2];
    hipLaunchKernelGGL(encode_brick_kernel, dim3(n), dim3(64), 0, stream, dense, dim[0], dim[1], dim[2], nb[0], nb[1], bshift[0], bshift[1], range, flag, slot_of, recs, atlas);
}
void launch_range_mip(const uint32_t* src, const int32_t sdim[3], uint32_t* dst, const int32_t ddim[3], hipStream_t stream) {
    const int32_t n = ddim[0] * ddim[1] * ddim[2];
    hipLaunchKernelGGL(range_mip_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, sdim[0], sdim[1], sdim[2], dst, ddim[0], ddim[1], ddim[2]);
}

// ---------------------------------------------------------------------------------------------------
// effective majorant of every cell of every level, written in the padded power-of-two layout that majorant_at indexes
// (vr_scene.h); cells beyond a level's real extent -- and levels the grid does not have -- hold 0
struct MajorantLayout { int32_t nb[3], mip_off[4], n_mips, mshift[3]; };
__global__ void __launch_bounds__(256)
majorant_kernel(const SceneParams P, const uint32_t* __restrict__ range_words, const MajorantLayout L, uint32_t n_padded, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_padded) return;
    const uint32_t k = (uint32_t)(L.mshift[0] + L.mshift[1] + L.mshift[2]);
    uint32_t mip = 0u;
    while (mip < 3u && i >= majorant_level_offset(k, mip + 1u)) ++mip;
    const uint32_t j = i - majorant_level_offset(k, mip);
    const uint32_t sx = (uint32_t)L.mshift[0] - mip, sy = (uint32_t)L.mshift[1] - mip;
    const uint32_t cx = j & ((1u << sx) - 1u), cy = (j >> sx) & ((1u << sy) - 1u), cz = j >> (sx + sy);
    const uint32_t rnd = (1u << mip) - 1u;
    const uint32_t dx = ((uint32_t)L.nb[0] + rnd) >> mip, dy = ((uint32_t)L.nb[1] + rnd) >> mip, dz = ((uint32_t)L.nb[2] + rnd) >> mip;
    float m = 0.0f;
    if ((int32_t)mip <= L.n_mips && cx < dx && cy < dy && cz < dz) {
        m = P.u.vol_density_scale * half2float(range_words[(uint32_t)L.mip_off[mip] + (cz * dy + cy) * dx + cx] >> 16);
        if (P.u.use_tf) {
            float rgba[4];
            tf_lookup(P, m * P.u.vol_inv_majorant, rgba);
            m = P.u.vol_majorant * rgba[3];
        }
    }
    out[i] = m;
}
void launch_majorants(const SceneParams& P, const uint32_t* range_words_all_mips, const int32_t nb[3], const int32_t mip_off[4], int32_t n_mips,
                      const int32_t mshift[3], float* out_padded, hipStream_t stream) {
    MajorantLayout L;
    for (int i = 0; i < 3; ++i) { L.nb[i] = nb[i]; L.mshift[i] = mshift[i]; }
    for (int i = 0; i < 4; ++i) L.mip_off[i] = mip_off[i];
    L.n_mips = n_mips;
    const uint32_t n = (uint32_t)majorant_padded_cells((uint32_t)(mshift[0] + mshift[1] + mshift[2]));
    hipLaunchKernelGGL(majorant_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, P, range_words_all_mips, L, n, out_padded);
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
    default: r = nan_(); break;
    }
    out[i] = r;
}
void launch_math_probe(int32_t fn, const float* a, const float* b, float* out, int32_t n, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fn, a, b, out, n);
}

}  // namespace vr
