// vr_pathtrace.h -- the path-tracing kernel (scheduling part; the per-path code is vr_trace.h).
//
// pathtrace_kernel<K, STATS>: persistent wavefronts pull (8x8 pixel tile x 8 samples) work units from an XCD-aware
// counter; each wavefront keeps a private pool of more path slots than it has lanes, so that the frequent march/collide
// code always finds lanes to fill and the rare, expensive events (new sample with the 32-round TEA hash, next-event
// estimation, scatter, escape) run as near-full-width batches of parked paths.
// K (TraceCfg, vr_trace.h) selects the variant at compile time; vr_pathtrace.hip is compiled once per variant.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "vr_trace.h"

namespace vr {

struct SchedParams {
    int32_t thr[ST_COUNT];     // minimum number of lanes that must wait in a state before its code runs
    uint32_t max_iters;        // watchdog: scheduler iterations a wavefront may run without finishing a path
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
// Watchdog (round 4: progress-based).  A wavefront gives up -- and the launch is reported as failed -- when it has neither finished a path nor pulled a
// work unit for kMaxIdleIters scheduler iterations or kMaxIdleTicks of shader clock (~3 s), whichever comes first.  Both counters restart with every
// finished path, so a launch may legitimately run for as long as it has work (round 3 measured the wavefront's LIFETIME against 8 s, which a
// 1000-bounce render of a thick cloud approaches); an input whose paths never end (a majorant that overflowed to +inf) still stops within seconds.
constexpr uint32_t kMaxIdleIters = 1u << 26;       // the heaviest wavefront seen runs ~10^6 iterations in its whole life
constexpr uint64_t kMaxIdleTicks = 7200000000ull;  // ~3 s at 2.4 GHz
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
#ifndef VR_MARCH_SPECULATIVE
#define VR_MARCH_SPECULATIVE 1
#endif
#ifndef VR_MARCH_LOADS_PINNED
#define VR_MARCH_LOADS_PINNED (VR_MARCH_STEPS == 2)
#endif
#ifndef VR_DIAG_PAD_VALU
#define VR_DIAG_PAD_VALU 0
#endif
#ifndef VR_DIAG_PAD_SLEEP
#define VR_DIAG_PAD_SLEEP 0
#endif
#ifndef VR_BATCH_REGS
#define VR_BATCH_REGS 1
#endif
// Instruction-arbiter priority (s_setprio) of a wavefront while it runs the hot pair / its event batches (round 5).  A SIMD's four wavefronts compete for the
// issue slots; the hot pair is short dependent chains that the issue-bound kernels live on, an event batch is long latency chains (cold line, nine table
// levels, texels) that lose little by waiting a cycle.  Hot pair above events (1 / 0): c2 +0.6 ... +1 %, c3 +1.9 %, c5full +1 %, c4 and c5cloud +-0
// (2 / 1, 3 / 0, 3 / 1 the same within the noise; events ABOVE the hot pair: c2 -0.4 %, c4 +0.5 %): profiles/r5i_*, r5j_*.
#ifndef VR_PRIO_HOT
#define VR_PRIO_HOT 1
#endif
#ifndef VR_PRIO_EVENTS
#define VR_PRIO_EVENTS 0
#endif
#ifndef VR_EMISSION_BY_POINTER
#define VR_EMISSION_BY_POINTER 1
#endif

// ---------------------------------------------------------------------------------------------------
// Wave-private path pool.
//
// A wavefront owns NSLOT path slots, more than it has lanes.  The 64 lanes hold, in registers, the hot state of the
// paths that are currently marching; every other path of the pool is parked: its hot state (NHOT dwords) sits in LDS
// and its slot id in one of the wave's LDS stacks -- READY (may march), NEE / POSTNEE / ESCAPE (wait for that event),
// FREE.  Cold path state lives in global memory (a 64-byte slot per path, see ColdGlobal) and in vector registers (ShleBanks); only the
// events touch it.
//   * a lane whose path reaches an event parks it (ds_write2 pairs + a stack push) and immediately resumes a READY path,
//     so the march/collide code runs with most lanes holding a path;
//   * an event's code runs when a (nearly) full-width batch of parked paths has piled up, or -- when the wave runs dry --
//     for its largest batch: lane i loads parked path i, runs the unchanged per-path code of vr_trace.h, stores it and
//     routes the slot to the stack of its new state.  The lanes double as batch workers; the batch path lives in its own
//     register set so the marching path stays put (VR_BATCH_REGS=0 swaps it through its LDS slot instead).
// Everything is wave-synchronous (ballots, mbcnt ranks, scalar counters): no atomics, no barriers, no spinning; the only
// global atomic is the work-queue head.  Which lane runs which path never changes a result.
// Pool size = LDS share of a wavefront (10 KiB at 4 wavefronts per SIMD) / bytes per slot.  Round 3: 1/dir of a parked path is no longer kept in its slot
// (15 dwords, 152 slots: VR_HOT_RI=1) but recomputed when the path is resumed (three exact reciprocals, vr_math.h rcp_exact): 12 dwords in 13-dword
// slots (odd stride) = 175 slots; 12-dword slots (188, VR_HOT_STRIDE=12) measure the same (profiles/r3f_*: the pool-size elasticity is gone beyond 152).
#ifndef VR_HOT_RI
#define VR_HOT_RI 0
#endif
#ifndef VR_HOT_STRIDE
#define VR_HOT_STRIDE (VR_HOT_RI ? 15 : 12)      /* round 5: 12-dword slots without the pad dword = 188 slots instead of 175 (13-dword, odd stride: rounds 3-4).  With the bookkeeping
                                                    amortised over up to four passes the larger pool is worth more than the odd stride: c2 +0.6 %, c4 +0.3 %, c5cloud +-0 (profiles/r5l_*);
                                                    150 slots: c2 -1.7 %, c4 -1.1 %; 1/dir stored again (15 dwords, 152 slots): c2 -1.2 %, c5cloud -3 % */
#endif
// Workgroup shape.  Default: four 4-wavefront workgroups per CU.  Build-time experiment (round 3, profiles/r3h_lds_resident_majorants.txt):
// -DVR_WG_WAVES=16 = ONE workgroup per CU whose 16 wavefronts share nothing but read-only tables in LDS -- the transfer-function LUT once per CU
// instead of four times and (VR_MAJ_LDS) the coarse levels of the majorant table, or all of it (smoke.brick's is 18 KiB).  Measured: the workgroup
// shape alone +-0.5 %; serving the majorant gathers from LDS c2 +-0 (its whole table resident: 28 % of the kernel's L1 accesses gone), c3 -4 %, c4 -3 %,
// c5full -5 % -- the select between the two sources costs what the L1-hit gathers cost.  Not adopted.
#ifndef VR_WG_WAVES
#define VR_WG_WAVES 4
#endif
#ifndef VR_MAJ_LDS
#define VR_MAJ_LDS 0
#endif
constexpr int32_t kWgWaves = VR_WG_WAVES;
constexpr int32_t kLdsPerWorkgroup = 163840 / 16 * kWgWaves;      // 160 KiB per CU
// The transfer-function kernels stage the LUT (up to kLutLdsEntries vec4 = 4 KiB) in LDS.
constexpr int32_t kLutLdsEntries = 256;
// cells of the majorant table's tail a kernel keeps in LDS: the brick kernel without transfer function 10 240 fp16 cells (20 KiB: all of a grid of up to
// ~8 000 bricks), the dense-grid and emission kernels 5 120 (10 KiB: levels 2-3 of a 512^3 grid, level 3 of a 1024^3 one), the transfer-function kernels
// 2 560 floats (10 KiB: their table holds TF-remapped floats)
template <class K> constexpr int32_t maj_lds_cells() { return !VR_MAJ_LDS ? 0 : (K::tf ? 2560 : ((K::dense == 0 && K::emission == 0) ? 10240 : 5120)); }
// Build-time experiment (round 3, -DVR_COLD_REGS=1; profiles/r3a_cold_state_in_registers.txt): cold path state in VECTOR REGISTERS
// (ColdBanks below) instead of global memory, for the dense-grid kernel, whose paths scatter 3.2 times per sample (c4) and spend a
// third of their memory-side traffic on the 64-byte cold slots.  Three wavefronts per SIMD instead of four leave each 168 registers:
// the fourth wavefront's share of the register file holds the cold state of all 192 slots of the other three (3 banks x 20 fields =
// 60 registers), its share of the LDS makes the pools 192 slots instead of 152.  No cold workspace, no cold traffic, bit-identical
// images.  Measured on c4: +7 % against the same 3 x 192 configuration with the cold state in memory, but the fourth wavefront is
// worth 10 %: 1.906 against 1.972 Gsamples/s for the default (4 x 152, cold state in memory).  Off by default.
#ifndef VR_COLD_REGS
#define VR_COLD_REGS 0
#endif
template <class K> constexpr bool cold_in_regs() { return VR_COLD_REGS != 0 && !K::tf && K::dense == 1 && K::emission == 0; }
template <class K> constexpr int32_t waves_per_simd() { return cold_in_regs<K>() ? 3 : VR_WAVES_PER_SIMD; }

enum PoolStack : int32_t { Q_READY = 0, Q_NEE = 1, Q_POST = 2, Q_ESC = 3, Q_FREE = 4, Q_COUNT = 5 };

// Lazy first cold line in the emission kernel too (round 3).  With an emission grid every tentative collision of a camera segment adds emitted light to the
// path's radiance, so until round 3 do_new wrote the whole cold line up front (80 bytes per sample) and every resume / park of a camera segment moved
// throughput and radiance through it -- also for the majority of samples that never scatter.  Now the radiance of a `first` path waits in three more dwords
// of its LDS slot (15-dword slots, 153 of them, in the emission kernel only) and its throughput is 1: such a path touches no cold line at all.
#ifndef VR_LAZY_EMISSION
#define VR_LAZY_EMISSION 1
#endif
template <class K> constexpr bool lazy_emission() { return VR_LAZY_EMISSION != 0 && !VR_HOT_RI && K::emission == 1; }
template <class K> constexpr int32_t hot_stride() { return lazy_emission<K>() ? 15 : VR_HOT_STRIDE; }      // dwords per slot in LDS (the parked fields, padded to an odd count)
constexpr int32_t HOT_EL = 12;                           // lazy_emission kernels: radiance of a `first` path, dwords 12..14 of its slot
// slots of a wavefront's pool: what is left of the workgroup's LDS after the shared tables, per wavefront, in slots of HOT_STRIDE dwords + Q_COUNT stack
// bytes; at most 192 (slot ids index three register banks, ShleBanks) -- or VR_NSLOT when a build pins it
template <class K> constexpr int32_t pool_slots() {
#ifdef VR_NSLOT
    return cold_in_regs<K>() ? 192 : VR_NSLOT;
#else
    if (cold_in_regs<K>()) return 192;
    const int32_t shared = maj_lds_cells<K>() * (K::tf ? 4 : 2) + (K::tf ? kLutLdsEntries * 16 + 16 : kWgWaves * 256);
    const int32_t n = (kLdsPerWorkgroup - shared) / kWgWaves / (hot_stride<K>() * 4 + Q_COUNT);
    return n > 192 ? 192 : n;
#endif
}
constexpr int32_t NSLOT = 192;             // upper bound of pool_slots (sizes the cold workspace); slot ids are bytes
constexpr int32_t HOT_COL = VR_HOT_RI ? 12 : 4;         // where the transfer-function kernels keep the colour of a real collision until its event: in the place of
                                                        // 1/dir (15-dword slots) or of dir (12-dword slots) -- both dead between the collision and the set-up of the next segment
// WORLD (vr_trace.h world_slot, -DVR_WORLD_SLOT=1): dwords 1..6 hold the segment's origin and direction in WORLD space (Hot::wpos / wdir) instead of the index-space
// ones; load_resume recomputes those with begin_segment's two transforms -- so that the collision event finds position and direction in the slot and reads no cold line
template <int32_t HOT_STRIDE, bool WORLD = false>
struct HotStoreT {                     // [slot][field]: a path's parked dwords are adjacent (ds_read2/ds_write2 pairs)
    static constexpr int32_t kStride = HOT_STRIDE;
    uint32_t* base;
    // lazy_emission kernels: the radiance a `first` path has gathered so far (Hot::eL) waits in its slot
    __device__ __forceinline__ void save_first_radiance(v3 eL, int32_t slot) const { uint32_t* p = base + slot * HOT_STRIDE + HOT_EL; p[0] = f2u(eL.x); p[1] = f2u(eL.y); p[2] = f2u(eL.z); }
    __device__ __forceinline__ v3 load_first_radiance(int32_t slot) const { const uint32_t* p = base + slot * HOT_STRIDE + HOT_EL; return v3{ u2f(p[0]), u2f(p[1]), u2f(p[2]) }; }
    float cam_ipos[3];                 // index-space position of the camera (wave-uniform): ipos of every `first` path, see FirstStash
    // mip (a multiple of 1/4 in [0,3]) rides in the flag word
    __device__ __forceinline__ static uint32_t flags(const Hot& h) { return (uint32_t)h.state | ((uint32_t)h.shadow << 8) | ((uint32_t)h.first << 12) | ((uint32_t)h.mipq << 16); }
    // a new path: everything, with the stash (world direction, sample slot) in the places of ipos and Tr
    __device__ __forceinline__ void save_new(const Hot& h, int32_t slot) const {
        uint32_t* p = base + slot * HOT_STRIDE;
        p[0] = h.seed;
        const v3 a = WORLD ? h.wpos : h.ipos, d = WORLD ? h.wdir : h.idir;
        p[1] = f2u(a.x); p[2] = f2u(a.y); p[3] = f2u(a.z);
        p[4] = f2u(d.x); p[5] = f2u(d.y); p[6] = f2u(d.z);
        p[7] = f2u(h.t); p[8] = f2u(h.far); p[9] = f2u(h.tau);
        p[10] = f2u(h.Tr);
        p[11] = flags(h);
        if (VR_HOT_RI) { p[12] = f2u(h.ri.x); p[13] = f2u(h.ri.y); p[14] = f2u(h.ri.z); }
    }
    // any later store: a `first` path's stash stays where it is
    __device__ __forceinline__ void save(const Hot& h, int32_t slot) const {
        uint32_t* p = base + slot * HOT_STRIDE;
        p[0] = h.seed;
        if (WORLD) {
            p[1] = f2u(h.wpos.x); p[2] = f2u(h.wpos.y); p[3] = f2u(h.wpos.z);
            p[4] = f2u(h.wdir.x); p[5] = f2u(h.wdir.y); p[6] = f2u(h.wdir.z);
            if (!h.first) p[10] = f2u(h.Tr);
        } else {
        if (!h.first) { p[1] = f2u(h.ipos.x); p[2] = f2u(h.ipos.y); p[3] = f2u(h.ipos.z); p[10] = f2u(h.Tr); }
        p[4] = f2u(h.idir.x); p[5] = f2u(h.idir.y); p[6] = f2u(h.idir.z);
        }
        p[7] = f2u(h.t); p[8] = f2u(h.far); p[9] = f2u(h.tau);
        p[11] = flags(h);
        if (VR_HOT_RI) { p[12] = f2u(h.ri.x); p[13] = f2u(h.ri.y); p[14] = f2u(h.ri.z); }
    }
    // a path that leaves the hot pair for an event: the march / collision code only changes seed, t, tau, Tr and the flag word
    // (state, mip); ray, far and 1/dir are still in the slot from the store that preceded the path's resume
    __device__ __forceinline__ void save_marched(const Hot& h, int32_t slot) const {
        uint32_t* p = base + slot * HOT_STRIDE;
        p[0] = h.seed;
        p[7] = f2u(h.t); p[9] = f2u(h.tau);
        if (!h.first) p[10] = f2u(h.Tr);
        p[11] = flags(h);
    }
    // for an event batch: the slot as it is (a `first` path's ipos / Tr = its stash, which is what do_nee / do_escape want)
    __device__ __forceinline__ void load(Hot& h, int32_t slot) const {
        const uint32_t* p = base + slot * HOT_STRIDE;
        h.seed = p[0];
        h.ipos = v3{ u2f(p[1]), u2f(p[2]), u2f(p[3]) };
        h.idir = v3{ u2f(p[4]), u2f(p[5]), u2f(p[6]) };
        if (WORLD) { h.wpos = h.ipos; h.wdir = h.idir; }            // (the events read wpos / wdir; a resumed path gets its index-space ray in load_resume)
        h.ri = VR_HOT_RI ? v3{ u2f(p[12 % HOT_STRIDE]), u2f(p[13 % HOT_STRIDE]), u2f(p[14 % HOT_STRIDE]) } : v3{ 0.0f, 0.0f, 0.0f };      // events do not read it (begin_segment sets it)
        h.t = u2f(p[7]); h.far = u2f(p[8]); h.tau = u2f(p[9]);
        h.Tr = u2f(p[10]);
        const uint32_t f = p[11];
        h.state = (int32_t)(f & 0xFFu); h.shadow = (int32_t)((f >> 8) & 0xFu); h.first = (int32_t)((f >> 12) & 0xFu);
        h.mipq = (int32_t)(f >> 16);
        h.majorant = 0.0f;
    }
    // for marching: a `first` path gets the real values of the two fields (= first_resume, vr_trace.h)
    __device__ __forceinline__ void load_resume(Hot& h, int32_t slot, const float* inv_transform = nullptr) const {
        load(h, slot);
        const bool first = h.first != 0;
        if (WORLD) {
            // begin_segment's two transforms on the values it was given (inv_transform = Uniforms::vol_density_inv_transform): the same index-space ray, bit for bit
            h.ipos = mat4_point(inv_transform, h.wpos);
            h.idir = mat4_dir(inv_transform, h.wdir);
        } else
        h.ipos = v3{ first ? cam_ipos[0] : h.ipos.x, first ? cam_ipos[1] : h.ipos.y, first ? cam_ipos[2] : h.ipos.z };
        h.Tr = first ? 1.0f : h.Tr;
        if (!VR_HOT_RI) h.ri = rcp3_exact(h.idir);             // as begin_segment computed it
    }
};
// Cold path state of one wavefront in global memory (vr_trace.h ColdField): a 64-byte slot per path -- half a cache line: the 16
// floats that live as long as the path -- and, in a separate compact array, 16 bytes per path: the radiance of the pending light
// sample (collision event -> scatter event) and the path's slot in the sample buffer.  An event touches one half-line of the
// big array (159 -> 80 MB for all resident wavefronts) and dirties one of its two sectors.  Measured against one 128-byte line per
// path with everything in it: c4 +3.3 %, c2 +1.1 %, same bytes moved (profiles/r2y_ab_cold_64_byte_slots.txt).  Earlier experiments:
// group-major [group][slot][4] (same speed, more traffic), non-temporal accesses (-21 %), everything in LDS (-28 ... -42 %: the
// pool slots it costs), profiles/r2j_layout_experiments.txt, r2m_cold_state_in_lds_experiments.txt.
constexpr uint32_t kStatsWaveBase = 32u;                 // statistics buffer: 32 counters, then (begin, queue empty, end) per wavefront of the launch
constexpr int32_t kMaxWorkgroups = 2048;                // the cold-state workspace is sized for this many resident 4-wavefront units = 8192 wavefronts (launch_pathtrace clamps the grid to it)
// SWAP (the kernels with VR_WORLD_SLOT, vr_trace.h world_slot): throughput and direction trade places in the slot and sh_pdf moves next to the direction -- sector 0 =
// [ unused | dir, sh_pdf ], sector 1 = [ L, n_paths | thr, f_p ] -- because those kernels' collision event reads nothing and writes ITS segment's direction and sh_pdf (the
// collision point stays with the path, the light sample's phase value is re-evaluated by the scatter event), and their scatter event writes L, n_paths, thr, f_p:
// every event dirties exactly ONE 32-byte sector, in whole 16-byte groups (do_nee / do_postnee, vr_trace.h)
#ifndef VR_COLD_NT_STORES
#define VR_COLD_NT_STORES 0
#endif
template <bool SWAP = false>
struct ColdGlobalT {
    // SWAP (the kernels with VR_WORLD_SLOT): [ -, - ][ dir, sh_pdf ][ L, n_paths ][ thr, f_p ] -- the collision event writes the second 16 bytes, the scatter event the
    // upper sector; pos and f_pl are not kept (vr_trace.h do_nee)
    static __device__ __forceinline__ constexpr int32_t phys(int32_t f) {
        return !SWAP ? f : ((f >= C_THR && f < C_THR + 3) ? f + (C_DIR - C_THR) : ((f >= C_DIR && f < C_DIR + 3) ? f - (C_DIR - C_THR) : (f == C_SHPDF ? C_FPL : (f == C_FPL ? C_SHPDF : f))));
    }
    float* base;                       // this slot's 16 floats in the wavefront's slice of the main array
    float* side;                       // this slot's 4 floats in the wavefront's slice of the side array
    float* col;                        // C_COL: the three dwords of the path's parked hot state (LDS) that hold 1/dir -- dead between a real collision and
                                       // the event that follows it, which is when the transfer-function kernels keep the collision's colour there
    __device__ __forceinline__ float ld(int32_t f) const {     // f is a compile-time constant at every call: the selection folds
        return f >= C_COL ? col[f - C_COL]
             : f < C_SIDE ? static_cast<const float*>(__builtin_assume_aligned(base, C_STRIDE * 4))[phys(f)]
                          : static_cast<const float*>(__builtin_assume_aligned(side, C_SIDE_STRIDE * 4))[f - C_SIDE];
    }
    __device__ __forceinline__ void st(int32_t f, float v) {
        if (f >= C_COL) col[f - C_COL] = v;
#if VR_COLD_NT_STORES
        // build-time experiment (round 6): the cold slot's stores as non-temporal ones -- with VR_WORLD_SLOT nobody reads a sector back before it has left the L2 anyway
        else if (f < C_SIDE) __builtin_nontemporal_store(v, static_cast<float*>(__builtin_assume_aligned(base, C_STRIDE * 4)) + phys(f));
#else
        else if (f < C_SIDE) static_cast<float*>(__builtin_assume_aligned(base, C_STRIDE * 4))[phys(f)] = v;
#endif
        else static_cast<float*>(__builtin_assume_aligned(side, C_SIDE_STRIDE * 4))[f - C_SIDE] = v;
    }
};
typedef ColdGlobalT<false> ColdGlobal;
// vr_trace.h ld4 for a slot in memory: one 16-byte load when the four fields are one aligned group of the (physical) layout -- the arguments are constants at every call
template <bool SWAP>
__device__ __forceinline__ Quad ld4(const ColdGlobalT<SWAP>& c, int32_t f3, int32_t f1) {
    const int32_t p3 = ColdGlobalT<SWAP>::phys(f3), p1 = ColdGlobalT<SWAP>::phys(f1);
    if (f3 + 2 < C_SIDE && f1 < C_SIDE && (p3 & 3) == 0 && p1 == p3 + 3 && ColdGlobalT<SWAP>::phys(f3 + 2) == p3 + 2) {
        const float4 v = *reinterpret_cast<const float4*>(static_cast<const float*>(__builtin_assume_aligned(c.base, C_STRIDE * 4)) + p3);
        return Quad{ v3{ v.x, v.y, v.z }, v.w };
    }
    return Quad{ ld3(c, f3), c.ld(f1) };
}
constexpr int32_t kColdWaveFloats = C_STRIDE * NSLOT, kColdSideWaveFloats = C_SIDE_STRIDE * NSLOT;
constexpr size_t kColdMainFloats = (size_t)kMaxWorkgroups * 4u * (size_t)kColdWaveFloats;       // the side arrays follow the main arrays of all wavefronts

// the lanes for which `cond` holds; the builtin takes the i1 as it is (HIP's __ballot goes through an int: v_cndmask + v_cmp per call)
__device__ __forceinline__ uint64_t wave_ballot(bool cond) { return __builtin_amdgcn_ballot_w64(cond); }
__device__ __forceinline__ int32_t popc(uint64_t mask) { return (int32_t)__popcll(mask); }     // int: min(long long, int) would go through double
// the same as one scalar instruction the optimiser cannot widen: `popc(m) < constant` otherwise becomes a 64-bit compare, which only the VECTOR unit has (v_cmp_lt_u64)
__device__ __forceinline__ int32_t popc_s(uint64_t mask) { int32_t r; asm("s_bcnt1_i32_b64 %0, %1" : "=s"(r) : "s"(mask) : "scc"); return r; }
__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Radiance of the pending light samples of a wavefront's parked paths (Hot::shle), in vector registers: slot s lives in lane
// s & 63 of bank s >> 6 -- 3 banks x 3 components = 9 registers hold it for all <= 192 slots.  The collision-event batch
// (lane i works on slot bs_i) moves its values to their home lanes: every lane learns through a 64-dword LDS row which batch lane
// holds "its" slot of each bank (byte k of the row's word d = 1 + the batch lane working on slot 64 k + d) and pulls the value
// with ds_bpermute; the scatter-event batch pulls them back from the home lanes.  Both run with all 64 lanes active.
// What this replaces: a 12-byte store and load per bounce in a global side array (one 32-byte sector written back and a 64-byte
// fetch each): profiles/r2z_*.
struct ShleBanks { v3 b[3]; uint32_t item[3]; };       // item: the paths' slots in the sample buffer, parked by a path's first collision event
__device__ __forceinline__ uint32_t lane_pull_u(uint32_t src_lane, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }
__device__ __forceinline__ float lane_pull(uint32_t src_lane, float v) { return u2f(lane_pull_u(src_lane, f2u(v))); }
// valid: this lane's batch path (slot bs) carries a light sample to park; with_item: also its sample-buffer slot (first collision)
template <bool ITEMS>
__device__ __forceinline__ void shle_park(ShleBanks& B, uint32_t* stage, int32_t lane, bool valid, int32_t bs, v3 val, bool with_item, uint32_t item) {
    // the row is how the lanes talk to each other: the fences keep a lane's read from being satisfied from its own earlier store
    // (wavefront scope: no instruction, the LDS executes a wavefront's accesses in order)
    stage[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (valid) reinterpret_cast<uint8_t*>(stage)[((bs & 63) << 2) + (bs >> 6)] = (uint8_t)((lane + 1) | (ITEMS && with_item ? 0x80 : 0));
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const uint32_t w = stage[lane];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t byte = (w >> (8 * k)) & 0xFFu, src = byte & 0x7Fu;      // src: 1 + the batch lane that holds slot 64 k + lane, or 0
        const uint32_t from = src ? src - 1u : (uint32_t)lane;
        const float x = lane_pull(from, val.x), y = lane_pull(from, val.y), z = lane_pull(from, val.z);
        B.b[k] = v3{ src ? x : B.b[k].x, src ? y : B.b[k].y, src ? z : B.b[k].z };
        if (ITEMS) { const uint32_t it = lane_pull_u(from, item); B.item[k] = (byte & 0x80u) ? it : B.item[k]; }
    }
}
__device__ __forceinline__ uint32_t bank_home(int32_t lane, int32_t bs) { return bs >= 0 ? (uint32_t)(bs & 63) : (uint32_t)lane; }
__device__ __forceinline__ v3 shle_fetch(const ShleBanks& B, int32_t lane, int32_t bs) {
    const uint32_t home = bank_home(lane, bs);
    const int32_t k = bs >> 6;
    v3 r[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) r[j] = v3{ lane_pull(home, B.b[j].x), lane_pull(home, B.b[j].y), lane_pull(home, B.b[j].z) };
    return v3{ k == 0 ? r[0].x : (k == 1 ? r[1].x : r[2].x), k == 0 ? r[0].y : (k == 1 ? r[1].y : r[2].y), k == 0 ? r[0].z : (k == 1 ? r[1].z : r[2].z) };
}
__device__ __forceinline__ uint32_t item_fetch(const ShleBanks& B, int32_t lane, int32_t bs) {
    const uint32_t home = bank_home(lane, bs);
    const int32_t k = bs >> 6;
    const uint32_t r0 = lane_pull_u(home, B.item[0]), r1 = lane_pull_u(home, B.item[1]), r2 = lane_pull_u(home, B.item[2]);
    return k == 0 ? r0 : (k == 1 ? r1 : r2);
}

// ---------------------------------------------------------------------------------------------------
// The whole cold state of a wavefront's paths in vector registers (cold_in_regs kernels): field f of slot s lives in lane s & 63 of
// register v[s >> 6][f] -- the 16 floats of a cold slot (ColdField, vr_trace.h) followed by the radiance of the pending light
// sample and the path's slot in the sample buffer.  An event batch (lane i works on slot bs_i) pulls the fields its event reads
// into a per-lane working copy (ColdLocal: the `Cold` the lane code of vr_trace.h is written against) with ds_bpermute -- one
// per bank, the lane's own bank selected afterwards -- and the home lanes pull the fields the event wrote back the same way,
// told by the LDS row of shle_park which batch lane holds "their" slot of each bank.  All of it runs with all 64 lanes active.
constexpr int32_t kBankFields = 20;
static_assert(C_SIDE == 16 && C_SHLE == 16 && C_ITEM == 19 && C_COL == kBankFields, "ColdLocal maps field f to v[f]");
struct ColdBanks { float v[3][kBankFields]; };
struct ColdLocal {
    float v[kBankFields];
    float* col;                        // C_COL (transfer-function kernels): the parked path's LDS slot, as in ColdGlobal
    __device__ __forceinline__ float ld(int32_t f) const { return f >= C_COL ? col[f - C_COL] : v[f]; }      // f is a compile-time constant at every call
    __device__ __forceinline__ void st(int32_t f, float x) { if (f >= C_COL) col[f - C_COL] = x; else v[f] = x; }
};
constexpr uint32_t cold_bits(int32_t first, int32_t n) { return ((1u << n) - 1u) << first; }
// what each event reads and writes (do_nee / do_postnee / do_escape, vr_trace.h)
constexpr uint32_t kColdNeeR = cold_bits(C_POS, 3) | cold_bits(C_THR, 3) | cold_bits(C_DIR, 3);
constexpr uint32_t kColdNeeW = cold_bits(C_POS, 8) | cold_bits(C_SHLE, 3);                                                // pos, sh_pdf, thr, f_p of the light sample, its radiance
constexpr uint32_t kColdNeeWFirst = cold_bits(C_L, 8) | cold_bits(C_ITEM, 1);                                             // a path's first collision: L, n_paths, dir, f_p, sample slot
constexpr uint32_t kColdPostR = cold_bits(C_POS, 15) | cold_bits(C_SHLE, 4);                                              // everything but C_FP
constexpr uint32_t kColdPostW = cold_bits(C_THR, 3) | cold_bits(C_L, 8);                                                  // thr (roulette), L, n_paths, dir, f_p
constexpr uint32_t kColdEscR = cold_bits(C_THR, 3) | cold_bits(C_L, 8) | cold_bits(C_ITEM, 1);
__device__ __forceinline__ void cold_clear(ColdLocal& c) {
#pragma unroll
    for (int f = 0; f < kBankFields; ++f) c.v[f] = 0.0f;
    c.col = nullptr;
}
template <uint32_t MASK>
__device__ __forceinline__ void cold_fetch(ColdLocal& c, const ColdBanks& B, int32_t lane, int32_t bs) {
    const uint32_t home = bank_home(lane, bs);
    const int32_t k = bs >> 6;
#pragma unroll
    for (int f = 0; f < kBankFields; ++f)
        if ((MASK >> f) & 1u) {
            const float r0 = lane_pull(home, B.v[0][f]), r1 = lane_pull(home, B.v[1][f]), r2 = lane_pull(home, B.v[2][f]);
            c.v[f] = k == 0 ? r0 : (k == 1 ? r1 : r2);
        }
}
// the row through which the home lanes learn who works on their slots: byte k of word d = 1 + the batch lane that holds slot 64 k + d
// (bit 7: that lane's `flag`), 0 = nobody.  The fences keep a lane's read from being satisfied from its own earlier store.
__device__ __forceinline__ uint32_t cold_stage(uint32_t* stage, int32_t lane, int32_t bs, bool flag) {
    stage[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (bs >= 0) reinterpret_cast<uint8_t*>(stage)[((bs & 63) << 2) + (bs >> 6)] = (uint8_t)((lane + 1) | (flag ? 0x80 : 0));
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    return stage[lane];
}
// FLAGGED: only from batch lanes that raised their flag
template <uint32_t MASK, bool FLAGGED>
__device__ __forceinline__ void cold_store(ColdBanks& B, uint32_t w, int32_t lane, const ColdLocal& c) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t byte = (w >> (8 * k)) & 0xFFu, src = byte & 0x7Fu;
        const uint32_t from = src ? src - 1u : (uint32_t)lane;
        const bool take = FLAGGED ? byte > 0x80u : src != 0u;
#pragma unroll
        for (int f = 0; f < kBankFields; ++f)
            if ((MASK >> f) & 1u) {
                const float x = lane_pull(from, c.v[f]);
                B.v[k][f] = take ? x : B.v[k][f];
            }
    }
}

// All kernel arguments travel as ONE struct so that the event code can address any of them through the kernarg pointer.
// The scene parameters alone are ~1 KiB of uniforms.  The hot pair (march / collide) reads its few fields from the by-value
// argument, which the compiler keeps in SGPRs; the event code (new sample, NEE, scatter, escape) reads everything else --
// camera, environment, transforms, the work queue -- through event_args(): the same bytes, addressed through a pointer the
// optimiser cannot see through, so that those ~200 dwords are fetched by scalar loads where an event needs them instead of
// being hoisted out of the scheduler loop and spilled (round 1: 217 SGPR spills = a v_readlane/v_writelane + s_nop per use).
struct KernelArgs {
    SceneParams P;
    LaunchDesc D;
    SchedParams S;
    float* sbuf;                 // sample pool
    float* cold_ws;              // cold path state of all resident wavefronts
    uint32_t* status;            // [0] bit 0: watchdog tripped, bit 1: a path ended in an impossible state
    unsigned long long* stats;   // STATS kernels only
};
typedef const __attribute__((address_space(4))) KernelArgs* KernargPtr;
__device__ __forceinline__ const KernelArgs& event_args() {
    KernargPtr p = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();      // the struct is the only kernel parameter: offset 0
    asm volatile("" : "+s"(p));
    return *(const KernelArgs*)p;
}

template <class K, bool STATS>
__global__ void __launch_bounds__(64 * kWgWaves, waves_per_simd<K>())
pathtrace_kernel(const KernelArgs A) {
#ifndef VR_PIN_TAP_POINTERS
#define VR_PIN_TAP_POINTERS 0
#endif
#if VR_PIN_TAP_POINTERS
    // experiment (round 5): the density grid's tap pointer as an opaque scalar value -- the register allocator treats a kernel-argument load as free to repeat and
    // re-loaded it in the collision code of every pass (s_load + wait in front of the tap)
    SceneParams Ppin = A.P;
    asm volatile("" : "+s"(Ppin.density.atlas), "+s"(Ppin.density.dense));
    const SceneParams& P = Ppin;
#else
    const SceneParams& P = A.P;           // hot pair only; events use event_args()
#endif
    const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // path slots of this kernel's wavefronts (shadows the global maximum below); the instrumented instances give three of them up for their counters (lds_stat)
    constexpr int32_t NS = pool_slots<K>() - (STATS ? 3 : 0);

    static_assert(!cold_in_regs<K>() || kWgWaves == 4, "the cold-state-in-registers experiment runs 3 workgroups of 4 wavefronts per CU");
    __shared__ uint8_t lds_q[kWgWaves * Q_COUNT * NS];
    uint8_t* const q = lds_q + wave * (Q_COUNT * NS);
    // per-wavefront slices of the workspace: the cold fields of its NSLOT paths
    const uint32_t wave_index = blockIdx.x * (uint32_t)kWgWaves + (uint32_t)wave;
    float* const cold_base = A.cold_ws + (size_t)wave_index * (size_t)kColdWaveFloats;
    float* const side_base = A.cold_ws + kColdMainFloats + (size_t)wave_index * (size_t)kColdSideWaveFloats;
    typedef ColdGlobalT<world_slot<K>()> ColdT;
#define VR_COLD(SLOT) ColdT{ cold_base + (SLOT) * C_STRIDE, side_base + (SLOT) * C_SIDE_STRIDE, reinterpret_cast<float*>(hs.base + (SLOT) * HS + HOT_COL) }
    // where the radiance of a parked path's pending light sample waits: vector registers (ShleBanks), or -- in the transfer-function
    // variants, which have no registers to spare (126 of 128) -- the side array
    constexpr bool kColdRegs = cold_in_regs<K>();            // the whole cold state in registers (ColdBanks); else:
    constexpr bool kShleInRegs = !K::tf && !kColdRegs;
    // the sample-buffer slot joins it there in the dense-grid kernel, where nearly every path scatters (c4 +1.3 %, memory-side traffic
    // 1.59x -> 1.52x); on smoke.brick two thirds of the escaping paths never scattered and the three extra ds_bpermute of every
    // escape batch cost more than the side-array accesses they save (c2 -0.7 %): profiles/r2z_*
    constexpr bool kItemInRegs = kShleInRegs && K::dense == 1;
    static_assert(!kItemInRegs || K::emission == 0, "with an emission grid do_new writes the sample-buffer slot to the side array (no stash to park it from)");
    static_assert(NS <= 192, "ShleBanks holds 3 x 64 slots");
    static_assert(!kColdRegs || (K::emission == 0 && !K::tf), "ColdBanks: no marching-path access to the cold state (EmissionCache), C_COL not wired");
    __shared__ uint32_t lds_stage[kShleInRegs || kColdRegs ? kWgWaves * 64 : 4];
    uint32_t* const stage = lds_stage + (kShleInRegs || kColdRegs ? wave * 64 : 0);
    ShleBanks banks;
    banks.b[0] = banks.b[1] = banks.b[2] = v3{ 0, 0, 0 };
    banks.item[0] = banks.item[1] = banks.item[2] = 0u;
    ColdBanks cb;
    if (kColdRegs) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int f = 0; f < kBankFields; ++f) cb.v[k][f] = 0.0f;
    }
    constexpr int32_t HS = hot_stride<K>();
    constexpr bool kLazyEm = lazy_emission<K>();
    static_assert(VR_BATCH_REGS || !kLazyEm, "the VR_BATCH_REGS=0 swap path moves a marching path's radiance through its cold line: a `first` path of a lazy-emission kernel has none");
    __shared__ uint32_t lds_hot[kWgWaves * HS * NS];
    constexpr bool kWorld = world_slot<K>();
    HotStoreT<HS, kWorld> hs;
    hs.base = lds_hot + wave * (HS * NS);
    {   // wave-uniform: keep it in scalar registers
        const v3 ci = mat4_point(P.u.vol_density_inv_transform, v3{ P.u.cam_pos[0], P.u.cam_pos[1], P.u.cam_pos[2] });      // == first_resume
        hs.cam_ipos[0] = u2f(__builtin_amdgcn_readfirstlane(f2u(ci.x)));
        hs.cam_ipos[1] = u2f(__builtin_amdgcn_readfirstlane(f2u(ci.y)));
        hs.cam_ipos[2] = u2f(__builtin_amdgcn_readfirstlane(f2u(ci.z)));
    }
    // transfer function: the LUT (tf_size x vec4, 128 B for lut.txt, 4 KiB for a 256-entry colour map) is read twice per
    // tentative collision; one copy per workgroup in LDS replaces those global gathers.  Larger LUTs stay in global memory.
    __shared__ float lds_lut[K::tf ? 4 * kLutLdsEntries : 4];
    const bool lut_in_lds = K::tf && P.u.tf_size <= (uint32_t)kLutLdsEntries;
    const bool emission_on = K::emission == 2 ? P.u.has_emission != 0 : K::emission == 1;
    // the tail of the majorant table: cells [maj_first, maj_end) = the coarsest levels that fit (level offsets: vr_scene.h); maj_first = maj_end: none
    constexpr int32_t kMajCells = maj_lds_cells<K>();
    typedef typename std::conditional<K::tf, float, uint16_t>::type MajT;
    __shared__ MajT lds_maj[kMajCells > 0 ? kMajCells : 1];
    int32_t maj_first = 0x7FFFFFFF;
    if (kMajCells > 0) {
        const uint32_t k = (uint32_t)(P.density.mshift[0] + P.density.mshift[1] + P.density.mshift[2]);
        const int32_t maj_end = (int32_t)majorant_table_cells(k);         // the "outside" cell included
        maj_first = maj_end;
#pragma unroll
        for (int mip = 3; mip >= 0; --mip) {
            const int32_t off = (int32_t)majorant_level_offset(k, (uint32_t)mip);
            if (maj_end - off <= kMajCells) maj_first = off;
        }
        maj_first = __builtin_amdgcn_readfirstlane(maj_first);
        for (int32_t i = (int32_t)threadIdx.x; i < maj_end - maj_first; i += 64 * kWgWaves)
            lds_maj[i] = K::tf ? (MajT)P.density.majorant[maj_first + i] : (MajT)P.density.majorant16[maj_first + i];
    }
    if (K::tf && lut_in_lds)
        for (uint32_t i = threadIdx.x; i < 4u * P.u.tf_size; i += 64u * (uint32_t)kWgWaves) lds_lut[i] = P.tf_lut[i];
    if (K::tf || kMajCells > 0) __syncthreads();      // the only workgroup barrier of the kernel: before the persistent loop

    // scheduler thresholds, one byte each in two scalars: batch sizes that trigger NEW / NEE / POSTNEE / ESCAPE, the low-water
    // mark of live paths ("hungry"), the slots in use (diagnostic cap)
    const int32_t pool = (A.S.thr[ST_BEGIN] > 0 && A.S.thr[ST_BEGIN] < NS) ? A.S.thr[ST_BEGIN] : NS;
    const uint32_t thr_a = (uint32_t)(A.S.thr[ST_NEW] & 255) | ((uint32_t)(A.S.thr[ST_NEE] & 255) << 8) | ((uint32_t)(A.S.thr[ST_POSTNEE] & 255) << 16) | ((uint32_t)(A.S.thr[ST_ESCAPE] & 255) << 24);
    const uint32_t thr_b = (uint32_t)(A.S.thr[ST_MARCH] & 255) | ((uint32_t)pool << 8) | ((uint32_t)(A.S.thr[ST_COLLIDE] & 255) << 16);
#define VR_THR_NEW ((int32_t)(thr_a & 255u))
#define VR_THR_NEE ((int32_t)((thr_a >> 8) & 255u))
#define VR_THR_POST ((int32_t)((thr_a >> 16) & 255u))
#define VR_THR_ESC ((int32_t)(thr_a >> 24))
#define VR_THR_HUNGRY ((int32_t)(thr_b & 255u))
#define VR_POOL ((int32_t)((thr_b >> 8) & 255u))
#define VR_THR_COLLIDE ((int32_t)(thr_b >> 16))
    int32_t cnt_ready = 0, cnt_nee = 0, cnt_post = 0, cnt_esc = 0, cnt_free = pool;     // stack heights (wave-uniform)
#if VR_READY_FIFO
    int32_t rdy_head = 0;                                                               // READY ring: position of its oldest entry (wave-uniform)
#endif
    for (int32_t i = lane; i < pool; i += 64) q[Q_FREE * NS + i] = (uint8_t)i;
    __builtin_amdgcn_wave_barrier();

    WorkUnit wu;                      // current unit; .out is filled in where a sample is written (event_args().sbuf)
    wu.px0 = wu.py0 = 0; wu.first_sample = 1; wu.n_items = 0; wu.base = 0u; wu.out = nullptr;
    uint32_t cursor = 0u;             // next item of the current unit (wave-uniform)
    bool exhausted = false;           // the global queue has no more units
    uint32_t seg_tries = 0u;          // queue segments this wavefront has found empty (wave-uniform)

    Hot l;
    hot_init(l);
    int32_t slot = -1;                // path held in this lane's registers (-1: none)

    // VR_STATS_LEVEL (round 6): what the instrumented (STATS) instances carry.  2 (default): everything -- executions and lanes per block, cycles per block, pool
    // occupancy, the wavefront's timeline; those instances spill 3-28 vector registers to scratch and run ~25 % slower than the production kernels.  1: executions and
    // lanes per block, resumes / parks / iterations only -- what tests/tools_issue_budget.py weights the static instruction stream with -- so that those counts come
    // from kernels that schedule like production (no scratch: profiles/r6_kernel_resources.txt); a library built with it reports zero cycles and occupancy
#ifndef VR_STATS_LEVEL
#define VR_STATS_LEVEL 2
#endif
    constexpr bool STATS_T = STATS && VR_STATS_LEVEL >= 2;
    const unsigned long long t_begin_rt = STATS_T ? __builtin_amdgcn_s_memrealtime() : 0ull;
    uint32_t iters = 0u, idle_iters = 0u;     // scheduler iterations: in all (statistics), since the last finished path or pulled unit (watchdog)
    if (VR_PRIO_EVENTS != VR_PRIO_HOT) __builtin_amdgcn_s_setprio(VR_PRIO_HOT);      // (the priority the loop starts with)
    uint32_t t_last = (uint32_t)__builtin_readcyclecounter(), t_elapsed = 0u;
    // STATS instances: the counters of a wavefront live in LDS -- 32 words, word i = its index in the statistics buffer (2k / 2k+1 executions / lanes of block k,
    // 14 / 15 resumes / parks, 16 iterations, 17 wavefronts, 18 + k cycles of block k, 26 + k pool occupancy); lane 0 adds, in wave-uniform control flow.  Round 5
    // kept them as ~40 scalar variables: the instrumented kernels spilled 94-139 scalar and 3-28 vector registers (to scratch) and ran ~25 % slower than the
    // production ones, whose schedule their counts are supposed to describe (verdict r5 #6).  Any value carried around the scheduler loop in a REGISTER does that --
    // one more loop-carried VGPR tips these kernels (114-127 vector registers) over the allocator's edge -- the LDS words do not: at VR_STATS_LEVEL 1 the instrumented
    // instances have the production kernels' registers and no scratch (profiles/r6_kernel_resources.txt); the timers of level 2 still cost 3-24 spilled VGPRs.
    // Cycles are summed in 32 bits: instrumented launches are meant to be short (a wavefront may run 1.7 s before a block's sum wraps); the wavefront's lifetime
    // [25] is taken in 64 bits at the end.
    __shared__ uint32_t lds_stat[STATS ? kWgWaves * 32 : 1];
    if (STATS) { if (lane < 32) lds_stat[wave * 32 + lane] = 0u; __builtin_amdgcn_wave_barrier(); }
    auto stat_add = [&](int i, uint32_t n) __attribute__((always_inline)) { if (lane == 0) lds_stat[wave * 32 + i] += n; };
    uint32_t t_blk = 0u;
    const unsigned long long t_start = STATS_T ? __builtin_readcyclecounter() : 0ull;
#ifndef VR_STAT_SCHED
#define VR_STAT_SCHED 0
#endif
#if VR_STAT_SCHED
    uint32_t t_tail = 0u;
    bool have_tail = false;
#endif
    unsigned long long t_exhausted = 0ull;                 // STATS: constant-rate clock (100 MHz) when this wavefront found the work queue empty
#define VR_STAT(ST, N) do { if (STATS) { stat_add(2 * (ST), 1u); stat_add(2 * (ST) + 1, (uint32_t)(N)); if (STATS_T) t_blk = (uint32_t)__builtin_readcyclecounter(); } } while (0)
#define VR_STAT_END(ST) do { if (STATS_T) { stat_add(18 + (ST), (uint32_t)__builtin_readcyclecounter() - t_blk); } } while (0)
// push the slots of all lanes where COND holds onto stack QI (wave-synchronous)
#define VR_PUSH(QI, CNT, COND, SLOTV) do { \
        const uint64_t m_ = wave_ballot(COND); \
        if (m_) { if (COND) q[(QI) * NS + (CNT) + (int32_t)lane_rank(m_)] = (uint8_t)(SLOTV); (CNT) += popc(m_); } \
    } while (0)

// READY as a queue instead of a stack (build-time experiment, round 4, -DVR_READY_FIFO=1): a ring of NS bytes, oldest entry at rdy_head.  The hypothesis was
// that paths parked early starve at the bottom of the stack until the end of the launch and ARE its 4-5 ms fixed cost.  They are not: the per-wavefront
// timeline (tests/tools_wave_timeline.py, profiles/r4f_*) shows EVERY wavefront taking ~1.6 ms (c2) to finish its pool once the work queue is empty -- the
// latency of the deepest of its 175 paths, ~40 bounces x ~12 scheduler passes x ~3 us, stack or queue -- and the queue measures -0.5 % on full frames.
#ifndef VR_READY_FIFO
#define VR_READY_FIFO 0
#endif
#if VR_READY_FIFO
#define VR_PUSH_READY(COND, SLOTV) do { \
        const uint64_t m_ = wave_ballot(COND); \
        if (m_) { if (COND) { int32_t p_ = rdy_head + cnt_ready + (int32_t)lane_rank(m_); p_ = p_ >= NS ? p_ - NS : p_; q[Q_READY * NS + p_] = (uint8_t)(SLOTV); } cnt_ready += popc(m_); } \
    } while (0)
#else
#define VR_PUSH_READY(COND, SLOTV) VR_PUSH(Q_READY, cnt_ready, COND, SLOTV)
#endif
// route the batch paths to the stack of their new state; an impossible state is reported and the slot recycled
#define VR_ROUTE_ST(BS, STV) do { \
        /* one integer per lane (its new state, or -1 without a batch path): every ballot below is then a single v_cmp */ \
        const int32_t s_ = (BS) >= 0 ? (STV) : -1; \
        VR_PUSH_READY((uint32_t)(s_ - ST_MARCH) < 2u, BS);                        /* ST_MARCH, ST_COLLIDE */ \
        VR_PUSH(Q_NEE, cnt_nee, s_ == ST_NEE, BS); \
        VR_PUSH(Q_POST, cnt_post, s_ == ST_POSTNEE, BS); \
        VR_PUSH(Q_ESC, cnt_esc, s_ == ST_ESCAPE, BS); \
        const bool lost_ = s_ == ST_BEGIN || s_ > ST_ESCAPE || s_ < -1; \
        if (wave_ballot(lost_)) { if (lost_) atomicOr(event_args().status, 2u); } \
        const int32_t free0_ = cnt_free; \
        VR_PUSH(Q_FREE, cnt_free, s_ == ST_NEW || lost_, BS); \
        if (cnt_free != free0_) { idle_iters = 0u; t_elapsed = 0u; }          /* a path has ended: the watchdog starts over */ \
    } while (0)
#define VR_ROUTE(BS) VR_ROUTE_ST(BS, l.state)
#define VR_ROUTE_B(BS) VR_ROUTE_ST(BS, b.state)

    for (;;) {
        // watchdog (kMaxIdleIters above): a kernel must never hang the GPU
        ++iters;
        bool give_up = ++idle_iters > kMaxIdleIters;
        if ((iters & 1023u) == 0u) {
            // elapsed shader-clock time in units of 1024 ticks, summed over 1024-iteration windows (low 32 bits of the counter:
            // a window is a few million ticks).  A window that appears to take more than 2^31 ticks is a counter discontinuity
            // -- a wavefront that was saved and restored on another XCD when several processes time-share the GPU reads a
            // different counter -- and is not counted.
            const uint32_t now = (uint32_t)__builtin_readcyclecounter(), d = now - t_last;
            t_last = now;
            if (d < (1u << 31)) t_elapsed += d >> 10;
            give_up = give_up || t_elapsed > (uint32_t)(kMaxIdleTicks >> 10);
        }
        if (give_up) {
            if (lane == 0) atomicOr(event_args().status, 1u);
            break;
        }
#if VR_STAT_SCHED
        // diagnostic build: the occupancy counters carry the cycles of the scheduler's sections instead (tests/tools_sched_stats.py --sections)
        uint32_t t_sec = STATS_T ? (uint32_t)__builtin_readcyclecounter() : 0u;
        if (STATS_T && have_tail) stat_add(26 + 4, t_sec - t_tail);         // loop tail + head (watchdog, exit test)
#define VR_SECTION(K) do { if (STATS_T) { const uint32_t n_ = (uint32_t)__builtin_readcyclecounter(); stat_add(26 + (K), n_ - t_sec); t_sec = n_; } } while (0)
#else
#define VR_SECTION(K) do { } while (0)
#endif
        // (1) idle lanes resume READY paths
        {
            const uint64_t idle = wave_ballot(slot < 0);
            const int32_t take = min(popc(idle), cnt_ready);
            if (take > 0) {
                if (STATS) stat_add(14, 1u);
                if (slot < 0) {
                    const int32_t r = (int32_t)lane_rank(idle);
                    if (r < take) {
#if VR_READY_FIFO
                        int32_t p_ = rdy_head + r; p_ = p_ >= NS ? p_ - NS : p_;
                        slot = q[Q_READY * NS + p_]; hs.load_resume(l, slot, kWorld ? event_args().P.u.vol_density_inv_transform : nullptr);
#else
                        slot = q[Q_READY * NS + cnt_ready - 1 - r]; hs.load_resume(l, slot, kWorld ? event_args().P.u.vol_density_inv_transform : nullptr);
#endif
                        if (emission_on && !l.shadow) {              // EmissionCache (vr_trace.h Hot): the collisions of this segment add to L
                            if (kLazyEm && l.first) { l.ethr = v3{ 1.0f, 1.0f, 1.0f }; l.eL = hs.load_first_radiance(slot); }      // no cold line yet
                            else { const ColdT c = VR_COLD(slot); l.ethr = ld3(c, C_THR); l.eL = ld3(c, C_L); }
                        }
                    }
                }
                cnt_ready -= take;
#if VR_READY_FIFO
                rdy_head += take; rdy_head = rdy_head >= NS ? rdy_head - NS : rdy_head;
#endif
            }
        }
        VR_SECTION(0);                                                   // resume
#if !VR_STAT_SCHED
        if (STATS_T) { stat_add(26, (uint32_t)popc(wave_ballot(slot >= 0))); stat_add(27, (uint32_t)cnt_ready); stat_add(28, (uint32_t)cnt_nee); stat_add(29, (uint32_t)cnt_post); stat_add(30, (uint32_t)cnt_esc); stat_add(31, (uint32_t)cnt_free); }
#endif
        // (2) the hot pair: two DDA steps for the marching lanes, then the collision code for every lane that now stands at a
        // tentative collision (after two steps that is most of them, so both blocks run nearly full width).  Two memory round
        // trips per pass instead of four: both majorants are loaded together (the second step is prepared speculatively,
        // march_prep), and a tap's brick record and voxel are loaded together (brick-linear atlas, tap_load).  The loads
        // themselves are unconditional straight-line code between the exec-masked blocks (march_load / collide_load).
        // VR_HOT_PAIRS (round 5): the pair runs up to that many times per scheduler iteration -- straight-line copies, no loop -- so that the bookkeeping around
        // it (resume, park, batch decision: ~17 % of the kernel's VALU issue cycles, profiles/r5_issue_budget.txt) is paid once per VR_HOT_PAIRS passes; a
        // further copy only runs while at least VR_HOT_PAIR_MIN lanes still hold a marching or colliding path (lanes whose path reached an event idle through
        // it).  Measured (profiles/r5c_*, r5d_*): 2 / 3 / 4 / 6 copies c2 +3.5 / +3.6 / +3.4 / +3.2 %, c3 +3.6 %, c4 +0.5 ... +1 %, c5cloud +1.2 ... +4 %,
        // c5full +6.3 / +8.1 / +8.8 / +9.2 % (sparse grids: several march passes per collision); MIN 32 / 44 / 54 within the noise.  Round 2 had tried the
        // same as a LOOP around the march block (profiles/r2ac_*): the loop form cost more than the repetitions brought.
#ifndef VR_HOT_PAIRS
#define VR_HOT_PAIRS 4
#endif
        // ... per kernel instance (round 6): the transfer-function kernel with an emission grid -- 8 corner taps, the LUT and a stochastic emission tap in one collision
        // block -- spills 34 scalar registers with four clean copies and the general one, 20 with VR_HOT_PAIRS_TF_EMISSION copies (profiles/r6_kernel_resources.txt)
#ifndef VR_HOT_PAIRS_TF_EMISSION
#define VR_HOT_PAIRS_TF_EMISSION 4
#endif
        constexpr int kHotPairs = (K::tf && K::emission == 1) ? (VR_HOT_PAIRS_TF_EMISSION < VR_HOT_PAIRS ? VR_HOT_PAIRS_TF_EMISSION : VR_HOT_PAIRS) : VR_HOT_PAIRS;

#ifndef VR_HOT_PAIR_MIN
#define VR_HOT_PAIR_MIN 44
#endif
        // VR_DRAIN (round 6, build-time experiment): once the work queue is empty a wavefront's pool only shrinks, and the launch ends with the latency of the deepest
        // path (profiles/r4f_*).  1: a draining wavefront runs its further copies of the hot pair for any number of lanes and, when hungry, EVERY non-empty event batch
        // per iteration instead of the largest only (nothing is left to fill the others up)
#ifndef VR_DRAIN
#define VR_DRAIN 0
#endif
        // VR_BALLOT_VALID (round 5): `slot` does not change inside the hot pair, so "the lane holds a path" is ONE ballot per scheduler iteration and the pair's
        // counts are ballots of a single compare ANDed with it on the scalar unit; a ballot of `slot >= 0 && state == X` costs a v_cndmask + v_cmp more each
#ifndef VR_BALLOT_VALID
#define VR_BALLOT_VALID 1
#endif
        const uint64_t holds_path = wave_ballot(slot >= 0);
        // Clean segments (vr_trace.h seg_clean, round 5): while every path the wavefront holds is on one -- always, outside degenerate scenes -- the pair runs in
        // its CLEAN form (integer inside test, v_min3 in the DDA step, no NaN guard on the density tap); otherwise ONE pass in the general form.  A lane's path, and
        // with it the flag in the sign of its `far`, only changes in the resume block above: one ballot per scheduler iteration.
        // (VR_CLEAN_FORMS=0, the everything-at-run-time variant: the general form only -- a second form costs that kernel registers it does not have)
#ifndef VR_CLEAN_FORMS
#define VR_CLEAN_FORMS VR_CLEAN_FLAG
#endif
        // (VR_CLEAN_FORMS_TF_EMISSION, round 6: whether the transfer-function kernel with an emission grid carries the clean form too)
#ifndef VR_CLEAN_FORMS_TF_EMISSION
#define VR_CLEAN_FORMS_TF_EMISSION 1
#endif
        constexpr bool kCleanForms = VR_CLEAN_FORMS && ((K::tf && K::emission == 1) ? VR_CLEAN_FORMS_TF_EMISSION != 0 : true);
        const bool all_clean = kCleanForms && (holds_path & wave_ballot((int32_t)f2u(l.far) < 0)) == 0ull;
        // VR_EARLY_MARCH (round 6, build-time experiment, profiles/r6l_*): the NEXT copy's march loads are issued between this copy's tap loads and the code that consumes
        // the taps.  After collide_prep a path's next DDA steps are known whatever the tap says -- a null collision leaves it where it stands, one level finer
        // (collide_finish: mip = max(0, mip - 2)); a real one takes it out of the hot pair, and its steps are dropped -- so march_prep runs on that state and both
        // majorants travel while the tap does: one exposed round trip per pass instead of two.  The same values by the same operations (all tests green); only where
        // the loads are issued changes.  1: as described; 2: without the pin that keeps the tap's decode below the majorant loads.  Measured: -5 % c2, -5 ... -7 % c4
        // either way -- the pair does not run at the speed of its round trips.  Off.
#ifndef VR_EARLY_MARCH
#define VR_EARLY_MARCH 0
#endif
        constexpr bool kEarlyMarch = VR_EARLY_MARCH != 0 && VR_MARCH_SPECULATIVE && VR_MARCH_STEPS == 2 && K::global == 0 && !K::tf && K::emission == 0 && !K::maj_reuse && maj_lds_cells<K>() == 0;
        MarchIO pre;                       // kEarlyMarch: the next copy's steps and (in flight) majorants
        march_idle(pre); pre.maj1 = pre.maj2 = 0u;
        auto hot_pair = [&](auto clean_tag, const int hot_rep_, const int n_copies_) __attribute__((always_inline)) -> bool {
            constexpr bool CLEAN = decltype(clean_tag)::value;
#if VR_BALLOT_VALID
            if (hot_rep_ > 0 && popc_s(holds_path & wave_ballot((uint32_t)(l.state - ST_MARCH) < 2u)) < (VR_DRAIN && exhausted ? 1 : VR_HOT_PAIR_MIN)) return false;
#else
            if (hot_rep_ > 0 && popc(wave_ballot(slot >= 0 && (uint32_t)(l.state - ST_MARCH) < 2u)) < VR_HOT_PAIR_MIN) return false;
#endif
            if (STATS_T) t_blk = (uint32_t)__builtin_readcyclecounter();
            const bool is_m = slot >= 0 && l.state == ST_MARCH;
#if VR_MARCH_SPECULATIVE
            MarchIO mio;
            if (kEarlyMarch && hot_rep_ > 0) mio = pre;      // prepared and loaded by the copy before this one (below), for every lane that is marching now
            else {
            march_idle(mio);
            if (is_m) march_prep<K::dense, K::majb, CLEAN>(l, P, mio);
            if constexpr (kMajCells > 0) march_load_lds<K::tf, MajT>(P, mio, lds_maj, maj_first);
            else if constexpr (K::maj_reuse) march_load_reuse<K::tf>(P, mio, l);
            else march_load<K::tf>(P, mio);
#if VR_MARCH_LOADS_PINNED
            // Both majorants must have been REQUESTED before the first is used.  Left alone, the compiler sinks each load into the
            // conditional block of march_finish that consumes it (load, wait, test, load, wait: two dependent round trips); an
            // empty asm that takes both values as operands keeps the two loads above it, back to back.
            asm volatile("" : "+v"(mio.maj1), "+v"(mio.maj2));
#endif
            }
            if (is_m) march_finish<K::tf, K::maj_reuse, CLEAN>(l, P, mio);
#else
            for (int32_t k = 0; k < 2; ++k)            // diagnostic: two plain steps, one majorant load each, only where a step runs
                if (slot >= 0 && l.state == ST_MARCH) do_march<K::tf, K::dense, K::majb>(l, P);
#endif
            if (STATS) { const int32_t nm = popc(wave_ballot(is_m)); if (nm) { stat_add(2 * ST_MARCH, 1u); stat_add(2 * ST_MARCH + 1, (uint32_t)nm); } }
            if (STATS_T) { const uint32_t t_now = (uint32_t)__builtin_readcyclecounter(); stat_add(18 + ST_MARCH, t_now - t_blk); t_blk = t_now; }
#if VR_DIAG_PAD_VALU > 0
            {   // diagnostic: VR_DIAG_PAD_VALU extra dependent-free vector instructions per pass -> how issue-bound is the pass?
                float pad_ = l.t;
#pragma unroll
                for (int k_ = 0; k_ < VR_DIAG_PAD_VALU; ++k_) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(pad_));
                asm volatile("" :: "v"(pad_));
            }
#endif
#if VR_DIAG_PAD_SLEEP > 0
            __builtin_amdgcn_s_sleep(VR_DIAG_PAD_SLEEP);       // diagnostic: 64 * n idle cycles per pass -> how latency-bound is the wavefront?
#endif
            // The collision code runs when enough lanes stand at a tentative collision -- or when no lane is left marching.  In a
            // dense medium two DDA steps take most marching lanes to one (smoke.brick: 43 of 59); in a sparse grid (c5: 7.6 steps per
            // collision) a pass would otherwise run the collision code, the most expensive block of the loop, for a dozen lanes.
            // Lanes that wait keep their path; the marching lanes of the next pass join them.
            // A wavefront that is running dry (end of the launch: a handful of deep paths) must not make them wait for each other:
            // the threshold is at most half the lanes that hold a marching or colliding path.
            const bool is_c = slot >= 0 && l.state == ST_COLLIDE;
#if VR_BALLOT_VALID
            const int32_t n_c = popc(holds_path & wave_ballot(l.state == ST_COLLIDE));
            const int32_t n_m = popc(holds_path & wave_ballot(l.state == ST_MARCH));
#else
            const int32_t n_c = popc(wave_ballot(is_c));
            const int32_t n_m = popc(wave_ballot(slot >= 0 && l.state == ST_MARCH));
#endif
            const bool run_c = n_c > 0 && n_c >= min(VR_THR_COLLIDE, (n_c + n_m + 1) >> 1);
            if (kEarlyMarch) {
                // the pipelined form of the block below: tap loads, then the next copy's march loads, then the code that waits for the taps
                CollideIO<K> cio;
                collide_idle<K>(cio);
                if (run_c) {
                    if (is_c) collide_prep<K, CLEAN>(l, P, P, cio);
                    collide_load<K>(P, P, cio);
                }
                const bool early = hot_rep_ + 1 < n_copies_;
                if (early) {
                    const bool spec = run_c && is_c;                                   // the tentative collision this pass resolves: assume it is a null one
                    Hot ls = l;
                    ls.mipq = spec ? (l.mipq > 8 ? l.mipq - 8 : 0) : l.mipq;             // collide_finish: mip = max(0, mip - 2)
                    march_idle(pre);
                    if (slot >= 0 && (l.state == ST_MARCH || spec)) march_prep<K::dense, K::majb, CLEAN>(ls, P, pre);
                    march_load<K::tf>(P, pre);
                    // the tap is used from HERE on as far as the compiler is concerned: left alone it hoists the tap's decode (a conversion and a select) up to the
                    // tap's load -- and waits for it there, before the majorants above have been requested
                    if (VR_EARLY_MARCH == 1) asm volatile("" : "+v"(cio.d.raw), "+v"(cio.d.rmin), "+v"(cio.d.rdiff));
                }
                if (run_c) {
                    if (is_c) { ColdT c = VR_COLD(slot); collide_finish<K, ColdT, true>(l, c, P, P, cio, P.tf_lut); }
                    if (STATS) { stat_add(2 * ST_COLLIDE, 1u); stat_add(2 * ST_COLLIDE + 1, (uint32_t)n_c); }
                }
                // (the majorants are used HERE as far as the compiler is concerned: it may neither sink the loads into the next copy's march_finish nor drop them)
                if (early) asm volatile("" : "+v"(pre.maj1), "+v"(pre.maj2));
            } else
            if (run_c) {
                CollideIO<K> cio;
                collide_idle<K>(cio);
                const SceneParams& PE = VR_EMISSION_BY_POINTER && K::emission != 0 ? event_args().P : P;      // see collide_prep
                // (VR_COLLIDE_BY_POINTER, round 6: the run-time variant with a transfer function evaluates its collisions on uniforms read through the kernarg pointer too --
                // scalar loads in the collision code instead of ~20 more scalar registers held through the whole scheduler loop)
#ifndef VR_COLLIDE_BY_POINTER
#define VR_COLLIDE_BY_POINTER 1
#endif
                const SceneParams& PF = (VR_COLLIDE_BY_POINTER && K::tf && (K::global == 2 || K::emission == 1)) ? event_args().P : P;
                if (is_c) collide_prep<K, CLEAN>(l, PF, PE, cio);
                collide_load<K>(PF, PE, cio);
                if (is_c) {
                    ColdT c = VR_COLD(slot);
                    if (lut_in_lds) collide_finish<K, ColdT, true>(l, c, PF, PE, cio, lds_lut);      // two instances: LDS reads need the address space at compile time
                    else collide_finish<K, ColdT, true>(l, c, PF, PE, cio, PF.tf_lut);
                }
                if (STATS) { stat_add(2 * ST_COLLIDE, 1u); stat_add(2 * ST_COLLIDE + 1, (uint32_t)n_c); }
            }
            // every load of the pass has been consumed or belongs to a lane that left early: say so, or the compiler carries
            // "possibly outstanding" around the loop and waits where nothing is pending
            __builtin_amdgcn_s_waitcnt(0x0F70);                                         // vmcnt(0)
            if (STATS_T) stat_add(18 + ST_COLLIDE, (uint32_t)__builtin_readcyclecounter() - t_blk);
            return true;
        };
        if constexpr (kCleanForms) {
            if (all_clean) {
#pragma unroll
                for (int hot_rep_ = 0; hot_rep_ < kHotPairs; ++hot_rep_) if (!hot_pair(std::true_type{}, hot_rep_, kHotPairs)) break;
            } else hot_pair(std::false_type{}, 0, 1);
        } else {
#pragma unroll
            for (int hot_rep_ = 0; hot_rep_ < kHotPairs; ++hot_rep_) if (!hot_pair(std::false_type{}, hot_rep_, kHotPairs)) break;
        }
        VR_SECTION(1);                                                   // hot pair (also in st_cyc[MARCH] + st_cyc[COLLIDE])
        // (3) park paths that reached an event
        {
            // the state of a lane's path if it has to be parked (the hot pair can only leave a path in NEE, POSTNEE or ESCAPE), else -1:
            // one integer, so that every ballot below is a single v_cmp
            const int32_t ps = (slot >= 0 && l.state != ST_MARCH && l.state != ST_COLLIDE) ? l.state : -1;
            if (wave_ballot(ps >= 0)) {
                if (STATS) stat_add(15, 1u);
                if (ps >= 0) {
                    hs.save_marched(l, slot);
                    if (emission_on && !l.shadow) {                                                    // EmissionCache: L back to the cold line, or -- a path without one -- to its slot
                        if (kLazyEm && l.first) hs.save_first_radiance(l.eL, slot);
                        else { ColdT c = VR_COLD(slot); st3(c, C_L, l.eL); }
                    }
                }
                VR_PUSH(Q_NEE, cnt_nee, ps == ST_NEE, slot);
                VR_PUSH(Q_POST, cnt_post, ps == ST_POSTNEE, slot);
                VR_PUSH(Q_ESC, cnt_esc, ps == ST_ESCAPE, slot);
                if (ps >= 0) slot = -1;
            }
        }
        VR_SECTION(2);                                                   // park
        // (4) event batches
        int32_t n;
        const int32_t n_live = popc(wave_ballot(slot >= 0)) + cnt_ready;
        const bool hungry = n_live < VR_THR_HUNGRY;                    // the hot pair is about to run under-filled
        // a batch runs when it is full enough; a hungry wave additionally runs its LARGEST batch (only that one, so that the
        // others keep filling up)
        const int32_t c_new = exhausted ? 0 : cnt_free;
        int32_t big = c_new;
        if (cnt_nee > big) big = cnt_nee;
        if (cnt_post > big) big = cnt_post;
        if (cnt_esc > big) big = cnt_esc;
        const bool drain_all = VR_DRAIN && exhausted;
        const bool want_new = c_new > 0 && (c_new >= VR_THR_NEW || (hungry && c_new == big));
        const bool want_nee = cnt_nee > 0 && (cnt_nee >= VR_THR_NEE || (hungry && (cnt_nee == big || drain_all)));
        const bool want_post = cnt_post > 0 && (cnt_post >= VR_THR_POST || (hungry && (cnt_post == big || drain_all)));
        const bool want_esc = cnt_esc > 0 && (cnt_esc >= VR_THR_ESC || (hungry && (cnt_esc == big || drain_all)));
        if (want_new || want_nee || want_post || want_esc) {
            // the lanes double as batch workers.  VR_BATCH_REGS=1: the batch path lives in its own register set `b` and the
            // marching path `l` stays put; =0: the marching path is saved to its LDS slot and `l` is reused (fewer VGPRs)
#if VR_BATCH_REGS
            Hot b;
#else
            const int32_t my_slot = slot;
            // a lane that WAITS at a tentative collision (fewer than the collide threshold stand there) needs its step's majorant back after the batch: the parked
            // hot state does not hold it (a parked path is never in that state).  Lost until round 4 -- the real / null decision then tested against 0: results
            // of this variant depended on the scheduler's thresholds (found by test_emission_grid_with_a_different_brick_layout)
            const float my_majorant = l.majorant;
            if (my_slot >= 0) { hs.save(l, my_slot); if (emission_on && !l.shadow) { ColdT c = VR_COLD(my_slot); st3(c, C_L, l.eL); } }
            __builtin_amdgcn_wave_barrier();
            Hot& b = l;
#endif
            if (VR_PRIO_EVENTS != VR_PRIO_HOT) __builtin_amdgcn_s_setprio(VR_PRIO_EVENTS);
            if (want_esc) {
                n = min(64, cnt_esc);
                VR_STAT(ST_ESCAPE, n);
                int32_t bs = -1;
                if (lane < n) { bs = q[Q_ESC * NS + cnt_esc - 1 - lane]; hs.load(b, bs); if (kLazyEm && b.first) b.eL = hs.load_first_radiance(bs); }
                if (kItemInRegs) b.item = item_fetch(banks, lane, bs);                      // all lanes
                if constexpr (kColdRegs) {
                    ColdLocal c;
                    cold_clear(c);
                    cold_fetch<kColdEscR>(c, cb, lane, bs);                                  // all lanes (a path that never scattered reads its slot's leftovers and discards them)
                    if (lane < n) {
                        const KernelArgs& E = event_args();
                        WorkUnit w; w.out = E.sbuf;
                        do_escape<ColdLocal, false>(b, c, E.P, w);
                    }
                } else
                if (lane < n) {
                    // a path that never scattered carries what it needs in its stash (FirstStash): its loads go to slot 0's line, shared by the batch
                    const ColdT c = VR_COLD(b.first ? 0 : bs);
                    const KernelArgs& E = event_args();
                    WorkUnit w; w.out = E.sbuf;
                    do_escape<ColdT, kItemInRegs, kLazyEm, kWorld>(b, c, E.P, w);          // writes the sample; the slot becomes free
                }
                cnt_esc -= n;
                VR_ROUTE_B(bs);                                            // ST_NEW: the slot is free again
                VR_STAT_END(ST_ESCAPE);
            }
            if (want_post) {
                n = min(64, cnt_post);
                VR_STAT(ST_POSTNEE, n);
                int32_t bs = -1;
                if (lane < n) { bs = q[Q_POST * NS + cnt_post - 1 - lane]; hs.load(b, bs); }
                if (kShleInRegs) b.shle = shle_fetch(banks, lane, bs);                     // all lanes: the values come from their home lanes
                if (kItemInRegs) b.item = item_fetch(banks, lane, bs);
                if constexpr (kColdRegs) {
                    ColdLocal c;
                    cold_clear(c);
                    cold_fetch<kColdPostR>(c, cb, lane, bs);
                    if (lane < n) {
                        const KernelArgs& E = event_args();
                        WorkUnit w; w.out = E.sbuf;
                        do_postnee<K, ColdLocal, false, false>(b, c, E.P, w);
                        hs.save(b, bs);
                    }
                    cold_store<kColdPostW, false>(cb, cold_stage(stage, lane, bs, false), lane, c);      // a path that has ended leaves leftovers in a free slot: harmless
                } else
                if (lane < n) {
                    ColdT c = VR_COLD(bs);
                    const KernelArgs& E = event_args();
                    WorkUnit w; w.out = E.sbuf;
                    do_postnee<K, ColdT, kShleInRegs, kItemInRegs>(b, c, E.P, w);
                    hs.save(b, bs);
                }
                cnt_post -= n;
                VR_ROUTE_B(bs);                                            // ST_NEW = path ended (bounce cap / roulette)
                VR_STAT_END(ST_POSTNEE);
            }
            if (want_new) {
                if (cursor == (uint32_t)wu.n_items) {
                    const KernelArgs& E = event_args();
                    uint32_t j = 0xFFFFFFFFu;
                    while (seg_tries < kQueueSegments) {                    // own segment first, then the following ones
                        const uint32_t k = ((blockIdx.x & (kQueueSegments - 1u)) + seg_tries) & (kQueueSegments - 1u);
                        const uint32_t lo = k * E.D.seg_len, hi = min(lo + E.D.seg_len, E.D.n_units);
                        uint32_t v = 0xFFFFFFFFu;
                        if (lo < hi) { if (lane == 0) v = atomicAdd(E.D.unit_counter + k, 1u); v = __builtin_amdgcn_readfirstlane(v); }
                        if (lo < hi && v < hi - lo) { j = lo + v; break; }
                        ++seg_tries;                                        // this segment is used up for good
                    }
                    if (j == 0xFFFFFFFFu) { exhausted = true; if (STATS_T) t_exhausted = __builtin_amdgcn_s_memrealtime(); }
                    else { wu = make_unit(E.D, E.P.u.resolution[0], j, nullptr); cursor = 0u; idle_iters = 0u; t_elapsed = 0u; }
                }
                n = min(min(64, cnt_free), (int32_t)((uint32_t)wu.n_items - cursor));
                if (n > 0) {
                    VR_STAT(ST_NEW, n);
                    int32_t bs = -1;
                    if (lane < n) {
                        bs = q[Q_FREE * NS + cnt_free - 1 - lane];
                        hot_init(b);
                        ColdT c = VR_COLD(bs);
                        do_new<K, ColdT, kLazyEm>(b, c, event_args().P, wu, cursor + (uint32_t)lane);
                        hs.save_new(b, bs);
                        if (kLazyEm) hs.save_first_radiance(v3{ 0.0f, 0.0f, 0.0f }, bs);
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
                bool was_first = false;
                uint32_t first_item = 0u;
                if (lane < n) { bs = q[Q_NEE * NS + cnt_nee - 1 - lane]; hs.load(b, bs); }
                if constexpr (kColdRegs) {
                    ColdLocal c;
                    cold_clear(c);
                    cold_fetch<kColdNeeR>(c, cb, lane, bs);
                    if (lane < n) {
                        was_first = b.first != 0;
                        do_nee<K, ColdLocal, false, false>(b, c, c, event_args().P);        // a first collision writes every field (its reads are discarded)
                        hs.save(b, bs);
                    }
                    const uint32_t w = cold_stage(stage, lane, bs, was_first);
                    cold_store<kColdNeeW, false>(cb, w, lane, c);
                    if (wave_ballot(was_first)) cold_store<kColdNeeWFirst, true>(cb, w, lane, c);
                } else
                if (lane < n) {
                    was_first = b.first != 0; first_item = f2u(b.Tr);      // a path's first collision: its sample-buffer slot is in the stash
                    if (kLazyEm && b.first) b.eL = hs.load_first_radiance(bs);
                    ColdT c = VR_COLD(bs);
                    const ColdT crd = VR_COLD(b.first ? 0 : bs);      // first scatter of a path: nothing to read yet (do_nee)
                    do_nee<K, ColdT, kShleInRegs, kItemInRegs, kLazyEm>(b, c, crd, event_args().P);
                    hs.save(b, bs);
                }
                if (kShleInRegs) shle_park<kItemInRegs>(banks, stage, lane, bs >= 0, bs, b.shle, was_first, first_item);      // all lanes: the light samples go to their slots' home lanes
                cnt_nee -= n;
                VR_ROUTE_B(bs);
                VR_STAT_END(ST_NEE);
            }
#if !VR_BATCH_REGS
            __builtin_amdgcn_wave_barrier();
            if (my_slot >= 0) { hs.load_resume(l, my_slot); l.majorant = my_majorant; if (emission_on && !l.shadow) { const ColdT c = VR_COLD(my_slot); l.ethr = ld3(c, C_THR); l.eL = ld3(c, C_L); } }
#endif
        }
        if (VR_PRIO_EVENTS != VR_PRIO_HOT) __builtin_amdgcn_s_setprio(VR_PRIO_HOT);
        VR_SECTION(3);                                                   // batch decision + event batches (the events' own cycles are in st_cyc)
#if VR_STAT_SCHED
        t_tail = t_sec; have_tail = true;
#endif
        if (exhausted && cnt_free == VR_POOL) break;                        // every path of the pool has finished
    }
    unsigned long long* const stats = A.stats;
    if (STATS && stats) {
        stat_add(16, iters);
        stat_add(17, 1u);
        __builtin_amdgcn_wave_barrier();
        if (lane < 32 && lane != 25) atomicAdd(&stats[lane], (unsigned long long)lds_stat[wave * 32 + lane]);
        if (STATS_T && lane == 0) {
            atomicAdd(&stats[25], __builtin_readcyclecounter() - t_start);
            // per wavefront (diagnostic, tests/tools_wave_timeline.py): when it started, found the queue empty and ended, on the GPU's constant 100 MHz clock
            stats[kStatsWaveBase + 3u * wave_index] = t_begin_rt;
            stats[kStatsWaveBase + 3u * wave_index + 1u] = t_exhausted;
            stats[kStatsWaveBase + 3u * wave_index + 2u] = __builtin_amdgcn_s_memrealtime();
        }
    }
#undef VR_STAT
#undef VR_COLD
#undef VR_THR_NEW
#undef VR_THR_NEE
#undef VR_THR_POST
#undef VR_THR_ESC
#undef VR_THR_HUNGRY
#undef VR_POOL
#undef VR_THR_COLLIDE
#undef VR_STAT_END
#undef VR_SECTION
#undef VR_PUSH
#undef VR_PUSH_READY
#undef VR_ROUTE
#undef VR_ROUTE_B
#undef VR_ROUTE_ST
}

}  // namespace vr
