// vr_trace.h -- the volumetric path tracer as a per-lane state machine.
//
// What it computes: exactly one pixel's samples of the reference kernels
// shader/pathtracer_brick.glsl:23-37 / pathtracer_brick_tf.glsl:24-38 -> common.glsl trace_path (:599-652),
// with the same RNG stream, the same draw order and the same arithmetic (see vr_math.h) as the reference's
// recursive/looping formulation.  How it is organised is different on purpose: the GLSL runs one dispatch per
// sample with nested data-dependent loops, which on a 64-wide wavefront leaves most lanes idle most of the
// time.  Here each lane carries an explicit state and the wavefront repeatedly executes ONE state's code for
// all lanes that are in it (scheduler in vr_kernels.hip), so that
//   * the DDA march step is shared by camera/scatter segments (sample_volumeDDA, :458-501) and shadow
//     segments (transmittanceDDA, :412-455): both are "mode" flags of one loop body,
//   * lanes are not tied to pixels: a wavefront owns a pool of (pixel, sample) items (one 8x8 tile x a chunk of
//     samples) and a lane that finishes a path immediately pulls the next item, so no lane idles because its
//     pixel was cheap; per-sample radiances go to a sample buffer and a second tiny kernel applies the running
//     mean mix(old, new, 1/s) of pathtracer_brick.glsl:36 in sample order (bit-identical to sequential dispatches), and
//   * rare, expensive events (NEE environment sampling, new-sample setup with the 32-round TEA hash, escape
//     lookups) are batched until enough lanes want them.
// The order in which lanes run their states never changes a result: every lane owns its RNG state.
#pragma once

#include "vr_math.h"
#include "vr_scene.h"

// Memory-access hooks of the HOST build used by tests/tools_l2_breakdown.py (round 6: which class of access misses the L2?).  Compiled to nothing everywhere else --
// the product and the device never see them; tests/hostkernel/host_kernel.cpp defines the sink when it is built with -DVR_HOST_TRACE.
#if defined(VR_HOST_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
namespace vr {
enum TraceClass : int32_t { TR_MAJORANT = 0, TR_TAP = 1, TR_ENV_WARP = 2, TR_ENV_TEXEL = 3 };
void host_trace(int32_t cls, const void* table, size_t a, size_t b);       // majorant: (cell index, -); tap: (cell, voxel offset); environment: (byte offset, bytes)
}
#define VR_TRACE(CLS, TABLE, A, B) ::vr::host_trace((CLS), (TABLE), (size_t)(A), (size_t)(B))
#else
#define VR_TRACE(CLS, TABLE, A, B) do { } while (0)
#endif

namespace vr {

enum LaneState : int32_t {
    ST_NEW = 0,      // needs a work item: seed, camera ray, first segment set-up
    ST_BEGIN = 1,    // (folded into NEW / NEE / POSTNEE: begin_segment) -- kept as an index for scheduler parameters
    ST_MARCH = 2,    // one DDA step over the majorant mips
    ST_COLLIDE = 3,  // tentative collision: density lookup, real/null decision
    ST_NEE = 4,      // real scatter: advance, sample the environment, set up the shadow segment
    ST_POSTNEE = 5,  // shadow segment done: add direct light, bounce cap, roulette, phase sample, next segment
    ST_ESCAPE = 6,   // path left the volume: environment lookup + MIS, write the sample
    ST_DONE = 7,
    ST_COUNT = 8
};

// Compile-time configuration of the lane code.  One kernel is instantiated per combination that matters for speed, so that
// code (and registers) of variants a scene does not use stay out of its kernel; 2 = "decided at run time from the uniforms"
// (the host harness and the rarely used variants).
//   TF        transfer-function kernel (pathtracer_brick_tf.glsl) or not (pathtracer_brick.glsl)
//   GLOBAL    0: DDA trackers (USE_DDA, both reference kernels)  1: global-majorant trackers (common.glsl:333-394)  2: run time
//   EMISSION  0: no emission grid bound (the 9 draws of lookup_emission are still consumed)  1: bound  2: run time
//   DENSE     0: density grid = bricks  1: dense fp16 voxels  2: run time
//   MAJB      layout of the majorant table's levels 0-1 (vr_scene.h majorant_cell_index): 0 linear, 1 in 4x4x4-cell blocks of one cache line, 2 run time
//             (GridView::maj_blocked, chosen per grid at commit(): round 5)
#ifndef VR_MARCH_STEPS
#define VR_MARCH_STEPS 2     /* DDA steps a march pass prepares and loads together (march_prep below) */
#endif
#ifndef VR_MAJ_REUSE
#define VR_MAJ_REUSE 0       /* build-time experiment (round 5, profiles/r5f_*): 0 never (default: c5cloud +-0, c2 -0.6 %), 1 in the kernel for blocked majorant tables (variant 4), 2 in every kernel */
#endif
template <bool TF, int GLOBAL, int EMISSION, int DENSE, int MAJB = 0>
struct TraceCfg {
    static constexpr bool tf = TF;
    static constexpr int global = GLOBAL, emission = EMISSION, dense = DENSE, majb = MAJB;
    static constexpr int edense = EMISSION == 1 ? DENSE : 2;      // a kernel with a compiled-in emission grid takes it in the same form as the density grid
    // that kernel, on brick grids, reads both grids from one paired atlas (vr_scene.h): component 1 = density, 2 = emission, 0 = a grid's own atlas
    static constexpr int pair_d = (VR_PAIRED_ATLAS && EMISSION == 1 && DENSE == 0) ? 1 : 0, pair_e = pair_d ? 2 : 0;
    // the kernel compiled for the large sparse grids (blocked majorant table: the grid whose gathers leave the caches) remembers the last majorant it used
    static constexpr bool maj_reuse = VR_MARCH_STEPS == 2 && (VR_MAJ_REUSE == 2 || (VR_MAJ_REUSE == 1 && MAJB == 1));
};

// VR_WORLD_SLOT (round 6, build-time experiment): a parked path keeps the WORLD origin and direction of its segment in its slot instead of the index-space ones
// (which the resume recomputes, as begin_segment computed them), so that the collision event needs nothing from the path's cold line -- position and direction are in
// the slot, and "throughput *= albedo" moves to the scatter event that always follows -- and reads none: one of a bounce's two cold line reads and the first dependent
// round trip of every collision event gone, for two transforms per resume and one more sector written per bounce.  Kernels of the DDA trackers without a transfer
// function only (a transfer function's collision colour waits in the slot's direction fields).
#ifndef VR_WORLD_SLOT
#define VR_WORLD_SLOT 1
#endif
#ifndef VR_WORLD_SLOT_EMISSION
#define VR_WORLD_SLOT_EMISSION 0     /* the emission kernels (127 vector registers as they are) spill 6 of them with it: measured separately (profiles/r6g_*) */
#endif
template <class K> constexpr bool world_slot() { return VR_WORLD_SLOT != 0 && !K::tf && K::global == 0 && (K::emission == 0 || VR_WORLD_SLOT_EMISSION != 0); }

// the pool of work items of one wavefront: pixel p = item & 63 of the 8x8 tile at (px0, py0), sample
// first_sample + (item >> 6) (1-based like the reference's current_sample)
struct WorkUnit {
    int32_t px0, py0;
    int32_t first_sample;
    int32_t n_items;
    uint32_t base;           // index of the unit's first item in the sample buffer
    float* out;              // sample buffer: RGBA32F per item, trace_path's vec4(L, alpha), unsanitised
};

// state of a path whose segment has ended without a real collision: a shadow segment (shadow = 1) goes to POSTNEE, a camera /
// scatter segment (0) to ESCAPE -- one subtraction instead of compare + select
static_assert(ST_POSTNEE == ST_ESCAPE - 1, "segment_end_state");
VR_HD int32_t segment_end_state(int32_t shadow) { return ST_ESCAPE - shadow; }

// Per-lane state is split by temperature.
// Hot: what the DDA march / collision loop touches every iteration -- lives in registers.
struct Hot {
    uint32_t seed;
    v3 ipos, idir, ri;       // index-space ray of the current segment
    v3 wpos, wdir;           // the same segment's origin and direction in world space (begin_segment's arguments): what a parked path keeps under VR_WORLD_SLOT
    float t, far, tau, majorant, Tr;
    int32_t mipq;            // 4 * mip: the DDA level moves in quarter steps (common.glsl:433,450), kept as an integer
    int32_t shadow;          // 0: sample_volumeDDA segment, 1: transmittanceDDA segment
    int32_t state;
    int32_t first;           // 1: the camera segment of a new path whose cold line has not been written yet (see FirstStash)
    // emission kernels on the device only (EmissionCache): the path's throughput and radiance while it marches a camera / scatter
    // segment -- every tentative collision there adds emitted light to L -- loaded when the path is resumed, L stored back when it
    // is parked; the same additions in the same order as on the cold line, without a load-load-store per collision
    v3 ethr, eL;
    // radiance of the pending light sample, do_nee -> do_postnee, for the schedulers that park it in vector registers
    // (SHLE_IN_HOT, vr_pathtrace.h ShleBanks); otherwise it goes through the side array (C_SHLE)
    v3 shle;
    uint32_t item;           // same schedulers: the path's slot in the sample buffer (otherwise C_ITEM of the side array)
    // kernels with majorant reuse (round 5, TraceCfg::maj_reuse): the table index and the raw table word of the majorant this LANE used last.  The table does not
    // change during a launch, so the pair stays valid whatever path the lane holds: a step that asks for the same cell again -- the walk restarting in place
    // after a null collision on the finest level -- takes the word from here and sends its load to cell 0, a line all such lanes share
    int32_t maj_idx;
    uint32_t maj_raw;
};
// A new path needs nothing of its cold line until its first event: position = the camera's, throughput 1, radiance 0, no
// scatter yet.  What it does need there -- its world direction and its slot in the sample buffer -- waits in the path's hot
// storage, in the places of two fields that are redundant during a camera segment: ipos (the same for every camera ray:
// first_resume recomputes it) and Tr (1).  So do_new writes no cold line, a path that escapes without scattering (or misses
// the box) never touches one, and the first scatter event writes the line without having to read it: for the bench scene
// 0.7 line fetches and 0.7 partial line writes fewer per sample, from memory that sits beyond the L2.
// On the GPU the stash is the parked path's LDS slot (HotStore, vr_pathtrace.h); the host harness keeps it in a FirstStash.
// (Kernels with VR_WORLD_SLOT keep every path's world direction in its slot anyway -- Hot::wdir -- and their events read it from there, first path or not: for them the
// stash is the sample slot in Tr alone.)
struct FirstStash { v3 dir; uint32_t item; };
// Cold: path state that only the rare events (new sample, scatter, escape) read or write.  On the GPU it is parked
// in LDS ([field][lane] dwords, conflict-free) so that it does not occupy registers while the lane marches; the
// host harness backs it with a plain array.  `Cold` is any type with float ld(int) / void st(int, float).
// The 16 floats that live as long as the path fill a 64-byte slot, half a cache line, in two 32-byte sectors (the granularity at
// which the L2 writes a dirty line back) sorted by WHO WRITES them; adjacent accesses fuse into dwordx4:
//   sector 0 (the collision event, NEE):    (pos, sh_pdf) (thr, f_p of the light sample)
//   sector 1 (the scatter event, POSTNEE):  (L, n_paths) (dir, f_p of the scattered direction)
// A separate compact array ("side", fields >= C_SIDE: 16 bytes per path; Cold types map the two ranges to their storage) holds the
// path's slot in the sample buffer, which only its first collision writes and its end reads, and has room for the radiance of the
// pending light sample (Hot::shle, 3 floats, collision event -> scatter event) for the schedulers that park it in memory.  The
// weight thr * mis * f_p of that sample is recomputed by the scatter event from thr, sh_pdf and f_p (same operations on the same
// values as when the collision event stored it).  History: one 128-byte line per path with all 24 floats (c4: 2.11x the
// algorithmic bytes moved), fields ordered by writer (1.80x), 64-byte slots (profiles/r2y_*).
#ifndef VR_C_STRIDE
#define VR_C_STRIDE 16          // floats between two slots of the cold workspace (a build-time experiment of round 6 spreads them: profiles/r6k_*)
#endif
enum ColdField : int32_t {
    C_POS = 0, C_SHPDF = 3, C_THR = 4, C_FPL = 7, C_L = 8, C_NPATHS = 11, C_DIR = 12, C_FP = 15,
    C_SIDE = 16, C_SHLE = 16, C_ITEM = 19,
    C_COL = 20,                 // transfer-function kernels: colour of the real collision, collide_finish -> do_nee (device: the parked path's LDS slot)
    C_COUNT = 23, C_STRIDE = VR_C_STRIDE, C_SIDE_STRIDE = 4
};
template <class Cold> VR_HD v3 ld3(const Cold& c, int32_t f) { return v3{ c.ld(f), c.ld(f + 1), c.ld(f + 2) }; }
template <class Cold> VR_HD void st3(Cold& c, int32_t f, v3 v) { c.st(f, v.x); c.st(f + 1, v.y); c.st(f + 2, v.z); }
template <class Cold> VR_HD uint32_t ldu(const Cold& c, int32_t f) { return f2u(c.ld(f)); }
// three fields and a fourth that share 16 bytes of the slot: ONE load where the slot is in memory (vr_pathtrace.h overloads this for its global-memory slots)
struct Quad { v3 a; float b; };
template <class Cold> VR_HD Quad ld4(const Cold& c, int32_t f3, int32_t f1) { return Quad{ ld3(c, f3), c.ld(f1) }; }
template <class Cold> VR_HD void stu(Cold& c, int32_t f, uint32_t v) { c.st(f, u2f(v)); }

// ---------------------------------------------------------------------------------------------------
// RNG  (common.glsl:40-67)
// Fully unrolled on the device (round 5): the round's key schedule s0 = n * delta folds into the literal of ONE add per half-round (v1 + s0 + k would otherwise be an
// add with the loop's scalar register -- 0.9 cycles dearer than a literal operand on gfx950 -- and a second one): 6 instead of 7 instructions per half-round, 384 in all
#ifndef VR_TEA_UNROLL
#define VR_TEA_UNROLL 32
#endif
VR_HD uint32_t tea32(uint32_t v0, uint32_t v1) {
    uint32_t s0 = 0u;
#pragma unroll VR_TEA_UNROLL
    for (int n = 0; n < 32; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xA341316Cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xC8013EA4u);
        v1 += ((v0 << 4) + 0xAD90777Du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7E95761Eu);
    }
    return v0;
}
VR_HD float rng(uint32_t& s) {
    s = s * 1664525u + 1013904223u;
    return (float)(s & 0x00FFFFFFu) * (1.0f / 16777216.0f);   // exact: same value as / float(0x01000000)
}
// advance the LCG by 9 draws (lookup_emission's stochastic filter when no emission grid is bound)
VR_HD void rng_skip9(uint32_t& s) {
    constexpr uint32_t a = 1664525u, c = 1013904223u;
    constexpr uint32_t a2 = a * a, a4 = a2 * a2, a8 = a4 * a4, a9 = a8 * a;
    constexpr uint32_t g2 = a + 1u, g4 = g2 * (a2 + 1u), g8 = g4 * (a4 + 1u), g9 = g8 * a + 1u;   // 1+a+...+a^8
    s = a9 * s + g9 * c;
}

// ---------------------------------------------------------------------------------------------------
// grids  (common.glsl:268-297); out-of-range fetches read 0 (GL: undefined)
//
// Every fetch is split in three: where the voxel lives (tap_addr: arithmetic only) -> the loads (tap_load) -> the value
// (tap_value).  The scheduler (vr_pathtrace.h) runs the first two for all lanes of a pass before anything waits, so that one
// memory round trip serves the DDA steps and the collisions of the whole wavefront.  The brick atlas is brick-linear
// (block of brick record i = bytes [640 i, 640 i + 640), vr_scene.h): range and voxel come from one cache line, no dependent pointer chase.
template <int DENSE>
VR_HD bool grid_is_dense(const GridView& g) { return DENSE == 2 ? g.dense != nullptr : DENSE == 1; }
struct TapAddr { uint32_t cell, off; bool in; };      // bricks: record index, byte inside the 8^3 block; dense: 4x4x4 block index, voxel inside it
struct TapData { float rmin, rdiff; uint32_t raw; };  // bricks: range of the brick + the u8; dense: the fp16 bits
template <int DENSE = 2>
VR_HD TapAddr tap_addr(const GridView& g, int32_t x, int32_t y, int32_t z) {
    TapAddr a;
    const bool nonneg = (x | y | z) >= 0;
    if (grid_is_dense<DENSE>(g)) {          // dense fp16 grid: one 2-byte load, no indirection
        a.in = nonneg && (uint32_t)x < (uint32_t)g.dim[0] && (uint32_t)y < (uint32_t)g.dim[1] && (uint32_t)z < (uint32_t)g.dim[2];
#if defined(VR_WHATIF_WRAP)
        // diagnostic build (tests/tools_whatif_wrap.py): the grid holds a field of period VR_WHATIF_WRAP voxels and the taps read its
        // first period only -- same values, a working set that fits a cache level: what would voxel taps cost if they never missed?
        x &= VR_WHATIF_WRAP - 1; y &= VR_WHATIF_WRAP - 1; z &= VR_WHATIF_WRAP - 1;
#endif
        // 4x4x4 blocks (vr_scene.h): neighbouring rays and the +-2-voxel stochastic taps share 128-byte lines; block counts < 2^14 per axis
        a.cell = (mul24((uint32_t)z >> 2, (uint32_t)g.dblk[1]) + ((uint32_t)y >> 2)) * (uint32_t)g.dblk[0] + ((uint32_t)x >> 2);
        a.off = (((uint32_t)z & 3u) << 4) | (((uint32_t)y & 3u) << 2) | ((uint32_t)x & 3u);
    } else {
        const uint32_t bx = (uint32_t)x >> 3, by = (uint32_t)y >> 3, bz = (uint32_t)z >> 3;
        a.in = nonneg && bx < (uint32_t)g.nb[0] && by < (uint32_t)g.nb[1] && bz < (uint32_t)g.nb[2];
        a.cell = mul24(mul24(bz, (uint32_t)g.nb[1]) + by, (uint32_t)g.nb[0]) + bx;      // two v_mad_u32_u24
        a.off = (((uint32_t)z & 7u) << 6) | (((uint32_t)y & 7u) << 3) | ((uint32_t)x & 7u);
    }
    if (!a.in) { a.cell = 0u; a.off = 0u; }       // the loads are unconditional: an outside tap reads cell 0 and is discarded
    return a;
}
#ifndef VR_TAP_LINE_INDEX
#define VR_TAP_LINE_INDEX 1
#endif
template <int DENSE = 2, int PAIR = 0>
VR_HD TapData tap_load(const GridView& g, TapAddr a) {
    TapData d;
    VR_TRACE(1, g.atlas ? (const void*)g.atlas : (const void*)g.dense, a.cell, a.off);
    if (PAIR != 0) {
        // paired atlas: ten lines of [rmin_d, rdiff_d, rmin_e, rdiff_e | 56 x (density, emission)] per brick
        const uint32_t line = pair_voxel_line(a.off);
#if VR_TAP_LINE_INDEX
        const uint8_t* ln = g.atlas + ((size_t)(a.cell * (kPairBlockBytes / 128u) + line) << 7);      // see the brick atlas below
#else
        const uint8_t* ln = g.atlas + ((size_t)a.cell * kPairBlockBytes + (size_t)(line * 128u));
#endif
        const float* rec = reinterpret_cast<const float*>(ln) + (PAIR == 2 ? 2 : 0);
#if defined(VR_TAP_NT) && defined(__HIP_DEVICE_COMPILE__)
        // build-time experiment (round 5): the paired atlas of a large grid streams through the L2 (450 MB touched on c5cloud); non-temporal taps would leave the L2
        // to the majorant and environment tables
        d.rmin = __builtin_nontemporal_load(rec); d.rdiff = __builtin_nontemporal_load(rec + 1);
        d.raw = __builtin_nontemporal_load(ln + (kPairLineHeader + 2u * (a.off - line * kPairLineVoxels) + (PAIR == 2 ? 1u : 0u)));
#else
        d.rmin = rec[0]; d.rdiff = rec[1];
        d.raw = ln[kPairLineHeader + 2u * (a.off - line * kPairLineVoxels) + (PAIR == 2 ? 1u : 0u)];
#endif
        return d;
    }
    if (grid_is_dense<DENSE>(g)) {
        d.rmin = 0.0f; d.rdiff = 0.0f;
#if defined(VR_DENSE_TAP_NT) && defined(__HIP_DEVICE_COMPILE__)
        d.raw = __builtin_nontemporal_load(g.dense + ((size_t)a.cell * 64u + a.off));      // build-time experiment (round 6, profiles/r6k_*): the dense grid's taps streaming through the L2
#else
        d.raw = g.dense[(size_t)a.cell * 64u + a.off];
#endif
    } else {
#if VR_BRICK_HEADERS
        // the voxel's line of the brick's block: [rmin, rdiff | 120 voxels] -- range and voxel come from one cache line
        const uint32_t line = brick_voxel_line(a.off);
#if VR_TAP_LINE_INDEX
        // the line's index in the atlas as ONE 32-bit number (cell * 5 + line: fewer than 2^32 lines = 512 GiB, checked at upload), shifted into the 64-bit
        // address once -- instead of a 64-bit product, two 64-bit selects and two 64-bit additions (round 5; fewer instructions, no measurable gain: profiles/r5p_*)
        const uint8_t* ln = g.atlas + ((size_t)(a.cell * (kBrickBlockBytes / 128u) + line) << 7);
#else
        const uint8_t* ln = g.atlas + ((size_t)a.cell * kBrickBlockBytes + (size_t)(line * 128u));
#endif
        const float* rec = reinterpret_cast<const float*>(ln);
        d.rmin = rec[0]; d.rdiff = rec[1];
        d.raw = ln[kBrickLineHeader + a.off - line * kBrickLineVoxels];
#else
        const float* rec = g.rng + 2u * (size_t)a.cell;          // compact (rmin, rdiff) pairs: twice as many bricks per cache line as BrickRec
        d.rmin = rec[0]; d.rdiff = rec[1];
        d.raw = g.atlas[(size_t)a.cell * 512u + a.off];
#endif
#if defined(VR_DIAG_EXTRA_RNG) && defined(__HIP_DEVICE_COMPILE__)
        {   // diagnostic (profiles/r4c_*): one more gather of the old range table's kind per tap, from the record of another brick -- what does such a gather cost?
            const uint32_t n_ = (uint32_t)g.nb[0] * (uint32_t)g.nb[1] * (uint32_t)g.nb[2];
            uint32_t c_ = a.cell * 2654435761u; c_ = c_ % n_;
            float x_ = g.rng[2u * (size_t)c_];
            asm volatile("" :: "v"(x_));
        }
#endif
    }
    return d;
}
template <int DENSE = 2>
VR_HD float tap_value(const GridView& g, TapData d, bool in) {
    const float v = grid_is_dense<DENSE>(g) ? half2float(d.raw) : d.rmin + unorm8(d.raw) * d.rdiff;
    return in ? v : 0.0f;
}
template <int DENSE = 2>
VR_HD float brick_value(const GridView& g, int32_t x, int32_t y, int32_t z) {
    const TapAddr a = tap_addr<DENSE>(g, x, y, z);
    return tap_value<DENSE>(g, tap_load<DENSE>(g, a), a.in);
}
// index of the majorant cell floor(ipos) >> (3 + mip) in the padded table, or -- outside the padded box, or NaN -- of the table's "outside" cell.
// The padded layout (vr_scene.h) holds the outside value in every cell beyond the real extent of a level, so only the padded extent -- the
// same for all levels -- is tested, on the floats (floor(x) in [0, n) <=> x in [0, n) for integer n; NaN fails), after
// which truncation equals floor.
// VR_MAJ_OUTSIDE_CELL (round 5, default): "outside" is the index of the table's last cell, which holds what the reference computes there (vr_scene.h
// majorant_table_cells), instead of -1 with the loaded value replaced by 0 afterwards: the load needs no clamp and the value no compare + select.  Either way the
// arithmetic on the value is the reference's: density_scale * 0 outside.  0: the -1 convention (kept for the A/B, profiles/r5o_*)
#ifndef VR_MAJ_OUTSIDE_CELL
#define VR_MAJ_OUTSIDE_CELL 1
#endif
#ifndef VR_MAJ_LEVEL_TEST
#define VR_MAJ_LEVEL_TEST 0
#endif
// ---- "clean" segments (round 5, VR_CLEAN_FLAG) ------------------------------------------------------------------------------------------------------------
// begin_segment marks a segment whose ray is finite and of moderate size -- |ipos| < 2^20, |idir| < 2^20 and |idir| * far < 2^20 per axis, |idir| >= 2^-20 on one
// axis at least; all comparisons that fail for NaN --
// in the sign of Hot::far (clean: far as computed, >= 0; not clean: -far; every reader takes |far|, seg_far).  On such a segment every point the trackers
// evaluate, p = ipos + t * idir with t < far, is finite with |p| < 2^21, and then:
//   * floor(p) converts to int exactly, so "inside the padded table" is an INTEGER test on the cell coordinates the index needs anyway (three right shifts, an
//     or3, one compare) instead of six float compares -- same set: 0 <= floor(x) < lim <=> 0 <= x < lim for the integer lim, and -0 is inside either way;
//   * the three candidates of a DDA step are never NaN (the bracket floor(p / dim) * dim + o - p is exact and its modulus >= 0.5, and 1 / idir is finite or
//     +-inf, never NaN): min(tx, min(ty, tz)) is ONE v_min3_f32 -- which differs from the reference's comparisons only in what it does with NaN;
//   * a tap's coordinates are finite -- or, when the ray parameter itself has become NaN (0 / 0 in the step back to the collision point), NaN on all three axes, where
//     the tap lands on index -1 by itself: no NaN guard on the voxel index (argument at nan_guard).
// The scheduler (vr_pathtrace.h) runs the hot pair in the CLEAN form while every marching path of the wavefront is clean and in the general form otherwise; the
// host harness picks the form per path (lane_step), so both are checked against the oracle on the CPU.
#ifndef VR_CLEAN_FLAG
#define VR_CLEAN_FLAG 1
#endif
constexpr float kCleanBound = 1048576.0f;      // 2^20
VR_HD float seg_far(const Hot& h) { return VR_CLEAN_FLAG ? abs_(h.far) : h.far; }
VR_HD bool seg_clean(const Hot& h) { return VR_CLEAN_FLAG ? (int32_t)f2u(h.far) >= 0 : false; }
// floor(x) as an int for |x| < 2^31 (clean segments: < 2^21)
VR_HD int32_t cvt_flr(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r;
#else
    return (int32_t)floor_(x);
#endif
}
template <int DENSE = 2, int MAJB = 2, bool CLEAN = false>
VR_HD int32_t majorant_index(const GridView& g, v3 ipos, int32_t mip) {
    const uint32_t sh = 3u + (uint32_t)mip;
    const uint32_t sx = (uint32_t)g.mshift[0] - (uint32_t)mip, sy = (uint32_t)g.mshift[1] - (uint32_t)mip;
    const uint32_t off = majorant_level_offset((uint32_t)(g.mshift[0] + g.mshift[1] + g.mshift[2]), (uint32_t)mip);
    // the layout majorant_kernel wrote the table in (GridView::maj_blocked, a property of the grid since round 5): known at compile time in the kernels built
    // for one layout -- a kernel is only launched on grids of its layout (vr_kernels.hip pathtrace_variant) -- read from the view otherwise (wave-uniform)
    const bool blocked = MAJB == 2 ? g.maj_blocked != 0 : MAJB == 1;
    if (CLEAN && VR_MAJ_OUTSIDE_CELL) {
        const uint32_t bx = (uint32_t)cvt_flr(ipos.x) >> sh, by = (uint32_t)cvt_flr(ipos.y) >> sh, bz = (uint32_t)cvt_flr(ipos.z) >> sh;
        const uint32_t sz = (uint32_t)g.mshift[2] - (uint32_t)mip;
        const bool inside = ((bx >> sx) | (by >> sy) | (bz >> sz)) == 0u;          // a negative coordinate leaves ones above the shifted-out bits
        return inside ? (int32_t)(off + majorant_cell_index(bx, by, bz, sx, sy, (uint32_t)mip, blocked)) : g.maj_outside;
    }
    // (a level the grid does not have -- mip > n_mips -- needs no test: the table has all four levels and the cells of a missing one hold the "outside" value,
    // vr_kernels.hip majorant_kernel; VR_MAJ_LEVEL_TEST=1 brings the compare back)
    const bool inside = (VR_MAJ_LEVEL_TEST ? (mip <= g.n_mips) : true) & (ipos.x >= 0.0f) & (ipos.x < g.mlim[0]) & (ipos.y >= 0.0f) & (ipos.y < g.mlim[1]) & (ipos.z >= 0.0f) & (ipos.z < g.mlim[2]);
    const uint32_t bx = (uint32_t)(int32_t)ipos.x >> sh, by = (uint32_t)(int32_t)ipos.y >> sh, bz = (uint32_t)(int32_t)ipos.z >> sh;
    return inside ? (int32_t)(off + majorant_cell_index(bx, by, bz, sx, sy, (uint32_t)mip, blocked)) : (VR_MAJ_OUTSIDE_CELL ? g.maj_outside : -1);
}
// Unconditional load (cell 0 when outside; the caller discards it then).  TF kernels read the float table (TF-remapped
// majorants); the others read the raw fp16 range maximum -- half the cache lines -- and scale it themselves (majorant_value).
template <bool TF>
VR_HD uint32_t majorant_fetch(const GridView& g, int32_t idx) {
    const int32_t i = VR_MAJ_OUTSIDE_CELL ? idx : (idx < 0 ? 0 : idx);
#if VR_MAJ_OUTSIDE_CELL && defined(__clang__)
    __builtin_assume(i >= 0);                        // a table index (at most 73/64 x 2^30 cells, vr_scene.h): zero- instead of sign-extended into the 64-bit address
#endif
    VR_TRACE(0, g.majorant16, i, TF ? 4 : 2);
    return TF ? f2u(g.majorant[i]) : (uint32_t)g.majorant16[i];
}
template <bool TF>
VR_HD float majorant_value(const SceneParams& P, uint32_t raw) { return TF ? u2f(raw) : P.u.vol_density_scale * half2float(raw); }
// the majorant of a step from the word march_load fetched for cell `idx`
template <bool TF>
VR_HD float majorant_of(const SceneParams& P, int32_t idx, uint32_t raw) {
    const float m = majorant_value<TF>(P, raw);
    return VR_MAJ_OUTSIDE_CELL ? m : (idx >= 0 ? m : 0.0f);
}
template <bool TF, int DENSE = 2, int MAJB = 2>
VR_HD float majorant_at(const SceneParams& P, v3 ipos, int32_t mip) {
    const int32_t idx = majorant_index<DENSE, MAJB>(P.density, ipos, mip);
    return majorant_of<TF>(P, idx, majorant_fetch<TF>(P.density, idx));
}
// A point of a CLEAN segment (seg_clean) needs no guard.  Its coordinates are finite with |p| < 2^21 -- or NaN on ALL THREE axes, when the ray parameter itself is NaN:
// the reference's `t += tau / majorant` is 0 / 0 when a free-flight draw of exactly 0 meets an empty first cell (2^-24 per segment: dozens of times in a bench frame),
// after which it still evaluates the collision there (every fetch outside, a null collision, the draws consumed) before the loop ends.  With NaN coordinates every
// filter test of the stochastic tap compares false (fast form: x = NaN, and the call's min |x| stays +inf: not "unsure"; reference's form: r < NaN), so the tap is
// floor + 0 - 1 with floor = (int)NaN = 0 on the device: index -1 on every axis, outside the grid, as the oracle's "NaN converts to outside"; and the trilinear lookup's
// weights are NaN, so its value is NaN whatever its corners hold.  tests/test_gpu_parity.py::test_nan_ray_parameter_on_a_clean_segment renders 2^26 segments of that kind
// of scene against the oracle.
// a NaN coordinate must read "outside": on the device voxel_index turns NaN into index o, so one index is forced negative
VR_HD int32_t nan_guard(int32_t ix, float fx, float fy, float fz) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ((fx != fx) | (fy != fy) | (fz != fz)) ? -1 : ix;
#else
    return ix;
#endif
}

// The 8 corner fetches of the trilinear lookup share their per-axis arithmetic: 2 cell coordinates, 2 in-cell offsets and 2
// validity tests per axis instead of 8 x 3.  A corner outside the grid reads 0 (as brick_value does); its address is clamped to
// cell 0 so that all 16 loads are unconditional and independent.
struct AxisCells { uint32_t c[2], o[2]; bool in[2]; };
VR_HD AxisCells axis_cells(int32_t i0, int32_t i1, uint32_t extent_voxels, uint32_t log2_cell) {
    AxisCells a;
    const int32_t i[2] = { i0, i1 };
    const uint32_t mask = (1u << log2_cell) - 1u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        a.in[t] = i[t] >= 0 && (uint32_t)i[t] < extent_voxels;
        a.c[t] = a.in[t] ? (uint32_t)i[t] >> log2_cell : 0u;
        a.o[t] = a.in[t] ? (uint32_t)i[t] & mask : 0u;
    }
    return a;
}
struct TriIO { TapAddr a[8]; TapData d[8]; float fx, fy, fz; uint32_t in_mask; };      // corners in [z][y][x] order
template <int DENSE = 2, bool GUARD = true>
VR_HD void trilinear_prep(const GridView& g, v3 ipos, TriIO& io) {
    const float qx = ipos.x - 0.5f, qy = ipos.y - 0.5f, qz = ipos.z - 0.5f;
    const float flx = floor_(qx), fly = floor_(qy), flz = floor_(qz);
    io.fx = qx - flx; io.fy = qy - fly; io.fz = qz - flz;
    const int32_t ix0 = voxel_index(flx, 0), ix1 = voxel_index(flx, 1);
    const int32_t ix = GUARD ? nan_guard(ix0, flx, fly, flz) : ix0, iy = voxel_index(fly, 0), iz = voxel_index(flz, 0);     // GUARD = false: a point of a clean segment (see nan_guard)
    const int32_t x1 = GUARD ? nan_guard(ix1, flx, fly, flz) : ix1, y1 = voxel_index(fly, 1), z1 = voxel_index(flz, 1);
    const bool dense = grid_is_dense<DENSE>(g);
    const uint32_t lg = dense ? 2u : 3u;
    const AxisCells X = axis_cells(ix, x1, dense ? (uint32_t)g.dim[0] : (uint32_t)g.nb[0] << 3, lg);
    const AxisCells Y = axis_cells(iy, y1, dense ? (uint32_t)g.dim[1] : (uint32_t)g.nb[1] << 3, lg);
    const AxisCells Z = axis_cells(iz, z1, dense ? (uint32_t)g.dim[2] : (uint32_t)g.nb[2] << 3, lg);
    io.in_mask = 0u;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t row = dense ? (mul24(Z.c[k], (uint32_t)g.dblk[1]) + Y.c[j]) * (uint32_t)g.dblk[0]
                                       : mul24(mul24(Z.c[k], (uint32_t)g.nb[1]) + Y.c[j], (uint32_t)g.nb[0]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                TapAddr& a = io.a[4 * k + 2 * j + i];
                a.cell = row + X.c[i];
                a.off = dense ? ((Z.o[k] << 4) | (Y.o[j] << 2) | X.o[i]) : ((Z.o[k] << 6) | (Y.o[j] << 3) | X.o[i]);
                a.in = true;
                io.in_mask |= (X.in[i] & Y.in[j] & Z.in[k]) ? 1u << (4 * k + 2 * j + i) : 0u;
            }
        }
}
// The 8 corner loads.  With a decoded float atlas (GridView::atlas_f32, brick grids under a transfer function) a corner is ONE
// 4-byte load of the value the byte path would compute (rmin + unorm8(b) * rdiff, evaluated once when the atlas is decoded).
template <int DENSE = 2, int PAIR = 0>
VR_HD void trilinear_load(const GridView& g, TriIO& io) {
    if (!grid_is_dense<DENSE>(g) && g.atlas_f32) {
#pragma unroll
        for (int n = 0; n < 8; ++n) { io.d[n].rmin = 0.0f; io.d[n].rdiff = 0.0f; io.d[n].raw = f2u(g.atlas_f32[(size_t)io.a[n].cell * 512u + io.a[n].off]); }
        return;
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) io.d[n] = tap_load<DENSE, PAIR>(g, io.a[n]);
}
VR_HD void trilinear_idle(TriIO& io) {           // addresses of a lane without a lookup: cell 0
#pragma unroll
    for (int n = 0; n < 8; ++n) { io.a[n].cell = 0u; io.a[n].off = 0u; io.a[n].in = false; }
    io.fx = io.fy = io.fz = 0.0f; io.in_mask = 0u;
}
template <int DENSE = 2>
VR_HD float trilinear_value(const GridView& g, const TriIO& io) {
    float v[8];
    if (!grid_is_dense<DENSE>(g) && g.atlas_f32) {
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = ((io.in_mask >> n) & 1u) ? u2f(io.d[n].raw) : 0.0f;
    } else {
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = tap_value<DENSE>(g, io.d[n], (io.in_mask >> n) & 1u);
    }
    const float lx0 = mix_(v[0], v[1], io.fx);
    const float lx1 = mix_(v[2], v[3], io.fx);
    const float hx0 = mix_(v[4], v[5], io.fx);
    const float hx1 = mix_(v[6], v[7], io.fx);
    return mix_(mix_(lx0, lx1, io.fy), mix_(hx0, hx1, io.fy), io.fz);
}
template <int DENSE = 2>
VR_HD float density_trilinear_raw(const GridView& g, v3 ipos) {
    TriIO io;
    trilinear_prep<DENSE>(g, ipos, io);
    trilinear_load<DENSE>(g, io);
    return trilinear_value<DENSE>(g, io);
}

// stochastic tricubic tap (common.glsl:221-244): 9 draws in the order tap2.xyz, tap3.xyz, tap4.xyz;
// tap k replaces the choice when draw < w_k / max(1e-3, w_1 + ... + w_k)
struct AxisWeights { float w2, s2, w3, s3, w4, s4, fl; };
VR_HD AxisWeights tricubic_axis_weights(float q) {
    AxisWeights a;
    const float fl = floor_(q);
    const float t = q - fl, t2 = t * t;
    const float k = 1.0f / 6.0f;
    float sum = k * (-t * t2 + 3.0f * t2 - 3.0f * t + 1.0f);
    a.w2 = k * (3.0f * t * t2 - 6.0f * t2 + 4.0f);
    sum = a.w2 + sum; a.s2 = max_(1e-3f, sum);
    a.w3 = k * (-3.0f * t * t2 + 3.0f * t2 + 3.0f * t + 1.0f);
    sum = a.w3 + sum; a.s3 = max_(1e-3f, sum);
    a.w4 = k * t * t2;
    sum = a.w4 + sum; a.s4 = max_(1e-3f, sum);
    a.fl = fl;
    return a;
}
// ---- fast decision of the nine filter tests (device; the host harness can switch it on to run the same code) ----------------
// Test j of an axis asks "r_j < w / s" with r_j = k_j * 2^-24 (k_j: the low 24 bits of the LCG state after j draws), w one of the
// weights above and s its partial sum.  The fast path evaluates the same polynomials the cheap way (6 w_i in Horner form, fused:
// 12 operations per axis instead of 28; s4 = 6 exactly), cross-multiplies instead of dividing, and decides only when k_j * s'
// lies outside a guard band of +-2^-18 (relative) around w' * 2^24.  The two evaluations' thresholds differ by at most 2^-21.3
// (relative) anywhere in [0, 1], so a draw outside the band is decided as the reference decides it: tests/tools_tricubic_band.cpp
// checks exactly that -- for EVERY float t in [0, 1] and all 2^24 values of a draw -- with this very code compiled for the host
// (3.2e9 (t, test) pairs, no exception; test_tricubic_fast_path_agrees_with_the_reference runs a strided pass).
// A draw inside a band (9 x ~4e-6 per call) sends the call to the exact code: the reference's weights and divisions.
// LCG jump-ahead: state_j = A_j * state_0 + C_j (mod 2^32) after j draws
struct LcgJump { uint32_t A, C; };
constexpr LcgJump lcg_jump(int j) {
    uint32_t A = 1u, C = 0u;
    for (int i = 0; i < j; ++i) { A = A * 1664525u; C = C * 1664525u + 1013904223u; }
    return LcgJump{ A, C };
}
// Round 5 (VR_TAP_ABS_BAND, default on): the same decision with fewer operations.  The weights come out of the Horner forms already scaled by 2^24 (the
// constants carry the factor: scaling by a power of two commutes with every rounding of the evaluation, so W = 2^24 x w bit for bit), the partial sums are
// s2 = RN(2^-24 W2 + u^3), s3 = RN(2^-24 W3 + s2) (one fma each, the same single rounding as the additions they replace), and a test is ONE fma and two
// compares: x = RN(k s - W);  yes: x < -G,  no: x > G,  with an ABSOLUTE band G = 160 (in units of k s; W <= 4 x 2^24, so G / W >= 2^-18.7 where the two
// evaluations' thresholds differ most, 2^-21.3 relative = 26 units).  Round 2's form -- x = k s against RN(W lo - eps) and RN(W hi + eps), a relative band --
// took a multiply and two fmas per test.  tests/tools_tricubic_band.cpp checks either form for every float t in [0, 1] and all 2^24 draws.
#ifndef VR_TAP_ABS_BAND
#define VR_TAP_ABS_BAND 1
#endif
// VR_TAP_PRESHIFT (round 5, default): the nine draws are taken from the LCG state SHIFTED LEFT BY 8 -- multiplying the recurrence by 2^8 commutes with it mod 2^32, and
// the top 24 bits of the shifted state are the draw's 24 bits, so its conversion to float is 2^8 k exactly (24 significant bits) and the mask per draw is gone.  The
// weights and the band carry the same factor (all of it powers of two: every rounding of the evaluation commutes with it, x' = 2^8 x bit for bit).  And a call
// needs the exact code when the SMALLEST |x| of its nine tests lies inside the band: min(|x|) chained through the tests (min3 for two of them) and ONE compare,
// instead of a second compare per test.  A test whose x is NaN (a non-finite coordinate on that axis) drops out of the minimum and decides "no" -- what the exact
// code's comparisons against NaN quotients decide for that axis as well.
#ifndef VR_TAP_PRESHIFT
#define VR_TAP_PRESHIFT 1
#endif
#ifndef VR_TAP_CHAIN
#define VR_TAP_CHAIN 0
#endif
#if VR_TAP_ABS_BAND
struct AxisFast { float w2, s2, w3, s3, w4, fl; };      // w*: 2^24 x 6 x the weights (2^32 x with VR_TAP_PRESHIFT); s*: 6 x the partial sums (s4 = 6)
constexpr float kTapDrawScale = VR_TAP_PRESHIFT ? 256.0f : 1.0f;        // a draw enters its test as kTapDrawScale x k, k its 24 bits
constexpr float kTapScale = 16777216.0f * kTapDrawScale, kTapInvScale = 1.0f / kTapScale, kTapBand = 160.0f * kTapDrawScale;
VR_HD AxisFast tricubic_axis_fast(float q) {
    AxisFast a;
    a.fl = floor_(q);
    const float t = q - a.fl, t2 = t * t, u = 1.0f - t;
    a.w4 = (t2 * t) * kTapScale;                                                                               // 2^24 t^3
    a.w2 = fma_(fma_(3.0f * kTapScale, t, -6.0f * kTapScale), t2, 4.0f * kTapScale);                           // 2^24 (3t^3 - 6t^2 + 4)
    a.w3 = fma_(fma_(fma_(-3.0f * kTapScale, t, 3.0f * kTapScale), t, 3.0f * kTapScale), t, kTapScale);        // 2^24 (-3t^3 + 3t^2 + 3t + 1)
    a.s2 = fma_(a.w2, kTapInvScale, (u * u) * u);                                                              // (1 - t)^3 + 6 w2
    a.s3 = fma_(a.w3, kTapInvScale, a.s2);
    return a;
}
// one test: yes / no as the reference decides, or neither (inside the band).  k: the draw's 24 bits as a float; w: 2^24 x the weight
VR_HD void tricubic_fast_test(float k, float w, float s, bool& yes, bool& no) {
    const float x = fma_(k, s, -w);
    yes = x < -kTapBand;
    no = x > kTapBand;
}
#else
struct AxisFast { float w2, s2, w3, s3, w4, fl; };      // 6 x the weights; s4 = 6
VR_HD AxisFast tricubic_axis_fast(float q) {
    AxisFast a;
    a.fl = floor_(q);
    const float t = q - a.fl, t2 = t * t, u = 1.0f - t;
    a.w4 = t2 * t;                                                    // t^3
    a.w2 = fma_(fma_(3.0f, t, -6.0f), t2, 4.0f);                      // 3t^3 - 6t^2 + 4
    a.w3 = fma_(fma_(fma_(-3.0f, t, 3.0f), t, 3.0f), t, 1.0f);        // -3t^3 + 3t^2 + 3t + 1
    a.s2 = (u * u) * u + a.w2;                                        // (1 - t)^3 + 6 w2
    a.s3 = a.s2 + a.w3;
    return a;
}
constexpr float kTapLo = 16777152.0f;       // 2^24 (1 - 2^-18)
constexpr float kTapHi = 16777280.0f;       // 2^24 (1 + 2^-18)
// one test: yes / no as the reference decides, or neither (inside the band).  k: the draw's 24 bits as a float
VR_HD void tricubic_fast_test(float k, float w, float s, bool& yes, bool& no) {
    const float x = k * s;
    // the absolute term keeps k = 0 out of "yes" where the reference's weight has underflowed to 0 (t < 1e-9) and the fast one has not
    yes = x < fma_(w, kTapLo, -1e-20f);
    no = x > fma_(w, kTapHi, 1e-20f);
}
#endif
#ifndef VR_TAP_FAST
#if defined(__HIP_DEVICE_COMPILE__)
#define VR_TAP_FAST 1
#else
#define VR_TAP_FAST 0
#endif
#endif

// FAST: the guarded fast decision first, the reference's code only for a call with a draw inside a band (the device; the host harness can switch it on);
// otherwise the reference's code always.  Both are compiled everywhere: tests/tools_tricubic_band.cpp runs one against the other.
// GUARD = false: a tap on a clean segment -- its coordinates are finite, or NaN on all three axes, which needs no guard (see nan_guard)
template <bool FAST, bool GUARD = true>
VR_HD void tricubic_tap_t(v3 ipos, uint32_t& seed, int32_t& tx, int32_t& ty, int32_t& tz) {
    int32_t jx = 0, jy = 0, jz = 0;
    float flx, fly, flz;
#if VR_FAST_DEVICE
    // tolerance mode: draw j is k_j * 2^-24 and "r_j < w / s" is taken as k_j * s < w * 2^24, unguarded (one rounding apart
    // from the reference's quotient; a decision flips only when the draw lands within that rounding of the threshold)
    const AxisWeights ax = tricubic_axis_weights(ipos.x - 0.5f);
    const AxisWeights ay = tricubic_axis_weights(ipos.y - 0.5f);
    const AxisWeights az = tricubic_axis_weights(ipos.z - 0.5f);
    flx = ax.fl; fly = ay.fl; flz = az.fl;
    {
        const uint32_t lo24 = seed & 0x00FFFFFFu;
#define VR_TAPF(J, V, W, S, N) do { \
            constexpr LcgJump g_ = lcg_jump(N); \
            const float k_ = (float)((mul24(lo24, g_.A & 0x00FFFFFFu) + g_.C) & 0x00FFFFFFu); \
            J = k_ * (S) < (W) * 16777216.0f ? V : J; \
        } while (0)
        VR_TAPF(jx, 1, ax.w2, ax.s2, 1); VR_TAPF(jy, 1, ay.w2, ay.s2, 2); VR_TAPF(jz, 1, az.w2, az.s2, 3);
        VR_TAPF(jx, 2, ax.w3, ax.s3, 4); VR_TAPF(jy, 2, ay.w3, ay.s3, 5); VR_TAPF(jz, 2, az.w3, az.s3, 6);
        VR_TAPF(jx, 3, ax.w4, ax.s4, 7); VR_TAPF(jy, 3, ay.w4, ay.s4, 8); VR_TAPF(jz, 3, az.w4, az.s4, 9);
#undef VR_TAPF
        rng_skip9(seed);
    }
#else
    bool exact = true;
    if constexpr (FAST) {
        // a draw only uses the low 24 bits of the state, and those of state_j = A_j * seed + C_j only need the low 24 bits of seed and
        // A_j: one 24-bit multiply-add per draw instead of a chained 32-bit multiply
        const AxisFast fx = tricubic_axis_fast(ipos.x - 0.5f), fy = tricubic_axis_fast(ipos.y - 0.5f), fz = tricubic_axis_fast(ipos.z - 0.5f);
        flx = fx.fl; fly = fy.fl; flz = fz.fl;
        const uint32_t seed0 = seed;
#if VR_TAP_ABS_BAND && VR_TAP_PRESHIFT
        const uint32_t sh8 = seed << 8;
        float amin = __builtin_inff();                               // the smallest |x| of the nine tests
#if VR_TAP_CHAIN
        // the nine shifted states one from the other, st_j = a st_(j-1) + 2^8 c: ONE multiplier in a scalar register instead of nine jump-ahead constants (the
        // kernels are short of scalar registers: the brick kernel re-loaded the atlas pointer from the kernel arguments in every collision pass).  The empty asm
        // keeps the optimiser from folding the chain back into nine constants.
        uint32_t st_ = sh8;
#if defined(__HIP_DEVICE_COMPILE__)
#define VR_TAP_KEEP(X) asm("" : "+v"(X))
#else
#define VR_TAP_KEEP(X) do { } while (0)
#endif
#define VR_TAP(J, V, W, S, N) do { \
            st_ = st_ * 1664525u + (1013904223u << 8); \
            VR_TAP_KEEP(st_); \
            const float x_ = fma_((float)st_, S, -(W));     /* tricubic_fast_test's x for the draw 2^8 k */ \
            J = x_ < -kTapBand ? V : J; \
            amin = __builtin_fminf(amin, __builtin_fabsf(x_)); \
        } while (0)
#else
#define VR_TAP(J, V, W, S, N) do { \
            constexpr LcgJump g_ = lcg_jump(N); \
            const float x_ = fma_((float)(sh8 * g_.A + (g_.C << 8)), S, -(W));     /* tricubic_fast_test's x for the draw 2^8 k */ \
            J = x_ < -kTapBand ? V : J; \
            amin = __builtin_fminf(amin, __builtin_fabsf(x_)); \
        } while (0)
#endif
#else
        const uint32_t lo24 = seed & 0x00FFFFFFu;
        bool unsure = false;
#define VR_TAP(J, V, W, S, N) do { \
            constexpr LcgJump g_ = lcg_jump(N); \
            bool yes_, no_; \
            tricubic_fast_test((float)((mul24(lo24, g_.A & 0x00FFFFFFu) + g_.C) & 0x00FFFFFFu), W, S, yes_, no_); \
            J = yes_ ? V : J; \
            unsure = unsure | !(yes_ | no_); \
        } while (0)
#endif
        VR_TAP(jx, 1, fx.w2, fx.s2, 1); VR_TAP(jy, 1, fy.w2, fy.s2, 2); VR_TAP(jz, 1, fz.w2, fz.s2, 3);
        VR_TAP(jx, 2, fx.w3, fx.s3, 4); VR_TAP(jy, 2, fy.w3, fy.s3, 5); VR_TAP(jz, 2, fz.w3, fz.s3, 6);
        VR_TAP(jx, 3, fx.w4, 6.0f, 7); VR_TAP(jy, 3, fy.w4, 6.0f, 8); VR_TAP(jz, 3, fz.w4, 6.0f, 9);
#undef VR_TAP
#if VR_TAP_ABS_BAND && VR_TAP_PRESHIFT
        const bool unsure = !(amin > kTapBand);                      // a test inside the band, or nine NaNs
#endif
        rng_skip9(seed);
        exact = unsure;
        if (unsure) { seed = seed0; jx = jy = jz = 0; }
    } else {
        flx = floor_(ipos.x - 0.5f); fly = floor_(ipos.y - 0.5f); flz = floor_(ipos.z - 0.5f);
    }
    if (exact) {
        // the reference's code (common.glsl:221-244)
        const AxisWeights ax = tricubic_axis_weights(ipos.x - 0.5f);
        const AxisWeights ay = tricubic_axis_weights(ipos.y - 0.5f);
        const AxisWeights az = tricubic_axis_weights(ipos.z - 0.5f);
        float r;
        r = rng(seed); if (r < ax.w2 / ax.s2) jx = 1;
        r = rng(seed); if (r < ay.w2 / ay.s2) jy = 1;
        r = rng(seed); if (r < az.w2 / az.s2) jz = 1;
        r = rng(seed); if (r < ax.w3 / ax.s3) jx = 2;
        r = rng(seed); if (r < ay.w3 / ay.s3) jy = 2;
        r = rng(seed); if (r < az.w3 / az.s3) jz = 2;
        r = rng(seed); if (r < ax.w4 / ax.s4) jx = 3;
        r = rng(seed); if (r < ay.w4 / ay.s4) jy = 3;
        r = rng(seed); if (r < az.w4 / az.s4) jz = 3;
    }
#endif
    tx = voxel_index(flx, jx - 1); ty = voxel_index(fly, jy - 1); tz = voxel_index(flz, jz - 1);
    if (GUARD) tx = nan_guard(tx, flx, fly, flz);
}
template <bool GUARD = true>
VR_HD void tricubic_tap(v3 ipos, uint32_t& seed, int32_t& tx, int32_t& ty, int32_t& tz) { tricubic_tap_t<VR_TAP_FAST != 0, GUARD>(ipos, seed, tx, ty, tz); }

// transfer function (common.glsl:203-212)
// `lut`: tf_size x vec4 -- the SSBO in global memory, or the copy the path-tracing kernel stages in LDS (vr_pathtrace.h)
VR_HD void tf_lookup_at(const Uniforms& u, const float* lut, float d, float rgba[4]) {
    const float tc = clamp_((d - u.tf_window_left) / u.tf_window_width, 0.0f, 1.0f - 1e-6f);
    const float tcs = tc * (float)u.tf_size;
    int32_t idx = floor2i(tcs);
    const float f = tcs - floor_(tcs);
    const int32_t n = (int32_t)u.tf_size;
    if (idx == kIntMin) idx = 0;
    idx = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
    const int32_t idx1 = idx + 1 < n - 1 ? idx + 1 : n - 1;
    const float* a = lut + 4 * idx;
    const float* b = lut + 4 * idx1;
#pragma unroll
    for (int k = 0; k < 4; ++k) rgba[k] = mix_(a[k], b[k], f);
}
VR_HD void tf_lookup(const SceneParams& P, float d, float rgba[4]) { tf_lookup_at(P.u, P.tf_lut, d, rgba); }

// ---------------------------------------------------------------------------------------------------
// environment (common.glsl:93-152)
VR_HD int32_t wrap_repeat(int32_t i, int32_t n) {
    // GL_REPEAT; texture coordinates in [0, 1] give i in [-1, n]: one conditional add/subtract, the integer division only
    // for coordinates further out
    if ((uint32_t)(i + n) < 3u * (uint32_t)n) return i < 0 ? i + n : (i >= n ? i - n : i);
    const int32_t m = i % n;
    return m < 0 ? m + n : m;
}
VR_HD int32_t clampi(int32_t i, int32_t lo, int32_t hi) { return i < lo ? lo : (i > hi ? hi : i); }

// two consecutive dwords (4-byte aligned) as one load
VR_HD void ld_pair(const uint32_t* p, uint32_t& a, uint32_t& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t Pair __attribute__((ext_vector_type(2), aligned(4)));
    const Pair v = *reinterpret_cast<const Pair*>(p);
    a = v.x; b = v.y;
#else
    a = p[0]; b = p[1];
#endif
}
// four consecutive floats (4-byte aligned) as one load
VR_HD void ld_quad(const float* p, float out[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float Quad4 __attribute__((ext_vector_type(4), aligned(4)));
    const Quad4 v = *reinterpret_cast<const Quad4*>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
#else
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3];
#endif
}
// scale of a compact environment texel: 2^(e - 136) for e >= 10 (exponent field e - 9), 0 for e = 0
VR_HD float rgbe_scale(uint32_t q) { const uint32_t e = q >> 24; return e ? u2f((e - 9u) << 23) : 0.0f; }
VR_HD v3 env_texture(const SceneParams& P, float u, float v) {
    const int32_t w = P.env_w, h = P.env_h;
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx = x - floor_(x), fy = y - floor_(y);
    int32_t ix = floor2i(x), iy = floor2i(y);
    if (ix == kIntMin || iy == kIntMin || ix > (1 << 28) || ix < -(1 << 28)) { ix = 0; iy = 0; fx = 0.0f; fy = 0.0f; }
    const int32_t x0 = wrap_repeat(ix, w), x1 = wrap_repeat(ix + 1, w);
    const int32_t y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
    if (P.env_rgbe) {
        // compact form (vr_scene.h SceneParams::env_rgbe): one dword per texel, decoded to the floats the float map holds -- (float)m * 2^(e - 136): an 8-bit integer
        // times a power of two, exact; texels with e = 0 are 0, texels with 0 < e < 10 (a subnormal scale) do not occur in a map that has this form (Environment::build)
        // the two texels of a row are neighbours in memory unless the lookup wraps around the map's seam: one 8-byte load per row then (two loads instead of four --
        // what a lookup costs the vector memory path is its number of (lane, instruction) pairs, profiles/r6i_*)
        uint32_t q00, q10, q01, q11;
        if (x1 == x0 + 1) {
            ld_pair(P.env_rgbe + ((size_t)y0 * w + x0), q00, q10);
            ld_pair(P.env_rgbe + ((size_t)y1 * w + x0), q01, q11);
        } else {
            q00 = P.env_rgbe[(size_t)y0 * w + x0]; q10 = P.env_rgbe[(size_t)y0 * w + x1]; q01 = P.env_rgbe[(size_t)y1 * w + x0]; q11 = P.env_rgbe[(size_t)y1 * w + x1];
        }
        VR_TRACE(3, P.env_rgbe, ((size_t)y0 * w + x0) * 4, 4); VR_TRACE(3, P.env_rgbe, ((size_t)y0 * w + x1) * 4, 4);
        VR_TRACE(3, P.env_rgbe, ((size_t)y1 * w + x0) * 4, 4); VR_TRACE(3, P.env_rgbe, ((size_t)y1 * w + x1) * 4, 4);
        const float s00 = rgbe_scale(q00), s10 = rgbe_scale(q10), s01 = rgbe_scale(q01), s11 = rgbe_scale(q11);
        v3 r;
        r.x = mix_(mix_((float)(q00 & 255u) * s00, (float)(q10 & 255u) * s10, fx), mix_((float)(q01 & 255u) * s01, (float)(q11 & 255u) * s11, fx), fy);
        r.y = mix_(mix_((float)((q00 >> 8) & 255u) * s00, (float)((q10 >> 8) & 255u) * s10, fx), mix_((float)((q01 >> 8) & 255u) * s01, (float)((q11 >> 8) & 255u) * s11, fx), fy);
        r.z = mix_(mix_((float)((q00 >> 16) & 255u) * s00, (float)((q10 >> 16) & 255u) * s10, fx), mix_((float)((q01 >> 16) & 255u) * s01, (float)((q11 >> 16) & 255u) * s11, fx), fy);
        return r;
    }
    const float* t00 = P.envmap + kEnvTexelFloats * ((size_t)y0 * w + x0);
    const float* t10 = P.envmap + kEnvTexelFloats * ((size_t)y0 * w + x1);
    const float* t01 = P.envmap + kEnvTexelFloats * ((size_t)y1 * w + x0);
    const float* t11 = P.envmap + kEnvTexelFloats * ((size_t)y1 * w + x1);
    VR_TRACE(3, P.envmap, (t00 - P.envmap) * 4, 16); VR_TRACE(3, P.envmap, (t10 - P.envmap) * 4, 16);
    VR_TRACE(3, P.envmap, (t01 - P.envmap) * 4, 16); VR_TRACE(3, P.envmap, (t11 - P.envmap) * 4, 16);
    v3 r;
    r.x = mix_(mix_(t00[0], t10[0], fx), mix_(t01[0], t11[0], fx), fy);
    r.y = mix_(mix_(t00[1], t10[1], fx), mix_(t01[1], t11[1], fx), fy);
    r.z = mix_(mix_(t00[2], t10[2], fx), mix_(t01[2], t11[2], fx), fy);
    return r;
}
// pyramid level `mip` starts at (4*dim^2 - 4*(dim>>mip)^2) / 3 floats
VR_HD int32_t imp_level_offset(int32_t dim, int32_t mip) { const int32_t d = dim >> mip; return (4 * dim * dim - 4 * d * d) / 3; }
VR_HD float imp_fetch(const SceneParams& P, int32_t x, int32_t y, int32_t mip) {
    const int32_t d = P.imp_dim >> mip;
    if (x < 0 || y < 0 || x >= d || y >= d) return 0.0f;
    return P.impmap[imp_level_offset(P.imp_dim, mip) + y * d + x];
}
// imp_fetch(P, 0, 0, base mip): the pyramid's coarsest value.  The renderer passes it along (SceneParams::env_avg_w: read back when the environment is built) -- fetched
// here it was one more dependent round trip at the end of every light sample and every escape, for a constant of the scene
VR_HD float env_average_importance(const SceneParams& P) { return P.env_avg_w_set ? P.env_avg_w : imp_fetch(P, 0, 0, P.u.env_imp_base_mip); }
VR_HD v3 lookup_environment(const SceneParams& P, v3 dir) {
    const v3 idir = mat3_mul(P.u.env_inv_transform, dir);
    const float u = atan2_(idir.z, idir.x) / (2.0f * kPi) + 0.5f;
    const float v = 1.0f - acos_(idir.y) / kPi;
    const v3 c = env_texture(P, u, v);
    return v3{ P.u.env_strength * c.x, P.u.env_strength * c.y, P.u.env_strength * c.z };
}
// The 2x2 warp of level `mip` only needs three numbers per parent cell: d = q0 / max(1e-8, q0 + q1), e0 = w0 / q0,
// e1 = w1 / q1 (common.glsl:116,126).  They depend on the importance pyramid alone, so they are tabulated once per
// environment (env_cdf_kernel, same operations) as one 12-byte record per cell: 1 load and 2 divisions per level instead
// of 4 loads and 4 divisions.  Two consecutive levels share a 64-byte block (vr_scene.h, env_cdf_index): the record of the child
// cell is in the line its parent's record came from.
// one level of the descent (common.glsl:118-131: "if (r < p) r /= p; else { pos += 1; r = (r - p) / (1 - p); }" per axis, written
// as operand selects + ONE division so that a wavefront whose lanes go both ways does not execute two); returns the child 0..3
// SAFE: the two quotients by vr_math.h div_core -- in the kernels compiled for one scene kind, which are only launched with an environment whose table passed the check
// when it was built (SceneParams::env_div_safe; vr_kernels.hip pathtrace_variant sends every other environment to the run-time variant, which divides in full).  The
// domain holds by induction over the levels: a threshold is NaN, 0 or in [2^-76, 1]; a coordinate p starts as k 2^-24 and stays +0, NaN or in [2^-99, 1] -- "left"
// (p < d, so d >= 2^-76) gives p / d in [p, 1); "right" gives (p - d) / (1 - d) with 1 - d >= 2^-24 (d <= p < 1) and p - d zero or at least an ulp of d, >= 2^-99,
// at most 1 - d.  The exceptions are NaN either way and nothing but their NaN-ness is used afterwards: a NaN threshold (an empty block's 0 / 0), and 0 / 0 when p has
// been rounded up to exactly 1 and meets a threshold of 1.
template <bool SAFE = false>
VR_HD int32_t env_warp_level(const float* rec, float& px, float& py, int32_t& posx, int32_t& posy) {
    const float d = rec[0], e0 = rec[1], e1 = rec[2];
    const bool right = !(px < d);
    const float e = right ? e1 : e0;
    const float nx = right ? px - d : px, dx = right ? 1.0f - d : d;
    px = SAFE ? div_core(nx, dx) : nx / dx;
    const bool up = !(py < e);
    const float ny = up ? py - e : py, dy = up ? 1.0f - e : e;
    py = SAFE ? div_core(ny, dy) : ny / dy;
    posx = 2 * posx + (right ? 1 : 0);
    posy = 2 * posy + (up ? 1 : 0);
    return (up ? 2 : 0) + (right ? 1 : 0);
}
#if defined(__HIP_DEVICE_COMPILE__)
// two levels from a 64-byte block held in registers: the parent's record, then the chosen child's, picked with selects
// children: c0 = (q0.w q1.x q1.y)  c1 = (q1.z q1.w q2.x)  c2 = (q2.y q2.z q2.w)  c3 = (q3.x q3.y q3.z)
template <bool ENVDC>
__device__ __forceinline__ void env_warp_block(float4 q0, float4 q1, float4 q2, float4 q3, float& px, float& py, int32_t& posx, int32_t& posy) {
    const float parent[3] = { q0.x, q0.y, q0.z };
    const int32_t c = env_warp_level<ENVDC>(parent, px, py, posx, posy);
    const bool c_right = (c & 1) != 0, c_up = (c & 2) != 0;
    const float lo_d = c_right ? q1.z : q0.w, lo_e0 = c_right ? q1.w : q1.x, lo_e1 = c_right ? q2.x : q1.y;
    const float hi_d = c_right ? q3.x : q2.y, hi_e0 = c_right ? q3.y : q2.z, hi_e1 = c_right ? q3.z : q2.w;
    const float child[3] = { c_up ? hi_d : lo_d, c_up ? hi_e0 : lo_e0, c_up ? hi_e1 : lo_e1 };
    env_warp_level<ENVDC>(child, px, py, posx, posy);
}
#endif
#ifndef VR_ENV_BLOCK_LOADS
#define VR_ENV_BLOCK_LOADS 1
#endif
// build-time experiments (round 6, profiles/r6j_*): levels 0 and 1 through the scalar cache; block loads only for the levels below VR_ENV_BLOCK_BELOW
#ifndef VR_ENV_SCALAR_TOP
#define VR_ENV_SCALAR_TOP 0
#endif
#ifndef VR_ENV_BLOCK_BELOW
#define VR_ENV_BLOCK_BELOW 64
#endif
// BLOCK: load a pair of levels' 64-byte block at once (below); off in the everything-at-run-time kernel, which has no registers for it
// ENVDC: the warp's quotients by div_core (env_warp_level<true>): only for an environment whose table passed the check (SceneParams::env_div_safe)
template <bool BLOCK = true, bool ENVDC = false>
VR_HD void sample_environment(const SceneParams& P, float r0, float r1, v3& w_i, v3& Le, float& pdf_out) {
    int32_t posx = 0, posy = 0;
    float px = r0, py = r1;
    const int32_t top = P.u.env_imp_base_mip - 1;              // levels 0 (one cell) .. top
    const float* blk = P.env_cdf;
    float w_texel = 0.0f;                                       // importance of the texel the descent ends in == imp_fetch(P, posx, posy, 0)
    int32_t k = 0;
    if ((top & 1) == 0) {                                       // odd number of levels: level 0 alone
        VR_TRACE(2, P.env_cdf, 0, 12);
        const int32_t c = env_warp_level<ENVDC>(blk, px, py, posx, posy);
        if (top == 0) w_texel = blk[3 + c];
        blk += kEnvCdfBlockFloats; k = 1;
    }
#if VR_ENV_BLOCK_LOADS && VR_ENV_SCALAR_TOP && defined(__HIP_DEVICE_COMPILE__)
    if (BLOCK && k == 0 && 1 < top) {
        // levels 0 and 1: the block is the same for every lane -- read through the scalar cache (a load in the constant address space: the table is not written while
        // the kernel runs) into scalar registers, no vector-memory round trip
        typedef const float4 __attribute__((address_space(4))) * ConstQ;
        uint64_t addr = reinterpret_cast<uint64_t>(blk);
        asm("" : "+s"(addr));                                   // (keeps the optimiser from folding the pointer back into the global address space and the load into a vector one)
        ConstQ cb = (ConstQ)(addr);
        env_warp_block<ENVDC>(cb[0], cb[1], cb[2], cb[3], px, py, posx, posy);
        blk += kEnvCdfBlockFloats; k = 2;
    }
#endif
    for (; k + 1 < top; k += 2) {                               // levels k and k + 1: parent record, then the chosen child's in the same block
        const float* b = blk + kEnvCdfBlockFloats * (size_t)((posy << k) + posx);
        VR_TRACE(2, P.env_cdf, (b - P.env_cdf) * 4, 64);
#if VR_ENV_BLOCK_LOADS && defined(__HIP_DEVICE_COMPILE__)
        if (BLOCK && k < VR_ENV_BLOCK_BELOW) {
        // the whole 64-byte block at once -- parent record and all four children's -- and the child picked in registers: one memory round trip per pair of
        // levels instead of two dependent ones (the second was a hit in the line the first had fetched, but a round trip all the same)
        const float4 q0 = reinterpret_cast<const float4*>(b)[0], q1 = reinterpret_cast<const float4*>(b)[1], q2 = reinterpret_cast<const float4*>(b)[2], q3 = reinterpret_cast<const float4*>(b)[3];
        env_warp_block<ENVDC>(q0, q1, q2, q3, px, py, posx, posy);
        } else
#endif
        {
        const int32_t c = env_warp_level<ENVDC>(b, px, py, posx, posy);
        env_warp_level<ENVDC>(b + 3 + 3 * c, px, py, posx, posy);
        }
        blk += (size_t)kEnvCdfBlockFloats << (2 * k);
    }
    if (k < top) {                                              // the last pair: one 128-byte line, the finest records carry their four texels
        const int32_t s = (top & 1) ? 0 : 1;
        const float* b = P.env_cdf + env_cdf_last_pair_floats(s, (top + 1 - s) / 2) + kEnvCdfLastBlockFloats * (size_t)((posy << k) + posx);
        VR_TRACE(2, P.env_cdf, (b - P.env_cdf) * 4, 128);
        const int32_t c = env_warp_level<ENVDC>(b, px, py, posx, posy);
        const float* child = b + kEnvCdfLastChild0 + kEnvCdfLastChildFloats * c;
        // the child's record AND its four texels' importances in one go (28 bytes: a 12- and a 16-byte load), the texel picked in registers: the importance no longer
        // waits for a third dependent round trip
        float w4[4];
        ld_quad(child + 3, w4);
        const int32_t c2 = env_warp_level<ENVDC>(child, px, py, posx, posy);
        w_texel = c2 == 0 ? w4[0] : (c2 == 1 ? w4[1] : (c2 == 2 ? w4[2] : w4[3]));
    }
    const float u = ((float)posx + px) * P.u.env_imp_inv_dim[0];
    const float v = ((float)posy + py) * P.u.env_imp_inv_dim[1];
    const float theta = saturate(1.0f - v) * kPi;
    const float phi = (saturate(u) * 2.0f - 1.0f) * kPi;
    float sin_t, cos_t, sin_p, cos_p;
    sincos_(theta, sin_t, cos_t);
    sincos_(phi, sin_p, cos_p);
    w_i = mat3_mul(P.u.env_transform, v3{ sin_t * cos_p, cos_t, sin_t * sin_p });
    const v3 c = env_texture(P, u, v);
    Le = v3{ P.u.env_strength * c.x, P.u.env_strength * c.y, P.u.env_strength * c.z };
    const float avg_w = env_average_importance(P);
    pdf_out = (w_texel / avg_w) * kInv4Pi;
}

// ---------------------------------------------------------------------------------------------------
// phase function (common.glsl:172-190), align (:25-33), MIS (:35)
VR_HD float phase_hg(float cos_t, float g) {
    const float denom = 1.0f + sqr(g) + 2.0f * g * cos_t;
    return kInv4Pi * (1.0f - sqr(g)) / (denom * sqrt_(denom));
}
VR_HD v3 align(v3 N, v3 v) {
    // common.glsl:27-28; both branches are "vector / sqrt(a*a + b*b)": select the operands, divide once
    const bool xa = abs_(N.x) > abs_(N.y);
    const float a = xa ? N.x : N.y;
    const v3 T = (xa ? v3{ -N.z, 0.0f, N.x } : v3{ 0.0f, N.z, -N.y }) / sqrt_(a * a + N.z * N.z);
    const v3 B = cross(N, T);
    return normalize(v3{ v.x * T.x + v.y * B.x + v.z * N.x,
                         v.x * T.y + v.y * B.y + v.z * N.y,
                         v.x * T.z + v.y * B.z + v.z * N.z });
}
VR_HD v3 sample_phase_hg(v3 dir, float g, float r0, float r1) {
    const float cos_t = abs_(g) < 1e-4f ? 1.0f - 2.0f * r0 :
        (1.0f + sqr(g) - sqr((1.0f - sqr(g)) / (1.0f - g + 2.0f * g * r0))) / (2.0f * g);
    const float sin_t = sqrt_(max_(0.0f, 1.0f - sqr(cos_t)));
    const float phi = 2.0f * kPi * r1;
    float sp, cp;
    sincos_(phi, sp, cp);
    return align(dir, v3{ sin_t * cp, sin_t * sp, cos_t });
}
VR_HD float power_heuristic(float a, float b) { return sqr(a) / (sqr(a) + sqr(b)); }

// (1 / v.x, 1 / v.y, 1 / v.z), each the correctly rounded quotient (vr_math.h rcp_exact); one range test for the three
VR_HD v3 rcp3_exact(v3 v) {
#if defined(__HIP_DEVICE_COMPILE__) && !VR_FAST_DEVICE
    v3 r = v3{ rcp_newton(v.x), rcp_newton(v.y), rcp_newton(v.z) };
    if (!((int)rcp_in_fast_range(v.x) & (int)rcp_in_fast_range(v.y) & (int)rcp_in_fast_range(v.z))) r = v3{ 1.0f / v.x, 1.0f / v.y, 1.0f / v.z };
    return r;
#else
    return v3{ 1.0f / v.x, 1.0f / v.y, 1.0f / v.z };
#endif
}

// box clip (common.glsl:157-165)
VR_HD bool intersect_box(v3 pos, v3 dir, const float* bmin, const float* bmax, float& tnear, float& tfar) {
    const v3 inv = rcp3_exact(dir);
    const v3 lo = (v3{ bmin[0], bmin[1], bmin[2] } - pos) * inv;
    const v3 hi = (v3{ bmax[0], bmax[1], bmax[2] } - pos) * inv;
    const v3 tmin = v3{ min_(lo.x, hi.x), min_(lo.y, hi.y), min_(lo.z, hi.z) };
    const v3 tmax = v3{ max_(lo.x, hi.x), max_(lo.y, hi.y), max_(lo.z, hi.z) };
    tnear = max_(0.0f, max_(tmin.x, max_(tmin.y, tmin.z)));
    tfar = min_(tmax.x, min_(tmax.y, tmax.z));
    return tnear <= tfar;
}

// one DDA step on mip (common.glsl:404-409)
#ifndef VR_DDA_INT_DIM
#define VR_DDA_INT_DIM 1
#endif
template <bool CLEAN = false>
VR_HD float step_dda(v3 p, v3 ri, int32_t mip) {
#if VR_DDA_INT_DIM
    const float dim = u2f((uint32_t)(130 + mip) << 23);        // 2^(3+mip) = (float)(8 << mip), from the exponent: shares the shift with idim (round 5: an add instead of shift + convert)
#else
    const float dim = (float)(8 << mip);
#endif
    const float idim = u2f((uint32_t)(124 - mip) << 23);       // 1 / dim exactly (dim = 2^(3+mip)): no division
    const float ox = ri.x >= 0.0f ? dim + 0.5f : -0.5f;
    const float oy = ri.y >= 0.0f ? dim + 0.5f : -0.5f;
    const float oz = ri.z >= 0.0f ? dim + 0.5f : -0.5f;
    const float tx = (floor_(p.x * idim) * dim + ox - p.x) * ri.x;
    const float ty = (floor_(p.y * idim) * dim + oy - p.y) * ri.y;
    const float tz = (floor_(p.z * idim) * dim + oz - p.z) * ri.z;
    if (CLEAN) return __builtin_fminf(tx, __builtin_fminf(ty, tz));       // no NaN among them (clean segment): one v_min3_f32
    return min_(tx, min_(ty, tz));
}

// ---------------------------------------------------------------------------------------------------
// state bodies

VR_HD void hot_init(Hot& h) {
    h.seed = 0u; h.item = 0u;
    h.ipos = h.idir = h.ri = h.ethr = h.eL = h.shle = h.wpos = h.wdir = v3{ 0, 0, 0 };
    h.t = h.far = h.tau = h.majorant = h.Tr = 0.0f;
    h.mipq = 0;
    h.shadow = 0;
    h.state = ST_NEW;
    h.first = 0;
    h.maj_idx = -2; h.maj_raw = 0u;
}

// VR_SAMPLE_NT (default since round 6): the per-sample radiance is written once and read once, by the accumulation pass, long after it has left the L2: a non-temporal
// store.  Round 4 measured +0.2 ... +0.5 % ("within the noise"), round 6 on the final kernels c2 +0.8 / +1.2 %, c4 +0.4 / +0.9 % in two rounds (profiles/r6h_*)
#ifndef VR_SAMPLE_NT
#define VR_SAMPLE_NT 1
#endif
// result of trace_path: vec4(L, clamp(n_paths, 0, 1)) -> the item's slot of the sample buffer
VR_HD void write_sample(const WorkUnit& wu, uint32_t item, v3 L, uint32_t n_paths) {
    float* o = wu.out + 4u * (size_t)item;
#if defined(__HIP_DEVICE_COMPILE__) && VR_SAMPLE_NT
    // build-time experiment (profiles/r4d_*): the sample pool is written once and read once by the accumulation pass -- non-temporal stores
    typedef float vr_f4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(vr_f4{ L.x, L.y, L.z, n_paths > 0u ? 1.0f : 0.0f }, reinterpret_cast<vr_f4*>(o));
#elif defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<float4*>(o) = make_float4(L.x, L.y, L.z, n_paths > 0u ? 1.0f : 0.0f);
#else
    o[0] = L.x; o[1] = L.y; o[2] = L.z; o[3] = n_paths > 0u ? 1.0f : 0.0f;
#endif
}

// pathtracer_brick.glsl:36: color = mix(color, sanitize(L), 1.f / current_sample)
VR_HD void accumulate_sample(float acc[4], const float L[4], int32_t current_sample) {
    const float a = 1.0f / (float)current_sample;
    acc[0] = mix_(acc[0], sanitize(L[0]), a);
    acc[1] = mix_(acc[1], sanitize(L[1]), a);
    acc[2] = mix_(acc[2], sanitize(L[2]), a);
    acc[3] = mix_(acc[3], sanitize(L[3]), a);
}

// head of sample_volumeDDA / transmittanceDDA (common.glsl:413-421, 459-468) for the ray (pos, d)
// returns false when the ray misses the volume's box altogether (nothing of the segment has been set up then)
template <class K>
VR_HD bool begin_segment(Hot& h, const SceneParams& P, v3 pos, v3 d, int32_t shadow) {
    h.wpos = pos; h.wdir = d;
    h.shadow = shadow;
    h.Tr = 1.0f;          // both set before the branch on purpose: conditional stores to different fields make the
    h.mipq = 12;          // mip = 3; compiler address-select between them, which forces the state into scratch memory
    float tnear, tfar;
    if (!intersect_box(pos, d, P.u.vol_bb_min, P.u.vol_bb_max, tnear, tfar)) {
        h.state = segment_end_state(shadow);
        return false;
    }
    h.ipos = mat4_point(P.u.vol_density_inv_transform, pos);
    h.idir = mat4_dir(P.u.vol_density_inv_transform, d);
    h.far = tfar;                                   // >= tnear >= 0: the sign is free (seg_clean)
    if (K::global == 2 ? P.u.integrator != 0 : K::global == 1) {
        // global-majorant delta / ratio tracking (common.glsl:333-394; compiled out in the reference by USE_DDA):
        // t = near - log(1 - xi) * vol_inv_majorant, then straight to the first tentative collision
        h.ri = v3{ 0, 0, 0 };
        h.tau = 0.0f;
        if (VR_CLEAN_FLAG) h.far = -tfar;            // these trackers' segments are never examined: not clean
        h.t = tnear + neg_log_1m(rng(h.seed)) * P.u.vol_inv_majorant;
        if (h.t < tfar) { h.majorant = P.u.vol_majorant; h.state = ST_COLLIDE; }
        else h.state = segment_end_state(shadow);
        return true;
    }
    h.ri = rcp3_exact(h.idir);
    h.t = tnear + 1e-6f;
    h.tau = neg_log_1m(rng(h.seed));
    h.state = ST_MARCH;
#if VR_CLEAN_FLAG
    {   // a finite ray of moderate size (see seg_clean): every comparison fails for NaN, and inf * 0 is NaN
        const float ax = abs_(h.idir.x), ay = abs_(h.idir.y), az = abs_(h.idir.z), amax = __builtin_fmaxf(ax, __builtin_fmaxf(ay, az));
        const bool clean = (int)(abs_(h.ipos.x) < kCleanBound) & (int)(abs_(h.ipos.y) < kCleanBound) & (int)(abs_(h.ipos.z) < kCleanBound) &
                           (int)(ax * tfar < kCleanBound) & (int)(ay * tfar < kCleanBound) & (int)(az * tfar < kCleanBound) &
                           // ... and for the quotient by div_core (march_finish): DDA steps of moderate length.  (The largest component: a NaN one drops out of the
                           // maximum and has failed its test above.  The majorants' size is the launch's business: vr_kernels.hip pathtrace_variant.)
                           (int)(amax < kCleanBound) & (int)(amax >= 1.0f / kCleanBound);
        h.far = clean ? tfar : -tfar;
    }
#endif
    return true;
}

// pathtracer_brick.glsl:27-30 + common.glsl:76-80; the lane has just been given `item` (< n_items)
// LAZY_EM: also with an emission grid no cold line is written (the scheduler keeps the radiance of a `first` path in its parked hot state: vr_pathtrace.h)
template <class K, class Cold, bool LAZY_EM = false>
VR_HD void do_new(Hot& h, Cold& c, const SceneParams& P, const WorkUnit& wu, uint32_t item) {
    const int32_t W = P.u.resolution[0], H = P.u.resolution[1];
    const int32_t px = wu.px0 + (int32_t)(item & 7u), py = wu.py0 + (int32_t)((item >> 3) & 7u);
    const int32_t smp = wu.first_sample + (int32_t)(item >> 6);
    if (px >= W || py >= H) return;                 // pixel outside a ragged frame: stay in ST_NEW, take the next item
    h.seed = tea32((uint32_t)P.u.seed * (uint32_t)(py * W + px), (uint32_t)smp);
    const float jx = rng(h.seed), jy = rng(h.seed);
    const float fx = (((float)px + jx) - (float)W * 0.5f) / (float)H;
    const float fy = (((float)py + jy) - (float)H * 0.5f) / (float)H;
    const v3 dir = normalize(mat3_mul(P.u.cam_transform, normalize(v3{ fx, fy, P.cam_z })));
    const v3 pos = v3{ P.u.cam_pos[0], P.u.cam_pos[1], P.u.cam_pos[2] };
    const bool hit = begin_segment<K>(h, P, pos, dir, 0);
    // With an emission grid the collision code accumulates into the cold line during the camera segment: then the line is
    // initialised here; otherwise not at all (FirstStash).  A ray that misses the box never collides: always a `first` path.
    const bool lazy = LAZY_EM || !hit || !(K::emission == 2 ? P.u.has_emission != 0 : K::emission == 1);
    if (!lazy) {
        st3(c, C_POS, pos); st3(c, C_DIR, dir);
        st3(c, C_L, v3{ 0, 0, 0 }); st3(c, C_THR, v3{ 1, 1, 1 });
        stu(c, C_NPATHS, 0u); c.st(C_FP, 0.0f); stu(c, C_ITEM, wu.base + item);    // C_ITEM: global slot in the sample buffer
    }
    // the stash travels in ipos / Tr until the path is stored (HotStore::save_new; host: lane_step)
    h.first = lazy ? 1 : 0;
    h.ipos = v3{ lazy ? dir.x : h.ipos.x, lazy ? dir.y : h.ipos.y, lazy ? dir.z : h.ipos.z };
    h.Tr = lazy ? u2f(wu.base + item) : h.Tr;
}
// a `first` path leaves its storage to march: the two fields that held the stash get their real values
VR_HD void first_resume(Hot& h, const SceneParams& P) {
    h.ipos = mat4_point(P.u.vol_density_inv_transform, v3{ P.u.cam_pos[0], P.u.cam_pos[1], P.u.cam_pos[2] });      // as begin_segment computed it
    h.Tr = 1.0f;
}

// Loop body of both DDA trackers up to the collision test (common.glsl:422-435, 469-482), kMarchSteps iterations at a time and
// in two phases.  march_prep does everything that needs no memory -- positions, DDA levels, step lengths of this iteration AND
// of the following ones, each taken as if its predecessors neither collide nor leave the box (step lengths do not depend on the
// majorant) -- and march_load fetches all their majorants together; march_finish replays the reference's loop on those
// values.  A step that an earlier one cancels costs one unused load; the arithmetic of a step that does run is the
// reference's, operation for operation.
#ifndef VR_MARCH_STEPS
#define VR_MARCH_STEPS 2
#endif
constexpr int32_t kMarchSteps = VR_MARCH_STEPS;
#if VR_MARCH_STEPS == 2
// the two-step form written out (the default; the generic loop below compiles ~1 % slower for the same arithmetic)
struct MarchIO { float dt1, dt2, t1; uint32_t maj1, maj2; int32_t i1, i2; bool go1, go2; };     // i*: majorant cell or -1 (outside: majorant 0); maj*: as loaded (majorant_fetch)
VR_HD void march_idle(MarchIO& io) { io.i1 = io.i2 = VR_MAJ_OUTSIDE_CELL ? 0 : -1; io.dt1 = io.dt2 = io.t1 = 0.0f; io.go1 = io.go2 = false; }      // a lane that is not marching (its loads: cell 0)
// CLEAN: every path the call runs for is on a clean segment (seg_clean).  The second step is prepared also when the first one leaves [near, far) (go2 false: its
// results are discarded); its point may then lie beyond the bound, which the clean forms tolerate -- an index is only formed for cells inside the table, and a
// discarded step's NaN is discarded with it
template <int DENSE = 2, int MAJB = 2, bool CLEAN = false>
VR_HD void march_prep(const Hot& h, const SceneParams& P, MarchIO& io) {
    const float far = seg_far(h);
    io.go1 = h.t < far;
    const v3 c1 = axpy(h.ipos, h.t, h.idir);
    const int32_t m1 = round_mip_q(h.mipq);
    io.i1 = majorant_index<DENSE, MAJB, CLEAN>(P.density, c1, m1);
    io.dt1 = step_dda<CLEAN>(c1, h.ri, m1);
    io.t1 = h.t + io.dt1;
    const int32_t q2 = h.mipq < 12 ? h.mipq + 1 : 12;              // mip = min(mip + 0.25, 3)
    const int32_t m2 = round_mip_q(q2);
    io.go2 = io.t1 < far;
    const v3 c2 = axpy(h.ipos, io.t1, h.idir);
    io.i2 = majorant_index<DENSE, MAJB, CLEAN>(P.density, c2, m2);
    io.dt2 = step_dda<CLEAN>(c2, h.ri, m2);
}
// the loads: unconditional and for every lane of the wavefront (an idle lane reads cell 0), so that they sit in straight-line
// code and the compiler's wait counts are exact
template <bool TF>
VR_HD void march_load(const SceneParams& P, MarchIO& io) {
    io.maj1 = majorant_fetch<TF>(P.density, io.i1);
    io.maj2 = majorant_fetch<TF>(P.density, io.i2);
}
// the same with the lane's remembered (index, word) pair (Hot::maj_idx): a first step into the remembered cell loads nothing new
template <bool TF>
VR_HD void march_load_reuse(const SceneParams& P, MarchIO& io, const Hot& h) {
    const bool same = io.i1 == h.maj_idx;                        // (-1 = outside never equals a remembered index: those are >= 0 or -2)
    const uint32_t m1 = majorant_fetch<TF>(P.density, same ? 0 : io.i1);
    io.maj1 = same ? h.maj_raw : m1;
    io.maj2 = majorant_fetch<TF>(P.density, io.i2);
}
// The same loads when the tail of the majorant table -- cells [first, end): the coarse levels, or the whole table of a small grid -- has been copied
// to LDS (vr_pathtrace.h): a lane whose cell lies there reads the copy and sends its global load to cell 0, which all such lanes share (one
// line); when the whole table is resident (first == 0, wave-uniform) no global load is issued at all.  T: uint16_t (raw fp16) or float (TF).
template <bool TF, class T>
VR_HD uint32_t majorant_fetch_lds(const GridView& g, int32_t idx, const T* lds, int32_t first, bool all_resident) {
    const int32_t i = idx < 0 ? 0 : idx;
    const bool in_lds = i >= first;
    const T s = lds[in_lds ? i - first : 0];
    const uint32_t sv = TF ? f2u((float)s) : (uint32_t)s;
    if (all_resident) return sv;
    const uint32_t gv = majorant_fetch<TF>(g, in_lds ? 0 : i);
    return in_lds ? sv : gv;
}
template <bool TF, class T>
VR_HD void march_load_lds(const SceneParams& P, MarchIO& io, const T* lds, int32_t first) {
    const bool all_resident = first == 0;
    io.maj1 = majorant_fetch_lds<TF, T>(P.density, io.i1, lds, first, all_resident);
    io.maj2 = majorant_fetch_lds<TF, T>(P.density, io.i2, lds, first, all_resident);
}
// CLEAN (and no transfer function): the step back to the collision point, tau / majorant, by vr_math.h div_core.  Its domain: the majorant is density_scale x an fp16
// range maximum, in [2^-40, 2^40] for the scales the kernels with a CLEAN form are launched with (2^-16 ... 2^24: vr_kernels.hip pathtrace_variant); tau = (what was left) - majorant x dt <= 0 with dt in [2^-22, 2^27] (a clean
// segment's |idir| lies in [2^-20, 2^20] on one axis at least and below 2^20 on all: step_dda's bracket is between 0.5 and 65), so tau is +0 or has a modulus of at
// least an ulp of the smaller operand, >= 2^-25 majorant dt >= 2^-87, and at most majorant dt: the quotient lies in [2^-48, 2^28].  A majorant of 0 only meets a
// tau of -0 (a free-flight draw of exactly 0 in an empty cell): NaN by either form, and only its NaN-ness is used.
template <bool TF, bool REUSE = false, bool CLEAN = false>
VR_HD void march_finish(Hot& h, const SceneParams& P, const MarchIO& io) {
    if (!io.go1) { h.state = segment_end_state(h.shadow); return; }
    float t = io.t1, maj = majorant_of<TF>(P, io.i1, io.maj1);
    float tau = h.tau - maj * io.dt1;
    int32_t q = h.mipq < 12 ? h.mipq + 1 : 12;
    // REUSE: the cell whose majorant the lane leaves the pass with (Hot::maj_idx; selects, not conditional stores)
    if (REUSE) { const bool keep = VR_MAJ_OUTSIDE_CELL || io.i1 >= 0; h.maj_idx = keep ? io.i1 : h.maj_idx; h.maj_raw = keep ? io.maj1 : h.maj_raw; }
    if (tau > 0.0f) {                                              // no tentative collision in the first cell: second step
        if (!io.go2) { h.t = t; h.tau = tau; h.mipq = q; h.state = segment_end_state(h.shadow); return; }
        maj = majorant_of<TF>(P, io.i2, io.maj2);
        if (REUSE) { const bool keep = VR_MAJ_OUTSIDE_CELL || io.i2 >= 0; h.maj_idx = keep ? io.i2 : h.maj_idx; h.maj_raw = keep ? io.maj2 : h.maj_raw; }
        t = io.t1 + io.dt2;
        tau = tau - maj * io.dt2;
        q = q < 12 ? q + 1 : 12;
        if (tau > 0.0f) { h.t = t; h.tau = tau; h.mipq = q; return; }
    }
    t += (CLEAN && !TF) ? div_core(tau, maj) : tau / maj;
    h.t = t; h.tau = tau; h.mipq = q;
    if (t >= seg_far(h)) { h.state = segment_end_state(h.shadow); return; }
    h.majorant = maj;
    h.state = ST_COLLIDE;
}
#else
struct MarchIO {
    float dt[kMarchSteps], t[kMarchSteps];      // step length; ray parameter after the step
    uint32_t maj[kMarchSteps];                  // majorant of the step's cell as loaded (majorant_fetch)
    int32_t idx[kMarchSteps];                   // its table index or -1 (outside: majorant 0)
    bool go[kMarchSteps];                       // the step starts inside [near, far)
};
VR_HD void march_idle(MarchIO& io) {             // a lane that is not marching
#pragma unroll
    for (int k = 0; k < kMarchSteps; ++k) { io.idx[k] = VR_MAJ_OUTSIDE_CELL ? 0 : -1; io.dt[k] = io.t[k] = 0.0f; io.go[k] = false; }
}
template <int DENSE = 2, int MAJB = 2, bool CLEAN = false>
VR_HD void march_prep(const Hot& h, const SceneParams& P, MarchIO& io) {
    float t = h.t;
    int32_t q = h.mipq;
#pragma unroll
    for (int k = 0; k < kMarchSteps; ++k) {
        io.go[k] = t < seg_far(h);
        const v3 c = axpy(h.ipos, t, h.idir);
        const int32_t m = round_mip_q(q);
        io.idx[k] = majorant_index<DENSE, MAJB, CLEAN>(P.density, c, m);
        io.dt[k] = step_dda<CLEAN>(c, h.ri, m);
        t = t + io.dt[k];
        io.t[k] = t;
        q = q < 12 ? q + 1 : 12;                                   // mip = min(mip + 0.25, 3)
    }
}
// the loads: unconditional and for every lane of the wavefront (an idle lane reads cell 0), so that they sit in straight-line
// code and the compiler's wait counts are exact
template <bool TF>
VR_HD void march_load(const SceneParams& P, MarchIO& io) {
#pragma unroll
    for (int k = 0; k < kMarchSteps; ++k) io.maj[k] = majorant_fetch<TF>(P.density, io.idx[k]);
}
template <bool TF, class T>
VR_HD void march_load_lds(const SceneParams&, MarchIO&, const T*, int32_t) { static_assert(sizeof(T) == 0, "VR_MAJ_LDS is written for VR_MARCH_STEPS == 2"); }
template <bool TF>
VR_HD void march_load_reuse(const SceneParams& P, MarchIO& io, const Hot&) { march_load<TF>(P, io); }      // (majorant reuse is written for VR_MARCH_STEPS == 2)
template <bool TF, bool REUSE = false, bool CLEAN = false>
VR_HD void march_finish(Hot& h, const SceneParams& P, const MarchIO& io) {
    static_assert(!REUSE, "majorant reuse is written for VR_MARCH_STEPS == 2");
    float tau = h.tau, maj = 0.0f, t = h.t;
    int32_t q = h.mipq;
#pragma unroll
    for (int k = 0; k < kMarchSteps; ++k) {
        if (!io.go[k]) { h.t = t; h.tau = tau; h.mipq = q; h.state = segment_end_state(h.shadow); return; }
        maj = majorant_of<TF>(P, io.idx[k], io.maj[k]);
        t = io.t[k];
        tau = tau - maj * io.dt[k];
        q = q < 12 ? q + 1 : 12;
        if (!(tau > 0.0f)) goto tentative_collision;
    }
    h.t = t; h.tau = tau; h.mipq = q;                              // still marching
    return;
tentative_collision:
    t += (CLEAN && !TF) ? div_core(tau, maj) : tau / maj;
    h.t = t; h.tau = tau; h.mipq = q;
    if (t >= seg_far(h)) { h.state = segment_end_state(h.shadow); return; }
    h.majorant = maj;
    h.state = ST_COLLIDE;
}
#endif
// one iteration (sequential form; the scheduler uses the two-phase form above)
template <bool TF, int DENSE = 2, int MAJB = 2>
VR_HD void do_march(Hot& h, const SceneParams& P) {
    if (!(h.t < seg_far(h))) { h.state = segment_end_state(h.shadow); return; }
    const v3 curr = axpy(h.ipos, h.t, h.idir);
    const int32_t m = round_mip_q(h.mipq);
    const float majorant = majorant_at<TF, DENSE, MAJB>(P, curr, m);
    const float dt = step_dda(curr, h.ri, m);
    h.t += dt;
    h.tau -= majorant * dt;
    h.mipq = h.mipq < 12 ? h.mipq + 1 : 12;                 // mip = min(mip + 0.25, 3)
    if (h.tau > 0.0f) return;
    h.t += h.tau / majorant;
    if (h.t >= seg_far(h)) { h.state = segment_end_state(h.shadow); return; }
    h.majorant = majorant;
    h.state = ST_COLLIDE;
}

// Tentative collision (common.glsl:436-452, 483-498; global-majorant trackers :342-359, 372-392), in two phases like the
// march: collide_prep draws the filter taps (the RNG draws of lookup_density / lookup_emission, in the reference's order),
// computes the voxel addresses and issues the loads; collide_finish evaluates density (and emission), makes the real / null
// decision and draws the next free-flight distance.
template <class K> struct CollideIO {
    TapAddr a; TapData d;        // no transfer function: the one stochastic-tricubic tap
    TriIO tri;                   // transfer function: the 8 trilinear corners
    TapAddr ea; TapData ed;      // emission tap (camera/scatter segments with an emission grid)
};
// PE: where the emission grid's view and transform are read from.  The scheduler passes the kernel arguments behind a pointer there
// (event_args(), vr_pathtrace.h): ~45 uniform dwords that only the emission tap needs are then fetched by scalar loads inside the collision
// code instead of living in scalar registers through the whole scheduler loop (the emission kernels were the ones spilling them).
// CLEAN: the path stands on a clean segment (seg_clean) -- its collision point is finite, or NaN on all three axes (t = NaN): the density tap needs no NaN guard
// (the emission tap keeps it: its point goes through one more transform)
template <class K, bool CLEAN = false>
VR_HD void collide_prep(Hot& h, const SceneParams& P, const SceneParams& PE, CollideIO<K>& io) {
    const v3 ip = axpy(h.ipos, h.t, h.idir);
    if (K::tf) {
        trilinear_prep<K::dense, !CLEAN>(P.density, ip, io.tri);
    } else {
        int32_t tx, ty, tz;
        tricubic_tap<!CLEAN>(ip, h.seed, tx, ty, tz);
        io.a = tap_addr<K::dense>(P.density, tx, ty, tz);
    }
    io.ea.cell = io.ea.off = 0u; io.ea.in = false;
    if (!h.shadow) {
        // lookup_emission's filter draws its 9 numbers whether or not an emission grid is bound
        if (K::emission == 2 ? P.u.has_emission != 0 : K::emission == 1) {
            const v3 ie = mat4_point(PE.emission_from_density, ip);
            int32_t ex, ey, ez;
            tricubic_tap(ie, h.seed, ex, ey, ez);
            io.ea = tap_addr<K::edense>(PE.emission, ex, ey, ez);
        } else {
            rng_skip9(h.seed);
        }
    }
}
template <class K>
VR_HD void collide_idle(CollideIO<K>& io) {      // a lane that is not colliding: its loads read cell 0
    io.a.cell = io.a.off = 0u; io.a.in = false;
    io.ea.cell = io.ea.off = 0u; io.ea.in = false;
    if (K::tf) trilinear_idle(io.tri);
}
template <class K, bool CLEAN = false>
VR_HD void collide_prep(Hot& h, const SceneParams& P, CollideIO<K>& io) { collide_prep<K, CLEAN>(h, P, P, io); }
template <class K>
VR_HD void collide_load(const SceneParams& P, const SceneParams& PE, CollideIO<K>& io) {      // unconditional, like march_load
    if (K::tf) trilinear_load<K::dense, K::pair_d>(P.density, io.tri);
    else io.d = tap_load<K::dense, K::pair_d>(P.density, io.a);
    if (K::emission == 2 ? P.u.has_emission != 0 : K::emission == 1) io.ed = tap_load<K::edense, K::pair_e>(PE.emission, io.ea);
}
template <class K>
VR_HD void collide_load(const SceneParams& P, CollideIO<K>& io) { collide_load<K>(P, P, io); }
// CACHED: throughput and radiance of the marching path are in h.ethr / h.eL (device scheduler, see Hot); otherwise on the cold line
template <class K, class Cold, bool CACHED = false>
VR_HD void collide_finish(Hot& h, Cold& c, const SceneParams& P, const SceneParams& PE, const CollideIO<K>& io, const float* tf_lut) {
    constexpr bool USE_TF = K::tf;
    const Uniforms& u = P.u;
    const bool global = K::global == 2 ? u.integrator != 0 : K::global == 1;
    float d;
    float rgba[4] = { 0, 0, 0, 0 };
    if (USE_TF) {
        tf_lookup_at(u, tf_lut, (u.vol_density_scale * trilinear_value<K::dense>(P.density, io.tri)) * u.vol_inv_majorant, rgba);
        d = u.vol_majorant * rgba[3];
    } else {
        d = u.vol_density_scale * tap_value<K::dense>(P.density, io.d, io.a.in);
    }
    if (!h.shadow) {
        const float P_real = d * u.vol_inv_majorant;                 // global trackers only
        if (K::emission == 2 ? u.has_emission != 0 : K::emission == 1) {
            // Le += throughput * (1 - albedo) * lookup_emission(...) * d * vol_inv_majorant
            const float tt = tap_value<K::edense>(PE.emission, io.ed, io.ea.in) * u.vol_emission_norm;
            const v3 e3 = v3{ tt, sqr(tt), sqr(sqr(tt)) };
            const v3 em = v3{ u.vol_emission_scale * sqr(e3.x), u.vol_emission_scale * sqr(e3.y), u.vol_emission_scale * sqr(e3.z) };
            const v3 oma = v3{ 1.0f - u.vol_albedo[0], 1.0f - u.vol_albedo[1], 1.0f - u.vol_albedo[2] };
            const v3 thr = CACHED ? h.ethr : ld3(c, C_THR);
            const v3 L0 = CACHED ? h.eL : ld3(c, C_L);
            const v3 L1 = global ? L0 + ((thr * oma) * em) * P_real : L0 + (((thr * oma) * em) * d) * u.vol_inv_majorant;
            if (CACHED) h.eL = L1; else st3(c, C_L, L1);
        }
        if (global ? rng(h.seed) < P_real : rng(h.seed) * h.majorant < d) {
            // real collision.  "throughput *= albedo [* rgba.rgb]" is applied by do_nee (the one event that follows): the
            // hot pair then never touches the path's cold state in global memory; a transfer-function colour travels there in C_COL
            if (USE_TF) st3(c, C_COL, v3{ rgba[0], rgba[1], rgba[2] });
            h.state = ST_NEE;
            return;
        }
    } else if (global) {
        h.Tr *= 1.0f - d * u.vol_inv_majorant;
        if (h.Tr < 0.1f) {
            const float prob = 1.0f - h.Tr;
            if (rng(h.seed) < prob) { h.Tr = 0.0f; h.state = ST_POSTNEE; return; }
            h.Tr /= 1.0f - prob;
        }
    } else {
        if (rng(h.seed) * h.majorant < d) {
            // (kernels of one scene kind without a transfer function: both majorants are density_scale x an fp16 number, density_scale in [2^-16, 2^24] -- vr_kernels.hip
            // pathtrace_variant --, the quotient in [1, 2^40]: div_core's domain.  The lane of a NaN collision point, whose cell majorant is 0, does not get here: 0 < 0.)
            const float ratio = (K::global == 0 && !K::tf) ? div_core(u.vol_majorant, h.majorant) : u.vol_majorant / h.majorant;
            h.Tr *= max_(0.0f, 1.0f - ratio);
            if (h.Tr < 0.1f) {
                const float prob = 1.0f - h.Tr;
                if (rng(h.seed) < prob) { h.Tr = 0.0f; h.state = ST_POSTNEE; return; }
                h.Tr /= 1.0f - prob;
            }
        }
    }
    if (global) {
        // stays in ST_COLLIDE while the ray is inside the box
        h.t = h.t + neg_log_1m(rng(h.seed)) * u.vol_inv_majorant;
        if (!(h.t < seg_far(h))) h.state = segment_end_state(h.shadow);
        return;
    }
    h.tau = neg_log_1m(rng(h.seed));
    h.mipq = h.mipq > 8 ? h.mipq - 8 : 0;                  // mip = max(0, mip - 2)
    h.state = ST_MARCH;
}
template <class K, class Cold, bool CACHED = false>
VR_HD void collide_finish(Hot& h, Cold& c, const SceneParams& P, const CollideIO<K>& io, const float* tf_lut) { collide_finish<K, Cold, CACHED>(h, c, P, P, io, tf_lut); }
template <class K, class Cold, bool CLEAN = false>
VR_HD void do_collide(Hot& h, Cold& c, const SceneParams& P) {
    CollideIO<K> io;
    collide_prep<K, CLEAN>(h, P, io);
    collide_load<K>(P, io);
    collide_finish<K>(h, c, P, io, P.tf_lut);
}

// real collision: common.glsl:611-626 up to (and including the set-up of) the transmittance call.
// `crd` is what the path's state is READ from: its own cold line, or -- for a `first` path, whose line holds nothing yet and whose
// loads are discarded -- any line that is cheap to read (the scheduler passes one that the whole batch shares).  The loads stay
// unconditional, followed by component-wise selects: a conditional block makes the compiler select between addresses and put
// the path state into scratch memory.
// SHLE_IN_HOT: the radiance of the light sample goes to h.shle (the scheduler parks it in registers) instead of the side array;
// ITEM_IN_HOT: likewise the path's slot in the sample buffer (h.item)
// FIRST_L_IN_HOT: the radiance a `first` path gathered on its camera segment (emission) is in h.eL instead of being 0
template <class K, class Cold, bool SHLE_IN_HOT = false, bool ITEM_IN_HOT = false, bool FIRST_L_IN_HOT = false>
VR_HD void do_nee(Hot& h, Cold& c, const Cold& crd, const SceneParams& P) {
    const bool first = h.first != 0;
    constexpr bool WS = world_slot<K>();
    if (!WS) {
        // The slot's values are REQUESTED first and touched last: the light sample below -- nine table levels and the texels, seven dependent round trips -- needs none of
        // them, so the line's latency (it has left the L2 since the path's last event, as a rule) passes under the sample's instead of in front of it (round 6).  No draw
        // moves: the code between here and the sample never drew.
        v3 dir = ld3(crd, C_DIR), pos0 = ld3(crd, C_POS), thr = ld3(crd, C_THR);
        h.first = 0;
        const float r0 = rng(h.seed), r1 = rng(h.seed);
        float pdf;
        v3 w_i, Le;
        sample_environment<K::global != 2, K::global != 2>(P, r0, r1, w_i, Le, pdf);      // (the kernels of one scene kind; the run-time variant loads record by record and divides in full)
        dir = v3{ first ? h.ipos.x : dir.x, first ? h.ipos.y : dir.y, first ? h.ipos.z : dir.z };
        pos0 = v3{ first ? P.u.cam_pos[0] : pos0.x, first ? P.u.cam_pos[1] : pos0.y, first ? P.u.cam_pos[2] : pos0.z };
        thr = v3{ first ? 1.0f : thr.x, first ? 1.0f : thr.y, first ? 1.0f : thr.z };
        const v3 pos = axpy(pos0, h.t, dir);
        // the real collision that led here: throughput *= albedo [* rgba.rgb] (common.glsl:383-388, 491-495; see collide_finish)
        const v3 alb = v3{ P.u.vol_albedo[0], P.u.vol_albedo[1], P.u.vol_albedo[2] };
        if (K::global == 2 ? P.u.integrator != 0 : K::global == 1) thr = K::tf ? thr * (ld3(c, C_COL) * alb) : thr * alb;
        else { thr = thr * alb; if (K::tf) thr = thr * ld3(c, C_COL); }
        if (first) {
            // what do_new left unwritten
            st3(c, C_L, FIRST_L_IN_HOT ? h.eL : v3{ 0, 0, 0 }); stu(c, C_NPATHS, 0u);      // (in the order of the layout: two 16-byte stores)
            st3(c, C_DIR, dir);
            c.st(C_FP, 0.0f);
            if (!ITEM_IN_HOT) stu(c, C_ITEM, f2u(h.Tr));       // else the scheduler takes it from the stash (h.Tr) before this call
        }
        const bool lit = pdf > 0.0f;                                       // (false for NaN; sh_pdf = 0 marks "no next-event estimate" for do_postnee)
        const float f_p = phase_hg(dot(-dir, w_i), P.u.vol_phase_g);       // with sh_pdf and thr: what do_postnee needs for the sample's weight (thr * mis) * f_p
        // (pos, sh_pdf) and (thr, f_pl): two 16-byte stores that fill sector 0 (profiles/r6i_*)
        st3(c, C_POS, pos); c.st(C_SHPDF, lit ? pdf : 0.0f);
        st3(c, C_THR, thr); c.st(C_FPL, lit ? f_p : 0.0f);
        if (lit) {
            if (SHLE_IN_HOT) h.shle = Le; else st3(c, C_SHLE, Le);
            begin_segment<K>(h, P, pos, w_i, 1);
        } else {
            h.shadow = 0;
            h.state = ST_POSTNEE;
        }
        return;
    }
    v3 dir, pos0, thr;
    if (WS) {
        // the segment's origin and direction are the path's own (begin_segment's arguments, parked with it): the values C_POS / C_DIR hold, without the line
        dir = h.wdir; pos0 = h.wpos; thr = v3{ 1.0f, 1.0f, 1.0f };
    } else {
        dir = ld3(crd, C_DIR); pos0 = ld3(crd, C_POS); thr = ld3(crd, C_THR);
        dir = v3{ first ? h.ipos.x : dir.x, first ? h.ipos.y : dir.y, first ? h.ipos.z : dir.z };
        pos0 = v3{ first ? P.u.cam_pos[0] : pos0.x, first ? P.u.cam_pos[1] : pos0.y, first ? P.u.cam_pos[2] : pos0.z };
        thr = v3{ first ? 1.0f : thr.x, first ? 1.0f : thr.y, first ? 1.0f : thr.z };
    }
    const v3 pos = axpy(pos0, h.t, dir);
    // the real collision that led here: throughput *= albedo [* rgba.rgb] (common.glsl:383-388, 491-495; see collide_finish)
    if (WS) {
        // ... is applied by do_postnee, the event that follows every collision event and holds the line anyway; a path's first collision starts the line with throughput 1
        if (first) st3(c, C_THR, thr);
        // (pos, sh_pdf) and (this segment's direction, f_pl) are written at the end, unconditionally and side by side: two 16-byte stores that fill sector 0 of the slot's
        // swapped layout (vr_pathtrace.h ColdGlobalT) -- this event dirties sector 0 only, the scatter event sector 1 only
    } else
    {
        const v3 alb = v3{ P.u.vol_albedo[0], P.u.vol_albedo[1], P.u.vol_albedo[2] };
        if (K::global == 2 ? P.u.integrator != 0 : K::global == 1) thr = K::tf ? thr * (ld3(c, C_COL) * alb) : thr * alb;
        else { thr = thr * alb; if (K::tf) thr = thr * ld3(c, C_COL); }
        // (pos, sh_pdf) and (thr, f_pl) are written at the end, as two 16-byte stores that fill sector 0 (profiles/r6i_*)
    }
    if (first) {
        // what do_new left unwritten
        st3(c, C_L, FIRST_L_IN_HOT ? h.eL : v3{ 0, 0, 0 }); stu(c, C_NPATHS, 0u);      // (in the order of the layout: two 16-byte stores)
        if (!WS) st3(c, C_DIR, dir);
        c.st(C_FP, 0.0f);
        if (!ITEM_IN_HOT) stu(c, C_ITEM, f2u(h.Tr));       // else the scheduler takes it from the stash (h.Tr) before this call
    }
    h.first = 0;
    const float r0 = rng(h.seed), r1 = rng(h.seed);
    float pdf;
    v3 w_i, Le;
    sample_environment<K::global != 2, K::global != 2>(P, r0, r1, w_i, Le, pdf);      // (the kernels of one scene kind; the run-time variant loads record by record and divides in full)
    if (WS) {
        const bool lit = pdf > 0.0f;                                   // (false for NaN: "no next-event estimate", sh_pdf = 0 for do_postnee)
        // ONE 16-byte store: this segment's direction and sh_pdf.  The collision point stays with the path (the shadow segment starts there: h.wpos), and the phase
        // function's value for the light sample is evaluated by do_postnee, from this direction and the shadow segment's (h.wdir = w_i): same operands, same value
        st3(c, C_DIR, dir); c.st(C_SHPDF, lit ? pdf : 0.0f);
        if (lit) {
            if (SHLE_IN_HOT) h.shle = Le; else st3(c, C_SHLE, Le);
            begin_segment<K>(h, P, pos, w_i, 1);
        } else {
            h.wpos = pos;                                              // (do_postnee takes the collision point from here)
            h.shadow = 0;
            h.state = ST_POSTNEE;
        }
        return;
    }
    const bool lit = pdf > 0.0f;                                       // (false for NaN; sh_pdf = 0 marks "no next-event estimate" for do_postnee)
    const float f_p = phase_hg(dot(-dir, w_i), P.u.vol_phase_g);       // with sh_pdf and thr: what do_postnee needs for the sample's weight (thr * mis) * f_p
    st3(c, C_POS, pos); c.st(C_SHPDF, lit ? pdf : 0.0f);
    st3(c, C_THR, thr); c.st(C_FPL, lit ? f_p : 0.0f);
    if (lit) {
        if (SHLE_IN_HOT) h.shle = Le; else st3(c, C_SHLE, Le);
        begin_segment<K>(h, P, pos, w_i, 1);
    } else {
        h.shadow = 0;
        h.state = ST_POSTNEE;
    }
}

// common.glsl:625-641, then the head of the next sample_volumeDDA call
template <class K, class Cold, bool SHLE_IN_HOT = false, bool ITEM_IN_HOT = false>
VR_HD void do_postnee(Hot& h, Cold& c, const SceneParams& P, const WorkUnit& wu) {
    // the whole slot, as the 16-byte groups its layout puts side by side: four loads (with VR_WORLD_SLOT three -- (dir, sh_pdf), (L, n_paths), (thr, f_p); vr_pathtrace.h
    // ColdGlobalT), all in flight at once (round 6; before: seven, the last two -- dir, pos -- issued only after the roulette)
    constexpr bool WS = world_slot<K>();
    const Quad q_l = ld4(c, C_L, C_NPATHS), q_p = WS ? Quad{ h.wpos, 0.0f } : ld4(c, C_POS, C_SHPDF);
    const Quad q_t = ld4(c, C_THR, WS ? C_FP : C_FPL), q_d = ld4(c, C_DIR, WS ? C_SHPDF : C_FP);
    v3 L = q_l.a;
    const float sh_pdf = WS ? q_d.b : q_p.b;
    v3 thr = q_t.a;
    if (WS) thr = thr * v3{ P.u.vol_albedo[0], P.u.vol_albedo[1], P.u.vol_albedo[2] };      // the real collision's "throughput *= albedo" (see do_nee): same operands, same product
    const float fpl_kept = WS ? 0.0f : q_t.b;
    if (sh_pdf > 0.0f) {
        // common.glsl:620-626: L += throughput * mis * f_p * Tr * Le / pdf, the factors of the light sample do_nee drew
        // (VR_WORLD_SLOT: f_p from the incoming direction and the shadow segment's, which the path still carries -- the expression do_nee's other form evaluates)
        const float f_p = WS ? phase_hg(dot(-q_d.a, h.wdir), P.u.vol_phase_g) : fpl_kept;
        const float mis = P.u.show_environment > 0 ? power_heuristic(sh_pdf, f_p) : 1.0f;
        L = L + ((((thr * mis) * f_p) * h.Tr) * (SHLE_IN_HOT ? h.shle : ld3(c, C_SHLE))) / sh_pdf;
    }                                    // (sector 1 is written once, at the end, by the paths that go on)
    const uint32_t n_paths = f2u(q_l.b) + 1u;
    if (n_paths >= (uint32_t)P.u.bounces) { write_sample(wu, ITEM_IN_HOT ? h.item : ldu(c, C_ITEM), L, n_paths); h.state = ST_NEW; return; }
    const float rr = luma(thr);
    if (rr < 0.1f) {
        const float prob = 1.0f - rr;
        if (rng(h.seed) < prob) { write_sample(wu, ITEM_IN_HOT ? h.item : ldu(c, C_ITEM), L, n_paths); h.state = ST_NEW; return; }
        thr = thr / (1.0f - prob);
        if (!WS) { st3(c, C_THR, thr); c.st(C_FPL, fpl_kept); }       // (16 bytes; only after a roulette: the one write of this event to sector 0)
    }
    const v3 dir = q_d.a;
    const float s0 = rng(h.seed), s1 = rng(h.seed);
    const v3 sd = sample_phase_hg(dir, P.u.vol_phase_g, s0, s1);
    const float f_p_next = phase_hg(dot(-dir, sd), P.u.vol_phase_g);
    if (WS) {
        // (L, n_paths) and (thr, f_p): two 16-byte stores that fill sector 1; the next collision event writes its segment's direction itself, an escape reads it from the path's slot
        st3(c, C_L, L); stu(c, C_NPATHS, n_paths);
        st3(c, C_THR, thr); c.st(C_FP, f_p_next);
    } else {
        st3(c, C_L, L); stu(c, C_NPATHS, n_paths);
        st3(c, C_DIR, sd); c.st(C_FP, f_p_next);
    }
    begin_segment<K>(h, P, q_p.a, sd, 0);
}

// common.glsl:644-651
// `c` of a `first` path (never scattered: L = 0, throughput 1; direction and sample slot in the stash) is only read and the
// values discarded: the scheduler points it at a line the batch shares (see do_nee)
template <class Cold, bool ITEM_IN_HOT = false, bool FIRST_L_IN_HOT = false, bool WS = false>
VR_HD void do_escape(Hot& h, const Cold& c, const SceneParams& P, const WorkUnit& wu) {
    const bool first = h.first != 0;
    v3 L = ld3(c, C_L), thr = ld3(c, C_THR), dir = ld3(c, C_DIR);
    uint32_t n_paths = ldu(c, C_NPATHS), item = ITEM_IN_HOT ? h.item : ldu(c, C_ITEM);
    const float f_p = c.ld(C_FP);
    // VR_WORLD_SLOT: the lookup needs nothing of the slot -- it runs BEFORE the slot's values are touched, the line's latency under the lookup's arithmetic and texel fetch
    // (round 6; before, the selects below waited for the line first: two round trips one after the other)
    v3 Le_early = v3{ 0.0f, 0.0f, 0.0f };
    if (WS && P.u.show_environment > 0) Le_early = lookup_environment(P, h.wdir);
    const v3 L_first = FIRST_L_IN_HOT ? h.eL : v3{ 0.0f, 0.0f, 0.0f };
    L = v3{ first ? L_first.x : L.x, first ? L_first.y : L.y, first ? L_first.z : L.z };
    thr = v3{ first ? 1.0f : thr.x, first ? 1.0f : thr.y, first ? 1.0f : thr.z };
    if (WS) dir = h.wdir;        // (VR_WORLD_SLOT: the escaping segment's own direction, first path or not)
    else dir = v3{ first ? h.ipos.x : dir.x, first ? h.ipos.y : dir.y, first ? h.ipos.z : dir.z };
    n_paths = first ? 0u : n_paths;
    item = first ? f2u(h.Tr) : item;
    h.first = 0;
    if (P.u.show_environment > 0) {
        const v3 Le = WS ? Le_early : lookup_environment(P, dir);
        float mis = 1.0f;
        if (n_paths > 0u) {
            const float avg_w = env_average_importance(P);
            const float pdf_env = (luma(Le) / avg_w) * kInv4Pi;
            mis = power_heuristic(f_p, pdf_env);
        }
        L = L + (thr * mis) * Le;
    }
    write_sample(wu, item, L, n_paths);
    h.state = ST_NEW;
}

// ---------------------------------------------------------------------------------------------------
// direct_volume_rendering (common.glsl:571-591): 64 jittered steps of emission-absorption compositing through the
// transfer function.  Dead code in the reference (no kernel calls it); offered as integrator = 2 (needs a LUT).  One
// call per (pixel, sample): no path state, no scheduler.  .w = opacity 1 - Tr (not defined by the reference).
VR_HD void dvr_sample(const SceneParams& P, int32_t px, int32_t py, int32_t smp, float out[4]) {
    const Uniforms& u = P.u;
    const int32_t W = u.resolution[0], H = u.resolution[1];
    uint32_t seed = tea32((uint32_t)u.seed * (uint32_t)(py * W + px), (uint32_t)smp);
    const float jx = rng(seed), jy = rng(seed);
    const float fx = (((float)px + jx) - (float)W * 0.5f) / (float)H;
    const float fy = (((float)py + jy) - (float)H * 0.5f) / (float)H;
    const v3 dir = normalize(mat3_mul(u.cam_transform, normalize(v3{ fx, fy, P.cam_z })));
    const v3 pos = v3{ u.cam_pos[0], u.cam_pos[1], u.cam_pos[2] };
    v3 L = v3{ 0, 0, 0 };
    float tnear, tfar;
    if (!intersect_box(pos, dir, u.vol_bb_min, u.vol_bb_max, tnear, tfar)) {
        const v3 e = lookup_environment(P, dir);
        out[0] = e.x; out[1] = e.y; out[2] = e.z; out[3] = 0.0f;
        return;
    }
    const v3 ipos = mat4_point(u.vol_density_inv_transform, pos);
    const v3 idir = mat4_dir(u.vol_density_inv_transform, dir);
    const float dt = (tfar - tnear) / 64.0f;
    tnear += rng(seed) * dt;
    float Tr = 1.0f;
    for (int32_t i = 0; i < 64; ++i) {
        float rgba[4];
        const v3 ip = axpy(ipos, min_(tnear + (float)i * dt, tfar), idir);
        tf_lookup(P, (u.vol_density_scale * density_trilinear_raw(P.density, ip)) * u.vol_inv_majorant, rgba);
        const float dtau = rgba[3] * u.vol_majorant * dt;
        L = L + (v3{ rgba[0], rgba[1], rgba[2] } * dtau) * Tr;
        Tr *= exp_(-dtau);
        if (Tr <= 1e-6f) { out[0] = L.x; out[1] = L.y; out[2] = L.z; out[3] = 1.0f - Tr; return; }
    }
    const v3 e = lookup_environment(P, dir);
    L = L + e * Tr;
    out[0] = L.x; out[1] = L.y; out[2] = L.z; out[3] = 1.0f - Tr;
}

// ---------------------------------------------------------------------------------------------------
// trace_path with the 64-step ray-marching trackers (common.glsl:506-566: transmittance_raymarch, sample_volume_raymarch;
// RAYMARCH_STEPS 64).  Dead code in the reference -- no kernel calls them; trace_path only switches between the DDA and the
// global-majorant pair -- offered as integrator = 3.  Every step costs a stochastic-tricubic tap (9 draws), also with a
// transfer function.  One call per (pixel, sample), sequential: no path state, no scheduler (like dvr_sample).
VR_HD float raymarch_density(const SceneParams& P, v3 ip, uint32_t& seed) {
    int32_t tx, ty, tz;
    tricubic_tap(ip, seed, tx, ty, tz);
    return P.u.vol_density_scale * brick_value<2>(P.density, tx, ty, tz);
}
VR_HD float transmittance_raymarch(const SceneParams& P, v3 wpos, v3 wdir, uint32_t& seed) {
    const Uniforms& u = P.u;
    float tnear, tfar;
    if (!intersect_box(wpos, wdir, u.vol_bb_min, u.vol_bb_max, tnear, tfar)) return 1.0f;
    const v3 ipos = mat4_point(u.vol_density_inv_transform, wpos);
    const v3 idir = mat4_dir(u.vol_density_inv_transform, wdir);
    const float dt = (tfar - tnear) / 64.0f;
    tnear += rng(seed) * dt;
    float tau = 0.0f;
    for (int32_t i = 0; i < 64; ++i) {
        const float d = raymarch_density(P, axpy(ipos, min_(tnear + (float)i * dt, tfar), idir), seed);
        if (u.use_tf) {
            float rgba[4];
            tf_lookup(P, d * u.vol_inv_majorant, rgba);
            tau += rgba[3] * u.vol_majorant * dt;
        } else {
            tau += d * dt;
        }
    }
    return exp_(-tau);
}
VR_HD bool sample_volume_raymarch(const SceneParams& P, v3 wpos, v3 wdir, float& t, v3& throughput, uint32_t& seed) {
    const Uniforms& u = P.u;
    float tnear, tfar;
    if (!intersect_box(wpos, wdir, u.vol_bb_min, u.vol_bb_max, tnear, tfar)) return false;
    const v3 ipos = mat4_point(u.vol_density_inv_transform, wpos);
    const v3 idir = mat4_dir(u.vol_density_inv_transform, wdir);
    const float tau_target = neg_log_1m(rng(seed));
    const float dt = (tfar - tnear) / 64.0f;
    tnear += rng(seed) * dt;
    float tau = 0.0f;
    for (int32_t i = 0; i < 64; ++i) {
        t = min_(tnear + (float)i * dt, tfar);
        const float d = raymarch_density(P, axpy(ipos, t, idir), seed);
        float rgba[4] = { 0, 0, 0, 0 };
        if (u.use_tf) {
            tf_lookup(P, d * u.vol_inv_majorant, rgba);
            tau += rgba[3] * u.vol_majorant * dt;
        } else {
            tau += d * dt;
        }
        if (tau >= tau_target) {
            // throughput *= albedo (the `pdf` output of the reference function has no consumer in trace_path)
            const v3 alb = u.use_tf ? v3{ rgba[0] * u.vol_albedo[0], rgba[1] * u.vol_albedo[1], rgba[2] * u.vol_albedo[2] }
                                    : v3{ u.vol_albedo[0], u.vol_albedo[1], u.vol_albedo[2] };
            throughput = throughput * alb;
            return true;
        }
    }
    return false;
}
// pathtracer_brick*.glsl main + trace_path (common.glsl:599-652) around the two trackers above
VR_HD void raymarch_path_sample(const SceneParams& P, int32_t px, int32_t py, int32_t smp, float out[4]) {
    const Uniforms& u = P.u;
    const int32_t W = u.resolution[0], H = u.resolution[1];
    uint32_t seed = tea32((uint32_t)u.seed * (uint32_t)(py * W + px), (uint32_t)smp);
    const float jx = rng(seed), jy = rng(seed);
    const float fx = (((float)px + jx) - (float)W * 0.5f) / (float)H;
    const float fy = (((float)py + jy) - (float)H * 0.5f) / (float)H;
    v3 dir = normalize(mat3_mul(u.cam_transform, normalize(v3{ fx, fy, P.cam_z })));
    v3 pos = v3{ u.cam_pos[0], u.cam_pos[1], u.cam_pos[2] };
    v3 L = v3{ 0, 0, 0 }, thr = v3{ 1, 1, 1 };
    bool free_path = true;
    uint32_t n_paths = 0u;
    float t = 0.0f, f_p = 0.0f;
    while (sample_volume_raymarch(P, pos, dir, t, thr, seed)) {
        pos = axpy(pos, t, dir);
        const float r0 = rng(seed), r1 = rng(seed);
        float pdf;
        v3 w_i, Le;
        sample_environment(P, r0, r1, w_i, Le, pdf);
        if (pdf > 0.0f) {
            f_p = phase_hg(dot(-dir, w_i), u.vol_phase_g);
            const float mis = u.show_environment > 0 ? power_heuristic(pdf, f_p) : 1.0f;
            const float Tr = transmittance_raymarch(P, pos, w_i, seed);
            L = L + ((((thr * mis) * f_p) * Tr) * Le) / pdf;
        }
        if (++n_paths >= (uint32_t)u.bounces) { free_path = false; break; }
        const float rr = luma(thr);
        if (rr < 0.1f) {
            const float prob = 1.0f - rr;
            if (rng(seed) < prob) { free_path = false; break; }
            thr = thr / (1.0f - prob);
        }
        const float s0 = rng(seed), s1 = rng(seed);
        const v3 sd = sample_phase_hg(dir, u.vol_phase_g, s0, s1);
        f_p = phase_hg(dot(-dir, sd), u.vol_phase_g);
        dir = sd;
    }
    if (free_path && u.show_environment > 0) {
        const v3 Le = lookup_environment(P, dir);
        float mis = 1.0f;
        if (n_paths > 0u) {
            const float avg_w = env_average_importance(P);
            mis = power_heuristic(f_p, (luma(Le) / avg_w) * kInv4Pi);
        }
        L = L + (thr * mis) * Le;
    }
    out[0] = L.x; out[1] = L.y; out[2] = L.z; out[3] = n_paths > 0u ? 1.0f : 0.0f;
}

// sequential driver (host harness / reference order): one state transition of one lane
template <class K, class Cold>
VR_HD void lane_step(Hot& h, Cold& c, const SceneParams& P, const WorkUnit& wu, uint32_t& next_item, FirstStash& stash) {
    // the events of a `first` path see its stash in ipos / Tr, as the GPU's event batches do after loading the parked path
    if (h.first && (h.state == ST_NEE || h.state == ST_ESCAPE)) { h.ipos = stash.dir; h.Tr = u2f(stash.item); }
    switch (h.state) {
    case ST_NEW:
        if (next_item >= (uint32_t)wu.n_items) { h.state = ST_DONE; break; }
        do_new<K>(h, c, P, wu, next_item++);
        if (h.first) { stash.dir = h.ipos; stash.item = f2u(h.Tr); first_resume(h, P); }      // = HotStore::save_new + load_resume
        break;
    // two DDA steps, as on the device; a path on a clean segment in the forms the device runs for a wavefront of such paths
    case ST_MARCH: { MarchIO io; if (seg_clean(h)) march_prep<K::dense, K::majb, true>(h, P, io); else march_prep<K::dense, K::majb, false>(h, P, io); march_load<K::tf>(P, io); if (seg_clean(h)) march_finish<K::tf, false, true>(h, P, io); else march_finish<K::tf, false, false>(h, P, io); break; }
    case ST_COLLIDE: if (seg_clean(h)) do_collide<K, Cold, true>(h, c, P); else do_collide<K, Cold, false>(h, c, P); break;
    case ST_NEE: do_nee<K>(h, c, c, P); break;
    case ST_POSTNEE: do_postnee<K>(h, c, P, wu); break;
    case ST_ESCAPE: do_escape<Cold, false, false, world_slot<K>()>(h, c, P, wu); break;
    default: break;
    }
}

}  // namespace vr
