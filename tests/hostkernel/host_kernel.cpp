// tests/hostkernel/host_kernel.cpp -- TEST HARNESS ONLY, never part of the product.
//
// Compiles the device code of the path tracer (volren_amd/csrc/vr_trace.h, vr_math.h -- the same headers the HIP
// kernel is built from) with the host compiler so that the lane state machine can be checked against the CPU
// oracle, and run under ASan/UBSan, in the GPU-less build container.  It is built and loaded by
// tests/test_host_kernel.py only; the product library (libvolren_amd.so) has no CPU path.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../volren_amd/csrc/vr_trace.h"

using namespace vr;

namespace {
struct HostGrid {
    std::vector<BrickRec> recs;
    std::vector<uint8_t> atlas;
    std::vector<float> majorant, rng, atlas_f32;
    std::vector<uint16_t> majorant16;
    GridView view{};
};

void build_grid(HostGrid& g, const Uniforms& u, const float* lut, const uint32_t nb[3], const uint32_t* indirection, const uint32_t* range,
                const uint32_t ad[3], const uint8_t* atlas, int n_mips, const uint32_t* const* mips, bool density, bool blocked = false) {
    g.view.maj_blocked = blocked ? 1 : 0;
    const size_t n = (size_t)nb[0] * nb[1] * nb[2];
    const uint32_t sx = ad[0] / 8, sy = ad[1] / 8, sz = ad[2] / 8;
    for (int i = 0; i < 3; ++i) { g.view.mshift[i] = ceil_log2(nb[i]) < 3 ? 3 : ceil_log2(nb[i]); g.view.mlim[i] = (float)(8u << g.view.mshift[i]); }
    g.recs.assign(n, BrickRec{ 0u, 0.f, 0.f, 0u });
    g.atlas.assign(g.recs.size() * (size_t)kBrickBlockBytes, 0);            // brick-linear blocks == brick_grid_to_device
    for (size_t i = 0; i < n; ++i) {
        const uint32_t ind = indirection[i], rg = range[i];
        const uint32_t px = ind >> 22, py = (ind >> 12) & 1023u, pz = (ind >> 2) & 1023u;
        const float lo = half2float(rg & 0xFFFFu), hi = half2float(rg >> 16);
        const size_t idx = i;
        BrickRec& r = g.recs[idx];
        r.slot = (uint32_t)idx; r.rmin = lo; r.rdiff = hi - lo; r.range = rg;
        uint8_t* dst = &g.atlas[idx * (size_t)kBrickBlockBytes];
        if (VR_BRICK_HEADERS)
            for (uint32_t l = 0; l < 5; ++l) { memcpy(dst + l * 128u, &r.rmin, 4); memcpy(dst + l * 128u + 4u, &r.rdiff, 4); }
        if (r.rdiff != 0.f && px < sx && py < sy && pz < sz)
            for (uint32_t z = 0; z < 8; ++z) for (uint32_t y = 0; y < 8; ++y) {
                const uint8_t* src = atlas + (((size_t)(pz * 8 + z) * ad[1] + (py * 8 + y)) * ad[0] + px * 8);
                for (uint32_t x = 0; x < 8; ++x) dst[brick_voxel_byte(z * 64 + y * 8 + x)] = src[x];
            }
    }
    std::vector<uint32_t> words(range, range + n);
    uint32_t mip_off[4] = { 0u, 0u, 0u, 0u };
    for (int m = 1; m <= n_mips; ++m) {
        const uint32_t rnd = (1u << m) - 1u;
        const size_t cnt = (size_t)((nb[0] + rnd) >> m) * ((nb[1] + rnd) >> m) * ((nb[2] + rnd) >> m);
        mip_off[m] = (uint32_t)words.size();
        words.insert(words.end(), mips[m - 1], mips[m - 1] + cnt);
    }
    const uint32_t k = (uint32_t)(g.view.mshift[0] + g.view.mshift[1] + g.view.mshift[2]);
    g.majorant.assign(majorant_table_cells(k), 0.0f);
    g.majorant16.assign(majorant_table_cells(k), 0);
    g.view.maj_outside = (int32_t)majorant_padded_cells(k);
    if (density) {
        SceneParams P{}; P.u = u; P.tf_lut = lut;
        {   // == majorant_kernel: every cell without a range word (beyond the real extent, missing level, the "outside" cell) holds density_scale * 0, TF-remapped
            float m0 = u.vol_density_scale * half2float(0u);
            if (u.use_tf) { float rgba[4]; tf_lookup(P, m0 * u.vol_inv_majorant, rgba); m0 = u.vol_majorant * rgba[3]; }
            g.majorant.assign(majorant_table_cells(k), m0);
        }
        for (int mip = 0; mip <= n_mips; ++mip) {            // == majorant_kernel of vr_kernels.hip
            const uint32_t rnd = (1u << mip) - 1u;
            const uint32_t dx = (nb[0] + rnd) >> mip, dy = (nb[1] + rnd) >> mip, dz = (nb[2] + rnd) >> mip;
            const uint32_t sxm = (uint32_t)g.view.mshift[0] - mip, sym = (uint32_t)g.view.mshift[1] - mip;
            for (uint32_t cz = 0; cz < dz; ++cz) for (uint32_t cy = 0; cy < dy; ++cy) for (uint32_t cx = 0; cx < dx; ++cx) {
                const uint32_t hw = words[mip_off[mip] + ((size_t)cz * dy + cy) * dx + cx] >> 16;
                const uint32_t cell = majorant_level_offset(k, mip) + majorant_cell_index(cx, cy, cz, sxm, sym, (uint32_t)mip, blocked);
                g.majorant16[cell] = (uint16_t)hw;
                float m = u.vol_density_scale * half2float(hw);
                if (u.use_tf) { float rgba[4]; tf_lookup(P, m * u.vol_inv_majorant, rgba); m = u.vol_majorant * rgba[3]; }
                g.majorant[cell] = m;
            }
        }
    }
    g.rng.resize(g.recs.size() * 2);
    for (size_t i = 0; i < g.recs.size(); ++i) { g.rng[2 * i] = g.recs[i].rmin; g.rng[2 * i + 1] = g.recs[i].rdiff; }
    g.view.bricks = g.recs.data(); g.view.atlas = g.atlas.data(); g.view.majorant = g.majorant.data();
    g.view.majorant16 = g.majorant16.data(); g.view.rng = g.rng.data();
    g.view.atlas_f32 = nullptr;
    if (density && u.use_tf) {                  // == RendererHIP::launch: decoded float atlas for transfer-function renders
        g.atlas_f32.resize(g.recs.size() * 512);
        for (size_t i = 0; i < g.atlas_f32.size(); ++i)
            g.atlas_f32[i] = g.rng[2 * (i >> 9)] + unorm8(g.atlas[(i >> 9) * (size_t)kBrickBlockBytes + brick_voxel_byte((uint32_t)(i & 511u))]) * g.rng[2 * (i >> 9) + 1];
        g.view.atlas_f32 = g.atlas_f32.data();
    }
    for (int i = 0; i < 3; ++i) g.view.nb[i] = (int32_t)nb[i];
    g.view.n_mips = n_mips;
}
}  // namespace

// ---- access trace (tests/tools_coherence_by_scope.py): which 128-byte line of the majorant table a marching path's next DDA step reads, which
// 4x4x4-voxel block (one line of the dense grid) / 8^3 brick a path at a tentative collision stands in.  One 32-bit word per lane-step:
// bits 31..30 = 0 march / 1 collide, bits 29..0 = line id.  Test infrastructure only.
static uint32_t* g_trace = nullptr;
static size_t g_trace_cap = 0, g_trace_n = 0;
static void trace_access(const Hot& h, const SceneParams& P) {
    if (g_trace_n >= g_trace_cap) return;
    const v3 c = axpy(h.ipos, h.t, h.idir);
    if (h.state == ST_MARCH) {
        if (!(h.t < seg_far(h))) return;                                               // (the sign of `far` carries the clean flag: vr_trace.h seg_clean)
        const int32_t idx = majorant_index<2>(P.density, c, round_mip_q(h.mipq));
        if (idx >= 0 && idx != P.density.maj_outside) g_trace[g_trace_n++] = (uint32_t)idx >> 6;      // 64 fp16 cells per 128-byte line; a step outside the table reads its one shared "outside" cell: not a gather
    } else {
        const int32_t x = (int32_t)floor_(c.x), y = (int32_t)floor_(c.y), z = (int32_t)floor_(c.z);
        if (x < 0 || y < 0 || z < 0) return;
        const uint32_t sh = P.density.dense ? 2u : 3u;                                  // dense: 4x4x4 block = one line; bricks: 8^3 block = 4 lines
        const uint32_t nx = P.density.dense ? (uint32_t)P.density.dblk[0] : (uint32_t)P.density.nb[0], ny = P.density.dense ? (uint32_t)P.density.dblk[1] : (uint32_t)P.density.nb[1];
        const uint32_t id = (((uint32_t)z >> sh) * ny + ((uint32_t)y >> sh)) * nx + ((uint32_t)x >> sh);
        g_trace[g_trace_n++] = (1u << 30) | (id & 0x3FFFFFFFu);
    }
}

extern "C" {

struct hk_grid_desc {
    uint32_t nb[3]; uint32_t atlas_dim[3]; int32_t n_mips;
    const uint32_t* indirection; const uint32_t* range; const uint8_t* atlas; const uint32_t* mips[3];
    const uint16_t* dense; uint32_t dim[3];
};

int hk_uniforms_size() { return (int)sizeof(Uniforms); }

// the scene as the lane code sees it (SceneParams + the arrays its views point into), built from the oracle's arrays exactly as the product builds its device copies
struct HostScene {
    SceneParams P{};
    HostGrid dg, eg;
    std::vector<uint16_t> blocked;
    std::vector<float> env, cdf;
};
static void build_scene(HostScene& S, const Uniforms* up, const hk_grid_desc* density, const hk_grid_desc* emission, const float* lut,
                        const float* env_rgb, int env_w, int env_h, const float* impmap, int imp_dim) {
    SceneParams& P = S.P;
    HostGrid& dg = S.dg; HostGrid& eg = S.eg;
    std::vector<uint16_t>& blocked = S.blocked;
    std::vector<float>& env = S.env; std::vector<float>& cdf = S.cdf;
    const Uniforms& u = *up;
    P.u = u;
    build_grid(dg, u, lut, density->nb, density->indirection, density->range, density->atlas_dim, density->atlas, density->n_mips, density->mips, true,
               // the majorant table's levels 0-1 in 4x4x4-cell blocks (a per-grid choice of the product since round 5; the lane code reads the view's flag at run time here)
               std::getenv("VR_HOST_MAJ_BLOCKED") != nullptr && std::getenv("VR_HOST_MAJ_BLOCKED")[0] == '1');
    P.density = dg.view;
    // (blocked: == dense_grid_to_device: 4x4x4 blocks)
    if (density->dense) {
        const uint32_t dx = density->dim[0], dy = density->dim[1], dz = density->dim[2];
        const uint32_t bx = (dx + 3u) / 4u, by = (dy + 3u) / 4u, bz = (dz + 3u) / 4u;
        blocked.assign((size_t)bx * by * bz * 64u, 0);
        for (uint32_t z = 0; z < dz; ++z) for (uint32_t y = 0; y < dy; ++y) for (uint32_t x = 0; x < dx; ++x)
            blocked[dense_blocked_index(x, y, z, bx, by)] = density->dense[((size_t)z * dy + y) * dx + x];
        P.density.dense = blocked.data();
        P.density.dblk[0] = (int32_t)bx; P.density.dblk[1] = (int32_t)by;
    }
    for (int i = 0; i < 3; ++i) P.density.dim[i] = (int32_t)density->dim[i];
    if (emission && u.has_emission) {
        build_grid(eg, u, lut, emission->nb, emission->indirection, emission->range, emission->atlas_dim, emission->atlas, emission->n_mips, emission->mips, false);
        P.emission = eg.view;
        // emission_from_density = vol_emission_inv_transform * vol_density_transform (same product as hostmath.h)
        const float* a = u.vol_emission_inv_transform; const float* b = u.vol_density_transform;
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r)
            P.emission_from_density[4 * c + r] = a[r] * b[4 * c] + a[4 + r] * b[4 * c + 1] + a[8 + r] * b[4 * c + 2] + a[12 + r] * b[4 * c + 3];
    }
    P.tf_lut = lut;
    env.assign((size_t)env_w * env_h * kEnvTexelFloats, 0.0f);
    for (size_t i = 0; i < (size_t)env_w * env_h; ++i) for (int k = 0; k < 3; ++k) env[kEnvTexelFloats * i + k] = env_rgb[3 * i + k];
    P.envmap = env.data(); P.env_w = env_w; P.env_h = env_h;
    P.impmap = impmap; P.imp_dim = imp_dim;
    int base = 0; while ((1 << base) < imp_dim) ++base;
    cdf.assign(env_cdf_table_floats(base - 1), 0.0f);
    {   // == env_cdf_kernel of vr_kernels.hip
        for (int mip = base - 1; mip >= 0; --mip) {
            const int d = imp_dim >> mip, hd = d >> 1;
            const float* level = impmap + imp_level_offset(imp_dim, mip);
            for (int y = 0; y < hd; ++y) for (int x = 0; x < hd; ++x) {
                const float w0 = level[(size_t)(2 * y) * d + 2 * x], w1 = level[(size_t)(2 * y) * d + 2 * x + 1];
                const float w2 = level[(size_t)(2 * y + 1) * d + 2 * x], w3 = level[(size_t)(2 * y + 1) * d + 2 * x + 1];
                const float q0 = w0 + w2, q1 = w1 + w3;
                float* o = cdf.data() + env_cdf_index(base - 1, base - 1 - mip, (uint32_t)x, (uint32_t)y);
                o[0] = q0 / max_(1e-8f, q0 + q1); o[1] = w0 / q0; o[2] = w1 / q1;
                if (mip == 0) { o[3] = w0; o[4] = w1; o[5] = w2; o[6] = w3; }
            }
        }
    }
    P.env_cdf = cdf.data();
    P.cam_z = -0.5f / tan_(0.5f * kPi * u.cam_fov / 180.f);
}

// env_rgb: texture order (row 0 bottom), 3 floats per texel.  Returns the number of lane steps executed.
long long hk_render(const Uniforms* up, const hk_grid_desc* density, const hk_grid_desc* emission, const float* lut,
                    const float* env_rgb, int env_w, int env_h, const float* impmap, int imp_dim,
                    float* fb, int x0, int y0, int x1, int y1, int first_sample, int n_samples) {
    HostScene S;
    build_scene(S, up, density, emission, lut, env_rgb, env_w, env_h, impmap, imp_dim);
    const SceneParams& P = S.P;
    const Uniforms& u = P.u;
    long long steps = 0;
    const int W = u.resolution[0], H = u.resolution[1];
    if (u.integrator == 2 && u.use_tf) {       // direct volume rendering: one call per (pixel, sample)
        for (int y = y0; y < y1; ++y) for (int x = x0; x < x1; ++x) {
            float* px = fb + 4 * ((size_t)y * W + x);
            for (int k = 0; k < n_samples; ++k) { float L[4]; dvr_sample(P, x, y, first_sample + k, L); accumulate_sample(px, L, first_sample + k); ++steps; }
        }
        return steps;
    }
    if (u.integrator == 3) {                   // trace_path with the ray-marching trackers: one call per (pixel, sample)
        for (int y = y0; y < y1; ++y) for (int x = x0; x < x1; ++x) {
            float* px = fb + 4 * ((size_t)y * W + x);
            for (int k = 0; k < n_samples; ++k) { float L[4]; raymarch_path_sample(P, x, y, first_sample + k, L); accumulate_sample(px, L, first_sample + k); ++steps; }
        }
        return steps;
    }
    // wave-sized work units exactly like the HIP kernel: 8x8 tile x chunk of samples -> sample buffer -> running mean
    const int spu = n_samples < 32 ? n_samples : 32;
    std::vector<float> sbuf((size_t)spu * 64 * 4);
    for (int ty = y0 & ~7; ty < y1; ty += 8)
        for (int tx = x0 & ~7; tx < x1; tx += 8)
            for (int c0 = 0; c0 < n_samples; c0 += spu) {
                WorkUnit wu;
                wu.px0 = tx; wu.py0 = ty; wu.first_sample = first_sample + c0;
                const int sc = (n_samples - c0) < spu ? (n_samples - c0) : spu;
                wu.n_items = sc * 64; wu.base = 0u; wu.out = sbuf.data();
                // 64 lanes advanced round-robin: exercises the item hand-out in a different order than the GPU does
                struct ColdHost {
                    float v[C_COUNT];
                    float ld(int32_t f) const { return v[f]; }
                    void st(int32_t f, float x) { v[f] = x; }
                };
                Hot lanes[64];
                ColdHost cold[64] = {};
                FirstStash stash[64] = {};
                for (auto& l : lanes) hot_init(l);
                uint32_t next_item = 0;
                bool live = true;
                while (live) {
                    live = false;
                    for (int i = 0; i < 64; ++i) {
                        Hot& l = lanes[i];
                        if (l.state == ST_DONE) continue;
                        live = true;
                        if (l.state == ST_NEW) for (float& v : cold[i].v) v = nan_();         // a new path must not depend on what its cold line held
                        if (g_trace && (l.state == ST_MARCH || l.state == ST_COLLIDE)) trace_access(l, P);
#if VR_WORLD_SLOT
                        // a build with -DVR_WORLD_SLOT=1 (tests/test_host_kernel.py::test_world_slot_*): the kernels of one scene kind that carry the experiment -- DDA trackers,
                        // no transfer function, no emission grid (the paired atlas of the emission kernels is not built here)
                        if (!u.use_tf && u.integrator == 0 && !u.has_emission) {
                            if (P.density.dense) lane_step<TraceCfg<false, 0, 0, 1, 2>>(l, cold[i], P, wu, next_item, stash[i]);
                            else lane_step<TraceCfg<false, 0, 0, 0, 2>>(l, cold[i], P, wu, next_item, stash[i]);
                        } else
#endif
                        if (u.use_tf) lane_step<TraceCfg<true, 2, 2, 2, 2>>(l, cold[i], P, wu, next_item, stash[i]); else lane_step<TraceCfg<false, 2, 2, 2, 2>>(l, cold[i], P, wu, next_item, stash[i]);
                        if (++steps > (1ll << 40)) return -1;
                    }
                }
                for (int p = 0; p < 64; ++p) {
                    const int x = tx + (p & 7), y = ty + (p >> 3);
                    if (x < x0 || x >= x1 || y < y0 || y >= y1 || x >= W || y >= H) continue;
                    float* px = fb + 4 * ((size_t)y * W + x);
                    for (int k = 0; k < sc; ++k) accumulate_sample(px, &sbuf[4 * ((size_t)k * 64 + p)], first_sample + c0 + k);
                }
            }
    return steps;
}

#if defined(VR_HOST_TRACE)
// ---- L2 model (tests/tools_l2_breakdown.py, round 6): which class of access leaves one XCD's L2? ------------------------------------------------------------------
// A population of `population` paths -- what one XCD keeps in flight: 1024 wavefronts x 188 pool slots -- works through the pixel-samples of a band of the frame (an
// XCD's segment of the work queue is a contiguous band of tiles), every live path advancing by one state transition per round (a march pass of two DDA steps, a
// tentative collision, an event), so that the paths' accesses interleave as a population's do.  Every access the DEVICE makes in that transition is presented, with
// its device-layout address, to a model of the XCD's L2: `l2_bytes` in 128-byte lines, `ways`-way set associative, LRU, write-allocate without fetch.  The read-only
// tables report through the hooks in vr_trace.h (VR_TRACE); the cold path state (64-byte slots, the accesses the device scheduler makes: vr_pathtrace.h) and the
// sample pool are generated here.  Per class: accesses, accesses to a line other than the path's previous one of that class (what an L1 cannot merge), L2 misses.
// Not modelled: the L1s, the Infinity Cache, the scheduler's batching (a path waits for its event batch on the device).
}  // extern "C" (reopened below)
namespace {
enum SimClass { SC_MAJ_FINE = 0, SC_MAJ_COARSE, SC_TAP_DENSITY, SC_TAP_EMISSION, SC_ENV_WARP, SC_ENV_TEXEL, SC_COLD_READ, SC_COLD_WRITE, SC_SAMPLE_WRITE, SC_COUNT };
struct L2Model {
    uint32_t sets = 0, ways = 0;
    std::vector<uint64_t> tag;      // [set][way], 0 = empty; line address + 1
    std::vector<uint32_t> age;
    std::vector<uint8_t> dirty;
    uint32_t clock = 0;
    unsigned long long writebacks = 0;
    void init(size_t bytes, uint32_t w) { ways = w; sets = (uint32_t)(bytes / 128u / w); tag.assign((size_t)sets * w, 0); age.assign((size_t)sets * w, 0); dirty.assign((size_t)sets * w, 0); }
    // returns true on a hit
    bool access(uint64_t line, bool write) {
        const uint64_t h = line ^ (line >> 11) ^ (line >> 23);
        const size_t base = (size_t)(h % sets) * ways;
        ++clock;
        size_t victim = base; uint32_t oldest = 0xFFFFFFFFu;
        for (size_t i = base; i < base + ways; ++i) {
            if (tag[i] == line + 1) { age[i] = clock; if (write) dirty[i] = 1; return true; }
            if (tag[i] == 0) { victim = i; oldest = 0; }
            else if (oldest != 0 && age[i] < oldest) { victim = i; oldest = age[i]; }
        }
        if (tag[victim] != 0 && dirty[victim]) ++writebacks;
        tag[victim] = line + 1; age[victim] = clock; dirty[victim] = write ? 1 : 0;
        return false;
    }
};
struct SimSink {
    L2Model l2;
    unsigned long long acc[SC_COUNT] = {}, newline[SC_COUNT] = {}, miss[SC_COUNT] = {};
    uint64_t* last = nullptr;        // the current path's last line per class
    const void* density_table = nullptr; const void* emission_table = nullptr; const void* maj_table = nullptr;
    uint32_t maj_coarse_from = 0;    // first cell of majorant level 2
    bool paired = false;             // density and emission taps read ONE paired atlas (vr_scene.h pair_voxel_line)
    void touch(int cls, uint64_t space, uint64_t byte, uint32_t bytes, bool write) {
        const uint64_t l0 = byte >> 7, l1 = (byte + bytes - 1) >> 7;
        for (uint64_t l = l0; l <= l1; ++l) {
            const uint64_t line = (space << 40) | l;
            ++acc[cls];
            if (last && last[cls] == line + 1) continue;          // the line this path touched last in this class, in this transition: one instruction (or the L1) serves it
            if (last) last[cls] = line + 1;
            ++newline[cls];
            if (!l2.access(line, write)) ++miss[cls];
        }
    }
};
SimSink* g_sink = nullptr;
}  // namespace
namespace vr {
void host_trace(int32_t cls, const void* table, size_t a, size_t b) {
    SimSink* S = g_sink;
    if (!S) return;
    if (cls == TR_MAJORANT) {
        if (table != S->maj_table) return;                          // (the emission grid's majorant table is never read)
        S->touch(a >= S->maj_coarse_from ? SC_MAJ_COARSE : SC_MAJ_FINE, 1, (uint64_t)a * b, (uint32_t)b, false);
    } else if (cls == TR_TAP) {
        const bool em = table == S->emission_table;
        if (!em && table != S->density_table) return;
        // device layout: the paired atlas (ten 128-byte lines per brick: header + 56 (density, emission) voxel pairs each) or the grid's own brick-linear atlas
        // (five lines per brick: header + 120 voxels each); header and voxel come from ONE line
        const uint64_t line = S->paired ? (uint64_t)a * (kPairBlockBytes / 128u) + pair_voxel_line((uint32_t)b) : (uint64_t)a * (kBrickBlockBytes / 128u) + brick_voxel_line((uint32_t)b);
        S->touch(em ? SC_TAP_EMISSION : SC_TAP_DENSITY, (S->paired || !em) ? 2 : 3, line << 7, 4, false);
    } else if (cls == TR_ENV_WARP) {
        S->touch(SC_ENV_WARP, 4, a, (uint32_t)b, false);
    } else if (cls == TR_ENV_TEXEL) {
        S->touch(SC_ENV_TEXEL, 5, a, (uint32_t)b, false);
    }
}
}  // namespace vr
extern "C" {
// out: SC_COUNT x 3 counters (accesses, new-line accesses, L2 misses), then [27] dirty lines written back, [28] samples finished, [29] lane steps
long long hk_l2_breakdown(const Uniforms* up, const hk_grid_desc* density, const hk_grid_desc* emission, const float* lut,
                          const float* env_rgb, int env_w, int env_h, const float* impmap, int imp_dim,
                          int x0, int y0, int x1, int y1, int spp, int population, long long l2_bytes, int ways, int paired, int lazy_emission,
                          unsigned long long* out, int n_bands) {
    // n_bands > 1: that many bands of y1 - y0 rows one after the other from y0 on (the frame's XCD segments), each with an L2 and a population of its own;
    // out then holds the bands' counters one after the other (32 words each)
    HostScene S;
    build_scene(S, up, density, emission, lut, env_rgb, env_w, env_h, impmap, imp_dim);
    const SceneParams& P = S.P;
    const Uniforms& u = P.u;
    const int band_rows = y1 - y0;
    long long total = 0;
    for (int band = 0; band < (n_bands > 0 ? n_bands : 1); ++band, y0 += band_rows, y1 += band_rows, out += 32) {
    SimSink sink;
    sink.l2.init((size_t)l2_bytes, (uint32_t)ways);
    sink.density_table = P.density.atlas; sink.emission_table = u.has_emission ? P.emission.atlas : nullptr; sink.maj_table = P.density.majorant16;
    sink.maj_coarse_from = majorant_level_offset((uint32_t)(P.density.mshift[0] + P.density.mshift[1] + P.density.mshift[2]), 2u);
    sink.paired = paired != 0;
    struct ColdHost {
        float v[C_COUNT];
        float ld(int32_t f) const { return v[f]; }
        void st(int32_t f, float x) { v[f] = x; }
    };
    struct Path { Hot h; ColdHost c; FirstStash stash; uint64_t last[SC_COUNT]; bool scattered; };
    std::vector<Path> paths((size_t)population);
    for (Path& p : paths) { hot_init(p.h); p.h.state = ST_NEW; memset(p.last, 0, sizeof p.last); p.scattered = false; memset(&p.c, 0, sizeof p.c); p.stash = FirstStash{}; }
    // work units: the 8x8 tiles of the band in raster order, all `spp` samples of a tile in one unit (the device's queue keeps a sub-tile's sample chunks adjacent)
    const int tx0 = x0 & ~7, ty0 = y0 & ~7;
    const int ntx = (x1 - tx0 + 7) / 8, nty = (y1 - ty0 + 7) / 8;
    const size_t n_units = (size_t)ntx * nty;
    std::vector<float> sbuf((size_t)64 * spp * 4, 0.0f);                       // (one unit's worth: the radiances themselves are not needed)
    size_t unit = 0;
    WorkUnit wu; uint32_t next_item = 0;
    auto open_unit = [&](size_t k) { wu.px0 = tx0 + 8 * (int)(k % ntx); wu.py0 = ty0 + 8 * (int)(k / ntx); wu.first_sample = 1; wu.n_items = 64 * spp; wu.base = 0u; wu.out = sbuf.data(); next_item = 0; };
    open_unit(0);
    const bool emission_on = u.has_emission != 0;
    unsigned long long finished = 0, steps = 0;
    g_sink = &sink;
    bool any = true;
    while (any) {
        any = false;
        for (size_t i = 0; i < paths.size(); ++i) {
            Path& p = paths[i];
            Hot& l = p.h;
            if (l.state == ST_DONE) continue;
            if (l.state == ST_NEW) {
                while (next_item >= (uint32_t)wu.n_items && unit + 1 < n_units) open_unit(++unit);
                if (next_item >= (uint32_t)wu.n_items) { l.state = ST_DONE; continue; }
                p.scattered = false;
                memset(p.last, 0, sizeof p.last);
            }
            any = true;
            memset(p.last, 0, sizeof p.last);                                  // merging only inside ONE transition (the dwordx4 loads of one block, a tap's header and voxel)
            sink.last = p.last;
            const uint64_t cold_byte = (uint64_t)i * 64u;                    // this slot's 64-byte cold slot in the XCD's share of the workspace
            const int32_t st = l.state;
            // the cold state as the DEVICE scheduler touches it (vr_pathtrace.h): a path has no cold line before its first scatter event (FirstStash; with an emission
            // grid: lazy_emission kernels); the collision event reads the line (not on a path's first) and writes sector 0, the scatter event reads it and writes
            // both sectors, the escape of a path that scattered reads it; emission kernels also read throughput and radiance when such a path is resumed for a
            // camera / scatter segment and write the radiance back when it is parked
            if (st == ST_NEE) { if (p.scattered) sink.touch(SC_COLD_READ, 6, cold_byte, 64, false); sink.touch(SC_COLD_WRITE, 6, cold_byte, p.scattered ? 32 : 64, true); }
            else if (st == ST_POSTNEE) { sink.touch(SC_COLD_READ, 6, cold_byte, 64, false); sink.touch(SC_COLD_WRITE, 6, cold_byte, 64, true); }
            else if (st == ST_ESCAPE) { if (p.scattered) sink.touch(SC_COLD_READ, 6, cold_byte, 64, false); sink.touch(SC_SAMPLE_WRITE, 7, finished * 16u, 16, true); }
            const int32_t shadow_before = l.shadow;
            if (u.use_tf) lane_step<TraceCfg<true, 2, 2, 2, 2>>(l, p.c, P, wu, next_item, p.stash); else lane_step<TraceCfg<false, 2, 2, 2, 2>>(l, p.c, P, wu, next_item, p.stash);
            ++steps;
            if (st == ST_NEE) p.scattered = true;
            if (emission_on && (lazy_emission ? p.scattered : true)) {
                // a camera / scatter segment begins (the path will be resumed: read thr + L) or ends (parked: write L) -- once per segment each
                if ((st == ST_POSTNEE || (st == ST_NEW && !lazy_emission)) && (l.state == ST_MARCH || l.state == ST_COLLIDE) && !l.shadow) sink.touch(SC_COLD_READ, 6, cold_byte, 32, false);
                if ((st == ST_MARCH || st == ST_COLLIDE) && !shadow_before && l.state != ST_MARCH && l.state != ST_COLLIDE) sink.touch(SC_COLD_WRITE, 6, cold_byte + 32u, 16, true);
            }
            if (st == ST_POSTNEE && l.state == ST_NEW) sink.touch(SC_SAMPLE_WRITE, 7, finished * 16u, 16, true);      // bounce cap / roulette: the scatter event writes the sample
            if (st == ST_ESCAPE || (st == ST_POSTNEE && l.state == ST_NEW)) { ++finished; }
            if (l.state == ST_NEW && st != ST_NEW) { /* the slot is free: it takes the next item in its next round */ }
        }
    }
    g_sink = nullptr;
    for (int c = 0; c < SC_COUNT; ++c) { out[3 * c] = sink.acc[c]; out[3 * c + 1] = sink.newline[c]; out[3 * c + 2] = sink.miss[c]; }
    out[27] = sink.l2.writebacks; out[28] = finished; out[29] = steps;
    total += (long long)finished;
    }
    return total;
}
#endif

void hk_set_trace(uint32_t* buf, unsigned long long cap) { g_trace = buf; g_trace_cap = (size_t)cap; g_trace_n = 0; }
unsigned long long hk_trace_count() { return (unsigned long long)g_trace_n; }

float hk_math(int fn, float x, float y) {
    switch (fn) {
    case 0: return log_(x); case 1: return sin_(x); case 2: return cos_(x); case 3: return tan_(x);
    case 4: return acos_(x); case 5: return atan2_(x, y); case 6: return exp_(x); case 7: return pow_(x, y);
    case 8: return asin_(x);
    case 13: { float s, c; sincos_(x, s, c); return s * y + c; }
    default: return nan_();
    }
}
}
