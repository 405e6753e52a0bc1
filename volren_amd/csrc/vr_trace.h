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
//   * a lane that finishes a path immediately starts its next sample (fused spp loop, the running mean
//     mix(old, new, 1/s) of pathtracer_brick.glsl:36 is kept in registers), and
//   * rare, expensive events (NEE environment sampling, new-sample setup with the 32-round TEA hash, escape
//     lookups) are batched until enough lanes want them.
// The order in which lanes run their states never changes a result: every lane owns its RNG state.
#pragma once

#include "vr_math.h"
#include "vr_scene.h"

namespace vr {

enum LaneState : int32_t {
    ST_NEW = 0,      // accumulate previous result, start next sample: seed, camera ray
    ST_BEGIN = 1,    // start a segment: clip box, index-space ray, first optical depth
    ST_MARCH = 2,    // one DDA step over the majorant mips
    ST_COLLIDE = 3,  // tentative collision: density lookup, real/null decision
    ST_NEE = 4,      // real scatter: advance, sample the environment, set up the shadow segment
    ST_POSTNEE = 5,  // shadow segment done: add direct light, bounce cap, roulette, phase sample
    ST_ESCAPE = 6,   // path left the volume: environment lookup + MIS, finish the sample
    ST_DONE = 7,
    ST_COUNT = 8
};

struct Lane {
    int32_t px, py;          // pixel (y up, like GL)
    int32_t s, s_end;        // current 1-based sample, last sample to run
    float acc[4];            // running mean (the RGBA32F texel)
    uint32_t seed;
    v3 pos, dir, thr, L;
    uint32_t n_paths;
    float f_p;
    // segment
    v3 ipos, idir, ri;
    float t, far, tau, mip, majorant;
    int32_t shadow;          // 0: sample_volumeDDA segment, 1: transmittanceDDA segment
    // pending next-event estimate
    v3 w_i, sh_a, sh_Le;
    float sh_pdf, Tr;
    int32_t has_nee;
    int32_t state;
    uint32_t steps;          // watchdog
};

// ---------------------------------------------------------------------------------------------------
// RNG  (common.glsl:40-67)
VR_HD uint32_t tea32(uint32_t v0, uint32_t v1) {
    uint32_t s0 = 0u;
#pragma unroll 4
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
VR_HD float brick_value(const GridView& g, int32_t x, int32_t y, int32_t z) {
    if ((x | y | z) < 0) return 0.0f;
    const uint32_t bx = (uint32_t)x >> 3, by = (uint32_t)y >> 3, bz = (uint32_t)z >> 3;
    if (bx >= (uint32_t)g.nb[0] || by >= (uint32_t)g.nb[1] || bz >= (uint32_t)g.nb[2]) return 0.0f;
    const BrickRec rec = g.bricks[(bz * (uint32_t)g.nb[1] + by) * (uint32_t)g.nb[0] + bx];
    const uint32_t b = g.atlas[(size_t)rec.slot * 512u + ((((uint32_t)z & 7u) << 6) | (((uint32_t)y & 7u) << 3) | ((uint32_t)x & 7u))];
    const float unorm = (float)b / 255.0f;
    return rec.rmin + unorm * rec.rdiff;
}
VR_HD float majorant_at(const GridView& g, v3 ipos, int32_t mip) {
    const int32_t x = floor2i(ipos.x), y = floor2i(ipos.y), z = floor2i(ipos.z);
    if ((x | y | z) < 0) return 0.0f;
    const uint32_t sh = 3u + (uint32_t)mip;
    const uint32_t bx = (uint32_t)x >> sh, by = (uint32_t)y >> sh, bz = (uint32_t)z >> sh;
    if (mip > g.n_mips) return 0.0f;
    const uint32_t rnd = (1u << mip) - 1u;
    const uint32_t dx = ((uint32_t)g.nb[0] + rnd) >> mip, dy = ((uint32_t)g.nb[1] + rnd) >> mip, dz = ((uint32_t)g.nb[2] + rnd) >> mip;
    if (bx >= dx || by >= dy || bz >= dz) return 0.0f;
    // lane-varying mip: select the level offset instead of indexing the kernel-argument array
    const int32_t off = mip == 0 ? g.mip_off[0] : (mip == 1 ? g.mip_off[1] : (mip == 2 ? g.mip_off[2] : g.mip_off[3]));
    return g.majorant[(uint32_t)off + (bz * dy + by) * dx + bx];
}
VR_HD int32_t offs_i(int32_t base, int32_t o) { return base == kIntMin ? kIntMin : base + o; }

VR_HD float density_trilinear_raw(const GridView& g, v3 ipos) {
    const float qx = ipos.x - 0.5f, qy = ipos.y - 0.5f, qz = ipos.z - 0.5f;
    const float fx = qx - floor_(qx), fy = qy - floor_(qy), fz = qz - floor_(qz);
    const int32_t ix = floor2i(qx), iy = floor2i(qy), iz = floor2i(qz);
    const int32_t x1 = offs_i(ix, 1), y1 = offs_i(iy, 1), z1 = offs_i(iz, 1);
    const float lx0 = mix_(brick_value(g, ix, iy, iz), brick_value(g, x1, iy, iz), fx);
    const float lx1 = mix_(brick_value(g, ix, y1, iz), brick_value(g, x1, y1, iz), fx);
    const float hx0 = mix_(brick_value(g, ix, iy, z1), brick_value(g, x1, iy, z1), fx);
    const float hx1 = mix_(brick_value(g, ix, y1, z1), brick_value(g, x1, y1, z1), fx);
    return mix_(mix_(lx0, lx1, fy), mix_(hx0, hx1, fy), fz);
}

// stochastic tricubic tap (common.glsl:221-244): 9 draws in the order tap2.xyz, tap3.xyz, tap4.xyz
VR_HD int32_t tricubic_axis_weights(float q, float& w1, float& c2, float& c3, float& c4) {
    // returns floor(q); thresholds c_k = w_k / max(1e-3, w_1 + ... + w_k)
    const float fl = floor_(q);
    const float t = q - fl, t2 = t * t;
    const float k = 1.0f / 6.0f;
    w1 = k * (-t * t2 + 3.0f * t2 - 3.0f * t + 1.0f);
    float sum = w1;
    float w = k * (3.0f * t * t2 - 6.0f * t2 + 4.0f);
    sum = w + sum; c2 = w / max_(1e-3f, sum);
    w = k * (-3.0f * t * t2 + 3.0f * t2 + 3.0f * t + 1.0f);
    sum = w + sum; c3 = w / max_(1e-3f, sum);
    w = k * t * t2;
    sum = w + sum; c4 = w / max_(1e-3f, sum);
    return floor2i(q);
}
VR_HD void tricubic_tap(v3 ipos, uint32_t& seed, int32_t& tx, int32_t& ty, int32_t& tz) {
    float w1, ax2, ax3, ax4, ay2, ay3, ay4, az2, az3, az4;
    const int32_t ix = tricubic_axis_weights(ipos.x - 0.5f, w1, ax2, ax3, ax4);
    const int32_t iy = tricubic_axis_weights(ipos.y - 0.5f, w1, ay2, ay3, ay4);
    const int32_t iz = tricubic_axis_weights(ipos.z - 0.5f, w1, az2, az3, az4);
    int32_t jx = 0, jy = 0, jz = 0;
    float r;
    r = rng(seed); if (r < ax2) jx = 1;
    r = rng(seed); if (r < ay2) jy = 1;
    r = rng(seed); if (r < az2) jz = 1;
    r = rng(seed); if (r < ax3) jx = 2;
    r = rng(seed); if (r < ay3) jy = 2;
    r = rng(seed); if (r < az3) jz = 2;
    r = rng(seed); if (r < ax4) jx = 3;
    r = rng(seed); if (r < ay4) jy = 3;
    r = rng(seed); if (r < az4) jz = 3;
    tx = offs_i(ix, jx - 1); ty = offs_i(iy, jy - 1); tz = offs_i(iz, jz - 1);
}

// transfer function (common.glsl:203-212)
VR_HD void tf_lookup(const SceneParams& P, float d, float rgba[4]) {
    const Uniforms& u = P.u;
    const float tc = clamp_((d - u.tf_window_left) / u.tf_window_width, 0.0f, 1.0f - 1e-6f);
    const float tcs = tc * (float)u.tf_size;
    int32_t idx = floor2i(tcs);
    const float f = tcs - floor_(tcs);
    const int32_t n = (int32_t)u.tf_size;
    if (idx == kIntMin) idx = 0;
    idx = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
    const int32_t idx1 = idx + 1 < n - 1 ? idx + 1 : n - 1;
    const float* a = P.tf_lut + 4 * idx;
    const float* b = P.tf_lut + 4 * idx1;
#pragma unroll
    for (int k = 0; k < 4; ++k) rgba[k] = mix_(a[k], b[k], f);
}

// ---------------------------------------------------------------------------------------------------
// environment (common.glsl:93-152)
VR_HD int32_t wrap_repeat(int32_t i, int32_t n) { const int32_t m = i % n; return m < 0 ? m + n : m; }
VR_HD int32_t clampi(int32_t i, int32_t lo, int32_t hi) { return i < lo ? lo : (i > hi ? hi : i); }

VR_HD v3 env_texture(const SceneParams& P, float u, float v) {
    const int32_t w = P.env_w, h = P.env_h;
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx = x - floor_(x), fy = y - floor_(y);
    int32_t ix = floor2i(x), iy = floor2i(y);
    if (ix == kIntMin || iy == kIntMin || ix > (1 << 28) || ix < -(1 << 28)) { ix = 0; iy = 0; fx = 0.0f; fy = 0.0f; }
    const int32_t x0 = wrap_repeat(ix, w), x1 = wrap_repeat(ix + 1, w);
    const int32_t y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
    const float* t00 = P.envmap + 4 * ((size_t)y0 * w + x0);
    const float* t10 = P.envmap + 4 * ((size_t)y0 * w + x1);
    const float* t01 = P.envmap + 4 * ((size_t)y1 * w + x0);
    const float* t11 = P.envmap + 4 * ((size_t)y1 * w + x1);
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
VR_HD v3 lookup_environment(const SceneParams& P, v3 dir) {
    const v3 idir = mat3_mul(P.u.env_inv_transform, dir);
    const float u = atan2_(idir.z, idir.x) / (2.0f * kPi) + 0.5f;
    const float v = 1.0f - acos_(idir.y) / kPi;
    const v3 c = env_texture(P, u, v);
    return v3{ P.u.env_strength * c.x, P.u.env_strength * c.y, P.u.env_strength * c.z };
}
VR_HD void sample_environment(const SceneParams& P, float r0, float r1, v3& w_i, v3& Le, float& pdf_out) {
    int32_t posx = 0, posy = 0;
    float px = r0, py = r1;
    for (int32_t mip = P.u.env_imp_base_mip - 1; mip >= 0; mip--) {
        posx *= 2; posy *= 2;
        const float w0 = imp_fetch(P, posx, posy, mip), w1 = imp_fetch(P, posx + 1, posy, mip);
        const float w2 = imp_fetch(P, posx, posy + 1, mip), w3 = imp_fetch(P, posx + 1, posy + 1, mip);
        const float q0 = w0 + w2, q1 = w1 + w3;
        const float d = q0 / max_(1e-8f, q0 + q1);
        float e;
        if (px < d) { px = px / d; e = w0 / q0; }
        else { posx += 1; px = (px - d) / (1.0f - d); e = w1 / q1; }
        if (py < e) { py = py / e; }
        else { posy += 1; py = (py - e) / (1.0f - e); }
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
    const float avg_w = imp_fetch(P, 0, 0, P.u.env_imp_base_mip);
    pdf_out = (imp_fetch(P, posx, posy, 0) / avg_w) * kInv4Pi;
}

// ---------------------------------------------------------------------------------------------------
// phase function (common.glsl:172-190), align (:25-33), MIS (:35)
VR_HD float phase_hg(float cos_t, float g) {
    const float denom = 1.0f + sqr(g) + 2.0f * g * cos_t;
    return kInv4Pi * (1.0f - sqr(g)) / (denom * sqrt_(denom));
}
VR_HD v3 align(v3 N, v3 v) {
    v3 T;
    if (abs_(N.x) > abs_(N.y)) T = v3{ -N.z, 0.0f, N.x } / sqrt_(N.x * N.x + N.z * N.z);
    else T = v3{ 0.0f, N.z, -N.y } / sqrt_(N.y * N.y + N.z * N.z);
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

// box clip (common.glsl:157-165)
VR_HD bool intersect_box(v3 pos, v3 dir, const float* bmin, const float* bmax, float& tnear, float& tfar) {
    const v3 inv = v3{ 1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z };
    const v3 lo = (v3{ bmin[0], bmin[1], bmin[2] } - pos) * inv;
    const v3 hi = (v3{ bmax[0], bmax[1], bmax[2] } - pos) * inv;
    const v3 tmin = v3{ min_(lo.x, hi.x), min_(lo.y, hi.y), min_(lo.z, hi.z) };
    const v3 tmax = v3{ max_(lo.x, hi.x), max_(lo.y, hi.y), max_(lo.z, hi.z) };
    tnear = max_(0.0f, max_(tmin.x, max_(tmin.y, tmin.z)));
    tfar = min_(tmax.x, min_(tmax.y, tmax.z));
    return tnear <= tfar;
}

// one DDA step on mip (common.glsl:404-409)
VR_HD float step_dda(v3 p, v3 ri, int32_t mip) {
    const float dim = (float)(8 << mip);
    const float idim = 1.0f / dim;
    const float ox = ri.x >= 0.0f ? dim + 0.5f : -0.5f;
    const float oy = ri.y >= 0.0f ? dim + 0.5f : -0.5f;
    const float oz = ri.z >= 0.0f ? dim + 0.5f : -0.5f;
    const float tx = (floor_(p.x * idim) * dim + ox - p.x) * ri.x;
    const float ty = (floor_(p.y * idim) * dim + oy - p.y) * ri.y;
    const float tz = (floor_(p.z * idim) * dim + oz - p.z) * ri.z;
    return min_(tx, min_(ty, tz));
}

// ---------------------------------------------------------------------------------------------------
// state bodies

VR_HD void lane_init(Lane& l, int32_t px, int32_t py, int32_t first_sample, int32_t n_samples, const float* texel) {
    l.px = px; l.py = py;
    l.s = first_sample - 1; l.s_end = first_sample + n_samples - 1;
    l.acc[0] = texel[0]; l.acc[1] = texel[1]; l.acc[2] = texel[2]; l.acc[3] = texel[3];
    l.state = ST_NEW; l.steps = 0u; l.n_paths = 0u;
    l.L = v3{ 0, 0, 0 }; l.thr = v3{ 1, 1, 1 }; l.f_p = 0.0f; l.seed = 0u;
    l.pos = l.dir = l.ipos = l.idir = l.ri = l.w_i = l.sh_a = l.sh_Le = v3{ 0, 0, 0 };
    l.t = l.far = l.tau = l.mip = l.majorant = l.sh_pdf = l.Tr = 0.0f;
    l.shadow = 0; l.has_nee = 0;
}

// result of trace_path is (L, clamp(n_paths,0,1)); pathtracer_brick.glsl:36 running mean
VR_HD void finish_sample(Lane& l) {
    const float a = 1.0f / (float)l.s;
    l.acc[0] = mix_(l.acc[0], sanitize(l.L.x), a);
    l.acc[1] = mix_(l.acc[1], sanitize(l.L.y), a);
    l.acc[2] = mix_(l.acc[2], sanitize(l.L.z), a);
    l.acc[3] = mix_(l.acc[3], l.n_paths > 0u ? 1.0f : 0.0f, a);
    l.state = ST_NEW;
}

// pathtracer_brick.glsl:27-30 + common.glsl:76-80
VR_HD void do_new(Lane& l, const SceneParams& P) {
    if (l.s >= l.s_end) { l.state = ST_DONE; return; }
    l.s += 1;
    const int32_t W = P.u.resolution[0], H = P.u.resolution[1];
    l.seed = tea32((uint32_t)P.u.seed * (uint32_t)(l.py * W + l.px), (uint32_t)l.s);
    const float jx = rng(l.seed), jy = rng(l.seed);
    const float fx = (((float)l.px + jx) - (float)W * 0.5f) / (float)H;
    const float fy = (((float)l.py + jy) - (float)H * 0.5f) / (float)H;
    l.dir = normalize(mat3_mul(P.u.cam_transform, normalize(v3{ fx, fy, P.cam_z })));
    l.pos = v3{ P.u.cam_pos[0], P.u.cam_pos[1], P.u.cam_pos[2] };
    l.L = v3{ 0, 0, 0 }; l.thr = v3{ 1, 1, 1 };
    l.n_paths = 0u; l.f_p = 0.0f;
    l.shadow = 0;
    l.state = ST_BEGIN;
}

// head of sample_volumeDDA / transmittanceDDA (common.glsl:413-421, 459-468)
VR_HD void do_begin(Lane& l, const SceneParams& P) {
    const v3 d = l.shadow ? l.w_i : l.dir;
    float tnear, tfar;
    if (!intersect_box(l.pos, d, P.u.vol_bb_min, P.u.vol_bb_max, tnear, tfar)) {
        if (l.shadow) { l.Tr = 1.0f; l.state = ST_POSTNEE; }
        else l.state = ST_ESCAPE;
        return;
    }
    l.ipos = mat4_point(P.u.vol_density_inv_transform, l.pos);
    l.idir = mat4_dir(P.u.vol_density_inv_transform, d);
    l.ri = v3{ 1.0f / l.idir.x, 1.0f / l.idir.y, 1.0f / l.idir.z };
    l.t = tnear + 1e-6f;
    l.far = tfar;
    l.Tr = 1.0f;
    l.tau = neg_log_1m(rng(l.seed));
    l.mip = 3.0f;
    l.state = ST_MARCH;
}

// loop body of both DDA trackers up to the collision test (common.glsl:422-435, 469-482)
VR_HD void do_march(Lane& l, const SceneParams& P) {
    if (!(l.t < l.far)) { l.state = l.shadow ? ST_POSTNEE : ST_ESCAPE; return; }
    const v3 curr = axpy(l.ipos, l.t, l.idir);
    const int32_t m = round_half_even(l.mip);
    const float majorant = majorant_at(P.density, curr, m);
    const float dt = step_dda(curr, l.ri, m);
    l.t += dt;
    l.tau -= majorant * dt;
    l.mip = min_(l.mip + 0.25f, 3.0f);
    if (l.tau > 0.0f) return;
    l.t += l.tau / majorant;
    if (l.t >= l.far) { l.state = l.shadow ? ST_POSTNEE : ST_ESCAPE; return; }
    l.majorant = majorant;
    l.state = ST_COLLIDE;
}

// tentative collision (common.glsl:436-452, 483-498)
template <bool USE_TF>
VR_HD void do_collide(Lane& l, const SceneParams& P) {
    const Uniforms& u = P.u;
    const v3 ip = axpy(l.ipos, l.t, l.idir);
    float d;
    float rgba[4] = { 0, 0, 0, 0 };
    if (USE_TF) {
        tf_lookup(P, (u.vol_density_scale * density_trilinear_raw(P.density, ip)) * u.vol_inv_majorant, rgba);
        d = u.vol_majorant * rgba[3];
    } else {
        int32_t tx, ty, tz;
        tricubic_tap(ip, l.seed, tx, ty, tz);
        d = u.vol_density_scale * brick_value(P.density, tx, ty, tz);
    }
    if (!l.shadow) {
        // Le += throughput * (1 - albedo) * lookup_emission(...) * d * vol_inv_majorant  (9 draws, always)
        if (u.has_emission) {
            const v3 ie = mat4_point(P.emission_from_density, ip);
            int32_t ex, ey, ez;
            tricubic_tap(ie, l.seed, ex, ey, ez);
            const float tt = brick_value(P.emission, ex, ey, ez) * u.vol_emission_norm;
            const v3 e3 = v3{ tt, sqr(tt), sqr(sqr(tt)) };
            const v3 em = v3{ u.vol_emission_scale * sqr(e3.x), u.vol_emission_scale * sqr(e3.y), u.vol_emission_scale * sqr(e3.z) };
            const v3 oma = v3{ 1.0f - u.vol_albedo[0], 1.0f - u.vol_albedo[1], 1.0f - u.vol_albedo[2] };
            l.L = l.L + (((l.thr * oma) * em) * d) * u.vol_inv_majorant;
        } else {
            rng_skip9(l.seed);
        }
        if (rng(l.seed) * l.majorant < d) {
            l.thr = l.thr * v3{ u.vol_albedo[0], u.vol_albedo[1], u.vol_albedo[2] };
            if (USE_TF) l.thr = l.thr * v3{ rgba[0], rgba[1], rgba[2] };
            l.state = ST_NEE;
            return;
        }
    } else {
        if (rng(l.seed) * l.majorant < d) {
            l.Tr *= max_(0.0f, 1.0f - u.vol_majorant / l.majorant);
            if (l.Tr < 0.1f) {
                const float prob = 1.0f - l.Tr;
                if (rng(l.seed) < prob) { l.Tr = 0.0f; l.state = ST_POSTNEE; return; }
                l.Tr /= 1.0f - prob;
            }
        }
    }
    l.tau = neg_log_1m(rng(l.seed));
    l.mip = max_(0.0f, l.mip - 2.0f);
    l.state = ST_MARCH;
}

// real collision: common.glsl:611-626 up to the transmittance call
VR_HD void do_nee(Lane& l, const SceneParams& P) {
    l.pos = axpy(l.pos, l.t, l.dir);
    const float r0 = rng(l.seed), r1 = rng(l.seed);
    float pdf;
    sample_environment(P, r0, r1, l.w_i, l.sh_Le, pdf);
    l.sh_pdf = pdf;
    if (pdf > 0.0f) {
        l.f_p = phase_hg(dot(-l.dir, l.w_i), P.u.vol_phase_g);
        const float mis = P.u.show_environment > 0 ? power_heuristic(pdf, l.f_p) : 1.0f;
        l.sh_a = (l.thr * mis) * l.f_p;
        l.has_nee = 1;
        l.shadow = 1;
        l.state = ST_BEGIN;
    } else {
        l.has_nee = 0;
        l.state = ST_POSTNEE;
    }
}

// common.glsl:625-641
VR_HD void do_postnee(Lane& l, const SceneParams& P) {
    if (l.has_nee) {
        const v3 a = ((l.sh_a * l.Tr) * l.sh_Le) / l.sh_pdf;
        l.L = l.L + a;
    }
    l.shadow = 0;
    if (++l.n_paths >= (uint32_t)P.u.bounces) { finish_sample(l); return; }
    const float rr = luma(l.thr);
    if (rr < 0.1f) {
        const float prob = 1.0f - rr;
        if (rng(l.seed) < prob) { finish_sample(l); return; }
        l.thr = l.thr / (1.0f - prob);
    }
    const float s0 = rng(l.seed), s1 = rng(l.seed);
    const v3 sd = sample_phase_hg(l.dir, P.u.vol_phase_g, s0, s1);
    l.f_p = phase_hg(dot(-l.dir, sd), P.u.vol_phase_g);
    l.dir = sd;
    l.state = ST_BEGIN;
}

// common.glsl:644-651
VR_HD void do_escape(Lane& l, const SceneParams& P) {
    if (P.u.show_environment > 0) {
        const v3 Le = lookup_environment(P, l.dir);
        float mis = 1.0f;
        if (l.n_paths > 0u) {
            const float avg_w = imp_fetch(P, 0, 0, P.u.env_imp_base_mip);
            const float pdf_env = (luma(Le) / avg_w) * kInv4Pi;
            mis = power_heuristic(l.f_p, pdf_env);
        }
        l.L = l.L + (l.thr * mis) * Le;
    }
    finish_sample(l);
}

template <bool USE_TF>
VR_HD void lane_step(Lane& l, const SceneParams& P) {
    switch (l.state) {
    case ST_NEW: do_new(l, P); break;
    case ST_BEGIN: do_begin(l, P); break;
    case ST_MARCH: do_march(l, P); break;
    case ST_COLLIDE: do_collide<USE_TF>(l, P); break;
    case ST_NEE: do_nee(l, P); break;
    case ST_POSTNEE: do_postnee(l, P); break;
    case ST_ESCAPE: do_escape(l, P); break;
    default: break;
    }
}

}  // namespace vr
