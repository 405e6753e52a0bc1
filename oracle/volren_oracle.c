/*
 * oracle/volren_oracle.c -- TEST INFRASTRUCTURE ONLY (see volren_oracle.h).
 *
 * Scalar CPU restatement of the reference's GLSL path tracer.  "ref:" comments
 * give the file:line in /root/reference that each function follows.  All
 * arithmetic is IEEE binary32; build with -O2 -ffp-contract=off -mfma.
 * Parity is pinned against outputs of the reference's own GLSL kernels run on Mesa llvmpipe
 * (tests/test_glsl_pin.py; see the header).
 *
 * Expression conventions (shared with the product's own statement of them):
 *   dot3(a,b)      = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   M*v (mat3)     = per row: fma(m2,v.z, fma(m1,v.y, m0*v.x))   (columns m0,m1,m2)
 *   M*(v,1) (mat4) = per row: fma(m2,v.z, fma(m1,v.y, fma(m0,v.x, m3)))
 *   M*(v,0) (mat4) = as mat3 on the upper-left 3x3
 *   a + t*b        = fma(t, b, a)
 *   normalize(v)   = v * (1 / sqrt(dot3(v,v)))
 *   everything else: one rounding per GLSL operator, evaluated left to right.
 */
#include "volren_oracle.h"
#include "oracle_math.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v3divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 v3neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 v3axpy(v3 a, float t, v3 b) { return V3(fmaf(t, b.x, a.x), fmaf(t, b.y, a.y), fmaf(t, b.z, a.z)); }
static inline v3 cross3(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 normalize3(v3 v) { float inv = 1.0f / sqrtf(dot3(v, v)); return v3scale(v, inv); }
static inline v3 mat3mul(const float* m, v3 v) {
    return V3(fmaf(m[6], v.z, fmaf(m[3], v.y, m[0] * v.x)),
              fmaf(m[7], v.z, fmaf(m[4], v.y, m[1] * v.x)),
              fmaf(m[8], v.z, fmaf(m[5], v.y, m[2] * v.x)));
}
static inline v3 mat4point(const float* m, v3 v) {
    return V3(fmaf(m[8],  v.z, fmaf(m[4], v.y, fmaf(m[0], v.x, m[12]))),
              fmaf(m[9],  v.z, fmaf(m[5], v.y, fmaf(m[1], v.x, m[13]))),
              fmaf(m[10], v.z, fmaf(m[6], v.y, fmaf(m[2], v.x, m[14]))));
}
static inline v3 mat4dir(const float* m, v3 v) {
    return V3(fmaf(m[8],  v.z, fmaf(m[4], v.y, m[0] * v.x)),
              fmaf(m[9],  v.z, fmaf(m[5], v.y, m[1] * v.x)),
              fmaf(m[10], v.z, fmaf(m[6], v.y, m[2] * v.x)));
}

/* ref: common.glsl:10-23 */
static inline float sqr(float x) { return x * x; }
static inline float luma(v3 c) { return dot3(c, V3(0.212671f, 0.715160f, 0.072169f)); }
static inline float saturate(float x) { return om_clamp(x, 0.0f, 1.0f); }
static inline float sanitize(float x) { return (x != x || fabsf(x) == INFINITY) ? 0.0f : x; }
/* ref: common.glsl:35 */
static inline float power_heuristic(float a, float b) { return sqr(a) / (sqr(a) + sqr(b)); }

#define INV_4PI (1.0f / (4.0f * OM_PI))

/* ------------------------------------------------------------------ */
/* RNG  ref: common.glsl:40-67 */

uint32_t orc_tea(uint32_t val0, uint32_t val1, uint32_t N) {
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (uint32_t n = 0; n < N; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xA341316Cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xC8013EA4u);
        v1 += ((v0 << 4) + 0xAD90777Du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7E95761Eu);
    }
    return v0;
}
float orc_rng(uint32_t* previous) {
    *previous = *previous * 1664525u + 1013904223u;
    return (float)(*previous & 0x00FFFFFFu) / (float)0x01000000u;
}
#define rng(seedp) orc_rng(seedp)

/* ------------------------------------------------------------------ */
/* render context */

typedef struct {
    const orc_params* p;
    const orc_scene* s;
    orc_counters c;
} ctx_t;

/* ------------------------------------------------------------------ */
/* camera  ref: common.glsl:76-80 */
static v3 view_dir(const orc_params* p, int x, int y, int w, int h, float jx, float jy) {
    const float px = (((float)x + jx) - (float)w * 0.5f) / (float)h;
    const float py = (((float)y + jy) - (float)h * 0.5f) / (float)h;
    const float z = -0.5f / om_tan(0.5f * OM_PI * p->cam_fov / 180.0f);
    return normalize3(mat3mul(p->cam_transform, normalize3(V3(px, py, z))));
}

/* ------------------------------------------------------------------ */
/* environment  ref: common.glsl:93-152 */

static inline int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
static inline int clamp_i(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

/* GL_LINEAR, LOD 0, texel centres at (i+.5)/N, GL_REPEAT in u, GL_CLAMP_TO_EDGE in v.
 * NaN/inf coordinates (GL: undefined) read texel (0,0) with weight 0 offsets. */
void orc_env_texture(const float* tex, int32_t w, int32_t h, float u, float v, float rgb[3]) {
    float x = u * (float)w - 0.5f;
    float y = v * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int32_t ix = om_floor2i(x), iy = om_floor2i(y);
    if (ix == INT32_MIN || iy == INT32_MIN || ix > (1 << 28) || ix < -(1 << 28)) { ix = 0; iy = 0; fx = 0.0f; fy = 0.0f; }
    int x0 = wrap_repeat(ix, w), x1 = wrap_repeat(ix + 1, w);
    int y0 = clamp_i(iy, 0, h - 1), y1 = clamp_i(iy + 1, 0, h - 1);
    const float* t00 = tex + 3 * ((size_t)y0 * w + x0);
    const float* t10 = tex + 3 * ((size_t)y0 * w + x1);
    const float* t01 = tex + 3 * ((size_t)y1 * w + x0);
    const float* t11 = tex + 3 * ((size_t)y1 * w + x1);
    for (int c = 0; c < 3; ++c)
        rgb[c] = om_mix(om_mix(t00[c], t10[c], fx), om_mix(t01[c], t11[c], fx), fy);
}

static inline size_t imp_level_offset(int dim, int mip) {
    size_t off = 0; int d = dim;
    for (int i = 0; i < mip; ++i) { off += (size_t)d * d; d >>= 1; }
    return off;
}
static inline float imp_fetch(const orc_scene* s, int x, int y, int mip) {
    int d = s->imp_dim >> mip;
    if (x < 0 || y < 0 || x >= d || y >= d) return 0.0f;   /* GL: undefined */
    return s->impmap[imp_level_offset(s->imp_dim, mip) + (size_t)y * d + x];
}

/* ref: common.glsl:93-98 */
static v3 lookup_environment(const ctx_t* c, v3 dir) {
    const orc_params* p = c->p;
    v3 idir = mat3mul(p->env_inv_transform, dir);
    float u = om_atan2(idir.z, idir.x) / (2.0f * OM_PI) + 0.5f;
    float v = 1.0f - om_acos(idir.y) / OM_PI;
    float rgb[3];
    orc_env_texture(c->s->envmap, c->s->env_w, c->s->env_h, u, v, rgb);
    return V3(p->env_strength * rgb[0], p->env_strength * rgb[1], p->env_strength * rgb[2]);
}

/* ref: common.glsl:100-146; returns (Le.rgb, pdf) */
static void sample_environment(const ctx_t* c, float r0, float r1, v3* w_i, float le_pdf[4]) {
    const orc_params* p = c->p;
    int posx = 0, posy = 0;
    float px = r0, py = r1;
    for (int mip = p->env_imp_base_mip - 1; mip >= 0; mip--) {
        posx *= 2; posy *= 2;
        float w[4];
        w[0] = imp_fetch(c->s, posx + 0, posy + 0, mip);
        w[1] = imp_fetch(c->s, posx + 1, posy + 0, mip);
        w[2] = imp_fetch(c->s, posx + 0, posy + 1, mip);
        w[3] = imp_fetch(c->s, posx + 1, posy + 1, mip);
        float q[2];
        q[0] = w[0] + w[2];
        q[1] = w[1] + w[3];
        int off_x;
        const float d = q[0] / om_max(1e-8f, q[0] + q[1]);
        if (px < d) { off_x = 0; px = px / d; }
        else        { off_x = 1; px = (px - d) / (1.0f - d); }
        posx += off_x;
        float e = w[off_x] / q[off_x];
        if (py < e) { py = py / e; }
        else        { posy += 1; py = (py - e) / (1.0f - e); }
    }
    const float u = ((float)posx + px) * p->env_imp_inv_dim[0];
    const float v = ((float)posy + py) * p->env_imp_inv_dim[1];
    const float theta = saturate(1.0f - v) * OM_PI;
    const float phi = (saturate(u) * 2.0f - 1.0f) * OM_PI;
    const float sin_t = om_sin(theta);
    *w_i = mat3mul(p->env_transform, V3(sin_t * om_cos(phi), om_cos(theta), sin_t * om_sin(phi)));
    float rgb[3];
    orc_env_texture(c->s->envmap, c->s->env_w, c->s->env_h, u, v, rgb);
    const float avg_w = imp_fetch(c->s, 0, 0, p->env_imp_base_mip);
    const float pdf = imp_fetch(c->s, posx, posy, 0) / avg_w;
    le_pdf[0] = p->env_strength * rgb[0];
    le_pdf[1] = p->env_strength * rgb[1];
    le_pdf[2] = p->env_strength * rgb[2];
    le_pdf[3] = pdf * INV_4PI;
}

/* ref: common.glsl:148-152 */
static float pdf_environment(const ctx_t* c, v3 dir) {
    const float avg_w = imp_fetch(c->s, 0, 0, c->p->env_imp_base_mip);
    const float pdf = luma(lookup_environment(c, dir)) / avg_w;
    return pdf * INV_4PI;
}

/* ------------------------------------------------------------------ */
/* ref: common.glsl:157-165 */
static int intersect_box(v3 pos, v3 dir, const float* bb_min, const float* bb_max, float* near, float* far) {
    const v3 inv_dir = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    const v3 lo = v3mul(v3sub(V3(bb_min[0], bb_min[1], bb_min[2]), pos), inv_dir);
    const v3 hi = v3mul(v3sub(V3(bb_max[0], bb_max[1], bb_max[2]), pos), inv_dir);
    const v3 tmin = V3(om_min(lo.x, hi.x), om_min(lo.y, hi.y), om_min(lo.z, hi.z));
    const v3 tmax = V3(om_max(lo.x, hi.x), om_max(lo.y, hi.y), om_max(lo.z, hi.z));
    *near = om_max(0.0f, om_max(tmin.x, om_max(tmin.y, tmin.z)));
    *far = om_min(tmax.x, om_min(tmax.y, tmax.z));
    return *near <= *far;
}

/* ------------------------------------------------------------------ */
/* phase function  ref: common.glsl:172-190, align :25-33 */
float orc_phase_hg(float cos_t, float g) {
    const float denom = 1.0f + sqr(g) + 2.0f * g * cos_t;
    return INV_4PI * (1.0f - sqr(g)) / (denom * sqrtf(denom));
}
static v3 align3(v3 N, v3 v) {
    v3 T;
    if (fabsf(N.x) > fabsf(N.y)) T = v3divs(V3(-N.z, 0.0f, N.x), sqrtf(N.x * N.x + N.z * N.z));
    else                         T = v3divs(V3(0.0f, N.z, -N.y), sqrtf(N.y * N.y + N.z * N.z));
    const v3 B = cross3(N, T);
    return normalize3(V3(v.x * T.x + v.y * B.x + v.z * N.x,
                         v.x * T.y + v.y * B.y + v.z * N.y,
                         v.x * T.z + v.y * B.z + v.z * N.z));
}
static v3 sample_phase_hg(v3 dir, float g, float r0, float r1) {
    const float cos_t = fabsf(g) < 1e-4f ? 1.0f - 2.0f * r0 :
        (1.0f + sqr(g) - sqr((1.0f - sqr(g)) / (1.0f - g + 2.0f * g * r0))) / (2.0f * g);
    const float sin_t = sqrtf(om_max(0.0f, 1.0f - sqr(cos_t)));
    const float phi = 2.0f * OM_PI * r1;
    return align3(dir, V3(sin_t * om_cos(phi), sin_t * om_sin(phi), cos_t));
}
void orc_sample_phase_hg(const float dir[3], float g, float r0, float r1, float out[3]) {
    v3 r = sample_phase_hg(V3(dir[0], dir[1], dir[2]), g, r0, r1);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------ */
/* transfer function  ref: common.glsl:203-212 */
static inline float tf_window(const orc_params* p, float d) {
    return om_clamp((d - p->tf_window_left) / p->tf_window_width, 0.0f, 1.0f - 1e-6f);
}
static void tf_lookup(ctx_t* c, float d, float rgba[4]) {
    const orc_params* p = c->p;
    const float tc = tf_window(p, d);
    const float tcs = tc * (float)p->tf_size;
    int idx = om_floor2i(tcs);
    const float f = tcs - floorf(tcs);
    int n = (int)p->tf_size;
    if (idx == INT32_MIN) idx = 0;                 /* NaN density (GL: undefined) */
    idx = clamp_i(idx, 0, n - 1);
    int idx1 = idx + 1 < n - 1 ? idx + 1 : n - 1;
    const float* a = c->s->tf_lut + 4 * idx;
    const float* b = c->s->tf_lut + 4 * idx1;
    for (int k = 0; k < 4; ++k) rgba[k] = om_mix(a[k], b[k], f);
    c->c.n_tf_lookup++;
}

/* ------------------------------------------------------------------ */
/* stochastic tricubic filter  ref: common.glsl:221-244 */
static void stochastic_tricubic_filter(v3 ipos, uint32_t* seed, int out[3]) {
    const float qx = ipos.x - 0.5f, qy = ipos.y - 0.5f, qz = ipos.z - 0.5f;
    const int ix = om_floor2i(qx), iy = om_floor2i(qy), iz = om_floor2i(qz);
    const v3 t = V3(qx - floorf(qx), qy - floorf(qy), qz - floorf(qz));   /* (ipos-0.5) - iipos */
    const v3 t2 = v3mul(t, t);
    const float k = 1.0f / 6.0f;
    float tt[3] = { t.x, t.y, t.z }, tt2[3] = { t2.x, t2.y, t2.z };
    float w[3], sumWt[3]; int idx[3] = { 0, 0, 0 };
    for (int a = 0; a < 3; ++a) { w[a] = k * (-tt[a] * tt2[a] + 3.0f * tt2[a] - 3.0f * tt[a] + 1.0f); sumWt[a] = w[a]; }
    /* second tap */
    for (int a = 0; a < 3; ++a) { w[a] = k * (3.0f * tt[a] * tt2[a] - 6.0f * tt2[a] + 4.0f); sumWt[a] = w[a] + sumWt[a]; }
    for (int a = 0; a < 3; ++a) { float r = rng(seed); if (r < w[a] / om_max(1e-3f, sumWt[a])) idx[a] = 1; }
    /* third tap */
    for (int a = 0; a < 3; ++a) { w[a] = k * (-3.0f * tt[a] * tt2[a] + 3.0f * tt2[a] + 3.0f * tt[a] + 1.0f); sumWt[a] = w[a] + sumWt[a]; }
    for (int a = 0; a < 3; ++a) { float r = rng(seed); if (r < w[a] / om_max(1e-3f, sumWt[a])) idx[a] = 2; }
    /* fourth tap */
    for (int a = 0; a < 3; ++a) { w[a] = k * tt[a] * tt2[a]; sumWt[a] = w[a] + sumWt[a]; }
    for (int a = 0; a < 3; ++a) { float r = rng(seed); if (r < w[a] / om_max(1e-3f, sumWt[a])) idx[a] = 3; }
    /* INT32_MIN (NaN position) stays far outside the grid after the small offset */
    out[0] = ix == INT32_MIN ? INT32_MIN : ix + idx[0] - 1;
    out[1] = iy == INT32_MIN ? INT32_MIN : iy + idx[1] - 1;
    out[2] = iz == INT32_MIN ? INT32_MIN : iz + idx[2] - 1;
}

/* ------------------------------------------------------------------ */
/* brick grid lookups  ref: common.glsl:268-281; out-of-range texelFetch (GL: undefined) reads 0 */
static inline void grid_extent(const orc_brickgrid* g, uint32_t e[3]) {
    for (int k = 0; k < 3; ++k) e[k] = g->extent[k] ? g->extent[k] : g->n_bricks[k] * 8u;
}

float orc_lookup_density_brick(const orc_brickgrid* g, int32_t x, int32_t y, int32_t z) {
    if (x < 0 || y < 0 || z < 0) return 0.0f;
    if (g->dense) {             /* dense fp16 grid: the voxel itself, 0 outside */
        if ((uint32_t)x >= g->extent[0] || (uint32_t)y >= g->extent[1] || (uint32_t)z >= g->extent[2]) return 0.0f;
        return om_half2float(g->dense[((size_t)z * g->extent[1] + (size_t)y) * g->extent[0] + (size_t)x]);
    }
    const uint32_t bx = (uint32_t)x >> 3, by = (uint32_t)y >> 3, bz = (uint32_t)z >> 3;
    if (bx >= g->n_bricks[0] || by >= g->n_bricks[1] || bz >= g->n_bricks[2]) return 0.0f;
    const size_t bi = ((size_t)bz * g->n_bricks[1] + by) * g->n_bricks[0] + bx;
    const uint32_t ind = g->indirection[bi];
    /* GL_UNSIGNED_INT_10_10_10_2: first component in the most significant bits (renderer.cpp:165-167) */
    const uint32_t ptrx = ind >> 22, ptry = (ind >> 12) & 1023u, ptrz = (ind >> 2) & 1023u;
    const uint32_t rg = g->range[bi];
    const float rmin = om_half2float((uint16_t)(rg & 0xFFFFu));
    const float rmax = om_half2float((uint16_t)(rg >> 16));
    const uint32_t ax = (ptrx << 3) + ((uint32_t)x & 7u), ay = (ptry << 3) + ((uint32_t)y & 7u), az = (ptrz << 3) + ((uint32_t)z & 7u);
    float unorm = 0.0f;
    if (ax < g->atlas_dim[0] && ay < g->atlas_dim[1] && az < g->atlas_dim[2])
        unorm = (float)g->atlas[((size_t)az * g->atlas_dim[1] + ay) * g->atlas_dim[0] + ax] / 255.0f;
    return rmin + unorm * (rmax - rmin);
}
/* range.y of the brick at mip (no density scale) */
float orc_lookup_majorant_raw(const orc_brickgrid* g, int32_t x, int32_t y, int32_t z, int32_t mip) {
    if (x < 0 || y < 0 || z < 0) return 0.0f;
    const uint32_t bx = (uint32_t)x >> (3 + mip), by = (uint32_t)y >> (3 + mip), bz = (uint32_t)z >> (3 + mip);
    const uint32_t* data; uint32_t dx, dy, dz;
    if (mip == 0) { data = g->range; dx = g->n_bricks[0]; dy = g->n_bricks[1]; dz = g->n_bricks[2]; }
    else {
        if ((uint32_t)mip > g->n_mips) return 0.0f;
        data = g->mips[mip - 1]; dx = g->mip_dim[mip - 1][0]; dy = g->mip_dim[mip - 1][1]; dz = g->mip_dim[mip - 1][2];
    }
    if (bx >= dx || by >= dy || bz >= dz) return 0.0f;
    return om_half2float((uint16_t)(data[((size_t)bz * dy + by) * dx + bx] >> 16));
}

static inline float lookup_majorant(const ctx_t* c, v3 ipos, int mip) {
    return c->p->vol_density_scale *
        orc_lookup_majorant_raw(c->s->density, om_floor2i(ipos.x), om_floor2i(ipos.y), om_floor2i(ipos.z), mip);
}
/* ref: common.glsl:284-286 on integer taps */
static inline float lookup_density_i(const ctx_t* c, int x, int y, int z) {
    return c->p->vol_density_scale * orc_lookup_density_brick(c->s->density, x, y, z);
}
/* ref: common.glsl:289-297 */
static float lookup_density_trilinear(const ctx_t* c, v3 ipos) {
    const float qx = ipos.x - 0.5f, qy = ipos.y - 0.5f, qz = ipos.z - 0.5f;
    const float fx = qx - floorf(qx), fy = qy - floorf(qy), fz = qz - floorf(qz);
    int ix = om_floor2i(qx), iy = om_floor2i(qy), iz = om_floor2i(qz);
    const orc_brickgrid* g = c->s->density;
    /* NaN/inf: every tap reads "outside" */
    int x1 = ix == INT32_MIN ? INT32_MIN : ix + 1, y1 = iy == INT32_MIN ? INT32_MIN : iy + 1, z1 = iz == INT32_MIN ? INT32_MIN : iz + 1;
    const float lx0 = om_mix(orc_lookup_density_brick(g, ix, iy, iz), orc_lookup_density_brick(g, x1, iy, iz), fx);
    const float lx1 = om_mix(orc_lookup_density_brick(g, ix, y1, iz), orc_lookup_density_brick(g, x1, y1, iz), fx);
    const float hx0 = om_mix(orc_lookup_density_brick(g, ix, iy, z1), orc_lookup_density_brick(g, x1, iy, z1), fx);
    const float hx1 = om_mix(orc_lookup_density_brick(g, ix, y1, z1), orc_lookup_density_brick(g, x1, y1, z1), fx);
    return c->p->vol_density_scale * om_mix(om_mix(lx0, lx1, fy), om_mix(hx0, hx1, fy), fz);
}
/* ref: common.glsl:300-304 */
static float lookup_density_stochastic(const ctx_t* c, v3 ipos, uint32_t* seed) {
    int tap[3];
    stochastic_tricubic_filter(ipos, seed, tap);
    return lookup_density_i(c, tap[0], tap[1], tap[2]);
}
/* ref: common.glsl:324-328.  Without an emission grid the reference reads unbound samplers
 * (renderer.cpp:117-124): the build defines that as 0 while still consuming the 9 draws. */
static v3 lookup_emission(const ctx_t* c, v3 ipos, uint32_t* seed) {
    const orc_params* p = c->p;
    if (!p->has_emission || !c->s->emission) {
        for (int i = 0; i < 9; ++i) (void)rng(seed);
        return V3(0.0f, 0.0f, 0.0f);
    }
    float M[16];
    orc_mat4_mul(p->vol_emission_inv_transform, p->vol_density_transform, M);
    const v3 ie = mat4point(M, ipos);
    int tap[3];
    stochastic_tricubic_filter(ie, seed, tap);
    const float t = orc_lookup_density_brick(c->s->emission, tap[0], tap[1], tap[2]) * p->vol_emission_norm;
    const v3 e = V3(t, sqr(t), sqr(sqr(t)));
    return V3(p->vol_emission_scale * sqr(e.x), p->vol_emission_scale * sqr(e.y), p->vol_emission_scale * sqr(e.z));
}

/* density at a tentative collision, both kernels: returns d and (TF) rgba */
static inline float collision_density(ctx_t* c, v3 ip, uint32_t* seed, float rgba[4]) {
    const orc_params* p = c->p;
    if (p->use_tf) {
        tf_lookup(c, lookup_density_trilinear(c, ip) * p->vol_inv_majorant, rgba);
        return p->vol_majorant * rgba[3];
    }
    return lookup_density_stochastic(c, ip, seed);
}

/* ------------------------------------------------------------------ */
/* global-majorant null-collision methods  ref: common.glsl:333-394 (compiled out in the reference: USE_DDA) */
static float transmittance_global(ctx_t* c, v3 wpos, v3 wdir, uint32_t* seed) {
    const orc_params* p = c->p;
    float near, far;
    if (!intersect_box(wpos, wdir, p->vol_bb_min, p->vol_bb_max, &near, &far)) return 1.0f;
    const v3 ipos = mat4point(p->vol_density_inv_transform, wpos);
    const v3 idir = mat4dir(p->vol_density_inv_transform, wdir);
    float t = near - om_log(1.0f - rng(seed)) * p->vol_inv_majorant, Tr = 1.0f;
    while (t < far) {
        float rgba[4];
        const float d = collision_density(c, v3axpy(ipos, t, idir), seed, rgba);
        c->c.n_coll_tr++;
        Tr *= 1.0f - d * p->vol_inv_majorant;
        if (Tr < 0.1f) {
            const float prob = 1.0f - Tr;
            if (rng(seed) < prob) return 0.0f;
            Tr /= 1.0f - prob;
        }
        t -= om_log(1.0f - rng(seed)) * p->vol_inv_majorant;
    }
    return Tr;
}
static int sample_volume_global(ctx_t* c, v3 wpos, v3 wdir, float* tout, v3* throughput, v3* Le, uint32_t* seed) {
    const orc_params* p = c->p;
    float near, far;
    if (!intersect_box(wpos, wdir, p->vol_bb_min, p->vol_bb_max, &near, &far)) return 0;
    const v3 ipos = mat4point(p->vol_density_inv_transform, wpos);
    const v3 idir = mat4dir(p->vol_density_inv_transform, wdir);
    float t = near - om_log(1.0f - rng(seed)) * p->vol_inv_majorant;
    while (t < far) {
        float rgba[4];
        const v3 ip = v3axpy(ipos, t, idir);
        const float d = collision_density(c, ip, seed, rgba);
        c->c.n_coll_sv++;
        const float P_real = d * p->vol_inv_majorant;
        const v3 em = lookup_emission(c, ip, seed);
        const v3 one_m_alb = V3(1.0f - p->vol_albedo[0], 1.0f - p->vol_albedo[1], 1.0f - p->vol_albedo[2]);
        *Le = v3add(*Le, v3scale(v3mul(v3mul(*throughput, one_m_alb), em), P_real));
        if (rng(seed) < P_real) {
            if (p->use_tf) *throughput = v3mul(*throughput, V3(rgba[0] * p->vol_albedo[0], rgba[1] * p->vol_albedo[1], rgba[2] * p->vol_albedo[2]));
            else           *throughput = v3mul(*throughput, V3(p->vol_albedo[0], p->vol_albedo[1], p->vol_albedo[2]));
            *tout = t;
            return 1;
        }
        t -= om_log(1.0f - rng(seed)) * p->vol_inv_majorant;
    }
    *tout = t;
    return 0;
}

/* ------------------------------------------------------------------ */
/* DDA  ref: common.glsl:399-409 */
#define MIP_START 3.0f
#define MIP_SPEED_UP 0.25f
#define MIP_SPEED_DOWN 2.0f

static float stepDDA(v3 pos, v3 inv_dir, int mip) {
    const float dim = (float)(8 << mip);
    const float idim = 1.0f / dim;
    const v3 offs = V3(inv_dir.x >= 0.0f ? dim + 0.5f : -0.5f,
                       inv_dir.y >= 0.0f ? dim + 0.5f : -0.5f,
                       inv_dir.z >= 0.0f ? dim + 0.5f : -0.5f);
    const v3 tmax = V3((floorf(pos.x * idim) * dim + offs.x - pos.x) * inv_dir.x,
                       (floorf(pos.y * idim) * dim + offs.y - pos.y) * inv_dir.y,
                       (floorf(pos.z * idim) * dim + offs.z - pos.z) * inv_dir.z);
    return om_min(tmax.x, om_min(tmax.y, tmax.z));
}

static inline float dda_majorant(ctx_t* c, v3 curr, int mip) {
    const orc_params* p = c->p;
    if (p->use_tf) {
        float rgba[4];
        tf_lookup(c, lookup_majorant(c, curr, mip) * p->vol_inv_majorant, rgba);
        return p->vol_majorant * rgba[3];
    }
    return lookup_majorant(c, curr, mip);
}

/* ref: common.glsl:412-455 */
static float transmittanceDDA(ctx_t* c, v3 wpos, v3 wdir, uint32_t* seed) {
    const orc_params* p = c->p;
    float near, far;
    if (!intersect_box(wpos, wdir, p->vol_bb_min, p->vol_bb_max, &near, &far)) return 1.0f;
    const v3 ipos = mat4point(p->vol_density_inv_transform, wpos);
    const v3 idir = mat4dir(p->vol_density_inv_transform, wdir);
    const v3 ri = V3(1.0f / idir.x, 1.0f / idir.y, 1.0f / idir.z);
    float t = near + 1e-6f, Tr = 1.0f, tau = -om_log(1.0f - rng(seed)), mip = MIP_START;
    while (t < far) {
        const v3 curr = v3axpy(ipos, t, idir);
        const int m = om_round_half_even(mip);
        const float majorant = dda_majorant(c, curr, m);
        const float dt = stepDDA(curr, ri, m);
        c->c.n_dda_tr++;
        t += dt;
        tau -= majorant * dt;
        mip = om_min(mip + MIP_SPEED_UP, 3.0f);
        if (tau > 0.0f) continue;
        t += tau / majorant;
        if (t >= far) break;
        float rgba[4];
        const float d = collision_density(c, v3axpy(ipos, t, idir), seed, rgba);
        c->c.n_coll_tr++;
        if (rng(seed) * majorant < d) {
            Tr *= om_max(0.0f, 1.0f - p->vol_majorant / majorant);
            if (Tr < 0.1f) {
                const float prob = 1.0f - Tr;
                if (rng(seed) < prob) return 0.0f;
                Tr /= 1.0f - prob;
            }
        }
        tau = -om_log(1.0f - rng(seed));
        mip = om_max(0.0f, mip - MIP_SPEED_DOWN);
    }
    return Tr;
}

/* ref: common.glsl:458-501 */
static int sample_volumeDDA(ctx_t* c, v3 wpos, v3 wdir, float* tout, v3* throughput, v3* Le, uint32_t* seed, int* hit_box) {
    const orc_params* p = c->p;
    float near, far;
    *hit_box = 0;
    if (!intersect_box(wpos, wdir, p->vol_bb_min, p->vol_bb_max, &near, &far)) return 0;
    *hit_box = 1;
    const v3 ipos = mat4point(p->vol_density_inv_transform, wpos);
    const v3 idir = mat4dir(p->vol_density_inv_transform, wdir);
    const v3 ri = V3(1.0f / idir.x, 1.0f / idir.y, 1.0f / idir.z);
    float t = near + 1e-6f;
    float tau = -om_log(1.0f - rng(seed)), mip = MIP_START;
    while (t < far) {
        const v3 curr = v3axpy(ipos, t, idir);
        const int m = om_round_half_even(mip);
        const float majorant = dda_majorant(c, curr, m);
        const float dt = stepDDA(curr, ri, m);
        c->c.n_dda_sv++;
        t += dt;
        tau -= majorant * dt;
        mip = om_min(mip + MIP_SPEED_UP, 3.0f);
        if (tau > 0.0f) continue;
        t += tau / majorant;
        if (t >= far) break;
        float rgba[4];
        const v3 ip = v3axpy(ipos, t, idir);
        const float d = collision_density(c, ip, seed, rgba);
        c->c.n_coll_sv++;
        /* Le += throughput * (1 - albedo) * emission * d * vol_inv_majorant (global inverse majorant: reference quirk) */
        const v3 em = lookup_emission(c, ip, seed);
        const v3 one_m_alb = V3(1.0f - p->vol_albedo[0], 1.0f - p->vol_albedo[1], 1.0f - p->vol_albedo[2]);
        *Le = v3add(*Le, v3scale(v3scale(v3mul(v3mul(*throughput, one_m_alb), em), d), p->vol_inv_majorant));
        if (rng(seed) * majorant < d) {
            *throughput = v3mul(*throughput, V3(p->vol_albedo[0], p->vol_albedo[1], p->vol_albedo[2]));
            if (p->use_tf) *throughput = v3mul(*throughput, V3(rgba[0], rgba[1], rgba[2]));
            *tout = t;
            return 1;
        }
        tau = -om_log(1.0f - rng(seed));
        mip = om_max(0.0f, mip - MIP_SPEED_DOWN);
    }
    *tout = t;
    return 0;
}

/* function-level probes (tests/glsl_pin_worker.py; probe.glsl modes 4, 5, 9) */
void orc_view_dir(const orc_params* p, int32_t x, int32_t y, int32_t w, int32_t h, float jx, float jy, float out[3]) {
    const v3 d = view_dir(p, x, y, w, h, jx, jy);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
int orc_intersect_box(const orc_params* p, const float pos[3], const float dir[3], float near_far[2]) {
    return intersect_box(V3(pos[0], pos[1], pos[2]), V3(dir[0], dir[1], dir[2]), p->vol_bb_min, p->vol_bb_max, &near_far[0], &near_far[1]);
}
/* one camera segment of the DDA tracker: returns real-collision flag; out = (t, throughput rgb) */
int orc_sample_volume(const orc_params* p, const orc_scene* s, const float pos[3], const float dir[3], uint32_t* seed, float out[4]) {
    ctx_t c; memset(&c, 0, sizeof c); c.p = p; c.s = s;
    v3 thr = V3(1, 1, 1), Le = V3(0, 0, 0);
    float t = 0.0f;
    int hit = 0;
    const int real = sample_volumeDDA(&c, V3(pos[0], pos[1], pos[2]), V3(dir[0], dir[1], dir[2]), &t, &thr, &Le, seed, &hit);
    out[0] = t; out[1] = thr.x; out[2] = thr.y; out[3] = thr.z;
    return real;
}

float orc_transmittance(const orc_params* p, const orc_scene* s, const float pos[3], const float dir[3], uint32_t* seed) {
    ctx_t c; memset(&c, 0, sizeof c); c.p = p; c.s = s;
    v3 P = V3(pos[0], pos[1], pos[2]), D = V3(dir[0], dir[1], dir[2]);
    return p->integrator == 0 ? transmittanceDDA(&c, P, D, seed) : transmittance_global(&c, P, D, seed);
}

/* ------------------------------------------------------------------ */
/* ray-marching trackers  ref: common.glsl:506-566 (RAYMARCH_STEPS 64).  Dead code in the reference (no kernel calls them;
 * trace_path only switches between the DDA and the global-majorant pair).  Offered as integrator = 3: trace_path with
 * sample_volume_raymarch / transmittance_raymarch in place of sample_volumeDDA / transmittanceDDA; the `pdf` output of
 * sample_volume_raymarch has no consumer in trace_path.  Both use lookup_density_stochastic, also with a transfer function. */
#define RAYMARCH_STEPS 64
static float transmittance_raymarch(ctx_t* c, v3 wpos, v3 wdir, uint32_t* seed) {
    const orc_params* p = c->p;
    float near, far;
    if (!intersect_box(wpos, wdir, p->vol_bb_min, p->vol_bb_max, &near, &far)) return 1.0f;
    const v3 ipos = mat4point(p->vol_density_inv_transform, wpos);
    const v3 idir = mat4dir(p->vol_density_inv_transform, wdir);
    const float dt = (far - near) / (float)RAYMARCH_STEPS;
    near += rng(seed) * dt;
    float tau = 0.0f;
    for (int i = 0; i < RAYMARCH_STEPS; ++i) {
        const float d = lookup_density_stochastic(c, v3axpy(ipos, om_min(near + (float)i * dt, far), idir), seed);
        c->c.n_coll_tr++;
        if (p->use_tf) {
            float rgba[4];
            tf_lookup(c, d * p->vol_inv_majorant, rgba);
            tau += rgba[3] * p->vol_majorant * dt;
        } else {
            tau += d * dt;
        }
    }
    return om_exp(-tau);
}
static int sample_volume_raymarch(ctx_t* c, v3 wpos, v3 wdir, float* tout, v3* throughput, float* pdf, uint32_t* seed, int* hit_box) {
    const orc_params* p = c->p;
    *pdf = 1.0f;
    float near, far;
    *hit_box = 0;
    if (!intersect_box(wpos, wdir, p->vol_bb_min, p->vol_bb_max, &near, &far)) return 0;
    *hit_box = 1;
    const v3 ipos = mat4point(p->vol_density_inv_transform, wpos);
    const v3 idir = mat4dir(p->vol_density_inv_transform, wdir);
    const float tau_target = -om_log(1.0f - rng(seed));
    const float dt = (far - near) / (float)RAYMARCH_STEPS;
    near += rng(seed) * dt;
    float tau = 0.0f;
    for (int i = 0; i < RAYMARCH_STEPS; ++i) {
        const float t = om_min(near + (float)i * dt, far);
        *tout = t;
        const float d = lookup_density_stochastic(c, v3axpy(ipos, t, idir), seed);
        c->c.n_coll_sv++;
        float rgba[4] = { 0, 0, 0, 0 };
        if (p->use_tf) {
            tf_lookup(c, d * p->vol_inv_majorant, rgba);
            tau += rgba[3] * p->vol_majorant * dt;
        } else {
            tau += d * dt;
        }
        if (tau >= tau_target) {
            const v3 albedo = p->use_tf ? V3(rgba[0] * p->vol_albedo[0], rgba[1] * p->vol_albedo[1], rgba[2] * p->vol_albedo[2])
                                        : V3(p->vol_albedo[0], p->vol_albedo[1], p->vol_albedo[2]);
            *pdf = ((((albedo.x + albedo.y) + albedo.z) / 3.0f) * d) * om_exp(-tau_target);      /* mean(albedo) * d * exp(-tau_target) */
            *throughput = v3mul(*throughput, albedo);
            return 1;
        }
    }
    *pdf = om_exp(-tau);
    return 0;
}

/* ------------------------------------------------------------------ */
/* ref: common.glsl:599-652 */
static void trace_path(ctx_t* c, v3 pos, v3 dir, uint32_t* seed, float out[4]) {
    const orc_params* p = c->p;
    v3 L = V3(0, 0, 0), throughput = V3(1, 1, 1);
    int free_path = 1;
    uint32_t n_paths = 0;
    float t = 0.0f, f_p = 0.0f;
    for (;;) {
        int hit_box = 1, real;
        float rm_pdf;
        if (p->integrator == 0)      real = sample_volumeDDA(c, pos, dir, &t, &throughput, &L, seed, &hit_box);
        else if (p->integrator == 3) real = sample_volume_raymarch(c, pos, dir, &t, &throughput, &rm_pdf, seed, &hit_box);
        else                         real = sample_volume_global(c, pos, dir, &t, &throughput, &L, seed);
        if (n_paths == 0 && !hit_box) c->c.n_primary_miss++;
        if (!real) break;
        pos = v3axpy(pos, t, dir);
        /* sample light source (environment) */
        v3 w_i; float le_pdf[4];
        const float r0 = rng(seed), r1 = rng(seed);
        sample_environment(c, r0, r1, &w_i, le_pdf);
        c->c.n_nee++;
        if (le_pdf[3] > 0.0f) {
            f_p = orc_phase_hg(dot3(v3neg(dir), w_i), p->vol_phase_g);
            const float mis_weight = p->show_environment > 0 ? power_heuristic(le_pdf[3], f_p) : 1.0f;
            const float Tr = p->integrator == 0 ? transmittanceDDA(c, pos, w_i, seed)
                           : (p->integrator == 3 ? transmittance_raymarch(c, pos, w_i, seed) : transmittance_global(c, pos, w_i, seed));
            /* L += throughput * mis_weight * f_p * Tr * Le_pdf.rgb / Le_pdf.w */
            v3 a = v3scale(v3scale(v3scale(throughput, mis_weight), f_p), Tr);
            a = v3mul(a, V3(le_pdf[0], le_pdf[1], le_pdf[2]));
            L = v3add(L, v3divs(a, le_pdf[3]));
        }
        if (++n_paths >= (uint32_t)p->bounces) { free_path = 0; break; }
        const float rr_val = luma(throughput);
        if (rr_val < 0.1f) {
            const float prob = 1.0f - rr_val;
            if (rng(seed) < prob) { free_path = 0; break; }
            throughput = v3divs(throughput, 1.0f - prob);
        }
        const float s0 = rng(seed), s1 = rng(seed);
        const v3 scatter_dir = sample_phase_hg(dir, p->vol_phase_g, s0, s1);
        f_p = orc_phase_hg(dot3(v3neg(dir), scatter_dir), p->vol_phase_g);
        dir = scatter_dir;
    }
    if (free_path && p->show_environment > 0) {
        const v3 Le = lookup_environment(c, dir);
        const float mis_weight = n_paths > 0 ? power_heuristic(f_p, pdf_environment(c, dir)) : 1.0f;
        L = v3add(L, v3mul(v3scale(throughput, mis_weight), Le));
        c->c.n_esc++;
    }
    out[0] = L.x; out[1] = L.y; out[2] = L.z;
    out[3] = n_paths > 0 ? 1.0f : 0.0f;     /* clamp(n_paths, 0.f, 1.f) */
}

/* ref: common.glsl:571-591 direct_volume_rendering: 64 jittered steps of emission-absorption compositing through the
 * transfer function (dead code in the reference: no kernel calls it; selectable here as integrator = 2, needs a LUT).
 * out[3] (not defined by the reference): opacity 1 - Tr. */
static void direct_volume_rendering(ctx_t* c, v3 pos, v3 dir, uint32_t* seed, float out[4]) {
    const orc_params* p = c->p;
    v3 L = V3(0, 0, 0);
    float near, far;
    if (!intersect_box(pos, dir, p->vol_bb_min, p->vol_bb_max, &near, &far)) {
        const v3 e = lookup_environment(c, dir);
        out[0] = e.x; out[1] = e.y; out[2] = e.z; out[3] = 0.0f;
        return;
    }
    const v3 ipos = mat4point(p->vol_density_inv_transform, pos);
    const v3 idir = mat4dir(p->vol_density_inv_transform, dir);
    const float dt = (far - near) / (float)RAYMARCH_STEPS;
    near += rng(seed) * dt;
    float Tr = 1.0f;
    for (int i = 0; i < RAYMARCH_STEPS; ++i) {
        float rgba[4];
        tf_lookup(c, lookup_density_trilinear(c, v3axpy(ipos, om_min(near + (float)i * dt, far), idir)) * p->vol_inv_majorant, rgba);
        const float dtau = rgba[3] * p->vol_majorant * dt;
        L = v3add(L, v3scale(v3scale(V3(rgba[0], rgba[1], rgba[2]), dtau), Tr));
        Tr *= om_exp(-dtau);
        if (Tr <= 1e-6f) { out[0] = L.x; out[1] = L.y; out[2] = L.z; out[3] = 1.0f - Tr; return; }
    }
    const v3 e = lookup_environment(c, dir);
    L = v3add(L, v3scale(e, Tr));
    out[0] = L.x; out[1] = L.y; out[2] = L.z; out[3] = 1.0f - Tr;
}

/* ref: pathtracer_brick.glsl:23-37 / pathtracer_brick_tf.glsl:24-38, one pixel, one sample */
static void pixel_sample(ctx_t* c, int x, int y, int sample, float out[4]) {
    const orc_params* p = c->p;
    const int W = p->resolution[0], H = p->resolution[1];
    /* GLSL int arithmetic wraps: seed * (y*W + x) */
    uint32_t seed = orc_tea((uint32_t)p->seed * (uint32_t)(y * W + x), (uint32_t)sample, 32);
    const v3 pos = V3(p->cam_pos[0], p->cam_pos[1], p->cam_pos[2]);
    const float jx = rng(&seed), jy = rng(&seed);
    const v3 dir = view_dir(p, x, y, W, H, jx, jy);
    if (p->integrator == 2 && p->use_tf) direct_volume_rendering(c, pos, dir, &seed, out);
    else trace_path(c, pos, dir, &seed, out);
    c->c.samples++;
}

void orc_trace_pixel_sample(const orc_params* p, const orc_scene* s, int32_t x, int32_t y, int32_t sample, float out[4]) {
    ctx_t c; memset(&c, 0, sizeof c); c.p = p; c.s = s;
    pixel_sample(&c, x, y, sample, out);
}

static void counters_add(orc_counters* a, const orc_counters* b) {
    a->samples += b->samples; a->n_dda_sv += b->n_dda_sv; a->n_dda_tr += b->n_dda_tr;
    a->n_coll_sv += b->n_coll_sv; a->n_coll_tr += b->n_coll_tr; a->n_nee += b->n_nee;
    a->n_esc += b->n_esc; a->n_primary_miss += b->n_primary_miss; a->n_tf_lookup += b->n_tf_lookup;
}

int32_t orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_render(const orc_params* p, const orc_scene* s, float* fb,
                int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                int32_t first_sample, int32_t n_samples, int32_t threads, orc_counters* counters) {
    const int W = p->resolution[0];
    orc_counters total; memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
    {
        ctx_t c; memset(&c, 0, sizeof c); c.p = p; c.s = s;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int y = y0; y < y1; ++y) {
            for (int x = x0; x < x1; ++x) {
                float* px = fb + 4 * ((size_t)y * W + x);
                for (int sm = first_sample; sm < first_sample + n_samples; ++sm) {
                    float L[4];
                    pixel_sample(&c, x, y, sm, L);
                    /* imageStore(color, pixel, mix(imageLoad(color, pixel), sanitize(L), 1.f / current_sample)) */
                    const float a = 1.0f / (float)sm;
                    for (int k = 0; k < 4; ++k) px[k] = om_mix(px[k], sanitize(L[k]), a);
                }
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        counters_add(&total, &c.c);
    }
    if (counters) counters_add(counters, &total);
}

/* ------------------------------------------------------------------ */
/* tonemap  ref: tonemap.glsl:13-36 */
static inline float hable(float x) {
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
void orc_tonemap(float* fb, int32_t w, int32_t h, float exposure, float gamma) {
    const float W = 11.2f;
    const float inv_gamma = 1.0f / gamma;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        float* px = fb + 4 * i;
        for (int k = 0; k < 3; ++k)
            px[k] = om_pow(hable(exposure * px[k]) / hable(W), inv_gamma);
        for (int k = 0; k < 4; ++k) px[k] = sanitize(px[k]);
    }
}

/* ------------------------------------------------------------------ */
/* Environment importance pyramid  ref: env_setup.glsl:18-34, environment.cpp:6-33 */
int32_t orc_impmap_floats(int32_t dim) {
    int32_t n = 0;
    for (int d = dim; d >= 1; d >>= 1) n += d * d;
    return n;
}
void orc_build_impmap(const float* env_tex, int32_t w, int32_t h, int32_t dim, float* out) {
    const int ns = 8;                                  /* sqrt(SAMPLES = 64) */
    const float inv_samples = 1.0f / (float)(ns * ns);
    const float oss = (float)(dim * ns);               /* output_size_samples */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int py = 0; py < dim; ++py)
        for (int px = 0; px < dim; ++px) {
            float importance = 0.0f;
            for (int y = 0; y < ns; ++y)
                for (int x = 0; x < ns; ++x) {
                    const float u = ((float)(px * ns) + ((float)x + 0.5f)) / oss;
                    const float v = ((float)(py * ns) + ((float)y + 0.5f)) / oss;
                    float rgb[3];
                    orc_env_texture(env_tex, w, h, u, v, rgb);
                    importance += luma(V3(rgb[0], rgb[1], rgb[2]));
                }
            out[(size_t)py * dim + px] = importance * inv_samples;
        }
    /* glGenerateMipmap: 2x2 box filter, (t00 + t10 + t01 + t11) * 0.25 */
    float* src = out; int d = dim;
    while (d > 1) {
        float* dst = src + (size_t)d * d; int hd = d >> 1;
        for (int y = 0; y < hd; ++y)
            for (int x = 0; x < hd; ++x) {
                const float a = src[(size_t)(2 * y) * d + 2 * x], b = src[(size_t)(2 * y) * d + 2 * x + 1];
                const float c = src[(size_t)(2 * y + 1) * d + 2 * x], e = src[(size_t)(2 * y + 1) * d + 2 * x + 1];
                dst[(size_t)y * hd + x] = ((a + b) + (c + e)) * 0.25f;
            }
        src = dst; d = hd;
    }
}

void orc_sample_environment(const orc_params* p, const orc_scene* s, float r0, float r1, float w_i[3], float le_pdf[4]) {
    ctx_t c; memset(&c, 0, sizeof c); c.p = p; c.s = s;
    v3 w; sample_environment(&c, r0, r1, &w, le_pdf);
    w_i[0] = w.x; w_i[1] = w.y; w_i[2] = w.z;
}

/* ------------------------------------------------------------------ */
/* TransferFunction  ref: transferfunc.cpp:33-58 */
int orc_lut_fixup(float* rgba, int32_t n) {
    int needs_cdf = 0;
    for (int i = 1; i < n; ++i)
        if (rgba[4 * (i - 1) + 3] > rgba[4 * i + 3]) { needs_cdf = 1; break; }
    if (!needs_cdf) return 0;
    for (int i = 1; i < n; ++i) rgba[4 * i + 3] += rgba[4 * (i - 1) + 3];
    const float integral = rgba[4 * (n - 1) + 3];
    for (int i = 0; i < n; ++i)
        rgba[4 * i + 3] = integral <= 0.0f ? (float)(i + 1) / (float)n : rgba[4 * i + 3] / integral;
    return 1;
}
/* ref: transferfunc.cpp:79-93: one row per line, also for a blank line; components that fail to parse are uninitialised there -- defined here as the
 * previous row's values (first row: 0), which is what the same stack slots hold in practice */
int orc_load_lut(const char* path, float* rgba, int32_t max_rows) {
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    char tmp[256]; int n = 0;
    float r = 0, g = 0, b = 0, a = 0;
    while (n < max_rows && fgets(tmp, sizeof tmp, f)) {
        sscanf(tmp, "%f, %f, %f, %f", &r, &g, &b, &a);
        rgba[4 * n + 0] = r; rgba[4 * n + 1] = g; rgba[4 * n + 2] = b; rgba[4 * n + 3] = a;
        n++;
    }
    fclose(f);
    return n;
}

/* ------------------------------------------------------------------ */
/* matrices (glm conventions, column-major) */
void orc_mat4_mul(const float a[16], const float b[16], float out[16]) {
    float r[16];
    for (int c = 0; c < 4; ++c)
        for (int rr = 0; rr < 4; ++rr)
            r[4 * c + rr] = a[rr] * b[4 * c] + a[4 + rr] * b[4 * c + 1] + a[8 + rr] * b[4 * c + 2] + a[12 + rr] * b[4 * c + 3];
    memcpy(out, r, sizeof r);
}
void orc_mat3_inverse(const float m[9], float out[9]) {
    /* cofactor expansion; m[c*3+r] */
    const float a = m[0], b = m[3], c = m[6], d = m[1], e = m[4], f = m[7], g = m[2], h = m[5], i = m[8];
    const float det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const float id = 1.0f / det;
    float r[9];
    r[0] = (e * i - f * h) * id; r[3] = -(b * i - c * h) * id; r[6] = (b * f - c * e) * id;
    r[1] = -(d * i - f * g) * id; r[4] = (a * i - c * g) * id; r[7] = -(a * f - c * d) * id;
    r[2] = (d * h - e * g) * id; r[5] = -(a * h - b * g) * id; r[8] = (a * e - b * d) * id;
    memcpy(out, r, sizeof r);
}
void orc_mat4_inverse(const float m[16], float out[16]) {
    float inv[16];
    inv[0]  =  m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4]  = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8]  =  m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1]  = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5]  =  m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9]  = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] =  m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2]  =  m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6]  = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] =  m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3]  = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7]  =  m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] =  m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    const float id = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * id;
}

/* ref: renderer.cpp:227-242; index_extent = n_bricks * 8 for a loaded BrickGrid */
void orc_unit_cube(const orc_brickgrid* g, float vt[16], float* density_scale) {
    const float* T = g->transform;
    uint32_t ge[3]; grid_extent(g, ge);
    const v3 ext_i = V3((float)ge[0], (float)ge[1], (float)ge[2]);
    const v3 c0 = mat4point(T, V3(0, 0, 0)), c1 = mat4point(T, ext_i);
    /* bb_min = min(FLT_MAX, c0); bb_max = max(FLT_MIN, c1) (reference quirk: FLT_MIN is the smallest positive float) */
    const v3 bb_min = V3(om_min(3.402823466e+38f, c0.x), om_min(3.402823466e+38f, c0.y), om_min(3.402823466e+38f, c0.z));
    const v3 bb_max = V3(om_max(1.175494351e-38f, c1.x), om_max(1.175494351e-38f, c1.y), om_max(1.175494351e-38f, c1.z));
    const v3 extent = v3sub(bb_max, bb_min);
    const float size = fmaxf(extent.x, fmaxf(extent.y, extent.z));
    for (int i = 0; i < 16; ++i) vt[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    if (size != 1.0f) {
        const float s = 1.0f / size;
        /* glm::translate(glm::scale(mat4(1), vec3(1/size)), -bb_min - 0.5f * extent): m[3] = m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3] */
        const v3 v = v3sub(v3neg(bb_min), v3scale(extent, 0.5f));
        vt[0] = s; vt[5] = s; vt[10] = s;
        vt[12] = s * v.x; vt[13] = s * v.y; vt[14] = s * v.z;
        *density_scale *= size;
    }
}

/* glm::lookAt(pos, pos+dir, up) then inverse(mat3(view)): columns right, up, -forward.
 * cppgl is not vendored; the orthonormal inverse is taken as the transpose. */
void orc_camera(const float pos[3], const float dir[3], const float up[3], float cam_transform[9]) {
    (void)pos;
    const v3 f = normalize3(V3(dir[0], dir[1], dir[2]));
    const v3 s = normalize3(cross3(f, V3(up[0], up[1], up[2])));
    const v3 u = cross3(s, f);
    cam_transform[0] = s.x; cam_transform[1] = s.y; cam_transform[2] = s.z;
    cam_transform[3] = u.x; cam_transform[4] = u.y; cam_transform[5] = u.z;
    cam_transform[6] = -f.x; cam_transform[7] = -f.y; cam_transform[8] = -f.z;
}

/* ref: main.cpp:382; glm::rotate about +y: [c 0 -s; 0 1 0; s 0 c] columns (c,0,-s),(0,1,0),(s,0,c) */
void orc_env_rotation(float deg, float m[9]) {
    const float a = deg * 0.01745329251994329576923690768489f;   /* glm::radians */
    const float c = om_cos(a), s = om_sin(a);
    m[0] = c; m[1] = 0; m[2] = -s;
    m[3] = 0; m[4] = 1; m[5] = 0;
    m[6] = s; m[7] = 0; m[8] = c;
}

/* ref: renderer.cpp:96-124 */
void orc_volume_uniforms(orc_params* p, const orc_brickgrid* density, const orc_brickgrid* emission,
                         const float vt[16], float density_scale,
                         const float clip_min[3], const float clip_max[3], float majorant_emission) {
    /* volume->AABB(): world-space box of the grid (voldata, unvendored): corners of [0, index_extent] through volume.transform * grid.transform */
    float M[16];
    orc_mat4_mul(vt, density->transform, M);
    uint32_t ge[3]; grid_extent(density, ge);
    const v3 ext_i = V3((float)ge[0], (float)ge[1], (float)ge[2]);
    v3 lo = V3(INFINITY, INFINITY, INFINITY), hi = V3(-INFINITY, -INFINITY, -INFINITY);
    for (int k = 0; k < 8; ++k) {
        const v3 cn = mat4point(M, V3((k & 1) ? ext_i.x : 0.0f, (k & 2) ? ext_i.y : 0.0f, (k & 4) ? ext_i.z : 0.0f));
        lo = V3(om_min(lo.x, cn.x), om_min(lo.y, cn.y), om_min(lo.z, cn.z));
        hi = V3(om_max(hi.x, cn.x), om_max(hi.y, cn.y), om_max(hi.z, cn.z));
    }
    const v3 ext = v3sub(hi, lo);
    p->vol_bb_min[0] = lo.x + clip_min[0] * ext.x; p->vol_bb_min[1] = lo.y + clip_min[1] * ext.y; p->vol_bb_min[2] = lo.z + clip_min[2] * ext.z;
    p->vol_bb_max[0] = lo.x + clip_max[0] * ext.x; p->vol_bb_max[1] = lo.y + clip_max[1] * ext.y; p->vol_bb_max[2] = lo.z + clip_max[2] * ext.z;
    p->vol_minorant = density->min_maj[0] * density_scale;
    p->vol_majorant = density->min_maj[1] * density_scale;
    p->vol_inv_majorant = 1.0f / (density->min_maj[1] * density_scale);
    p->vol_density_scale = density_scale;
    p->vol_emission_norm = majorant_emission > 0.0f ? 1.0f / fmaxf(majorant_emission, 1e-4f) : 1.0f;
    memcpy(p->vol_density_transform, M, sizeof M);
    orc_mat4_inverse(M, p->vol_density_inv_transform);
    if (emission) {
        float E[16];
        orc_mat4_mul(vt, emission->transform, E);
        memcpy(p->vol_emission_transform, E, sizeof E);
        orc_mat4_inverse(E, p->vol_emission_inv_transform);
        p->has_emission = 1;
    } else {
        p->has_emission = 0;
    }
}

/* ------------------------------------------------------------------ */
/* .brick reader */
static int rd(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n ? 0 : -1; }

static int read_buf3d(FILE* f, size_t elem, uint32_t dim[3], void** data) {
    uint64_t count;
    if (rd(f, dim, 12) || rd(f, &count, 8)) return -1;
    if (count != (uint64_t)dim[0] * dim[1] * dim[2] || count > ((uint64_t)1 << 34)) return -1;
    *data = malloc((size_t)count * elem + 16);
    if (!*data) return -1;
    return rd(f, *data, (size_t)count * elem);
}

int orc_load_brick(const char* path, orc_brickgrid* g) {
    memset(g, 0, sizeof *g);
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    uint8_t endian; int err = 0;
    uint32_t dim[3];
    err |= rd(f, &endian, 1);
    if (err || endian != 1) { fclose(f); return -2; }
    err |= rd(f, g->transform, 64);
    err |= rd(f, g->n_bricks, 12);
    err |= rd(f, g->min_maj, 8);
    err |= rd(f, &g->brick_counter, 8);
    if (!err) err |= read_buf3d(f, 4, dim, (void**)&g->indirection);
    if (!err && (dim[0] != g->n_bricks[0] || dim[1] != g->n_bricks[1] || dim[2] != g->n_bricks[2])) err = -3;
    if (!err) err |= read_buf3d(f, 4, dim, (void**)&g->range);
    if (!err && (dim[0] != g->n_bricks[0] || dim[1] != g->n_bricks[1] || dim[2] != g->n_bricks[2])) err = -3;
    if (!err) err |= read_buf3d(f, 1, g->atlas_dim, (void**)&g->atlas);
    uint64_t nm = 0;
    if (!err) err |= rd(f, &nm, 8);
    if (!err && nm > 8) err = -4;
    g->n_mips = (uint32_t)nm;
    for (uint32_t i = 0; !err && i < g->n_mips; ++i) err |= read_buf3d(f, 4, g->mip_dim[i], (void**)&g->mips[i]);
    if (!err) { uint8_t extra; if (fread(&extra, 1, 1, f) != 0) err = -5; }   /* must end exactly */
    fclose(f);
    if (err) { orc_free_brick(g); return err; }
    return 0;
}
void orc_free_brick(orc_brickgrid* g) {
    free(g->indirection); free(g->range); free(g->atlas);
    for (int i = 0; i < 8; ++i) free(g->mips[i]);
    memset(g, 0, sizeof *g);
}
void orc_free(void* p) { free(p); }

/* ------------------------------------------------------------------ */
/* Radiance .hdr (RGBE, new-style RLE).  value = mantissa * 2^(e-136) */
static float rgbe_scale(uint8_t e) { return e ? ldexpf(1.0f, (int)e - 136) : 0.0f; }

int orc_load_hdr(const char* path, float** out, int32_t* w, int32_t* h) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    char line[512]; int have_fmt = 0;
    if (!fgets(line, sizeof line, f) || strncmp(line, "#?", 2) != 0) { fclose(f); return -2; }
    for (;;) {
        if (!fgets(line, sizeof line, f)) { fclose(f); return -2; }
        if (line[0] == '\n') break;
        if (strncmp(line, "FORMAT=32-bit_rle_rgbe", 22) == 0) have_fmt = 1;
    }
    if (!fgets(line, sizeof line, f)) { fclose(f); return -2; }
    int W = 0, H = 0;
    if (sscanf(line, "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0 || !have_fmt) { fclose(f); return -3; }
    float* img = (float*)malloc((size_t)W * H * 3 * sizeof(float));
    uint8_t* scan = (uint8_t*)malloc((size_t)W * 4);
    int err = 0;
    for (int y = 0; y < H && !err; ++y) {
        uint8_t hd[4];
        if (rd(f, hd, 4)) { err = -4; break; }
        if (W >= 8 && W < 32768 && hd[0] == 2 && hd[1] == 2 && !(hd[2] & 0x80)) {
            if (((int)hd[2] << 8 | hd[3]) != W) { err = -5; break; }
            for (int ch = 0; ch < 4 && !err; ++ch) {
                int x = 0;
                while (x < W) {
                    int cnt = fgetc(f);
                    if (cnt == EOF) { err = -4; break; }
                    if (cnt > 128) {
                        cnt -= 128; int val = fgetc(f);
                        if (val == EOF || x + cnt > W) { err = -4; break; }
                        while (cnt--) scan[4 * (x++) + ch] = (uint8_t)val;
                    } else {
                        if (cnt == 0 || x + cnt > W) { err = -4; break; }
                        while (cnt--) { int val = fgetc(f); if (val == EOF) { err = -4; break; } scan[4 * (x++) + ch] = (uint8_t)val; }
                    }
                }
            }
        } else {            /* flat scanline */
            memcpy(scan, hd, 4);
            if (rd(f, scan + 4, (size_t)(W - 1) * 4)) { err = -4; break; }
        }
        for (int x = 0; x < W; ++x) {
            const float s = rgbe_scale(scan[4 * x + 3]);
            float* px = img + 3 * ((size_t)y * W + x);
            px[0] = (float)scan[4 * x + 0] * s; px[1] = (float)scan[4 * x + 1] * s; px[2] = (float)scan[4 * x + 2] * s;
        }
    }
    free(scan); fclose(f);
    if (err) { free(img); return err; }
    *out = img; *w = W; *h = H;
    return 0;
}

void orc_flip_rows(float* img, int32_t w, int32_t h, int32_t ch) {
    size_t row = (size_t)w * ch;
    float* tmp = (float*)malloc(row * sizeof(float));
    for (int y = 0; y < h / 2; ++y) {
        float* a = img + (size_t)y * row; float* b = img + (size_t)(h - 1 - y) * row;
        memcpy(tmp, a, row * sizeof(float)); memcpy(a, b, row * sizeof(float)); memcpy(b, tmp, row * sizeof(float));
    }
    free(tmp);
}

float orc_math(int32_t fn, float a, float b) {
    switch (fn) {
    case 0: return om_log(a);
    case 1: return om_sin(a);
    case 2: return om_cos(a);
    case 3: return om_tan(a);
    case 4: return om_acos(a);
    case 5: return om_atan2(a, b);
    case 6: return om_exp(a);
    case 7: return om_pow(a, b);
    case 8: return om_asin(a);
    default: return NAN;
    }
}
