// vr_scene.h -- the kernel argument block: the reference's uniforms (src/renderer.cpp:88-138) plus device views
// of the scene arrays.  Plain data, shared by the host marshalling code and the kernels.
#pragma once

#include <stdint.h>

namespace vr {

// One brick of a voldata::BrickGrid, repacked for HBM (DESIGN.md "Data layout"):
//   slot  : index of the brick's 8x8x8 u8 block in the atlas (512 contiguous bytes per block).  The atlas is brick-LINEAR:
//           slot == the record's own index, so the hot path computes the voxel address without reading it
//   rmin  : float(range.x); rdiff = float(range.y) - float(range.x)  (both exact/IEEE, done once at commit)
//   range : the file's 2 x fp16 word, low = min, high = max (GL_RG16F in renderer.cpp:181-183), kept for reference
// one 16-byte load per tap;
// replaces the indirection (RGB10_A2UI) + range + 3D-atlas texture triple of common.glsl:268-275.
struct alignas(16) BrickRec { uint32_t slot; float rmin; float rdiff; uint32_t range; };

#if defined(__HIPCC__)
#define VR_SCENE_HD __host__ __device__ inline
#else
#define VR_SCENE_HD inline
#endif

// Index arithmetic of the hot loops is shifts and 24-bit multiply-adds only (one instruction each on gfx950): brick records -- and with
// them the 512-byte atlas blocks -- are stored with their EXACT pitches (a grid of 130^3 bricks takes 130^3 blocks, not the 256 x 256 x 130
// that power-of-two pitches cost in round 2), majorant cells with power-of-two pitches.
//   brick record of brick (bx, by, bz):  (bz * nb[1] + by) * nb[0] + bx  as two v_mad_u32_u24 (nb[1] * nb[2] < 2^24, checked at upload)
//   majorant of cell (cx, cy, cz) of mip m:  level_offset(m) + (((cz << (mshift[1] - m)) + cy) << (mshift[0] - m)) + cx,
//     level 0 has 2^mshift[i] >= max(nb[i], 8) cells per axis, every level half of that; cells beyond the real
//     ceil(nb / 2^m) hold 0 (= "outside the grid reads 0"), so the whole padded extent may be indexed
VR_SCENE_HD int32_t ceil_log2(uint32_t v) { int32_t s = 0; while (s < 31 && (1u << s) < v) ++s; return s; }
// offset of level m = S0 * (0, 1, 9/8, 73/64)[m], S0 = 2^k cells on level 0 (k >= 9)
VR_SCENE_HD uint32_t majorant_level_offset(uint32_t k, uint32_t mip) { const uint32_t s = 9u - 3u * mip; return ((0x49u >> s) << s) << (k - 6u); }
VR_SCENE_HD size_t majorant_padded_cells(uint32_t k) { return (size_t)majorant_level_offset(k, 3u) + ((size_t)1 << (k - 9u)); }
// The table has one cell more than its levels, at index majorant_padded_cells(k) (GridView::maj_outside): the answer to "outside the grid" -- what the reference's
// lookup_majorant makes of an out-of-range texelFetch, density_scale * 0, TF-remapped when a LUT is bound (common.glsl:278-281, 425).  A DDA step outside the
// padded box reads that cell instead of selecting 0 after the load (round 5: one clamp, one compare and one select less per step).
VR_SCENE_HD size_t majorant_table_cells(uint32_t k) { return majorant_padded_cells(k) + 1u; }
// Cell (cx, cy, cz) of level `mip` inside its level.  Linear: x fastest with power-of-two pitches.  Blocked (GridView::maj_blocked,
// used for the large tables of dense grids): levels 0 and 1 -- at least 4 cells per axis -- are stored as 4x4x4-cell blocks of 64
// consecutive cells (128 bytes of fp16 = one cache line), so that a DDA step to ANY neighbouring cell usually stays in the line;
// levels 2 and 3 (a few hundred cells) stay linear.  sx, sy: log2 of the level's padded extent in x and y.
// Measured on c4 (64^3 cells): -2 % (profiles/r2j_layout_experiments.txt), so the blocked layout is a build-time experiment
// (-DVR_MAJORANT_BLOCKED=1), not a run-time switch: a run-time flag costs every DDA step a divergent branch.
#ifndef VR_MAJORANT_BLOCKED
#define VR_MAJORANT_BLOCKED 0
#endif
VR_SCENE_HD uint32_t majorant_cell_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t sx, uint32_t sy, uint32_t mip, bool blocked) {
    if (blocked && mip <= 1u)
        return (((((cz >> 2) << (sy - 2u)) + (cy >> 2)) << (sx - 2u)) + (cx >> 2)) * 64u + (((cz & 3u) << 4) | ((cy & 3u) << 2) | (cx & 3u));
    return (((cz << sy) + cy) << sx) + cx;
}

// Atlas block of one brick (round 4): FIVE cache lines of 128 bytes, each starting with the brick's decode range (rmin, rdiff: two floats, the
// same in all five) followed by 120 of the brick's 512 u8 voxels in index order (x & 7) + 8 (y & 7) + 64 (z & 7); the last line holds 32.  A tap
// reads the range and the voxel from ONE line -- until round 3 the range was a second gather, into a table of its own (8 bytes per brick, 16 MiB
// for 1024^3 voxels: on BASELINE configs[4]'s grids a third of the kernel's misses beyond the L2; profiles/r4c_*).  Costs a quarter more atlas.
// Build-time switch (-DVR_BRICK_HEADERS=0: 512-byte blocks and the separate table, as in round 3) for the A/B.
#ifndef VR_BRICK_HEADERS
#define VR_BRICK_HEADERS 1
#endif
constexpr uint32_t kBrickLineVoxels = VR_BRICK_HEADERS ? 120u : 128u, kBrickLineHeader = VR_BRICK_HEADERS ? 8u : 0u;
constexpr uint32_t kBrickBlockBytes = VR_BRICK_HEADERS ? 640u : 512u;
// line (0..4) that holds voxel `off` (0..511) of a brick, and the voxel's byte offset inside the block
VR_SCENE_HD uint32_t brick_voxel_line(uint32_t off) { return VR_BRICK_HEADERS ? (off * 547u) >> 16 : off >> 7; }      // off / 120 for off < 512 (547 / 65536 = 1 / 119.81)
VR_SCENE_HD uint32_t brick_voxel_byte(uint32_t off) { return off + kBrickLineHeader * (brick_voxel_line(off) + 1u); }  // = 128 line + 8 + (off - 120 line)

// Paired atlas blocks (round 4, -DVR_PAIRED_ATLAS=0 switches it off): when a frame's density grid and its emission grid are both brick grids of the same brick
// layout -- BASELINE configs[4] -- the kernel compiled for that case (variant 2) reads ONE atlas in which a brick's block is ten cache lines of
// [rmin_d, rdiff_d, rmin_e, rdiff_e | 56 x (density voxel, emission voxel)]: the emission tap of a tentative collision (a second stochastic tricubic tap around the
// same point) lands in the line the density tap fetched in 38 % of the cases (same brick and the same run of 56 voxels: profiles/r4g_*), instead of always
// fetching a line of its own from a second atlas.  Same bytes per brick as two 640-byte blocks.
#ifndef VR_PAIRED_ATLAS
#define VR_PAIRED_ATLAS 1
#endif
constexpr uint32_t kPairLineVoxels = 56u, kPairLineHeader = 16u, kPairBlockBytes = 1280u;
VR_SCENE_HD uint32_t pair_voxel_line(uint32_t off) { return (off * 1171u) >> 16; }                    // off / 56 for off < 512 (checked for all 512)
// byte of voxel `off` of component c (0 density, 1 emission) inside the paired block
VR_SCENE_HD uint32_t pair_voxel_byte(uint32_t off, uint32_t c) { const uint32_t l = pair_voxel_line(off); return l * 128u + kPairLineHeader + 2u * (off - l * kPairLineVoxels) + c; }

// element index of voxel (x, y, z) in the blocked dense layout (see GridView::dense)
VR_SCENE_HD size_t dense_blocked_index(uint32_t x, uint32_t y, uint32_t z, uint32_t blocks_x, uint32_t blocks_y) {
    return (((size_t)(z >> 2) * blocks_y + (y >> 2)) * blocks_x + (x >> 2)) * 64u + (((z & 3u) << 4) | ((y & 3u) << 2) | (x & 3u));
}

// Records of the environment arrays are packed to 12 bytes (no alpha / padding word): 10.7 instead of 8 records per cache line.
// The kernel is bound by the L1 misses a CU can keep in flight; these gathers are a quarter of them.
#ifndef VR_ENV_TEXEL_FLOATS
#define VR_ENV_TEXEL_FLOATS 3
#endif
constexpr int32_t kEnvTexelFloats = VR_ENV_TEXEL_FLOATS;      // envmap texel: RGB (3) or RGBA (4) floats
// Warp table of sample_environment: one record (d, e0, e1) per 2x2 block of every importance-pyramid level (level k = 0 .. top has
// 2^k x 2^k records; top = base mip - 1).  The descent reads one record per level, each chosen by the previous one: nine
// dependent gathers for the 512^2 map.  Records of TWO consecutive levels share a block -- the parent's record followed by those
// of its four children (child index 2 * (y & 1) + (x & 1)) -- so the second gather of a pair hits the line the first one brought
// in.  Levels are paired from the finest up, (top-1, top), (top-3, top-2), ...; with an odd number of levels, level 0 (one
// record) sits alone in block 0.  Pair p (upper level ku = 2p + s, s = 1 if level 0 is alone) starts at block
// s + 4^s (16^p - 1) / 15 and holds one block per cell of its upper level, x fastest.  Blocks are 64 bytes,
// [parent | c0 | c1 | c2 | c3 | pad]; the LAST pair's are 128 bytes, one cache line, [parent | pad | c0 .. c3] with 28-byte child
// records (d, e0, e1, w0, w1, w2, w3): the finest level also carries the four importance texels its cell splits into, which is the
// value the pdf needs (imp_fetch(pos, 0)) -- one gather into a 1 MiB map less per next-event estimate.
constexpr int32_t kEnvCdfBlockFloats = 16, kEnvCdfLastBlockFloats = 32, kEnvCdfLastChildFloats = 7, kEnvCdfLastChild0 = 4;
VR_SCENE_HD size_t env_cdf_pair_base(int32_t s, int32_t p) { return (size_t)s + (((size_t)1 << (2 * s)) * ((((size_t)1) << (4 * p)) - 1)) / 15; }
// float offset of the last pair's first block: after the 64-byte blocks, on a 128-byte boundary
VR_SCENE_HD size_t env_cdf_last_pair_floats(int32_t s, int32_t n_pairs) {
    const size_t f = (size_t)kEnvCdfBlockFloats * env_cdf_pair_base(s, n_pairs - 1);
    return (f + (size_t)kEnvCdfLastBlockFloats - 1u) / (size_t)kEnvCdfLastBlockFloats * (size_t)kEnvCdfLastBlockFloats;
}
VR_SCENE_HD size_t env_cdf_index(int32_t top, int32_t k, uint32_t x, uint32_t y) {      // float index of the record of cell (x, y) of level k
    const int32_t s = (top & 1) ? 0 : 1;
    if (k < s) return 0;
    const int32_t kk = k - s, p = kk >> 1, ku = 2 * p + s, n_pairs = (top + 1 - s) / 2;
    const uint32_t c = 2u * (y & 1u) + (x & 1u);
    if (p == n_pairs - 1) {
        const size_t base = env_cdf_last_pair_floats(s, n_pairs);
        if (kk & 1) return base + (size_t)kEnvCdfLastBlockFloats * (((size_t)(y >> 1) << ku) + (x >> 1)) + (size_t)kEnvCdfLastChild0 + (size_t)kEnvCdfLastChildFloats * c;
        return base + (size_t)kEnvCdfLastBlockFloats * (((size_t)y << ku) + x);
    }
    const size_t blk = env_cdf_pair_base(s, p);
    if (kk & 1) return (size_t)kEnvCdfBlockFloats * (blk + ((size_t)(y >> 1) << ku) + (x >> 1)) + 3u + 3u * c;
    return (size_t)kEnvCdfBlockFloats * (blk + ((size_t)y << ku) + x);
}
VR_SCENE_HD size_t env_cdf_table_floats(int32_t top) {
    if (top < 1) return kEnvCdfBlockFloats;                      // no level, or level 0 alone (its record carries the texels: 7 floats)
    const int32_t s = (top & 1) ? 0 : 1, n_pairs = (top + 1 - s) / 2, ku = 2 * (n_pairs - 1) + s;
    return env_cdf_last_pair_floats(s, n_pairs) + ((size_t)kEnvCdfLastBlockFloats << (2 * ku));
}

struct GridView {
    const BrickRec* bricks;      // nb[0] * nb[1] * nb[2] records, x fastest (see above)
    const uint8_t* atlas;        // one kBrickBlockBytes block per brick record (same index): 5 lines of [rmin, rdiff | 120 voxels], voxel (x&7) + 8*(y&7) + 64*(z&7) (brick_voxel_byte)
    const float* majorant;       // all mips, padded (see above): "effective" majorant = density_scale * range.y, TF-remapped when a LUT is bound
    const uint16_t* majorant16;  // the same cells as raw fp16 range.y (0 outside): what the kernels WITHOUT a transfer function read --
                                 // density_scale * half2float(.) is one multiply, and the table is half as many cache lines
    const float* atlas_f32;      // optional: the atlas decoded to float (rmin + unorm8(b) * rdiff, 2 KiB per brick), built for transfer-function
                                 // renders -- their 8 corner taps then cost one 4-byte load each instead of record + byte; nullptr otherwise
    const float* rng;            // (rmin, rdiff) per brick record, 8 bytes, same index as `bricks`: what decode_atlas_kernel reads (a tap reads the copy in its voxel's line)
    int32_t nb[3];               // bricks per axis (mip 0); mip m has ceil(nb / 2^m) cells per axis
    int32_t mshift[3];           // log2 of the padded level-0 majorant extent per axis (each >= 3)
    float mlim[3];               // the same extent in voxels, (float)(8 << mshift[i]): the inside test of the DDA compares against it
    int32_t n_mips;              // range mips available above level 0 (reference: 3)
    int32_t maj_blocked;         // majorant levels 0 and 1 in 4x4x4-cell blocks (majorant_cell_index): set for dense grids
    int32_t maj_outside;         // index of the table's last cell, which holds the majorant of "outside the grid" (majorant_table_cells)
    const uint16_t* dense;       // dense fp16 voxels in 4x4x4 blocks of 128 contiguous bytes (one cache line): block (x>>2, y>>2, z>>2),
                                 // x fastest over dblk[0] x dblk[1] x ceil(dim.z/4) blocks, voxel (x&3) + 4*(y&3) + 16*(z&3) inside; or nullptr
    int32_t dim[3];              // voxel extent of the dense grid
    int32_t dblk[2];             // blocks per axis (x, y) = ceil(dim / 4)
};

struct Uniforms {                // names follow the GLSL uniforms
    int32_t bounces, seed, show_environment;
    float cam_pos[3], cam_fov, cam_transform[9];
    float vol_bb_min[3], vol_bb_max[3];
    float vol_minorant, vol_majorant, vol_inv_majorant;
    float vol_albedo[3], vol_phase_g, vol_density_scale, vol_emission_scale, vol_emission_norm;
    float vol_density_transform[16], vol_density_inv_transform[16];
    float vol_emission_transform[16], vol_emission_inv_transform[16];
    uint32_t tf_size; float tf_window_left, tf_window_width;
    float env_transform[9], env_inv_transform[9], env_strength, env_imp_inv_dim[2];
    int32_t env_imp_base_mip;
    int32_t resolution[2];
    int32_t use_tf, has_emission, integrator;
};

struct SceneParams {
    Uniforms u;
    GridView density;
    GridView emission;
    float emission_from_density[16];   // vol_emission_inv_transform * vol_density_transform (common.glsl:325)
    int32_t paired;                    // host side only (which kernel variant serves the scene): density.atlas and emission.atlas are ONE paired atlas (kPairBlockBytes per brick)
    const float* tf_lut;               // tf_size x vec4 (std430 SSBO binding 4)
    const float* envmap;               // env_w*env_h texels of kEnvTexelFloats floats (RGB), row 0 = v~0
    const uint32_t* env_rgbe;          // round 6: the same texels as r | g << 8 | b << 16 | e << 24 (value = float(m) * 2^(e - 136), 0 for e = 0) when EVERY texel of the map
                                       // is exactly such a number -- a Radiance .hdr file's always are --, else null: a quarter of the bytes behind every texel fetch
                                       // (the 1024 x 512 map: 2 MiB instead of 8, smaller than an XCD's L2); env_texture decodes to the same floats, bit for bit
    int32_t env_w, env_h;
    float env_avg_w;                   // round 6: impmap's coarsest value, imp_fetch(0, 0, base mip) -- the divisor of every light sample's pdf and MIS weight -- as an argument
    int32_t env_avg_w_set;             // (1: env_avg_w is that value; 0 -- a SceneParams filled by other code: the lane code fetches it)
    const float* impmap;               // importance pyramid, level 0 (dim^2) first, 2x2 box mips after it
    int32_t imp_dim;
    const float* env_cdf;              // warp table: (d, e0, e1) per 2x2 block of every pyramid level, two levels per block; the finest with its texels (env_cdf_index)
    int32_t env_div_safe;              // every d, e0, e1 of the table is NaN, 0 or in [2^-76, 1] (checked when the table is built): the warp's quotients may use div_core --
                                       // which the kernels compiled for one scene kind do unconditionally: any other environment is rendered by the run-time variant
    float cam_z;                       // -.5f / tan(.5f * M_PI * cam_fov / 180.f), common.glsl:78 (uniform per frame)
};

}  // namespace vr
