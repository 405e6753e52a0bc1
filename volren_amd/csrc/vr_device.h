// vr_device.h -- host-callable launchers of the HIP kernels in vr_kernels.hip (all asynchronous on `stream`).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vr_scene.h"

namespace vr {

// Path tracing: runs samples first_sample .. first_sample+n_samples-1 (1-based, the reference's current_sample)
// for every pixel of the listed 16x16 tiles (kernel 1: per-sample radiances into `sample_pool`) and folds them in
// sample order into the RGBA32F running mean `fb` (kernel 2; W*H texels, row 0 at the bottom).
// tiles == nullptr: all tiles of the frame (n_tiles = their count).  sample_pool must hold
// pathtrace_pool_floats(n_tiles, n_samples) floats; unit_counter is 8 device words (the work queue heads, one per XCD segment).  status[0] is set non-zero if a wavefront trips the watchdog.
// Tuning state of ONE renderer (nothing about a launch is process-global: two renderers, on one device or two, never share it).
struct PathtraceTuning {
    // scheduler thresholds, indexed like LaneState (vr_trace.h): NEW (free slots that trigger a NEW batch), [1] diagnostic cap on the slots in use
    // (0 = all), MARCH (= low-water mark of live paths: below it every non-empty batch runs), COLLIDE (lanes that must stand at a tentative
    // collision before the collision code runs while others still march; 0 = per kernel), NEE, POSTNEE, ESCAPE (batch sizes that trigger the event)
    int32_t thr[8] = { 64, 0, 56, 0, 60, 60, 64, 0 };
    unsigned long long* stats = nullptr;      // device buffer of 32 counters: the launch uses the instrumented (STATS) kernels; or null
    int32_t samples_per_unit = 0;             // samples of a work unit; 0 = per kernel variant
    int32_t blocks_per_cu = 0;                // resident workgroups per CU; 0 = from the occupancy query (cached per device)
};
PathtraceTuning default_tuning();             // the defaults, with the diagnostic overrides VR_SPU / VR_BLOCKS_PER_CU of the environment (read once)
size_t pathtrace_pool_floats(const PathtraceTuning& T, int32_t n_tiles, int32_t n_samples);
size_t pathtrace_workspace_floats();      // cold path state of all resident wavefronts
void launch_pathtrace(const PathtraceTuning& T, const SceneParams& P, float* fb, float* sample_pool, float* workspace, uint32_t* unit_counter, const int32_t* tiles, int32_t n_tiles,
                      int32_t first_sample, int32_t n_samples, uint32_t* status, hipStream_t stream, bool fast_math = false,
                      hipEvent_t ev_kernel_begin = nullptr, hipEvent_t ev_kernel_end = nullptr);      // optional: bracket the path-tracing kernel alone
// which compiled kernel variant (vr_pathtrace.hip: 0 bricks, 1 dense fp16, 2 / 4 bricks + emission grid, 3 everything at run time) serves a scene, and -- *why, a mask --
// what sent it to the run-time variant (0: nothing, the scene has a kernel of its own kind)
enum PathtraceVariantReason : int {
    VR_VARIANT_INTEGRATOR = 1,        // a global-majorant / ray-marching integrator was asked for
    VR_VARIANT_ENV_DIVISION = 2,      // the environment's warp table failed env_cdf_kernel's check (thresholds below 2^-76: vr_math.h div_core does not apply)
    VR_VARIANT_DENSITY_SCALE = 4,     // density scale outside [2^-16, 2^24] (the clean march divides by majorants without rescaling)
    VR_VARIANT_GRID_FORMS = 8         // emission grid with a dense grid on either side, or brick grids of different layouts (no paired atlas)
};
int pathtrace_variant_of(const SceneParams& P, int* why);
// fast_math: the opt-in tolerance-mode kernels (hardware transcendentals, reciprocal division; vr_math.h VR_FAST_MATH); the default
// kernels are bit-identical to the CPU oracle

// env_setup.glsl:18-34 + glGenerateMipmap (environment.cpp:27-31): importance pyramid of a dim x dim map
void launch_build_impmap(const float* envmap_rgba, int32_t env_w, int32_t env_h, int32_t dim, float* pyramid, hipStream_t stream);

// warp table of sample_environment: float4 per 2x2 block of every pyramid level (coarsest first); (dim^2 - 1) / 3 records
void launch_build_env_cdf(const float* pyramid, int32_t dim, float* table, uint32_t* unsafe_flag, hipStream_t stream);

// effective majorant of every cell of every range mip:
//   m = density_scale * float(range.y);  with a LUT: m = vol_majorant * tf_lookup(m * vol_inv_majorant).a
// (common.glsl:278-281, 425, 472)
void launch_majorants(const SceneParams& P, const uint32_t* range_words_all_mips, const int32_t nb[3], const int32_t mip_off[4], int32_t n_mips,
                      const int32_t mshift[3], float* out_padded, uint16_t* out16_padded, hipStream_t stream);      // out16: the raw fp16 range maxima, same layout

// Dense -> brick encoder on the device (Volume::to_brick_grid / commit(), src/renderer.cpp:63); see vr_kernels.hip.
// ranges: range[nb] (fp16x2 words), flag[nb] (range is not a single value: the voxels matter)
void launch_encode_ranges(const float* dense, const int32_t dim[3], const int32_t nb[3], uint32_t* range, uint32_t* flag, hipStream_t stream);
void launch_encode_bricks(const float* dense, const int32_t dim[3], const int32_t nb[3], const uint32_t* range, const uint32_t* flag,
                          BrickRec* recs, float* rng, uint8_t* atlas, hipStream_t stream);      // rng: compact (rmin, rdiff) pairs, same index as recs
void launch_range_mip(const uint32_t* src, const int32_t sdim[3], uint32_t* dst, const int32_t ddim[3], hipStream_t stream);

// paired atlas of a density and an emission brick grid with the same brick layout (vr_scene.h kPairBlockBytes per brick); needs VR_BRICK_HEADERS blocks as input
void launch_pair_atlas(const uint8_t* atlas_density, const uint8_t* atlas_emission, uint8_t* out, size_t n_records, hipStream_t stream);

// decoded float atlas (one float per atlas byte: rmin + unorm8(b) * rdiff of its brick), used by transfer-function renders
void launch_decode_atlas(const float* rng, const uint8_t* atlas, float* out, size_t n_records, hipStream_t stream);

// tonemap.glsl:29-36 in place
void launch_tonemap(float* fb, int32_t w, int32_t h, float exposure, float gamma, hipStream_t stream);

// multi-GPU shard helpers: copy owned 16x16 tiles frame <-> compact tile-major buffer (256 texels per tile)
void launch_pack_tiles(const float* fb, int32_t w, int32_t h, const int32_t* tiles, int32_t n_tiles, float* packed, hipStream_t stream);
void launch_unpack_tiles(const float* packed, const int32_t* tiles, int32_t n_tiles, float* fb, int32_t w, int32_t h, hipStream_t stream);

// unit-test probe: out[i] = f(a[i], b[i]) with the device build of vr_math.h
// fn: 0 log 1 sin 2 cos 3 tan 4 acos 5 atan2 6 exp 7 pow 8 asin 9 a/b 10 sqrt 11 fma(a,b,a) 12 float(u8)/255 13 sincos 14 a*b+a 15 half->float
//     16 rcp_exact(a) 17 rcp3_exact((a, b, a)).y
void launch_math_probe(int32_t fn, const float* a, const float* b, float* out, int32_t n, hipStream_t stream);

}  // namespace vr
