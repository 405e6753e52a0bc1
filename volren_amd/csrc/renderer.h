// renderer.h -- RendererHIP: the MI355X drop-in for the reference's RendererOpenGL (src/renderer.h:16-63).
// Same public fields, same call protocol (mutate fields -> commit() after changing the volume -> reset() ->
// trace() once per sample, result = running mean in `color`, RGBA32F, row 0 at the bottom).  Differences that
// are visible to a caller are additions only:
//   * render(spp): all remaining samples in ONE fused launch (what bindings.cpp:124-132 loops over trace());
//   * trace() still advances `sample` by exactly one, but consecutive trace() calls on an unchanged scene are COALESCED
//     (round 5): the call records what it would launch -- a byte snapshot of every launch input -- and a following
//     trace() that finds the same bytes only adds to a count; the samples go out as ONE fused launch at the next
//     point where anyone could observe or change the frame (flush_pending()).  The reference's loop
//     `while (sample < sppx) trace();` (src/main.cpp:533-537, src/bindings.cpp:124-132) therefore runs at render(n)
//     speed, with the same bits,
//   * the camera and the resolution are explicit members instead of cppgl globals
//     (current_camera(), Context::resolution(): renderer.cpp:47,93-95,137),
//   * set_tiles(): restrict a renderer to a subset of 16x16 framebuffer tiles (multi-GPU sharding).
#pragma once

#include <memory>
#include <vector>

#include "devmem.h"
#include "environment.h"
#include "grids.h"
#include "transferfunc.h"
#include "vr_device.h"
#include "vr_scene.h"

namespace vr {

// stand-in for cppgl's camera (pos/dir/up/fov_degree are what the renderer reads: renderer.cpp:93-95)
struct Camera {
    vec3 pos{ 1.f, 0.f, 1.f };                 // main.cpp:458
    vec3 dir = normalize(-pos);                // main.cpp:459 (note dir.y = -0.f, as in the reference)
    vec3 up{ 0.f, 1.f, 0.f };
    float fov_degree = 70.f;                   // cppgl default (unverified): always pass --cam_fov
    mat3 view_inverse() const;                 // inverse(mat3(lookAt(pos, pos+dir, up))): columns right, up, -forward
};

// replaces BrickGridGL (renderer.h:9-14): the three textures become three device arrays (+ majorant cache)
struct BrickGridHIP {
    DeviceBufferPtr bricks;        // BrickRec per brick
    DeviceBufferPtr atlas;         // brick-major u8 voxels, 512 B per slot
    DeviceBufferPtr range_words;   // fp16x2 range of every cell of mips 0..n_mips (input of the majorant kernel)
    DeviceBufferPtr majorant;      // effective majorants (float), padded power-of-two layout (vr_scene.h)
    DeviceBufferPtr majorant16;    // raw fp16 range maxima in the same layout (read by the kernels without a transfer function)
    DeviceBufferPtr rng;           // compact (rmin, rdiff) float pairs, same index as `bricks` (what a tap reads)
    DeviceBufferPtr atlas_f32;     // decoded float atlas, built on the first render with a transfer function (4x the atlas; dropped by commit())
    bool atlas_f32_failed = false; // its allocation failed once: not retried until commit() or a tf_float_atlas toggle (the byte atlas serves)
    DeviceBufferPtr atlas_paired;  // this grid's voxels interleaved with those of the frame's other grid (density + emission grids of one brick layout: vr_scene.h);
                                   // one buffer, held by both grids of the frame; built by commit()
    DeviceBufferPtr dense;         // dense fp16 voxels in 4x4x4 blocks (DenseGridF16), then bricks/atlas are empty
    int32_t dim[3] = { 0, 0, 0 };
    int32_t dblk[2] = { 0, 0 };            // 4x4x4 blocks per axis (x, y) of the dense layout
    int32_t nb[3] = { 0, 0, 0 };
    int32_t mip_off[4] = { 0, 0, 0, 0 };   // word offset of each level inside range_words (compact)
    int32_t n_mips = 0;
    int32_t n_cells = 0;                   // words in range_words
    int32_t mshift[3] = { 3, 3, 3 };       // padded power-of-two extent of `majorant` (vr_scene.h)
    size_t n_active = 0;                   // bricks whose voxels matter (range not a single value): what the choice of the majorant layout looks at
    bool maj_blocked = false;              // commit()'s choice for this grid: majorant levels 0-1 in 4x4x4-cell blocks (vr_scene.h majorant_cell_index)
    mat4 transform;
};

struct RendererHIP {
    // Renderer interface
    void init();
    void resize(uint32_t w, uint32_t h);
    void commit();
    void trace();
    void draw();
    void reset();

    // all of `n` further samples in one launch (n <= 0: up to sppx)
    void render(int n = 0);

    // helper to convert brick grid to device arrays
    BrickGridHIP brick_grid_to_device(const std::shared_ptr<BrickGrid>& grid);
    BrickGridHIP dense_grid_to_device(const std::shared_ptr<DenseGridF16>& grid);
    BrickGridHIP grid_to_device(const Volume::GridPtr& grid);      // dense fp16 stays dense, everything else becomes bricks
    BrickGridHIP dense_to_bricks_on_device(const std::shared_ptr<DenseGrid>& grid);   // to_brick_grid + upload, all on the GPU
    bool gpu_encoder = true;                                       // DenseGrid -> bricks on the device (false: host encoder)
    void grid_checksums(const BrickGridHIP& g, uint64_t out[3]) const;   // FNV-1a of bricks / atlas / range words (tests)
    // scale and move volume to fit into [-0.5, 0.5] unit cube
    void scale_and_move_to_unit_cube();

    // General settings
    int sample = 0;
    int sppx = 1024;
    int seed = 42;
    int bounces = 100;
    float tonemap_exposure = 5.f;
    float tonemap_gamma = 2.2f;
    bool tonemapping = true;
    bool show_environment = true;

    // Volume settings
    vec3 albedo = vec3(0.9f);           // volume albedo
    float phase = 0.f;                  // volume phase (henyey-greenstein g parameter)
    float density_scale = 1.f;          // volume density scaling factor
    float emission_scale = 100.f;       // volume emission scaling factor

    // device data
    DeviceBufferPtr color;              // RGBA32F running mean, W*H texels, row 0 = bottom
    DeviceBufferPtr display;            // tonemapped copy written by draw()
    std::vector<BrickGridHIP> density_grids;
    std::vector<BrickGridHIP> emission_grids;
    float majorant_emission = 0.f;

    // Volume data
    std::shared_ptr<Volume> volume;

    // Volume clip planes
    vec3 vol_clip_min = vec3(0.f);
    vec3 vol_clip_max = vec3(1.f);

    // Scene data
    std::shared_ptr<Environment> environment;
    std::shared_ptr<TransferFunction> transferfunc;

    // ---- additions ----
    Camera camera;
    ivec2 resolution{ 0, 0 };
    hipStream_t stream = nullptr;
    int integrator = 0;                               // 0: DDA tracking (both reference kernels), 1: global-majorant tracking (common.glsl:333-394),
                                                      // 2: direct volume rendering (:571-591, needs a LUT), 3: 64-step ray-marching trackers (:506-566)
    bool tf_float_atlas = true;                       // transfer-function renders decode the brick atlas to floats once (4x its size): one load per corner tap
    int order_tiles = 1;                              // a launch works through its tiles costliest first (chord of the pixel rays through the volume's box), so that what
                                                      // is left when the work queue runs empty are short paths (profiles/r4f_*): 0 never, 1 when the renderer has a tile
                                                      // subset (a rank's share: +0.5 ... +7 %), 2 always (full frames measure +-0.5 %: raster order stays their default).
                                                      // Which tile runs when never changes a result
    int majorant_layout = -1;                         // layout of the majorant table's levels 0-1 for the frames the two-brick-grid kernel serves: -1 = per grid, chosen at
                                                      // commit() (blocked when more than kBlockedMajorantBricks bricks carry voxels: the large, well filled sparse grids of
                                                      // BASELINE configs[4], +3 % there, -2 % on small or thinly filled ones: profiles/r4d_*, r5_*), 0 = linear, 1 = blocked.
                                                      // Results never depend on it
    bool fast_math = false;                           // opt-in tolerance mode: hardware log/sin/cos/rcp instead of the specified arithmetic
                                                      // (not bit-reproducible; without a transfer function within 1e-3 relative L2 of the default --
                                                      // with one bound the renderer refuses it: DESIGN.md 3)
    PathtraceTuning tuning = default_tuning();        // scheduler thresholds, work-unit size, statistics buffer of THIS renderer's launches
    int last_launches = 0;                            // path-tracing sub-launches of the last trace()/render()
    int launch_target_ms = 2000;                      // a sub-launch is planned to take at most this long, from the rate the renderer measured on its last launch (a short
                                                      // probe launch when it has none for the current settings and the request is large); 0 = plan by the sample pool alone.
                                                      // Correctness does not depend on it (the kernel's watchdog is progress-based): it bounds how long one launch holds the GPU
    size_t sample_pool_bytes = (size_t)64 << 30;      // HBM budget of the per-sample radiance pool (16 B per pixel-sample; sized for 288 GB HBM3E: 64 GiB = the 2^32 items a sub-launch
                                                      // can index; allocated on demand, only as large as a launch needs, halved when it does not fit the free memory).  Round 5: 16 -> 64 GiB,
                                                      // a 2048^2 x 4096-spp frame is 5 sub-launches instead of 16 and each one's drain (4-8 ms) is paid that much less often: c5full +2.2 %,
                                                      // c4 at 1920x1080x4096 +0.9 % (tests/tools_pool_ab.py)

    void set_tiles(const std::vector<int32_t>& tile_ids);     // empty = whole frame
    void fill_params(SceneParams& P);                          // renderer.cpp:88-138
    void download(float* rgba);                                // color -> host
    void download_display(float* rgba) const;
    void synchronize();
    // Launches the samples that coalesced trace() calls have recorded (no-op without any).  Every member function that reads or replaces the
    // framebuffer, the device grids, the tile set or the timing state calls it first (render, draw, download, synchronize, commit, resize,
    // set_tiles, last_*_ms, sched_stats, watchdog_status, ...); a caller that reads `color` through its raw device pointer calls it itself
    // (the C ABI does: vr_framebuffer_device, vr_pack_tiles, vr_unpack_tiles, vr_set_stream).  Changes of PUBLIC FIELDS between two trace()
    // calls need no flush by the caller: the next trace() sees bytes that differ from the recorded ones and launches the recorded samples first,
    // with the values they were recorded with.
    void flush_pending();
    int pending_samples() const { return pending_n_; }         // samples recorded by trace() and not launched yet
    bool coalesce_trace = true;                                // false: every trace() is its own launch (round 4's behaviour; A/B and tests)
    double last_kernel_ms();                                    // HIP-event time of the last launch (a render(), or the trace() calls coalesced into one): all sub-launches, path tracing + accumulation (waits for it)
    double last_pathtrace_ms();                                 // HIP-event time of the path-tracing kernel alone, summed over the sub-launches of the last trace()/render()
                                                                // (0 when that call launched none: integrators 2 / 3)
    void sched_stats(bool enable, unsigned long long out[32]);
    void wave_timeline(unsigned long long* out, size_t n_words);  // diagnostics: the per-wavefront (begin, queue empty, end) triples of the last instrumented launch  // diagnostics: out (may be null) receives the counters gathered so far; enable starts (zeroed) or stops counting
    uint32_t watchdog_status();
    ~RendererHIP();

private:
    // majorant cache key
    struct MajKey { float density_scale = -1.f; uint64_t tf_version = ~0ull; float wl = 0, ww = 0; size_t frame = ~(size_t)0; int blocked = -1; };   // tf_version: TransferFunction::version (unique per upload), 0 = no LUT
    // Everything a launch reads from the renderer's mutable state, as one trace()/render() call found it.  `P` is zero-filled before it is written
    // (fill_params), so two snapshots are compared byte by byte; the handles keep alive what P points into (a caller may replace the environment or
    // re-upload the LUT between two trace() calls: the recorded samples still see the old arrays, as the reference's already issued dispatches do).
    struct LaunchInputs {
        SceneParams P;
        MajKey maj;
        size_t frame = 0;
        PathtraceTuning tuning;
        int order_tiles = 0, launch_target_ms = 0, fast_math = 0;
        size_t sample_pool_bytes = 0;
        hipStream_t stream = nullptr;
        std::shared_ptr<Environment> env;
        std::shared_ptr<TransferFunction> tf;
        DeviceBufferPtr keep[5];               // envmap, impmap, env_cdf, lut, compact envmap
        bool same_launch_as(const LaunchInputs& o) const;
    };
    void capture(LaunchInputs& in);            // validates, builds the decoded float atlas when a LUT needs it, fills `in` from the current fields
    void submit(const LaunchInputs& in, int first, int n);      // samples first+1 .. first+n
    int samples_per_launch(const LaunchInputs& in, int n_tiles) const;
    LaunchInputs pending_;
    int pending_n_ = 0, pending_first_ = 0, pending_cap_ = 0;
    void update_majorants(const LaunchInputs& in, BrickGridHIP& g);
    std::vector<int32_t> tiles_host_;
    DeviceBufferPtr tiles_dev_;
    // the launch's own order of those tiles: the costliest first (launch(): tile_order)
    DeviceBufferPtr order_dev_;
    uint64_t order_key_ = 0;
    const int32_t* tile_order(const SceneParams& P, int n_tiles);
    DeviceBufferPtr status_;
    DeviceBufferPtr pool_;
    DeviceBufferPtr workspace_;
    DeviceBufferPtr stats_;                            // 32 counters of the instrumented kernels (sched_stats)
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    std::vector<hipEvent_t> pt_events_;                // (begin, end) around the path-tracing kernel of every sub-launch
    size_t pt_events_used_ = 0;
    double last_ms_ = 0.0, last_pathtrace_ms_ = 0.0;
    bool timing_pending_ = false;
    // launch sizing: samples per millisecond of the last finished path-tracing sub-launch, and a fingerprint of the settings it ran with
    double rate_samples_per_ms_ = 0.0;
    uint64_t rate_key_ = 0, rate_pending_key_ = 0;
    double rate_pending_samples_ = 0.0;               // samples of the last sub-launch enqueued (its events: the last pair of pt_events_)
    void harvest_rate(bool wait);
    MajKey maj_key_;
};

using Renderer = RendererHIP;

}  // namespace vr
