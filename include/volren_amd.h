/*
 * volren_amd.h -- C ABI of libvolren_amd.so, the MI355X (gfx950, HIP) drop-in for the offline path-tracing path of
 * nihofm/volren.
 *
 * The reference exposes this path as an in-process C++ object API (struct RendererOpenGL, class Environment,
 * class TransferFunction) that is bound to Python by pybind11 (src/bindings.cpp:64-209).  This header is the same
 * surface flattened to C so that any FFI (ctypes, cgo, JNI, N-API, or a pybind11 module like the reference's
 * `volpy`) can bind it: plain pointers and sizes, int return codes (0 = ok, non-zero = the std::runtime_error the
 * C++ method threw; text via vr_last_error()).  Each entry point names the reference interface it replaces
 * (paths relative to the reference repository).  INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions kept from the reference: framebuffers are RGBA32F with row 0 at the BOTTOM (GL image order); matrices
 * are column-major (glm); `sample` counts completed samples per pixel; the running mean of
 * shader/pathtracer_brick.glsl:36 is what the framebuffer holds.
 *
 * There is no CPU path: every compute entry point fails with VR_ERR_NO_DEVICE when no HIP device is present.
 */
#ifndef VOLREN_AMD_H
#define VOLREN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VR_OK 0
#define VR_ERR 1              /* std::runtime_error / std::out_of_range from the C++ layer */
#define VR_ERR_NO_DEVICE 2
#define VR_ERR_ARG 3

typedef struct vr_renderer vr_renderer;

/* text of the last error on this thread ("" if none) */
const char* vr_last_error(void);
/* library version string */
const char* vr_version(void);
/* number of HIP devices visible (does not initialise a device context) */
int vr_device_count(void);

/* --- lifetime: std::make_shared<RendererOpenGL>() + RendererOpenGL::init()  (src/main.cpp:445-446, src/renderer.cpp:29-50)
 *     device: HIP device ordinal; width/height: Context::resolution() (src/renderer.cpp:47) */
int vr_create(vr_renderer** out, int device, int width, int height);
void vr_destroy(vr_renderer* r);
/* RendererOpenGL::resize (src/renderer.cpp:52-54); clears the framebuffer */
int vr_resize(vr_renderer* r, int width, int height);

/* --- scene loading: load_volume / load_envmap / load_transferfunc of src/main.cpp:37-81 (same side effects:
 *     load_volume sets density_scale=1, scale_and_move_to_unit_cube(), commit(), sample=0; load_envmap resets
 *     transform/strength; load_transferfunc sets show_environment=false) */
int vr_load_volume(vr_renderer* r, const char* path);              /* .brick / .dense / .raw file, or a folder of such frames */
int vr_load_envmap(vr_renderer* r, const char* path);              /* Radiance .hdr */
/* `renderer.volume = Volume(path)` of the pybind11 module (src/bindings.cpp:82,176): replaces the volume and nothing else --
 * no density_scale reset, no unit cube, no commit; the caller runs vr_scale_and_move_to_unit_cube() / vr_commit() in the
 * reference's order (scripts/datagen_colmap.py:57-63, datagen_denoise.py:85-86) */
int vr_set_volume_path(vr_renderer* r, const char* path);
/* voldata::Volume::AABB(name) / minorant_majorant(name) (src/bindings.cpp:91,93; used by scripts/datagen_*.py to place the
 * camera): world-space box, out = min xyz, max xyz */
int vr_volume_aabb(vr_renderer* r, const char* name, float out[6]);
int vr_volume_minorant_majorant(vr_renderer* r, const char* name, float out[2]);
int vr_load_transferfunc(vr_renderer* r, const char* path);        /* "%f, %f, %f, %f" rows */

/* --- scene from memory: voldata::DenseGrid(w,h,d,float*) + Volume(grid) (src/main.cpp:470-472, bindings.cpp Volume ctors);
 *     transform: grid index->model, 16 floats column-major (NULL = identity).  name: "density" | "temperature" | "flame" | "flames".
 *     unit_cube != 0 applies load_volume's density_scale=1 + scale_and_move_to_unit_cube().  Follow with vr_commit(). */
int vr_set_volume_dense(vr_renderer* r, const char* name, const float* voxels, int nx, int ny, int nz, const float* transform, int unit_cube);
/* voldata::Volume::add_grid_frame / update_grid_frame / n_grid_frames (src/bindings.cpp:89-90; voldata is not vendored: call sites
 * src/main.cpp:47, src/renderer.cpp:61-75): append an animation frame holding the dense float grid `name`, or replace grid `name` of frame
 * `frame`.  Follow with vr_commit(); select the frame to render with vr_set_int "grid_frame_counter". */
int vr_volume_add_grid_frame_dense(vr_renderer* r, const char* name, const float* voxels, int nx, int ny, int nz, const float* transform);
int vr_volume_update_grid_frame_dense(vr_renderer* r, int frame, const char* name, const float* voxels, int nx, int ny, int nz, const float* transform);
int vr_volume_n_grid_frames(vr_renderer* r, int* n);
/* dense fp16 grid that STAYS dense on the device (no brick conversion; north_star "dense fp16 grid"): voxels are IEEE
 * binary16, x fastest.  No reference counterpart (the reference bricks every grid in commit(), src/renderer.cpp:63). */
int vr_set_volume_dense_f16(vr_renderer* r, const char* name, const uint16_t* voxels, int nx, int ny, int nz, const float* transform, int unit_cube);
/* voldata::BrickGrid fields as stored in a .brick file; mips may be NULL/0 */
int vr_set_volume_brick(vr_renderer* r, const char* name, const float* transform, const uint32_t n_bricks[3], const float min_maj[2],
                        const uint32_t* indirection, const uint32_t* range, const uint32_t atlas_dim[3], const uint8_t* atlas,
                        int n_mips, const uint32_t* const* mips, const uint32_t (*mip_dims)[3], int unit_cube);
/* Environment(Texture2D) (src/environment.cpp:11-33): float RGB, rows top first */
int vr_set_envmap(vr_renderer* r, const float* rgb, int width, int height);
/* TransferFunction(std::vector<glm::vec4>) (src/transferfunc.cpp:19-22); n = 0 removes the transfer function */
int vr_set_transferfunc(vr_renderer* r, const float* rgba, int n);

/* --- public fields of RendererOpenGL / Environment / TransferFunction / camera (src/renderer.h:30-62, environment.h:20-21,
 *     transferfunc.h:39, src/main.cpp:360-435).  Names: "sample" "sppx" "seed" "bounces" "show_environment" "tonemapping"
 *     "gpu_encoder" (dense grids are bricked on the device, default 1)
 *     "integrator" (0 DDA tracking = both reference kernels, 1 global-majorant tracking = common.glsl:333-394, 2 direct volume rendering
 *     = common.glsl:571-591, needs a transfer function, 3 trace_path around the 64-step ray-marching trackers = common.glsl:506-566)
 *     "fast_math" (0 = the specified, bit-reproducible arithmetic; 1 = opt-in tolerance mode: hardware log/sin/cos/rcp, within 1e-3
 *     relative L2 of the default -- refused with VR_ERR while a transfer function is bound, where it misses that bound)
 *     "tf_float_atlas" (default 1: transfer-function renders of brick grids decode the atlas to floats once, 4x its size; 0 = read the bytes)
 *     "grid_frame_counter"
 *     "sample_pool_mb" (HBM budget of the per-sample radiance pool, 16 .. 65536, default 65536: allocated only as large as a launch needs)
 *     "launch_target_ms" (default 2000: a vr_render is split into sub-launches planned to take at most this long each, from the rate this
 *     renderer measured last -- a short probe launch, one synchronisation, when it has none for the current settings and the request is
 *     large; 0 = split by the sample pool alone.  Results never depend on the split)
 *     "order_tiles" (a launch works through its tiles costliest first -- longest chord of the pixel rays through the volume's box -- so that short
 *     paths are what is left when its work queue runs empty: 0 = never (raster order), 1 = when a tile subset is set (vr_set_tiles / a sharded
 *     renderer's parts; default), 2 = always.  Results never depend on the order) (int);  "tonemap_exposure" "tonemap_gamma" "albedo"(3) "phase" "density_scale"
 *     "emission_scale" "vol_clip_min"(3) "vol_clip_max"(3) "env_strength" "env_transform"(9) "env_rot"(1, degrees about +y,
 *     main.cpp:382) "tf_window_left" "tf_window_width" "cam_pos"(3) "cam_dir"(3) "cam_up"(3) "cam_fov" "volume_transform"(16) (float) */
/* read-only through vr_get_int: "kernel_variant" (the compiled path-tracing kernel the next launch uses: 0 brick grid, 1 dense fp16 grid, 2 / 4 brick grid +
 *     emission grid, 3 everything decided at run time -- correct for every scene, up to an order of magnitude slower) and "kernel_variant_reason" (what sent the
 *     scene to variant 3, a mask: 1 integrator != 0, 2 the environment's warp table has thresholds below 2^-76 ("env_div_safe" = 0), 4 density scale outside
 *     [2^-16, 2^24], 8 emission grid with a dense grid / brick grids of different layouts; 0: the scene has a kernel of its own kind).  Reasons 2 and 4 are also
 *     said once per process on stderr: a caller cannot see them coming;  "env_compact" (1: every texel of the environment map is exactly an RGBE number -- a
 *     Radiance file's always are -- and the path tracer fetches them as one dword each; the values are the float map's, bit for bit) */
int vr_set_int(vr_renderer* r, const char* name, int value);
int vr_get_int(vr_renderer* r, const char* name, int* value);
int vr_set_float(vr_renderer* r, const char* name, const float* values, int count);
int vr_get_float(vr_renderer* r, const char* name, float* values, int count);

/* RendererOpenGL::commit / reset / scale_and_move_to_unit_cube (src/renderer.cpp:56-76,155-157,227-242) */
int vr_commit(vr_renderer* r);
int vr_reset(vr_renderer* r);
int vr_scale_and_move_to_unit_cube(vr_renderer* r);

/* --- the hot path.  vr_trace = RendererOpenGL::trace (src/renderer.cpp:78-145): ONE more sample per pixel.
 *     vr_render = the Python binding's render(spp) loop (src/bindings.cpp:124-132) / the offline loop (src/main.cpp:533-537)
 *     fused into one launch: `spp` more samples per pixel (spp <= 0: up to sppx).  Both are asynchronous on the renderer's
 *     stream; vr_synchronize waits and reports a tripped kernel watchdog as an error.
 *     The reference's own loop `while (sample < sppx) trace();` runs at vr_render's speed (round 5): a vr_trace records the launch
 *     inputs it found ("sample" advances by one, invalid state is reported at the call) and consecutive calls that find the same
 *     bytes are launched TOGETHER -- at the next call of this header that could observe or change the frame (vr_synchronize,
 *     vr_framebuffer*, vr_draw / vr_display / vr_save_png, vr_render, vr_commit, vr_resize, vr_set_tiles, vr_pack_tiles /
 *     vr_unpack_tiles, vr_set_stream, vr_last_*_ms, vr_sched_stats), at a vr_trace that finds changed inputs (any vr_set_*
 *     in between: the recorded samples are launched with the values they were recorded with, as the reference's already issued
 *     dispatches are), or when a full sub-launch has been recorded.  vr_flush launches what has been recorded without waiting;
 *     vr_set_int "coalesce_trace" 0 makes every vr_trace its own launch again; vr_get_int "pending_samples" reads the count. */
int vr_trace(vr_renderer* r);
int vr_flush(vr_renderer* r);
int vr_render(vr_renderer* r, int spp);
int vr_synchronize(vr_renderer* r);
/* duration of the last path-tracing launch in ms, measured with HIP events on the renderer's stream (waits for it) */
int vr_last_kernel_ms(vr_renderer* r, double* ms);
/* duration of the path-tracing kernel alone, summed over the sub-launches of the last vr_trace / vr_render (without the accumulation
 * passes); 0 if that call launched none (integrators 2 and 3 run their own kernels) */
int vr_last_pathtrace_ms(vr_renderer* r, double* ms);

/* --- results.  vr_framebuffer = fbo_data() without the alpha drop (src/bindings.cpp:141-148): W*H*4 floats, row 0 bottom.
 *     vr_draw = RendererOpenGL::draw (src/renderer.cpp:147-153) into a separate tonemapped buffer (shader/tonemap.glsl);
 *     vr_save_png = tonemap + Texture2D::save_ldr of the offline loop (src/main.cpp:540-555): RGBA8 PNG, top row first. */
int vr_framebuffer(vr_renderer* r, float* rgba_out);
int vr_framebuffer_device(vr_renderer* r, void** device_ptr);
int vr_draw(vr_renderer* r);
int vr_display(vr_renderer* r, float* rgba_out);
int vr_save_png(vr_renderer* r, const char* path);

/* --- additions for multi-GPU and measurement (no reference counterpart) */
/* restrict rendering to these 16x16 tiles (raster tile ids, row 0 = bottom); n = 0 -> whole frame */
int vr_set_tiles(vr_renderer* r, const int32_t* tile_ids, int n);
/* use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream */
int vr_set_stream(vr_renderer* r, void* hip_stream);
/* pack the owned tiles of the framebuffer into a compact device buffer (n_tiles*256*4 floats) / scatter a gathered buffer back */
int vr_pack_tiles(vr_renderer* r, const int32_t* tile_ids_device, int n_tiles, void* packed_device);
int vr_unpack_tiles(vr_renderer* r, const int32_t* tile_ids_device, int n_tiles, const void* packed_device);
/* --- ONE frame on SEVERAL devices, in one process (no reference counterpart: the reference drives one GL context, src/main.cpp:524-557;
 *     SURVEY.md 8e; volren_amd/csrc/sharded.h).  vr_sharded_create makes n_parts renderers, part i on HIP device devices[i] (vr_create each).
 *     The scene is REPLICATED by the caller: apply the scene calls of this header (vr_load_volume, vr_set_float, ...) to every
 *     vr_sharded_part(s, i).  vr_sharded_render deals the frame's 16x16 tiles diagonally (owner = (tx + ty) mod n_parts), lets every part
 *     render `spp` more samples of its tiles on its own stream, and gathers the accumulated radiance with ONE collective per frame
 *     (RCCL over xGMI; librccl.so.1 is opened at run time): grouped ncclSend / ncclRecv to part 0 ("gather", the default: only part 0 needs -- and
 *     allocates -- the whole frame), or a grouped ncclAllGather with VR_SHARDED_COLLECTIVE=allgather (rounds 4-5); vr_sharded_collective says which; afterwards part 0's framebuffer (vr_framebuffer / vr_save_png on
 *     vr_sharded_part(s, 0)) holds the whole frame, bit-identical to a single-device render.  Parts that share a device (logical shards:
 *     devices = {0, 0, 0}) exchange their tiles with device-to-device copies instead -- vr_sharded_transport says which: "rccl", "copy", or
 *     "none" for one part; environment VR_SHARDED_TRANSPORT=copy|rccl overrides (rccl also with ONE part: a one-rank communicator).
 *     Asynchronous like vr_render; vr_sharded_synchronize waits for every part and reports a tripped watchdog.  The parts belong to the
 *     sharded renderer: never vr_destroy one, never vr_set_stream / vr_set_tiles on one. */
typedef struct vr_sharded vr_sharded;
int vr_sharded_create(vr_sharded** out, const int* devices, int n_parts, int width, int height);
void vr_sharded_destroy(vr_sharded* s);
int vr_sharded_parts(vr_sharded* s);
vr_renderer* vr_sharded_part(vr_sharded* s, int i);
const char* vr_sharded_transport(vr_sharded* s);
const char* vr_sharded_collective(vr_sharded* s);         /* "gather" | "allgather": what the rccl transport runs per frame */
int vr_sharded_reset(vr_sharded* s);                      /* vr_reset on every part */
int vr_sharded_render(vr_sharded* s, int spp);
int vr_sharded_synchronize(vr_sharded* s);
/* the tile deal itself (host only, needs no device): owner_out[t] = the part (0 .. n_parts-1) that renders raster tile t of a width x height frame,
 * t = ty * ceil(width / 16) + tx, row 0 = bottom; n_tiles must be ceil(width / 16) * ceil(height / 16) */
int vr_tile_owners(int width, int height, int n_parts, int32_t* owner_out, int n_tiles);
/* the uniform block the next launch would use (struct vr::Uniforms of volren_amd/csrc/vr_scene.h, `bytes` must match) */
int vr_get_uniforms(vr_renderer* r, void* out, int bytes);
int vr_uniforms_size(void);
/* importance pyramid of the current environment (floats, level 0 first); count from vr_impmap_floats */
int vr_impmap_floats(vr_renderer* r);
int vr_get_impmap(vr_renderer* r, float* out, int count);
/* scheduler thresholds of THIS renderer's path-tracing launches (8 ints, vr::PathtraceTuning::thr in volren_amd/csrc/vr_device.h);
 * tuning state is per renderer: two renderers in one process, on one device or two, never share it */
int vr_set_sched(vr_renderer* r, const int32_t thresholds[8]);
/* FNV-1a checksums of the committed density grid's device arrays: [0] brick records, [1] atlas, [2] range words (tests: the
 * device encoder and the host encoder must agree) */
int vr_grid_checksums(vr_renderer* r, uint64_t out[3]);
/* scheduler statistics of this renderer's launches (diagnostics): enable != 0 starts counting (instrumented kernels); out (32 x uint64,
 * may be NULL) receives, per state, [block executions, active lanes], then [16] wave iterations, [17] waves, [18..24] cycles per
 * state, [25] summed wave lifetime, [26..31] summed pool occupancy */
int vr_sched_stats(vr_renderer* r, int enable, unsigned long long* out);
/* diagnostics: after an instrumented launch (vr_sched_stats enable), out[3 i .. 3 i + 2] = when wavefront i started, found the work queue empty and ended
 * (ticks of the device's constant 100 MHz clock; 0 = no such wavefront); n_words <= 3 * 8192 */
int vr_wave_timeline(vr_renderer* r, unsigned long long* out, int n_words);
/* test hook: device allocations above `mb` MiB fail as if the device were out of memory (the fall-back paths can then be exercised on a
 * shared GPU); mb < 0 removes the cap.  Initial value: environment variable VR_TEST_MAX_ALLOC_MB, read once per process. */
int vr_test_alloc_cap_mb(long long mb);
/* unit-test probe of the device math (volren_amd/csrc/vr_math.h): host arrays in/out */
int vr_math_probe(int fn, const float* a, const float* b, float* out, int n);
/* voldata::Volume::to_brick_grid + BrickGrid serialisation: encode a dense float grid (x fastest) and write it as a .brick
 * container (SURVEY.md 2.3 layout); transform may be NULL (identity).  Host only, needs no device. */
int vr_write_brick_from_dense(const float* voxels, int nx, int ny, int nz, const float* transform, const char* path);
/* writes this build's ".dense" container (the serialized dense grid main.cpp:44 can load; voldata's own layout is not
 * vendored): u8 voxels, x fastest, value = lo + u8 / 255 * (hi - lo) */
int vr_write_dense(const uint8_t* voxels, int nx, int ny, int nz, float lo, float hi, const float* transform, const char* path);
/* host-side helpers exposed for tests: dense->brick encoder statistics */
int vr_encode_dense_stats(const float* voxels, int nx, int ny, int nz, uint32_t n_bricks_out[3], uint64_t* brick_counter, float min_maj_out[2]);

#ifdef __cplusplus
}
#endif
#endif /* VOLREN_AMD_H */
