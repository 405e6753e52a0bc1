/*
 * oracle/volren_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of nihofm/volren's ray-marching hot path
 * (shader/pathtracer_brick*.glsl + shader/common.glsl) and of the host steps on
 * either side of it (src/renderer.cpp, src/environment.cpp, src/transferfunc.cpp,
 * shader/env_setup.glsl, shader/tonemap.glsl).  Every function cites the reference
 * file:line it follows.
 *
 * PARITY PINNED AGAINST OUTPUTS OF THE REFERENCE ITSELF, RUN HERE.  The reference has no
 * tests or golden vectors and its host program cannot be built (cppgl/voldata are not
 * vendored), but its hot path is GLSL, and Mesa's software rasteriser is in the image:
 * oracle/glref runs the reference's own shader files (read from /root/reference/shader)
 * on llvmpipe, tests/golden/make_golden_glsl.py stores the outputs (tests/golden/
 * glsl_golden.npz) and tests/test_glsl_pin.py checks this oracle against them: whole
 * renders (path tracer with and without transfer function, emission) to ~1e-7 relative L2
 * for the unfused build of this file (llvmpipe never fuses multiply-add), >= 99.5 % of
 * the pixels to 1e-5 for the standard build; tea/rng, majorant fetches and a complete
 * transmittanceDDA segment bit for bit.  In addition: structural known-answers on the
 * reference's data files (tests/golden/known_answers.json) and the reference's only
 * output artefact imgs/example.jpg (tests/golden/example_64.npy).  DESIGN.md 2 has the
 * details, including the three points GLSL/GL leave to the driver (precision of
 * log/acos/atan, multiply-add contraction, the generic GL_COMPRESSED_RED atlas format).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (volren_amd/) never includes, links or calls it.
 */
#ifndef VOLREN_ORACLE_H
#define VOLREN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* voldata::BrickGrid as stored in a .brick file */
typedef struct {
    float    transform[16];     /* glm::mat4, column-major */
    uint32_t n_bricks[3];
    float    min_maj[2];
    uint64_t brick_counter;
    uint32_t* indirection;      /* n_bricks.x*y*z words, GL_RGB10_A2UI packing */
    uint32_t* range;            /* 2 x fp16: low half = min, high half = max */
    uint32_t atlas_dim[3];
    uint8_t* atlas;
    uint32_t n_mips;
    uint32_t mip_dim[8][3];
    uint32_t* mips[8];          /* range mip levels 1..n_mips */
    /* build-only extension (no reference counterpart): dense fp16 voxels [z][y][x] instead of indirection+atlas
     * (north_star "dense fp16 grid"); extent = voxel dimensions (0 -> n_bricks*8 as for a loaded BrickGrid) */
    uint32_t extent[3];
    const uint16_t* dense;
} orc_brickgrid;

/* the uniform block RendererOpenGL::trace uploads (src/renderer.cpp:88-138) */
typedef struct {
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
    /* build-only switches */
    int32_t use_tf;          /* pathtracer_brick_tf.glsl vs pathtracer_brick.glsl */
    int32_t has_emission;    /* emission grid bound (renderer.cpp:117-124) */
    int32_t integrator;      /* 0 = USE_DDA (both reference kernels), 1 = global-majorant delta/ratio tracking (common.glsl:333-394),
                              * 2 = direct volume rendering, 64-step ray marcher (common.glsl:571-591; needs a LUT),
                              * 3 = trace_path with the 64-step ray-marching trackers (common.glsl:506-566) */
} orc_params;

typedef struct {
    const orc_brickgrid* density;
    const orc_brickgrid* emission;     /* may be NULL */
    const float* tf_lut;               /* tf_size x 4 (already CDF-fixed), may be NULL */
    const float* envmap;               /* env_w*env_h*3, texture order: row 0 = v~0 (bottom) */
    int32_t env_w, env_h;
    const float* impmap;               /* importance pyramid, level 0 first */
    int32_t imp_dim;                   /* 512 */
} orc_scene;

/* per-sample event counters for SURVEY 8(d)'s algorithmic-bytes formula */
typedef struct {
    uint64_t samples;
    uint64_t n_dda_sv, n_dda_tr;       /* DDA loop iterations in sample_volumeDDA / transmittanceDDA */
    uint64_t n_coll_sv, n_coll_tr;     /* tentative collisions */
    uint64_t n_nee;                    /* real scatter events (NEE evaluations) */
    uint64_t n_esc;                    /* escaped paths that fetch the environment */
    uint64_t n_primary_miss;           /* camera rays that miss the clip box */
    uint64_t n_tf_lookup;              /* tf_lookup calls (on-chip traffic) */
} orc_counters;

/* ---- loaders ---- */
int  orc_load_brick(const char* path, orc_brickgrid* out);           /* 0 ok */
void orc_free_brick(orc_brickgrid* g);
/* Radiance RGBE -> float RGB, rows in FILE order (top row first) */
int  orc_load_hdr(const char* path, float** rgb, int32_t* w, int32_t* h);
/* LUT text file "%f, %f, %f, %f" per row (src/transferfunc.cpp:79-93); returns rows, -1 on error */
int  orc_load_lut(const char* path, float* rgba, int32_t max_rows);
void orc_free(void* p);

/* ---- host steps ---- */
/* TransferFunction::upload_gpu + compute_lut_cdf (src/transferfunc.cpp:33-58); in place; returns 1 if the CDF fix-up ran */
int  orc_lut_fixup(float* rgba, int32_t n);
/* flip rows so that row 0 = bottom (texture order) */
void orc_flip_rows(float* rgb, int32_t w, int32_t h, int32_t ch);
/* Environment ctor: env_setup.glsl + glGenerateMipmap (src/environment.cpp:11-33); out has orc_impmap_floats(dim) floats */
int32_t orc_impmap_floats(int32_t dim);
void orc_build_impmap(const float* env_tex, int32_t w, int32_t h, int32_t dim, float* out);
/* bilinear envmap fetch used everywhere (GL_LINEAR, repeat u, clamp v) */
void orc_env_texture(const float* env_tex, int32_t w, int32_t h, float u, float v, float rgb[3]);

/* RendererOpenGL::scale_and_move_to_unit_cube (src/renderer.cpp:227-242): volume_transform out, density_scale in/out */
void orc_unit_cube(const orc_brickgrid* g, float volume_transform[16], float* density_scale);
/* camera: cam_transform = inverse(mat3(lookAt(pos,pos+dir,up))) (src/renderer.cpp:95) */
void orc_camera(const float pos[3], const float dir[3], const float up[3], float cam_transform[9]);
/* glm::mat3(glm::rotate(mat4(1), radians(deg), (0,1,0))) (src/main.cpp:382) */
void orc_env_rotation(float deg, float m[9]);
/* fills the volume/grid dependent uniforms exactly like RendererOpenGL::trace (src/renderer.cpp:96-124) */
void orc_volume_uniforms(orc_params* p, const orc_brickgrid* density, const orc_brickgrid* emission,
                         const float volume_transform[16], float density_scale,
                         const float clip_min[3], const float clip_max[3], float majorant_emission);
void orc_mat3_inverse(const float m[9], float out[9]);
void orc_mat4_inverse(const float m[16], float out[16]);
void orc_mat4_mul(const float a[16], const float b[16], float out[16]);

/* ---- the hot path ---- */
/* Runs samples first_sample..first_sample+n_samples-1 (1-based, like current_sample) for the pixel rectangle
 * [x0,x1) x [y0,y1) of a W x H image and applies the running-mean update of pathtracer_brick.glsl:36 to
 * fb (RGBA32F, W*H*4, row 0 = bottom).  threads <= 0 -> all cores.  counters may be NULL. */
void orc_render(const orc_params* p, const orc_scene* s, float* fb,
                int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                int32_t first_sample, int32_t n_samples, int32_t threads, orc_counters* counters);

/* one path, for unit tests: returns vec4(L, alpha) of trace_path for pixel (x,y), sample s */
void orc_trace_pixel_sample(const orc_params* p, const orc_scene* s, int32_t x, int32_t y, int32_t sample, float out[4]);

/* tonemap.glsl:13-36 in place on RGBA32F */
void orc_tonemap(float* fb, int32_t w, int32_t h, float exposure, float gamma);

/* ---- small exports for unit tests ---- */
uint32_t orc_tea(uint32_t v0, uint32_t v1, uint32_t n);
float    orc_rng(uint32_t* state);
float    orc_lookup_density_brick(const orc_brickgrid* g, int32_t x, int32_t y, int32_t z);
float    orc_lookup_majorant_raw(const orc_brickgrid* g, int32_t x, int32_t y, int32_t z, int32_t mip);
void     orc_sample_environment(const orc_params* p, const orc_scene* s, float r0, float r1, float w_i[3], float le_pdf[4]);
void     orc_view_dir(const orc_params* p, int32_t x, int32_t y, int32_t w, int32_t h, float jx, float jy, float out[3]);
int      orc_intersect_box(const orc_params* p, const float pos[3], const float dir[3], float near_far[2]);
int      orc_sample_volume(const orc_params* p, const orc_scene* s, const float pos[3], const float dir[3], uint32_t* seed, float out[4]);
void     orc_sample_phase_hg(const float dir[3], float g, float r0, float r1, float out[3]);
float    orc_phase_hg(float cos_t, float g);
float    orc_transmittance(const orc_params* p, const orc_scene* s, const float pos[3], const float dir[3], uint32_t* seed);
float    orc_math(int32_t fn, float a, float b);   /* 0 log 1 sin 2 cos 3 tan 4 acos 5 atan2 6 exp 7 pow 8 asin */
int32_t  orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
