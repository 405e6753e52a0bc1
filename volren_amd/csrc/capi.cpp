// capi.cpp -- extern "C" surface of libvolren_amd.so (include/volren_amd.h) over the C++ classes.
#include "../../include/volren_amd.h"

#include <cstring>
#include <filesystem>
#include <functional>
#include <iostream>
#include <stdexcept>
#include <memory>
#include <string>
#include <vector>

#include "imageio.h"
#include "renderer.h"
#include "sharded.h"
#include "vr_device.h"

struct vr_renderer {
    vr::RendererHIP impl;
    int device = 0;
};

// N renderers on N devices behind one frame (sharded.h).  The parts are ordinary vr_renderer objects owned by this object: every scene
// call of this header applies to them one by one (vr_sharded_part), which is how the scene is replicated.
struct vr_sharded {
    std::vector<vr_renderer*> parts;
    std::unique_ptr<vr::ShardedRenderer> impl;
    std::string transport, collective;
};

static thread_local std::string g_last_error;

static int guard(const std::function<void()>& fn) {
    try {
        fn();
        g_last_error.clear();
        return VR_OK;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return VR_ERR;
    } catch (...) {
        g_last_error = "unknown error";
        return VR_ERR;
    }
}
static int fail(int code, const char* msg) { g_last_error = msg; return code; }
#define NEED(r) do { if (!(r)) return fail(VR_ERR_ARG, "null renderer"); } while (0)

static void use_device(vr_renderer* r) { VR_HIP(hipSetDevice(r->device)); }

extern "C" {

const char* vr_last_error(void) { return g_last_error.c_str(); }
const char* vr_version(void) { return "volren_amd 0.1 (gfx950)"; }

int vr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vr_create(vr_renderer** out, int device, int width, int height) {
    if (!out) return fail(VR_ERR_ARG, "null out pointer");
    *out = nullptr;
    if (vr_device_count() <= 0) return fail(VR_ERR_NO_DEVICE, "no HIP device available (libvolren_amd has no CPU path)");
    vr_renderer* r = nullptr;
    const int rc = guard([&] {
        if (width <= 0 || height <= 0) throw std::runtime_error("vr_create: resolution must be positive");
        r = new vr_renderer();
        r->device = device;
        use_device(r);
        r->impl.resolution = { width, height };
        r->impl.init();
    });
    if (rc != VR_OK) { delete r; return rc; }
    *out = r;
    return VR_OK;
}

void vr_destroy(vr_renderer* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    delete r;
}

int vr_resize(vr_renderer* r, int w, int h) {
    NEED(r);
    return guard([&] { use_device(r); if (w <= 0 || h <= 0) throw std::runtime_error("vr_resize: resolution must be positive"); r->impl.resize((uint32_t)w, (uint32_t)h); r->impl.sample = 0; });
}

// main.cpp:37-62
int vr_load_volume(vr_renderer* r, const char* path) {
    NEED(r);
    if (!path) return fail(VR_ERR_ARG, "null path");
    return guard([&] {
        use_device(r);
        std::cout << "load volume: " << path << std::endl;
        if (std::filesystem::is_directory(path)) r->impl.volume = vr::Volume::load_folder(path);
        else r->impl.volume = std::make_shared<vr::Volume>(std::string(path));
        r->impl.density_scale = 1.f;
        r->impl.scale_and_move_to_unit_cube();
        r->impl.commit();
        r->impl.sample = 0;
    });
}

// renderer.volume = std::make_shared<voldata::Volume>(path) as the Python scripts do it (bindings.cpp:82,176;
// datagen_colmap.py:57, datagen_denoise.py:85): ONLY replaces the volume -- density_scale, the unit-cube transform and the
// device grids are untouched until the caller runs scale_and_move_to_unit_cube() / commit() itself
int vr_set_volume_path(vr_renderer* r, const char* path) {
    NEED(r);
    if (!path) return fail(VR_ERR_ARG, "null path");
    return guard([&] {
        if (std::filesystem::is_directory(path)) r->impl.volume = vr::Volume::load_folder(path);
        else r->impl.volume = std::make_shared<vr::Volume>(std::string(path));
    });
}

// voldata::Volume::AABB(name) / minorant_majorant(name) of the renderer's volume (bindings.cpp:91,93): world-space box of the
// current frame's grid under the volume transform, out = min xyz, max xyz
int vr_volume_aabb(vr_renderer* r, const char* name, float out[6]) {
    NEED(r);
    if (!out) return fail(VR_ERR_ARG, "null output");
    return guard([&] {
        if (!r->impl.volume || r->impl.volume->grids.empty()) throw std::runtime_error("vr_volume_aabb: no volume");
        const auto bb = r->impl.volume->AABB(name ? name : "density");
        out[0] = bb.first.x; out[1] = bb.first.y; out[2] = bb.first.z; out[3] = bb.second.x; out[4] = bb.second.y; out[5] = bb.second.z;
    });
}
int vr_volume_minorant_majorant(vr_renderer* r, const char* name, float out[2]) {
    NEED(r);
    if (!out) return fail(VR_ERR_ARG, "null output");
    return guard([&] {
        if (!r->impl.volume || r->impl.volume->grids.empty()) throw std::runtime_error("vr_volume_minorant_majorant: no volume");
        const auto mm = r->impl.volume->minorant_majorant(name ? name : "density");
        out[0] = mm.first; out[1] = mm.second;
    });
}

// main.cpp:64-71
int vr_load_envmap(vr_renderer* r, const char* path) {
    NEED(r);
    if (!path) return fail(VR_ERR_ARG, "null path");
    return guard([&] { use_device(r); r->impl.environment = std::make_shared<vr::Environment>(std::string(path)); r->impl.sample = 0; });
}

// main.cpp:73-81
int vr_load_transferfunc(vr_renderer* r, const char* path) {
    NEED(r);
    if (!path) return fail(VR_ERR_ARG, "null path");
    return guard([&] {
        use_device(r);
        r->impl.transferfunc = std::make_shared<vr::TransferFunction>(std::string(path));
        r->impl.show_environment = false;
        r->impl.sample = 0;
    });
}

static void install_grid(vr_renderer* r, const char* name, const std::shared_ptr<vr::Grid>& grid, int unit_cube) {
    const std::string n = name ? name : "density";
    auto& R = r->impl;
    if (n == "density") {
        R.volume = std::make_shared<vr::Volume>(grid);
        if (unit_cube) { R.density_scale = 1.f; R.scale_and_move_to_unit_cube(); }
    } else {
        if (!R.volume || R.volume->grids.empty()) throw std::runtime_error("set the density grid before '" + n + "'");
        R.volume->update_grid_frame(R.volume->grid_frame_counter, grid, n);
    }
    R.sample = 0;
}

int vr_set_volume_dense(vr_renderer* r, const char* name, const float* voxels, int nx, int ny, int nz, const float* transform, int unit_cube) {
    NEED(r);
    if (!voxels || nx <= 0 || ny <= 0 || nz <= 0) return fail(VR_ERR_ARG, "bad dense grid arguments");
    return guard([&] {
        use_device(r);
        auto g = std::make_shared<vr::DenseGrid>((uint32_t)nx, (uint32_t)ny, (uint32_t)nz, voxels);
        if (transform) memcpy(g->transform.m, transform, 64);
        install_grid(r, name, g, unit_cube);
    });
}

// voldata::Volume::add_grid_frame / update_grid_frame (src/bindings.cpp:89-90) for dense float grids: a further animation frame holding
// `name`, or grid `name` of frame `frame` replaced.  Takes effect at the next vr_commit(), like every change of the volume.
int vr_volume_add_grid_frame_dense(vr_renderer* r, const char* name, const float* voxels, int nx, int ny, int nz, const float* transform) {
    NEED(r);
    if (!voxels || nx <= 0 || ny <= 0 || nz <= 0) return fail(VR_ERR_ARG, "bad dense grid arguments");
    return guard([&] {
        auto g = std::make_shared<vr::DenseGrid>((uint32_t)nx, (uint32_t)ny, (uint32_t)nz, voxels);
        if (transform) memcpy(g->transform.m, transform, 64);
        if (!r->impl.volume) r->impl.volume = std::make_shared<vr::Volume>();
        r->impl.volume->add_grid_frame(g, name ? name : "density");
        r->impl.sample = 0;
    });
}
int vr_volume_update_grid_frame_dense(vr_renderer* r, int frame, const char* name, const float* voxels, int nx, int ny, int nz, const float* transform) {
    NEED(r);
    if (!voxels || nx <= 0 || ny <= 0 || nz <= 0 || frame < 0) return fail(VR_ERR_ARG, "bad dense grid arguments");
    return guard([&] {
        if (!r->impl.volume || (size_t)frame >= r->impl.volume->n_grid_frames()) throw std::out_of_range("vr_volume_update_grid_frame_dense: no such frame");
        auto g = std::make_shared<vr::DenseGrid>((uint32_t)nx, (uint32_t)ny, (uint32_t)nz, voxels);
        if (transform) memcpy(g->transform.m, transform, 64);
        r->impl.volume->update_grid_frame((size_t)frame, g, name ? name : "density");
        r->impl.sample = 0;
    });
}
int vr_volume_n_grid_frames(vr_renderer* r, int* n) {
    NEED(r);
    if (!n) return fail(VR_ERR_ARG, "null output");
    *n = r->impl.volume ? (int)r->impl.volume->n_grid_frames() : 0;
    return VR_OK;
}

int vr_set_volume_dense_f16(vr_renderer* r, const char* name, const uint16_t* voxels, int nx, int ny, int nz, const float* transform, int unit_cube) {
    NEED(r);
    if (!voxels || nx <= 0 || ny <= 0 || nz <= 0) return fail(VR_ERR_ARG, "bad dense grid arguments");
    return guard([&] {
        use_device(r);
        auto g = std::make_shared<vr::DenseGridF16>((uint32_t)nx, (uint32_t)ny, (uint32_t)nz, voxels);
        if (transform) memcpy(g->transform.m, transform, 64);
        install_grid(r, name, g, unit_cube);
    });
}

int vr_set_volume_brick(vr_renderer* r, const char* name, const float* transform, const uint32_t nb[3], const float min_maj[2],
                        const uint32_t* indirection, const uint32_t* range, const uint32_t atlas_dim[3], const uint8_t* atlas,
                        int n_mips, const uint32_t* const* mips, const uint32_t (*mip_dims)[3], int unit_cube) {
    NEED(r);
    if (!nb || !min_maj || !indirection || !range || !atlas_dim || !atlas || n_mips < 0 || n_mips > 3) return fail(VR_ERR_ARG, "bad brick grid arguments");
    return guard([&] {
        use_device(r);
        auto g = std::make_shared<vr::BrickGrid>();
        if (transform) memcpy(g->transform.m, transform, 64);
        g->n_bricks = { nb[0], nb[1], nb[2] };
        g->min_maj = { min_maj[0], min_maj[1] };
        const size_t n = (size_t)nb[0] * nb[1] * nb[2];
        g->indirection = vr::Buf3D<uint32_t>(nb[0], nb[1], nb[2]);
        g->range = vr::Buf3D<uint32_t>(nb[0], nb[1], nb[2]);
        memcpy(g->indirection.data.data(), indirection, n * 4);
        memcpy(g->range.data.data(), range, n * 4);
        if ((atlas_dim[0] % 8) || (atlas_dim[1] % 8) || (atlas_dim[2] % 8)) throw std::runtime_error("atlas dimensions must be multiples of 8");
        g->atlas = vr::Buf3D<uint8_t>(atlas_dim[0], atlas_dim[1], atlas_dim[2]);
        memcpy(g->atlas.data.data(), atlas, g->atlas.data.size());
        uint64_t cnt = 0;
        for (size_t i = 0; i < n; ++i) cnt += (vr::half2float(range[i] & 0xFFFFu) != vr::half2float(range[i] >> 16));
        g->brick_counter = cnt;
        for (int m = 0; m < n_mips; ++m) {
            vr::Buf3D<uint32_t> b(mip_dims[m][0], mip_dims[m][1], mip_dims[m][2]);
            memcpy(b.data.data(), mips[m], b.data.size() * 4);
            g->range_mipmaps.push_back(std::move(b));
        }
        install_grid(r, name, g, unit_cube);
    });
}

int vr_set_envmap(vr_renderer* r, const float* rgb, int w, int h) {
    NEED(r);
    if (!rgb || w <= 0 || h <= 0) return fail(VR_ERR_ARG, "bad envmap arguments");
    return guard([&] { use_device(r); r->impl.environment = std::make_shared<vr::Environment>(rgb, w, h); r->impl.sample = 0; });
}

int vr_set_transferfunc(vr_renderer* r, const float* rgba, int n) {
    NEED(r);
    return guard([&] {
        use_device(r);
        if (n <= 0 || !rgba) { r->impl.transferfunc.reset(); r->impl.sample = 0; return; }
        std::vector<vr::vec4> lut;
        for (int i = 0; i < n; ++i) lut.emplace_back(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]);
        r->impl.transferfunc = std::make_shared<vr::TransferFunction>(lut);
        r->impl.sample = 0;
    });
}

int vr_set_int(vr_renderer* r, const char* name, int v) {
    NEED(r);
    if (!name) return fail(VR_ERR_ARG, "null name");
    return guard([&] {
        auto& R = r->impl;
        const std::string n = name;
        if (n == "sample") R.sample = v;
        else if (n == "sppx") R.sppx = v;
        else if (n == "seed") R.seed = v;
        else if (n == "bounces") R.bounces = v;
        else if (n == "show_environment") R.show_environment = v != 0;
        else if (n == "tonemapping") R.tonemapping = v != 0;
        else if (n == "integrator") R.integrator = v;
        else if (n == "fast_math") R.fast_math = v != 0;
        else if (n == "coalesce_trace") { R.flush_pending(); R.coalesce_trace = v != 0; }
        else if (n == "majorant_layout") { if (v < -1 || v > 1) throw std::runtime_error("majorant_layout: -1 (per grid, chosen at commit), 0 (linear), 1 (4x4x4-cell blocks)"); R.majorant_layout = v; }
        else if (n == "tf_float_atlas") R.tf_float_atlas = v != 0;
        else if (n == "gpu_encoder") R.gpu_encoder = v != 0;
        else if (n == "sample_pool_mb") { if (v < 16 || v > 65536) throw std::runtime_error("sample_pool_mb must be in [16, 65536] (item indices of a sub-launch are 32-bit: < 2^32 RGBA32F items)"); R.sample_pool_bytes = (size_t)v << 20; }
        else if (n == "launch_target_ms") { if (v < 0) throw std::runtime_error("launch_target_ms must be >= 0 (0 = no sizing by time)"); R.launch_target_ms = v; }
        else if (n == "order_tiles") { if (v < 0 || v > 2) throw std::runtime_error("order_tiles: 0 (never), 1 (tile subsets), 2 (always)"); R.order_tiles = v; }
        else if (n == "grid_frame_counter") {
            if (!R.volume || v < 0 || (size_t)v >= R.volume->n_grid_frames()) throw std::runtime_error("grid_frame_counter out of range");
            R.volume->grid_frame_counter = (size_t)v;
        } else throw std::runtime_error("unknown int parameter: " + n);
    });
}

int vr_get_int(vr_renderer* r, const char* name, int* v) {
    NEED(r);
    if (!name || !v) return fail(VR_ERR_ARG, "null argument");
    return guard([&] {
        auto& R = r->impl;
        const std::string n = name;
        if (n == "sample") *v = R.sample;
        else if (n == "sppx") *v = R.sppx;
        else if (n == "seed") *v = R.seed;
        else if (n == "bounces") *v = R.bounces;
        else if (n == "show_environment") *v = R.show_environment ? 1 : 0;
        else if (n == "tonemapping") *v = R.tonemapping ? 1 : 0;
        else if (n == "integrator") *v = R.integrator;
        else if (n == "fast_math") *v = R.fast_math ? 1 : 0;
        else if (n == "coalesce_trace") *v = R.coalesce_trace ? 1 : 0;
        else if (n == "majorant_layout") *v = R.majorant_layout;
        else if (n == "majorant_blocked") {          // what the current frame's next launch will use
            vr::SceneParams P; R.fill_params(P); *v = P.density.maj_blocked;
        }
        else if (n == "kernel_variant" || n == "kernel_variant_reason") {      // which compiled kernel the next launch uses, and what sent it to the run-time one (vr_device.h PathtraceVariantReason)
            vr::SceneParams P; R.fill_params(P);
            int why = 0; const int variant = vr::pathtrace_variant_of(P, &why);
            *v = n == "kernel_variant" ? variant : why;
        }
        else if (n == "env_div_safe") *v = R.environment && R.environment->cdf_div_safe ? 1 : 0;      // the environment's warp table passed env_cdf_kernel's check (vr_math.h div_core)
        else if (n == "env_compact") *v = R.environment && R.environment->envmap_rgbe ? 1 : 0;      // the path tracer fetches the map's texels as RGBE dwords (vr_scene.h SceneParams::env_rgbe)
        else if (n == "pending_samples") *v = R.pending_samples();
        else if (n == "tf_float_atlas") *v = R.tf_float_atlas ? 1 : 0;
        else if (n == "gpu_encoder") *v = R.gpu_encoder ? 1 : 0;
        else if (n == "sample_pool_mb") *v = (int)(R.sample_pool_bytes >> 20);
        else if (n == "launch_target_ms") *v = R.launch_target_ms;
        else if (n == "order_tiles") *v = R.order_tiles;
        else if (n == "grid_frame_counter") *v = R.volume ? (int)R.volume->grid_frame_counter : 0;
        else if (n == "n_grid_frames") *v = R.volume ? (int)R.volume->n_grid_frames() : 0;
        else if (n == "last_launches") *v = R.last_launches;
        else if (n == "width") *v = R.resolution.x;
        else if (n == "height") *v = R.resolution.y;
        else throw std::runtime_error("unknown int parameter: " + n);
    });
}

namespace {
struct FloatField { float* ptr; int count; };
FloatField float_field(vr::RendererHIP& R, const std::string& n) {
    if (n == "tonemap_exposure") return { &R.tonemap_exposure, 1 };
    if (n == "tonemap_gamma") return { &R.tonemap_gamma, 1 };
    if (n == "albedo") return { &R.albedo.x, 3 };
    if (n == "phase") return { &R.phase, 1 };
    if (n == "density_scale") return { &R.density_scale, 1 };
    if (n == "emission_scale") return { &R.emission_scale, 1 };
    if (n == "vol_clip_min") return { &R.vol_clip_min.x, 3 };
    if (n == "vol_clip_max") return { &R.vol_clip_max.x, 3 };
    if (n == "cam_pos") return { &R.camera.pos.x, 3 };
    if (n == "cam_dir") return { &R.camera.dir.x, 3 };
    if (n == "cam_up") return { &R.camera.up.x, 3 };
    if (n == "cam_fov") return { &R.camera.fov_degree, 1 };
    if (n == "env_strength") { if (!R.environment) throw std::runtime_error("no environment"); return { &R.environment->strength, 1 }; }
    if (n == "env_transform") { if (!R.environment) throw std::runtime_error("no environment"); return { R.environment->transform.m, 9 }; }
    if (n == "tf_window_left") { if (!R.transferfunc) throw std::runtime_error("no transfer function"); return { &R.transferfunc->window_left, 1 }; }
    if (n == "tf_window_width") { if (!R.transferfunc) throw std::runtime_error("no transfer function"); return { &R.transferfunc->window_width, 1 }; }
    if (n == "volume_transform") { if (!R.volume) throw std::runtime_error("no volume"); return { R.volume->transform.m, 16 }; }
    throw std::runtime_error("unknown float parameter: " + n);
}
}  // namespace

int vr_set_float(vr_renderer* r, const char* name, const float* values, int count) {
    NEED(r);
    if (!name || !values) return fail(VR_ERR_ARG, "null argument");
    return guard([&] {
        auto& R = r->impl;
        const std::string n = name;
        if (n == "env_rot") {                       // main.cpp:381-382
            if (count != 1) throw std::runtime_error("env_rot takes 1 value");
            if (!R.environment) throw std::runtime_error("no environment");
            R.environment->transform = vr::rotation_axis(values[0], 1);
            return;
        }
        if (n == "albedo" && count == 1) { R.albedo = vr::vec3(values[0]); return; }      // main.cpp:371-372
        const FloatField f = float_field(R, n);
        if (count != f.count) throw std::runtime_error("wrong value count for " + n);
        memcpy(f.ptr, values, sizeof(float) * (size_t)count);
    });
}

int vr_get_float(vr_renderer* r, const char* name, float* values, int count) {
    NEED(r);
    if (!name || !values) return fail(VR_ERR_ARG, "null argument");
    return guard([&] {
        const FloatField f = float_field(r->impl, name);
        if (count != f.count) throw std::runtime_error(std::string("wrong value count for ") + name);
        memcpy(values, f.ptr, sizeof(float) * (size_t)count);
    });
}

int vr_commit(vr_renderer* r) { NEED(r); return guard([&] { use_device(r); r->impl.commit(); }); }
int vr_reset(vr_renderer* r) { NEED(r); return guard([&] { r->impl.reset(); }); }
int vr_scale_and_move_to_unit_cube(vr_renderer* r) { NEED(r); return guard([&] { r->impl.scale_and_move_to_unit_cube(); }); }

int vr_trace(vr_renderer* r) { NEED(r); return guard([&] { use_device(r); r->impl.trace(); }); }
int vr_flush(vr_renderer* r) { NEED(r); return guard([&] { use_device(r); r->impl.flush_pending(); }); }
int vr_render(vr_renderer* r, int spp) { NEED(r); return guard([&] { use_device(r); r->impl.render(spp); }); }

int vr_synchronize(vr_renderer* r) {
    NEED(r);
    return guard([&] {
        use_device(r);
        r->impl.synchronize();
        if (r->impl.watchdog_status() != 0) throw std::runtime_error("path-tracing kernel watchdog tripped (a wavefront exceeded its step budget)");
    });
}

int vr_last_kernel_ms(vr_renderer* r, double* ms) {
    NEED(r);
    if (!ms) return fail(VR_ERR_ARG, "null argument");
    return guard([&] { use_device(r); *ms = r->impl.last_kernel_ms(); });
}

int vr_last_pathtrace_ms(vr_renderer* r, double* ms) {
    NEED(r);
    if (!ms) return fail(VR_ERR_ARG, "null output");
    return guard([&] { use_device(r); *ms = r->impl.last_pathtrace_ms(); });
}

int vr_framebuffer(vr_renderer* r, float* out) {
    NEED(r);
    if (!out) return fail(VR_ERR_ARG, "null argument");
    return guard([&] { use_device(r); r->impl.download(out); });
}
int vr_framebuffer_device(vr_renderer* r, void** p) {
    NEED(r);
    if (!p) return fail(VR_ERR_ARG, "null argument");
    return guard([&] { if (!r->impl.color) throw std::runtime_error("no framebuffer"); use_device(r); r->impl.flush_pending(); *p = r->impl.color->get(); });
}
int vr_draw(vr_renderer* r) { NEED(r); return guard([&] { use_device(r); r->impl.draw(); }); }
int vr_display(vr_renderer* r, float* out) {
    NEED(r);
    if (!out) return fail(VR_ERR_ARG, "null argument");
    return guard([&] { use_device(r); r->impl.download_display(out); });
}
int vr_save_png(vr_renderer* r, const char* path) {
    NEED(r);
    if (!path) return fail(VR_ERR_ARG, "null path");
    return guard([&] {
        use_device(r);
        auto& R = r->impl;
        R.draw();
        std::vector<float> fb((size_t)R.resolution.x * R.resolution.y * 4);
        R.download_display(fb.data());
        std::vector<uint8_t> rgba;
        vr::framebuffer_to_rgba8(fb.data(), R.resolution.x, R.resolution.y, rgba);
        vr::save_png_rgba8(path, rgba.data(), R.resolution.x, R.resolution.y);
    });
}

int vr_set_tiles(vr_renderer* r, const int32_t* ids, int n) {
    NEED(r);
    return guard([&] {
        use_device(r);
        std::vector<int32_t> t;
        if (n > 0 && ids) t.assign(ids, ids + n);
        r->impl.set_tiles(t);
    });
}
int vr_set_stream(vr_renderer* r, void* s) { NEED(r); return guard([&] { use_device(r); r->impl.flush_pending(); r->impl.stream = (hipStream_t)s; }); }

int vr_pack_tiles(vr_renderer* r, const int32_t* ids_dev, int n, void* packed) {
    NEED(r);
    return guard([&] {
        use_device(r);
        auto& R = r->impl;
        if (!R.color) throw std::runtime_error("no framebuffer");
        R.flush_pending();
        vr::launch_pack_tiles(R.color->as<float>(), R.resolution.x, R.resolution.y, ids_dev, n, (float*)packed, R.stream);
        VR_HIP(hipGetLastError());
    });
}
int vr_unpack_tiles(vr_renderer* r, const int32_t* ids_dev, int n, const void* packed) {
    NEED(r);
    return guard([&] {
        use_device(r);
        auto& R = r->impl;
        if (!R.color) throw std::runtime_error("no framebuffer");
        R.flush_pending();
        vr::launch_unpack_tiles((const float*)packed, ids_dev, n, R.color->as<float>(), R.resolution.x, R.resolution.y, R.stream);
        VR_HIP(hipGetLastError());
    });
}

int vr_grid_checksums(vr_renderer* r, uint64_t out[3]) {
    NEED(r);
    if (!out) return fail(VR_ERR_ARG, "null argument");
    return guard([&] {
        use_device(r);
        auto& R = r->impl;
        if (!R.volume || R.density_grids.empty()) throw std::runtime_error("no committed volume");
        R.grid_checksums(R.density_grids.at(R.volume->grid_frame_counter), out);
    });
}

// ---- one frame on several devices (sharded.h) ------------------------------------------------------------------------------
int vr_sharded_create(vr_sharded** out, const int* devices, int n_parts, int width, int height) {
    if (!out) return fail(VR_ERR_ARG, "null out pointer");
    *out = nullptr;
    if (!devices || n_parts <= 0 || n_parts > 64) return fail(VR_ERR_ARG, "vr_sharded_create: 1..64 parts, one device ordinal each");
    if (vr_device_count() <= 0) return fail(VR_ERR_NO_DEVICE, "no HIP device available (libvolren_amd has no CPU path)");
    vr_sharded* s = new vr_sharded();
    const int rc = guard([&] {
        std::vector<vr::RendererHIP*> impls;
        std::vector<int> devs(devices, devices + n_parts);
        for (int i = 0; i < n_parts; ++i) {
            vr_renderer* r = nullptr;
            if (vr_create(&r, devices[i], width, height) != VR_OK) throw std::runtime_error(g_last_error);
            s->parts.push_back(r);
            impls.push_back(&r->impl);
        }
        s->impl = std::make_unique<vr::ShardedRenderer>(impls, devs);
        s->transport = s->impl->transport();
        s->collective = s->impl->collective();
    });
    if (rc != VR_OK) { const std::string keep = g_last_error; vr_sharded_destroy(s); g_last_error = keep; return rc; }
    *out = s;
    return VR_OK;
}
void vr_sharded_destroy(vr_sharded* s) {
    if (!s) return;
    s->impl.reset();                                  // waits for the parts' streams, gives them their default stream back
    for (vr_renderer* r : s->parts) vr_destroy(r);
    delete s;
}
int vr_sharded_parts(vr_sharded* s) { return s ? (int)s->parts.size() : 0; }
vr_renderer* vr_sharded_part(vr_sharded* s, int i) { return (s && i >= 0 && i < (int)s->parts.size()) ? s->parts[(size_t)i] : nullptr; }
const char* vr_sharded_transport(vr_sharded* s) { return s ? s->transport.c_str() : ""; }
const char* vr_sharded_collective(vr_sharded* s) { return s ? s->collective.c_str() : ""; }
int vr_sharded_reset(vr_sharded* s) {
    if (!s) return fail(VR_ERR_ARG, "null sharded renderer");
    return guard([&] { s->impl->reset(); });
}
int vr_sharded_render(vr_sharded* s, int spp) {
    if (!s) return fail(VR_ERR_ARG, "null sharded renderer");
    return guard([&] { s->impl->render(spp); });
}
int vr_sharded_synchronize(vr_sharded* s) {
    if (!s) return fail(VR_ERR_ARG, "null sharded renderer");
    return guard([&] { s->impl->synchronize(); });
}

int vr_tile_owners(int width, int height, int n_parts, int32_t* owner_out, int n_tiles) {
    if (!owner_out || width <= 0 || height <= 0 || n_parts <= 0) return fail(VR_ERR_ARG, "vr_tile_owners: bad arguments");
    if (n_tiles != ((width + 15) / 16) * ((height + 15) / 16)) return fail(VR_ERR_ARG, "vr_tile_owners: n_tiles must be ceil(width / 16) * ceil(height / 16)");
    return guard([&] {
        const auto lists = vr::tile_owner_lists(width, height, n_parts);
        for (size_t p = 0; p < lists.size(); ++p)
            for (int32_t t : lists[p]) owner_out[t] = (int32_t)p;
    });
}

int vr_uniforms_size(void) { return (int)sizeof(vr::Uniforms); }
int vr_get_uniforms(vr_renderer* r, void* out, int bytes) {
    NEED(r);
    if (!out || bytes != (int)sizeof(vr::Uniforms)) return fail(VR_ERR_ARG, "bad uniforms buffer");
    return guard([&] { use_device(r); vr::SceneParams P; r->impl.fill_params(P); memcpy(out, &P.u, sizeof(vr::Uniforms)); });
}

int vr_impmap_floats(vr_renderer* r) {
    if (!r || !r->impl.environment) return 0;
    return (int)(r->impl.environment->impmap->size_bytes() / sizeof(float));
}
int vr_get_impmap(vr_renderer* r, float* out, int count) {
    NEED(r);
    if (!out || count != vr_impmap_floats(r)) return fail(VR_ERR_ARG, "bad impmap buffer");
    return guard([&] { use_device(r); const auto v = r->impl.environment->download_impmap(); memcpy(out, v.data(), v.size() * sizeof(float)); });
}

int vr_set_sched(vr_renderer* r, const int32_t thr[8]) {
    NEED(r);
    if (!thr) return fail(VR_ERR_ARG, "null argument");
    // the kernel packs each threshold into a byte: a batch size is a number of lanes (NEW / MARCH = low-water mark / COLLIDE / NEE / POSTNEE / ESCAPE:
    // 0..64, 0 = the default of COLLIDE), [1] caps the slots of a wavefront's pool (0 = all, else 1..192).  A NEW threshold above the pool size
    // would never trigger a batch and end in the watchdog: refused here instead.
    for (int i = 0; i < 8; ++i) {
        const int hi = i == 1 ? 192 : 64;
        if (thr[i] < 0 || thr[i] > hi) return fail(VR_ERR_ARG, i == 1 ? "vr_set_sched: the slot cap [1] must be in 0..192" : "vr_set_sched: batch thresholds must be in 0..64");
    }
    if (thr[1] > 0 && thr[0] > thr[1]) return fail(VR_ERR_ARG, "vr_set_sched: the NEW threshold [0] exceeds the slot cap [1]");
    for (int i = 0; i < 8; ++i) r->impl.tuning.thr[i] = thr[i];
    g_last_error.clear();
    return VR_OK;
}

// scheduler statistics of this renderer's path-tracing launches: enable != 0 zeroes its 32 device counters and switches its launches to the
// instrumented kernels; out (32 x u64, may be NULL) receives what was counted so far (layout: include/volren_amd.h)
int vr_sched_stats(vr_renderer* r, int enable, unsigned long long* out) {
    NEED(r);
    return guard([&] { use_device(r); r->impl.sched_stats(enable != 0, out); });
}

// diagnostics: (begin, queue empty, end) of every wavefront of the last instrumented launch, 100 MHz ticks; n_words = 3 x wavefronts wanted (<= 3 x 8192)
int vr_wave_timeline(vr_renderer* r, unsigned long long* out, int n_words) {
    NEED(r);
    if (!out || n_words <= 0) return fail(VR_ERR_ARG, "bad arguments");
    return guard([&] { use_device(r); r->impl.wave_timeline(out, (size_t)n_words); });
}

// test hook (devmem.h): device allocations above `mb` MiB fail as if the device were out of memory; mb < 0 removes the cap
int vr_test_alloc_cap_mb(long long mb) {
    vr::test_alloc_cap().store(mb < 0 ? ~(size_t)0 : (size_t)mb << 20);
    return VR_OK;
}

int vr_math_probe(int fn, const float* a, const float* b, float* out, int n) {
    if (!a || !b || !out || n <= 0) return fail(VR_ERR_ARG, "bad arguments");
    if (vr_device_count() <= 0) return fail(VR_ERR_NO_DEVICE, "no HIP device available");
    return guard([&] {
        vr::DeviceBuffer da((size_t)n * 4), db((size_t)n * 4), dout((size_t)n * 4);
        da.upload(a, (size_t)n * 4); db.upload(b, (size_t)n * 4);
        vr::launch_math_probe(fn, da.as<float>(), db.as<float>(), dout.as<float>(), n, nullptr);
        VR_HIP(hipGetLastError());
        dout.download(out, (size_t)n * 4);
    });
}

int vr_write_brick_from_dense(const float* voxels, int nx, int ny, int nz, const float* transform, const char* path) {
    if (!voxels || !path || nx <= 0 || ny <= 0 || nz <= 0) return fail(VR_ERR_ARG, "bad arguments");
    return guard([&] {
        auto d = std::make_shared<vr::DenseGrid>((uint32_t)nx, (uint32_t)ny, (uint32_t)nz, voxels);
        if (transform) memcpy(d->transform.m, transform, 64);
        vr::Volume::to_brick_grid(d)->write(path);
    });
}

// writes this build's ".dense" container (grids.cpp): u8 voxels with value = lo + u8 / 255 * (hi - lo)
int vr_write_dense(const uint8_t* voxels, int nx, int ny, int nz, float lo, float hi, const float* transform, const char* path) {
    if (!voxels || !path || nx <= 0 || ny <= 0 || nz <= 0) return fail(VR_ERR_ARG, "bad dense grid arguments");
    return guard([&] {
        vr::mat4 t;
        if (transform) memcpy(t.m, transform, 64);
        vr::write_dense_file(path, t, (uint32_t)nx, (uint32_t)ny, (uint32_t)nz, lo, hi, voxels);
    });
}

int vr_encode_dense_stats(const float* voxels, int nx, int ny, int nz, uint32_t nb[3], uint64_t* counter, float mm[2]) {
    if (!voxels || !nb || !counter || !mm) return fail(VR_ERR_ARG, "null argument");
    return guard([&] {
        auto d = std::make_shared<vr::DenseGrid>((uint32_t)nx, (uint32_t)ny, (uint32_t)nz, voxels);
        auto b = vr::Volume::to_brick_grid(d);
        nb[0] = b->n_bricks.x; nb[1] = b->n_bricks.y; nb[2] = b->n_bricks.z;
        *counter = b->brick_counter;
        mm[0] = b->min_maj.first; mm[1] = b->min_maj.second;
    });
}

}  // extern "C"
