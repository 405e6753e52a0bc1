// environment.cpp -- see environment.h.  Follows src/environment.cpp:6-33.
#include "environment.h"

#include "imageio.h"
#include "vr_device.h"

namespace vr {

// importance map parameters (power of two!) -- environment.cpp:6-7
static const uint32_t DIMENSION = 512;

static std::vector<float> load_rgb(const std::string& path, int& w, int& h) {
    std::vector<float> rgb;
    load_hdr(path, rgb, w, h);
    return rgb;
}

Environment::Environment(const std::string& path) : transform(1.0f), strength(1.0f) {
    int w, h;
    const std::vector<float> rgb = load_rgb(path, w, h);
    build(rgb.data(), w, h);
}

Environment::Environment(const float* rgb_top_first, int w, int h) : transform(1.0f), strength(1.0f) {
    build(rgb_top_first, w, h);
}

Environment::~Environment() {}

void Environment::build(const float* rgb, int w, int h) {
    if (w <= 0 || h <= 0) throw std::runtime_error("Environment: empty image");
    width = w; height = h;
    // texture order: image rows are stored top first, GL's v runs bottom-up -> flip; kEnvTexelFloats floats per texel (vr_scene.h)
    std::vector<float> tex((size_t)w * h * kEnvTexelFloats);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float* s = rgb + 3 * ((size_t)(h - 1 - y) * w + x);
            float* d = &tex[kEnvTexelFloats * ((size_t)y * w + x)];
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
            if (kEnvTexelFloats > 3) d[3] = 1.0f;
        }
    envmap = make_device_buffer(tex.size() * sizeof(float));
    envmap->upload(tex.data(), tex.size() * sizeof(float));
    size_t n = 0;
    for (uint32_t d = DIMENSION; d >= 1; d >>= 1) n += (size_t)d * d;
    impmap = make_device_buffer(n * sizeof(float));
    // build importance map: env_setup.glsl + glGenerateMipmap
    launch_build_impmap(envmap->as<float>(), w, h, (int)DIMENSION, impmap->as<float>(), nullptr);
    VR_HIP(hipGetLastError());
    cdf = make_device_buffer(env_cdf_table_floats((int32_t)num_mip_levels() - 2) * sizeof(float));      // levels 0 .. base mip - 1
    DeviceBufferPtr unsafe = make_device_buffer(sizeof(uint32_t));
    launch_build_env_cdf(impmap->as<float>(), (int)DIMENSION, cdf->as<float>(), unsafe->as<uint32_t>(), nullptr);
    VR_HIP(hipGetLastError());
    VR_HIP(hipStreamSynchronize(nullptr));
    uint32_t flag = 1u;
    unsafe->download(&flag, sizeof flag);
    cdf_div_safe = flag == 0u;
}

uint32_t Environment::num_mip_levels() const { return 1 + (uint32_t)std::floor(std::log2((float)DIMENSION)); }
uint32_t Environment::dimension() const { return DIMENSION; }

std::vector<float> Environment::download_impmap() const {
    std::vector<float> out(impmap->size_bytes() / sizeof(float));
    impmap->download(out.data(), impmap->size_bytes());
    return out;
}

}  // namespace vr
