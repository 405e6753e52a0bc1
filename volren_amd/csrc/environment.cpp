// environment.cpp -- see environment.h.  Follows src/environment.cpp:6-33.
#include "environment.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

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
    // Compact form for the path tracer's texel fetches (round 6): a texel whose three components are m_c * 2^(e - 136) with 8-bit integers m_c and one shared e in
    // [10, 255] -- every texel of a Radiance file is; e >= 10 keeps the scale a normal float -- packs into one dword that decodes to the same three floats exactly.
    // One texel that is not such a number (a map handed over as floats usually has some) and the map keeps its float form only.  VR_ENV_RGBE=0 switches it off.
    {
        std::vector<uint32_t> packed((size_t)w * h);
        bool exact = !(std::getenv("VR_ENV_RGBE") && std::getenv("VR_ENV_RGBE")[0] == '0');
        for (size_t i = 0; exact && i < packed.size(); ++i) {
            const float* c = &tex[kEnvTexelFloats * i];
            const float mx = std::max(c[0], std::max(c[1], c[2]));
            if (!(c[0] >= 0.0f && c[1] >= 0.0f && c[2] >= 0.0f) || std::signbit(c[0]) || std::signbit(c[1]) || std::signbit(c[2]) || !std::isfinite(mx)) { exact = false; break; }
            if (mx == 0.0f) { packed[i] = 0u; continue; }
            int k = 0;
            (void)std::frexp(mx, &k);                                   // mx = f * 2^k, f in [0.5, 1): its mantissa as an integer below 256 needs the scale 2^(k - 8)
            const int e = k - 8 + 136;
            if (e < 10 || e > 255) { exact = false; break; }
            const float scale = std::ldexp(1.0f, e - 136);
            uint32_t q = (uint32_t)e << 24;
            for (int j = 0; j < 3; ++j) {
                const float m = c[j] / scale;                            // exact: a power of two
                const uint32_t mi = (uint32_t)m;
                if (m != (float)mi || mi > 255u || (float)mi * scale != c[j]) { exact = false; break; }
                q |= mi << (8 * j);
            }
            packed[i] = q;
        }
        if (exact) {
            envmap_rgbe = make_device_buffer(packed.size() * sizeof(uint32_t));
            envmap_rgbe->upload(packed.data(), packed.size() * sizeof(uint32_t));
        }
    }
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
    {
        const int32_t base_mip = (int32_t)std::floor(std::log2((float)DIMENSION));      // = Uniforms::env_imp_base_mip (renderer.cpp)
        const size_t at = (size_t)(4 * (int32_t)DIMENSION * (int32_t)DIMENSION - 4 * ((int32_t)DIMENSION >> base_mip) * ((int32_t)DIMENSION >> base_mip)) / 3;      // vr_trace.h imp_level_offset
        VR_HIP(hipMemcpy(&avg_importance, impmap->as<float>() + at, sizeof(float), hipMemcpyDeviceToHost));
    }
}

uint32_t Environment::num_mip_levels() const { return 1 + (uint32_t)std::floor(std::log2((float)DIMENSION)); }
uint32_t Environment::dimension() const { return DIMENSION; }

std::vector<float> Environment::download_impmap() const {
    std::vector<float> out(impmap->size_bytes() / sizeof(float));
    impmap->download(out.data(), impmap->size_bytes());
    return out;
}

}  // namespace vr
