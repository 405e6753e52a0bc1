// environment.h -- mirror of the reference's Environment (src/environment.h:7-23): an HDR environment map plus
// the 512x512 importance map whose mip pyramid is the hierarchical CDF sampled by sample_environment
// (common.glsl:100-146).  Device memory replaces the two GL textures.
#pragma once

#include <string>
#include <vector>

#include "devmem.h"
#include "hostmath.h"

namespace vr {

class Environment {
public:
    explicit Environment(const std::string& path);                       // Radiance .hdr
    Environment(const float* rgb_top_first, int w, int h);               // Environment(Texture2D) equivalent
    virtual ~Environment();

    explicit operator bool() const { return envmap && impmap; }
    uint32_t num_mip_levels() const;
    uint32_t dimension() const;

    // data (names as in the reference)
    mat3 transform;
    float strength;
    DeviceBufferPtr envmap;      // RGBA32F texels, row 0 = bottom of the image (GL texture order)
    DeviceBufferPtr envmap_rgbe; // the same texels, one dword each (vr_scene.h SceneParams::env_rgbe), when every texel is exactly an RGBE number; else null
    DeviceBufferPtr impmap;      // R32F pyramid: 512^2, 256^2, ..., 1
    DeviceBufferPtr cdf;         // per-2x2-block warp thresholds derived from the pyramid (see vr_trace.h sample_environment)
    int width = 0, height = 0;
    float avg_importance = 0.0f; // the pyramid's coarsest value (what the kernels' MIS weights divide by: common.glsl:147's textureLod at the base mip), read back once so that
                                 // the kernels get it as an argument instead of fetching it at the end of every light sample and escape (round 6)
    bool cdf_div_safe = false;   // every threshold of `cdf` is NaN, 0 or in [2^-76, 1]: the kernels' warp may use the division without its guard instructions (vr_math.h div_core)

    std::vector<float> download_impmap() const;

private:
    void build(const float* rgb_top_first, int w, int h);
};

}  // namespace vr
