// transferfunc.h -- mirror of the reference's TransferFunction (src/transferfunc.h:9-42): an RGBA LUT whose
// alpha is forced monotone (density CDF) before upload, with a window (left, width).
// Not mirrored: randomize() and tinycolormap presets (UI conveniences).
#pragma once

#include <string>
#include <vector>

#include "devmem.h"
#include "hostmath.h"

namespace vr {

class TransferFunction {
public:
    TransferFunction();
    explicit TransferFunction(const std::string& path);
    explicit TransferFunction(const std::vector<vec4>& lut);
    virtual ~TransferFunction();

    // compute density-CDF lut from given lut
    static std::vector<vec4> compute_lut_cdf(const std::vector<vec4>& lut);
    // push (cdf-fixed) lut data to the device
    void upload_gpu();
    // load LUT from file (format: %f, %f, %f, %f per line)
    void load_from_file(const std::string& path);
    // write current LUT to (text-)file
    void write_to_file(const std::string& filename);

    uint32_t size() const { return (uint32_t)lut_gpu.size(); }   // tf_size uniform

    // data
    float window_left, window_width;
    std::vector<vec4> lut;
    std::vector<vec4> lut_gpu;        // what was uploaded (after the CDF fix-up)
    DeviceBufferPtr lut_ssbo;
    uint64_t version = 0;             // process-wide unique id of the last upload (majorant cache key): never reused, so a new
                                      // TransferFunction that happens to land at a freed one's address cannot match a stale key
};

}  // namespace vr
