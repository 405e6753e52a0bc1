// transferfunc.cpp -- see transferfunc.h.  Behaviour follows src/transferfunc.cpp:33-58 (CDF fix-up + upload)
// and :79-106 (text IO); written from the behaviour, not from the text.
#include "transferfunc.h"

#include <atomic>

#include <cstdio>
#include <filesystem>
#include <iostream>

namespace vr {

namespace {
std::vector<vec4> linear_ramp(int bins) {
    std::vector<vec4> ramp;
    for (int i = 0; i < bins; ++i) {
        const float f = (float)i / (float)(bins - 1);
        ramp.emplace_back(f, f, f, f);
    }
    return ramp;
}
bool alpha_is_monotone(const std::vector<vec4>& l) {
    for (size_t i = 1; i < l.size(); ++i)
        if (l[i].w < l[i - 1].w) return false;
    return true;
}
}  // namespace

// The reference's default constructor fills 8 random bins (transferfunc.cpp:62-67, UI convenience, out of scope);
// this build starts from a linear ramp so that a default-constructed object is deterministic.
TransferFunction::TransferFunction() : window_left(0.f), window_width(1.f), lut(linear_ramp(8)) { upload_gpu(); }

TransferFunction::TransferFunction(const std::string& path) : window_left(0.f), window_width(1.f) { load_from_file(path); }

TransferFunction::TransferFunction(const std::vector<vec4>& entries) : window_left(0.f), window_width(1.f), lut(entries) { upload_gpu(); }

TransferFunction::~TransferFunction() = default;

// alpha_i <- (sum_{j<=i} alpha_j) / (sum_j alpha_j); a non-positive total falls back to (i+1)/N.  rgb untouched.
std::vector<vec4> TransferFunction::compute_lut_cdf(const std::vector<vec4>& in) {
    std::vector<vec4> out(in);
    const size_t n = out.size();
    float running = 0.f;
    for (size_t i = 0; i < n; ++i) {
        running = i == 0 ? out[0].w : out[i].w + running;     // same association as a[i] += a[i-1]
        out[i].w = running;
    }
    const float total = running;
    for (size_t i = 0; i < n; ++i)
        out[i].w = total <= 0.f ? (float)(i + 1) / (float)n : out[i].w / total;
    return out;
}

void TransferFunction::upload_gpu() {
    if (lut.empty()) throw std::runtime_error("TransferFunction: empty LUT");
    // the DDA majorant remap (common.glsl:425,472) needs alpha to be nondecreasing
    lut_gpu = alpha_is_monotone(lut) ? lut : compute_lut_cdf(lut);
    lut_ssbo = make_device_buffer(lut_gpu.size() * sizeof(vec4));
    lut_ssbo->upload(lut_gpu.data(), lut_gpu.size() * sizeof(vec4));
    static std::atomic<uint64_t> next_upload_id{ 1 };
    version = next_upload_id.fetch_add(1);
}

void TransferFunction::load_from_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) throw std::runtime_error("Unable to read file: " + path);
    std::cout << "Loading LUT: " << path << std::endl;
    // One row per LINE, as the reference's getline loop pushes them (transferfunc.cpp:86-91) -- also for a blank or malformed line, whose unparsed
    // components are uninitialised there; in practice they are the previous row's values (the same stack slots), which is what this build defines
    // (first row: 0).  tf_size counts such rows, so skipping them would change every lookup.
    std::vector<vec4> rows;
    char line[256];
    vec4 e(0.f, 0.f, 0.f, 0.f);
    while (std::fgets(line, sizeof line, f)) {
        std::sscanf(line, "%f, %f, %f, %f", &e.x, &e.y, &e.z, &e.w);
        rows.push_back(e);
    }
    std::fclose(f);
    lut.swap(rows);
    upload_gpu();
}

void TransferFunction::write_to_file(const std::string& filename) {
    const std::string target = std::filesystem::path(filename).replace_extension(".txt").string();   // always text
    FILE* f = std::fopen(target.c_str(), "w");
    if (!f) return;
    for (const vec4& e : lut) std::fprintf(f, "%f, %f, %f, %f\n", e.x, e.y, e.z, e.w);
    std::fclose(f);
}

}  // namespace vr
