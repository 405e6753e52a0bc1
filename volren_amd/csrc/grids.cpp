// grids.cpp -- .brick container IO, dense->brick encoder, Volume helpers (see grids.h).
#include "grids.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <sstream>
#include <stdexcept>

namespace vr {

// ---------------------------------------------------------------------------------------------------
// fp16 helpers (range texture is GL_RG16F)

uint16_t float_to_half_round_down(float f) { return (uint16_t)float_to_half_down(f); }
uint16_t float_to_half_round_up(float f) { return (uint16_t)float_to_half_up(f); }

// ---------------------------------------------------------------------------------------------------
DenseGrid::DenseGrid(uint32_t w, uint32_t h, uint32_t d, const float* data) : dim{ w, h, d }, voxels(data, data + (size_t)w * h * d) {
    if (voxels.empty()) throw std::runtime_error("DenseGrid: empty grid");
    float lo = voxels[0], hi = voxels[0];
    for (float v : voxels) { lo = std::min(lo, v); hi = std::max(hi, v); }
    min_maj = { lo, hi };
}

// (min of mins, max of maxes) over 2x2x2 children, halves kept as the children's own fp16 words
static void build_range_mips(const Buf3D<uint32_t>& base, std::vector<Buf3D<uint32_t>>& mips) {
    mips.clear();
    mips.reserve(3);
    const Buf3D<uint32_t>* src = &base;
    for (int m = 0; m < 3; ++m) {
        const uvec3 s = src->stride;
        Buf3D<uint32_t> dst((s.x + 1) / 2, (s.y + 1) / 2, (s.z + 1) / 2);
        for (uint32_t z = 0; z < dst.stride.z; ++z)
            for (uint32_t y = 0; y < dst.stride.y; ++y)
                for (uint32_t x = 0; x < dst.stride.x; ++x) {
                    float lo = INFINITY, hi = -INFINITY; uint16_t hlo = 0, hhi = 0;
                    for (uint32_t c = 0; c < 8; ++c) {
                        const uint32_t cx = 2 * x + (c & 1), cy = 2 * y + ((c >> 1) & 1), cz = 2 * z + (c >> 2);
                        if (cx >= s.x || cy >= s.y || cz >= s.z) continue;
                        const uint32_t rg = (*src)(cx, cy, cz);
                        const float l = half2float(rg & 0xFFFFu), h = half2float(rg >> 16);
                        if (l < lo) { lo = l; hlo = (uint16_t)(rg & 0xFFFFu); }
                        if (h > hi) { hi = h; hhi = (uint16_t)(rg >> 16); }
                    }
                    dst(x, y, z) = (uint32_t)hlo | ((uint32_t)hhi << 16);
                }
        mips.push_back(std::move(dst));
        src = &mips.back();
    }
}

DenseGridF16::DenseGridF16(uint32_t w, uint32_t h, uint32_t d, const uint16_t* data) : dim{ w, h, d }, voxels(data, data + (size_t)w * h * d) {
    if (voxels.empty()) throw std::runtime_error("DenseGridF16: empty grid");
    const uint32_t nbx = (w + 7) / 8, nby = (h + 7) / 8, nbz = (d + 7) / 8;
    range = Buf3D<uint32_t>(nbx, nby, nbz);
    // separable dilated min/max would be faster; the direct form is the specification
    std::vector<float> f(voxels.size());
    float gmin = INFINITY, gmax = -INFINITY;
    for (size_t i = 0; i < voxels.size(); ++i) { f[i] = half2float(voxels[i]); gmin = std::min(gmin, f[i]); gmax = std::max(gmax, f[i]); }
    min_maj = { gmin, gmax };
    auto at = [&](int64_t x, int64_t y, int64_t z) -> float {
        if (x < 0 || y < 0 || z < 0 || x >= w || y >= h || z >= d) return 0.f;
        return f[((size_t)z * h + y) * w + x];
    };
    for (uint32_t bz = 0; bz < nbz; ++bz)
        for (uint32_t by = 0; by < nby; ++by)
            for (uint32_t bx = 0; bx < nbx; ++bx) {
                float lo = INFINITY, hi = -INFINITY;
                const int64_t x0 = (int64_t)bx * 8 - 2, y0 = (int64_t)by * 8 - 2, z0 = (int64_t)bz * 8 - 2;
                for (int64_t z = z0; z < z0 + 12; ++z)
                    for (int64_t y = y0; y < y0 + 12; ++y)
                        for (int64_t x = x0; x < x0 + 12; ++x) {
                            const float v = at(x, y, z);
                            lo = std::min(lo, v); hi = std::max(hi, v);
                        }
                range(bx, by, bz) = (uint32_t)float_to_half_round_down(lo) | ((uint32_t)float_to_half_round_up(hi) << 16);
            }
    build_range_mips(range, range_mipmaps);
}

// ---------------------------------------------------------------------------------------------------
// .brick container (little endian, leading endianness byte): SURVEY.md 2.3
namespace {
struct Reader {
    FILE* f;
    std::string path;
    void read(void* dst, size_t n) {
        if (fread(dst, 1, n, f) != n) throw std::runtime_error("Unable to read brick grid (truncated): " + path);
    }
    template <typename T> void buf3d(Buf3D<T>& b) {
        uint32_t dim[3]; uint64_t count;
        read(dim, 12); read(&count, 8);
        if (count != (uint64_t)dim[0] * dim[1] * dim[2] || count > (1ull << 36))
            throw std::runtime_error("Unable to read brick grid (bad buffer header): " + path);
        b.stride = { dim[0], dim[1], dim[2] };
        b.data.resize((size_t)count);
        read(b.data.data(), (size_t)count * sizeof(T));
    }
};
struct Writer {
    FILE* f;
    void write(const void* src, size_t n) { if (fwrite(src, 1, n, f) != n) throw std::runtime_error("write failed"); }
    template <typename T> void buf3d(const Buf3D<T>& b) {
        const uint32_t dim[3] = { b.stride.x, b.stride.y, b.stride.z };
        const uint64_t count = b.data.size();
        write(dim, 12); write(&count, 8); write(b.data.data(), (size_t)count * sizeof(T));
    }
};
}  // namespace

BrickGrid::BrickGrid(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Unable to read file: " + path);
    Reader r{ f, path };
    try {
        uint8_t endian;
        r.read(&endian, 1);
        if (endian != 1) throw std::runtime_error("Unable to read brick grid (not a little-endian .brick): " + path);
        r.read(transform.m, 64);
        uint32_t nb[3]; r.read(nb, 12);
        n_bricks = { nb[0], nb[1], nb[2] };
        float mm[2]; r.read(mm, 8);
        min_maj = { mm[0], mm[1] };
        r.read(&brick_counter, 8);
        r.buf3d(indirection);
        r.buf3d(range);
        r.buf3d(atlas);
        uint64_t nm; r.read(&nm, 8);
        if (nm > 16) throw std::runtime_error("Unable to read brick grid (mip count): " + path);
        range_mipmaps.resize((size_t)nm);
        for (auto& m : range_mipmaps) r.buf3d(m);
        uint8_t extra;
        if (fread(&extra, 1, 1, f) != 0) throw std::runtime_error("Unable to read brick grid (trailing bytes): " + path);
        auto same = [&](const uvec3& s) { return s.x == n_bricks.x && s.y == n_bricks.y && s.z == n_bricks.z; };
        if (!same(indirection.stride) || !same(range.stride) || (atlas.stride.x % 8) || (atlas.stride.y % 8) || (atlas.stride.z % 8))
            throw std::runtime_error("Unable to read brick grid (inconsistent dimensions): " + path);
    } catch (...) { fclose(f); throw; }
    fclose(f);
}

void BrickGrid::write(const std::string& path) const {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("Unable to write file: " + path);
    Writer w{ f };
    try {
        const uint8_t endian = 1; w.write(&endian, 1);
        w.write(transform.m, 64);
        const uint32_t nb[3] = { n_bricks.x, n_bricks.y, n_bricks.z }; w.write(nb, 12);
        const float mm[2] = { min_maj.first, min_maj.second }; w.write(mm, 8);
        w.write(&brick_counter, 8);
        w.buf3d(indirection); w.buf3d(range); w.buf3d(atlas);
        const uint64_t nm = range_mipmaps.size(); w.write(&nm, 8);
        for (const auto& m : range_mipmaps) w.buf3d(m);
    } catch (...) { fclose(f); throw; }
    fclose(f);
}

float BrickGrid::lookup(uint32_t x, uint32_t y, uint32_t z) const {
    const uint32_t bx = x >> 3, by = y >> 3, bz = z >> 3;
    if (bx >= n_bricks.x || by >= n_bricks.y || bz >= n_bricks.z) return 0.f;
    const uint32_t ind = indirection(bx, by, bz), rg = range(bx, by, bz);
    const uint32_t ax = ((ind >> 22) << 3) + (x & 7), ay = (((ind >> 12) & 1023u) << 3) + (y & 7), az = (((ind >> 2) & 1023u) << 3) + (z & 7);
    const float lo = half2float(rg & 0xFFFFu), hi = half2float(rg >> 16);
    float un = 0.f;
    if (ax < atlas.stride.x && ay < atlas.stride.y && az < atlas.stride.z) un = (float)atlas(ax, ay, az) / 255.0f;
    return lo + un * (hi - lo);
}

// ---------------------------------------------------------------------------------------------------
// Serialized dense grid, ".dense".  voldata's own serialisation is not vendored, so this build defines the
// container by analogy with the .brick one it could pin: little endian,
//   u8 endian_flag = 1; f32[16] transform (column-major); u32[3] n_voxels; f32[2] min_maj;
//   Buf3D<u8> voxels: u32[3] stride = n_voxels, u64 count, u8 data[count]  (x fastest; value = min + u8/255 * (maj - min))
// "parity unpinned" for files written by the reference's voldata; files written by write_dense_file() round-trip exactly.
static std::shared_ptr<DenseGrid> load_dense_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Unable to read file: " + path);
    Reader r{ f, path };
    try {
        uint8_t endian;
        r.read(&endian, 1);
        if (endian != 1) throw std::runtime_error("Unable to read dense grid (not a little-endian .dense): " + path);
        mat4 transform;
        r.read(transform.m, 64);
        uint32_t n[3]; r.read(n, 12);
        float mm[2]; r.read(mm, 8);
        Buf3D<uint8_t> vox;
        r.buf3d(vox);
        if (vox.stride.x != n[0] || vox.stride.y != n[1] || vox.stride.z != n[2]) throw std::runtime_error("Unable to read dense grid (extent mismatch): " + path);
        std::vector<float> v(vox.data.size());
        for (size_t i = 0; i < v.size(); ++i) v[i] = mm[0] + ((float)vox.data[i] / 255.f) * (mm[1] - mm[0]);
        fclose(f);
        auto g = std::make_shared<DenseGrid>(n[0], n[1], n[2], v.data());
        g->transform = transform;
        return g;
    } catch (...) { fclose(f); throw; }
}
void write_dense_file(const std::string& path, const mat4& transform, uint32_t nx, uint32_t ny, uint32_t nz, float lo, float hi, const uint8_t* voxels) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("Unable to write file: " + path);
    Writer w{ f };
    try {
        const uint8_t endian = 1; w.write(&endian, 1);
        w.write(transform.m, 64);
        const uint32_t n[3] = { nx, ny, nz }; w.write(n, 12);
        const float mm[2] = { lo, hi }; w.write(mm, 8);
        Buf3D<uint8_t> b(nx, ny, nz);
        memcpy(b.data.data(), voxels, b.data.size());
        w.buf3d(b);
        fclose(f);
    } catch (...) { fclose(f); throw; }
}
// Headerless raw volume, "<name>_<nx>x<ny>x<nz>_<uint8|uint16|float32>.raw" (the Open SciVis naming convention): x fastest,
// little endian; integer types are normalised to [0, 1].  Identity transform (index space = model space).
static std::shared_ptr<DenseGrid> load_raw_file(const std::string& path) {
    const std::string stem = std::filesystem::path(path).stem().string();
    unsigned nx = 0, ny = 0, nz = 0;
    char type[16] = { 0 };
    const size_t us = stem.rfind('_');
    const size_t ds = us == std::string::npos ? std::string::npos : stem.rfind('_', us - 1);
    if (us == std::string::npos || ds == std::string::npos || sscanf(stem.c_str() + ds + 1, "%ux%ux%u_%15s", &nx, &ny, &nz, type) != 4 || !nx || !ny || !nz)
        throw std::runtime_error("Unable to load raw volume (expected <name>_<nx>x<ny>x<nz>_<uint8|uint16|float32>.raw): " + path);
    const std::string t = type;
    const size_t bpv = t == "uint8" ? 1 : (t == "uint16" ? 2 : (t == "float32" ? 4 : 0));
    if (!bpv) throw std::runtime_error("Unable to load raw volume (voxel type " + t + "): " + path);
    const size_t n = (size_t)nx * ny * nz;
    std::vector<uint8_t> raw(n * bpv);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Unable to read file: " + path);
    const size_t got = fread(raw.data(), 1, raw.size(), f);
    fclose(f);
    if (got != raw.size()) throw std::runtime_error("Unable to load raw volume (file shorter than its name says): " + path);
    std::vector<float> v(n);
    if (bpv == 1) for (size_t i = 0; i < n; ++i) v[i] = (float)raw[i] / 255.f;
    else if (bpv == 2) for (size_t i = 0; i < n; ++i) { uint16_t u; memcpy(&u, &raw[2 * i], 2); v[i] = (float)u / 65535.f; }
    else memcpy(v.data(), raw.data(), n * 4);
    return std::make_shared<DenseGrid>(nx, ny, nz, v.data());
}

// voldata::Volume(path) (main.cpp:44): grid "density" of frame 0 from a file.  This build reads the serialized sparse (.brick)
// and dense (.dense) grids and headerless .raw volumes; OpenVDB / NanoVDB / DICOM need libraries the image does not have.
Volume::Volume(const std::string& path) {
    const std::string ext = std::filesystem::path(path).extension().string();
    if (ext == ".brick") add_grid_frame(std::make_shared<BrickGrid>(path), "density");
    else if (ext == ".dense") add_grid_frame(load_dense_file(path), "density");
    else if (ext == ".raw") add_grid_frame(load_raw_file(path), "density");
    else throw std::runtime_error("Unable to load volume (this build reads .brick, .dense and .raw grids): " + path);
}

std::shared_ptr<Volume> Volume::load_folder(const std::string& path) {
    std::vector<std::string> files;
    for (const auto& e : std::filesystem::directory_iterator(path))
        if (e.is_regular_file() && (e.path().extension() == ".brick" || e.path().extension() == ".dense" || e.path().extension() == ".raw")) files.push_back(e.path().string());
    std::sort(files.begin(), files.end());
    if (files.empty()) throw std::runtime_error("Unable to load volume (no .brick / .dense / .raw files in folder): " + path);
    auto vol = std::make_shared<Volume>();
    for (const auto& f : files) vol->add_grid_frame(Volume(f).current_grid(), "density");
    return vol;
}

std::pair<vec3, vec3> Volume::AABB(const std::string& name) const {
    const GridPtr g = current_grid(name);
    const mat4 M = transform * g->transform;
    const uvec3 e = g->index_extent();
    vec3 lo(INFINITY), hi(-INFINITY);
    for (int k = 0; k < 8; ++k) {
        const vec3 c = transform_point(M, vec3((k & 1) ? (float)e.x : 0.f, (k & 2) ? (float)e.y : 0.f, (k & 4) ? (float)e.z : 0.f));
        lo = vmin(lo, c); hi = vmax(hi, c);
    }
    return { lo, hi };
}

std::string Volume::to_string(const std::string& indent) const {
    std::ostringstream s;
    s << indent << "frames: " << grids.size() << ", current: " << grid_frame_counter << "\n";
    if (!grids.empty()) {
        const auto g = current_grid();
        const uvec3 e = g->index_extent();
        const auto mm = g->minorant_majorant();
        s << indent << "index extent: " << e.x << " x " << e.y << " x " << e.z << "\n";
        s << indent << "minorant / majorant: " << mm.first << " / " << mm.second << "\n";
    }
    return s.str();
}

// ---------------------------------------------------------------------------------------------------
// Dense -> brick encoder.  voldata's own encoder is not available (un-vendored, unpinned); this one produces
// grids that satisfy every invariant observed on data/smoke.brick:
//   * n_bricks rounded up to a multiple of 8 per axis (3 range mips),
//   * range = (min, max) of the brick's voxels dilated by 2 voxels (covers the tricubic filter's taps),
//     min rounded down / max rounded up to fp16,
//   * bricks whose dilated range is a single value carry no atlas block (indirection 0),
//   * atlas = n_bricks.x x n_bricks.y x ceil(count / (nx*ny)) blocks, voxel = round((v - min) / (max - min) * 255),
//   * mips = (min of mins, max of maxes) over 2x2x2 children, min_maj = (min of mins, max of maxes).
std::shared_ptr<BrickGrid> Volume::to_brick_grid(const GridPtr& grid) {
    if (auto b = std::dynamic_pointer_cast<BrickGrid>(grid)) return b;
    auto dense = std::dynamic_pointer_cast<DenseGrid>(grid);
    if (!dense) {
        const auto f16 = std::dynamic_pointer_cast<DenseGridF16>(grid);
        if (!f16) throw std::runtime_error("to_brick_grid: unsupported grid type");
        std::vector<float> f(f16->voxels.size());
        for (size_t i = 0; i < f.size(); ++i) f[i] = half2float(f16->voxels[i]);
        dense = std::make_shared<DenseGrid>(f16->dim.x, f16->dim.y, f16->dim.z, f.data());
        dense->transform = f16->transform;
    }
    const uvec3 dim = dense->dim;
    auto up8 = [](uint32_t v) { return ((v + 7u) / 8u + 7u) / 8u * 8u; };
    auto out = std::make_shared<BrickGrid>();
    out->transform = dense->transform;
    out->n_bricks = { up8(dim.x), up8(dim.y), up8(dim.z) };
    const uvec3 nb = out->n_bricks;
    out->indirection = Buf3D<uint32_t>(nb.x, nb.y, nb.z);
    out->range = Buf3D<uint32_t>(nb.x, nb.y, nb.z);
    const float* vox = dense->voxels.data();
    auto at = [&](int64_t x, int64_t y, int64_t z) -> float {
        if (x < 0 || y < 0 || z < 0 || x >= dim.x || y >= dim.y || z >= dim.z) return 0.f;
        return vox[((size_t)z * dim.y + y) * dim.x + x];
    };
    // pass 1: ranges
    std::vector<uint8_t> alloc((size_t)nb.x * nb.y * nb.z, 0);
    uint64_t count = 0;
    float gmin = INFINITY, gmax = -INFINITY;
    for (uint32_t bz = 0; bz < nb.z; ++bz)
        for (uint32_t by = 0; by < nb.y; ++by)
            for (uint32_t bx = 0; bx < nb.x; ++bx) {
                float lo = INFINITY, hi = -INFINITY;
                const int64_t x0 = (int64_t)bx * 8 - 2, y0 = (int64_t)by * 8 - 2, z0 = (int64_t)bz * 8 - 2;
                // fully outside the data (with dilation): constant 0
                if (x0 >= dim.x || y0 >= dim.y || z0 >= dim.z) { lo = hi = 0.f; }
                else
                    for (int64_t z = z0; z < z0 + 12; ++z)
                        for (int64_t y = y0; y < y0 + 12; ++y)
                            for (int64_t x = x0; x < x0 + 12; ++x) {
                                const float v = at(x, y, z);
                                lo = std::min(lo, v); hi = std::max(hi, v);
                            }
                const uint16_t hlo = float_to_half_round_down(lo), hhi = float_to_half_round_up(hi);
                out->range(bx, by, bz) = (uint32_t)hlo | ((uint32_t)hhi << 16);
                const float flo = half2float(hlo), fhi = half2float(hhi);
                gmin = std::min(gmin, flo); gmax = std::max(gmax, fhi);
                if (fhi != flo) { alloc[out->range.index(bx, by, bz)] = 1; ++count; }
            }
    out->min_maj = { gmin, gmax };
    out->brick_counter = count;
    const uint64_t per_layer = (uint64_t)nb.x * nb.y;
    const uint32_t layers = (uint32_t)std::max<uint64_t>(1, (count + per_layer - 1) / per_layer);
    if (nb.x > 1023 || nb.y > 1023 || layers > 1023) throw std::runtime_error("to_brick_grid: grid too large for 10-bit brick pointers");
    out->atlas = Buf3D<uint8_t>(nb.x * 8, nb.y * 8, layers * 8);
    // pass 2: quantise allocated bricks, sequential slot order
    uint64_t k = 0;
    for (uint32_t bz = 0; bz < nb.z; ++bz)
        for (uint32_t by = 0; by < nb.y; ++by)
            for (uint32_t bx = 0; bx < nb.x; ++bx) {
                if (!alloc[out->range.index(bx, by, bz)]) { out->indirection(bx, by, bz) = 0; continue; }
                const uint32_t px = (uint32_t)(k % nb.x), py = (uint32_t)((k / nb.x) % nb.y), pz = (uint32_t)(k / per_layer);
                ++k;
                out->indirection(bx, by, bz) = (px << 22) | (py << 12) | (pz << 2);
                const uint32_t rg = out->range(bx, by, bz);
                const float lo = half2float(rg & 0xFFFFu), hi = half2float(rg >> 16);
                const float inv = 255.0f / (hi - lo);
                for (uint32_t z = 0; z < 8; ++z)
                    for (uint32_t y = 0; y < 8; ++y)
                        for (uint32_t x = 0; x < 8; ++x) {
                            const float v = at((int64_t)bx * 8 + x, (int64_t)by * 8 + y, (int64_t)bz * 8 + z);
                            float q = std::floor((v - lo) * inv + 0.5f);
                            q = q < 0.f ? 0.f : (q > 255.f ? 255.f : q);
                            out->atlas(px * 8 + x, py * 8 + y, pz * 8 + z) = (uint8_t)q;
                        }
            }
    // mips
    out->range_mipmaps.reserve(3);
    const Buf3D<uint32_t>* src = &out->range;
    for (int m = 0; m < 3; ++m) {
        const uvec3 s = src->stride;
        Buf3D<uint32_t> dst((s.x + 1) / 2, (s.y + 1) / 2, (s.z + 1) / 2);
        for (uint32_t z = 0; z < dst.stride.z; ++z)
            for (uint32_t y = 0; y < dst.stride.y; ++y)
                for (uint32_t x = 0; x < dst.stride.x; ++x) {
                    float lo = INFINITY, hi = -INFINITY; uint16_t hlo = 0, hhi = 0;
                    for (uint32_t c = 0; c < 8; ++c) {
                        const uint32_t cx = 2 * x + (c & 1), cy = 2 * y + ((c >> 1) & 1), cz = 2 * z + (c >> 2);
                        if (cx >= s.x || cy >= s.y || cz >= s.z) continue;
                        const uint32_t rg = (*src)(cx, cy, cz);
                        const float l = half2float(rg & 0xFFFFu), h = half2float(rg >> 16);
                        if (l < lo) { lo = l; hlo = (uint16_t)(rg & 0xFFFFu); }
                        if (h > hi) { hi = h; hhi = (uint16_t)(rg >> 16); }
                    }
                    dst(x, y, z) = (uint32_t)hlo | ((uint32_t)hhi << 16);
                }
        out->range_mipmaps.push_back(std::move(dst));
        src = &out->range_mipmaps.back();
    }
    return out;
}

}  // namespace vr
