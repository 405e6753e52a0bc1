// grids.h -- the slice of the (un-vendored) voldata library that the renderer consumes:
// Buf3D, Grid, DenseGrid, BrickGrid (+ the .brick reader), Volume and Volume::to_brick_grid.
// Call sites in the reference: src/renderer.cpp:32,61-73,97-98,159-225,230-239; src/main.cpp:42-50,465-472.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "hostmath.h"

namespace vr {

// voldata::Buf3D<T>: linear index (z*stride.y + y)*stride.x + x, handed to glTexImage3D as is (renderer.cpp:161-215)
template <typename T>
struct Buf3D {
    uvec3 stride;
    std::vector<T> data;
    Buf3D() = default;
    Buf3D(uint32_t x, uint32_t y, uint32_t z) : stride{ x, y, z }, data((size_t)x * y * z) {}
    size_t index(uint32_t x, uint32_t y, uint32_t z) const { return ((size_t)z * stride.y + y) * stride.x + x; }
    T& operator()(uint32_t x, uint32_t y, uint32_t z) { return data[index(x, y, z)]; }
    const T& operator()(uint32_t x, uint32_t y, uint32_t z) const { return data[index(x, y, z)]; }
};

struct Grid {
    mat4 transform;                                          // index space -> model space
    virtual ~Grid() = default;
    virtual uvec3 index_extent() const = 0;
    virtual std::pair<float, float> minorant_majorant() const = 0;
    virtual float lookup(uint32_t x, uint32_t y, uint32_t z) const = 0;
};

// voldata::DenseGrid(w, h, d, const float*) (main.cpp:470): x fastest
struct DenseGrid : Grid {
    uvec3 dim;
    std::vector<float> voxels;
    std::pair<float, float> min_maj{ 0.f, 0.f };
    DenseGrid(uint32_t w, uint32_t h, uint32_t d, const float* data);
    uvec3 index_extent() const override { return dim; }
    std::pair<float, float> minorant_majorant() const override { return min_maj; }
    float lookup(uint32_t x, uint32_t y, uint32_t z) const override { return voxels[((size_t)z * dim.y + y) * dim.x + x]; }
};

// Dense fp16 grid (north_star: "dense fp16 grid ... lives in HBM3E"): voxels stay dense, x fastest; the DDA still needs
// local majorants, so 8^3 macro cells (dilated by 2 voxels for the tricubic taps) and their 3 min/max mips are derived
// on construction.  No reference counterpart: the reference turns every grid into a BrickGrid at commit().
struct DenseGridF16 : Grid {
    uvec3 dim;
    std::vector<uint16_t> voxels;                // IEEE binary16
    std::pair<float, float> min_maj{ 0.f, 0.f };
    Buf3D<uint32_t> range;                       // per 8^3 cell: 2 x fp16 (min, max), ceil(dim / 8) cells per axis
    std::vector<Buf3D<uint32_t>> range_mipmaps;
    DenseGridF16(uint32_t w, uint32_t h, uint32_t d, const uint16_t* data);
    uvec3 index_extent() const override { return dim; }
    std::pair<float, float> minorant_majorant() const override { return min_maj; }
    float lookup(uint32_t x, uint32_t y, uint32_t z) const override { return half2float(voxels[((size_t)z * dim.y + y) * dim.x + x]); }
};

// voldata::BrickGrid, fields as in the .brick file
struct BrickGrid : Grid {
    uvec3 n_bricks;
    std::pair<float, float> min_maj{ 0.f, 0.f };
    uint64_t brick_counter = 0;
    Buf3D<uint32_t> indirection;                 // GL_RGB10_A2UI: ptr.x = v>>22, ptr.y = (v>>12)&1023, ptr.z = (v>>2)&1023
    Buf3D<uint32_t> range;                       // 2 x fp16: low = min, high = max
    Buf3D<uint8_t> atlas;                        // unorm8 voxels, 8^3 per brick
    std::vector<Buf3D<uint32_t>> range_mipmaps;  // (min of mins, max of maxes) over 2x2x2 children

    explicit BrickGrid() = default;
    explicit BrickGrid(const std::string& path);             // .brick reader; throws std::runtime_error
    uvec3 index_extent() const override { return { n_bricks.x * 8, n_bricks.y * 8, n_bricks.z * 8 }; }
    std::pair<float, float> minorant_majorant() const override { return min_maj; }
    float lookup(uint32_t x, uint32_t y, uint32_t z) const override;   // common.glsl:268-275
    void write(const std::string& path) const;               // same container format
};

struct Volume {
    using GridPtr = std::shared_ptr<Grid>;
    using GridFrame = std::map<std::string, GridPtr>;

    std::vector<GridFrame> grids;
    mat4 transform;                    // model -> world (set by scale_and_move_to_unit_cube)
    size_t grid_frame_counter = 0;

    Volume() = default;
    explicit Volume(const std::string& path);                 // single .brick / .dense / .raw file as frame 0, grid "density"
    explicit Volume(const GridPtr& density) { add_grid_frame(density, "density"); }

    void clear() { grids.clear(); grid_frame_counter = 0; }
    size_t n_grid_frames() const { return grids.size(); }
    GridPtr current_grid(const std::string& name = "density") const { return grids.at(grid_frame_counter).at(name); }
    void add_grid_frame(const GridPtr& grid, const std::string& name = "density") { grids.push_back({ { name, grid } }); }
    void update_grid_frame(size_t i, const GridPtr& grid, const std::string& name) { grids.at(i)[name] = grid; }
    // folder of .brick files in alphanumerical order = animation frames (main.cpp:40-42)
    static std::shared_ptr<Volume> load_folder(const std::string& path);

    std::pair<vec3, vec3> AABB(const std::string& name = "density") const;          // world space
    std::pair<float, float> minorant_majorant(const std::string& name = "density") const { return current_grid(name)->minorant_majorant(); }
    std::string to_string(const std::string& indent = "") const;

    // dense -> brick encoder (a BrickGrid passes through unchanged)
    static std::shared_ptr<BrickGrid> to_brick_grid(const GridPtr& grid);
};

// this build's ".dense" container (see grids.cpp): u8 voxels, value = lo + u8 / 255 * (hi - lo)
void write_dense_file(const std::string& path, const mat4& transform, uint32_t nx, uint32_t ny, uint32_t nz, float lo, float hi, const uint8_t* voxels);

uint16_t float_to_half_round_down(float f);   // largest fp16 <= f
uint16_t float_to_half_round_up(float f);     // smallest fp16 >= f

}  // namespace vr
