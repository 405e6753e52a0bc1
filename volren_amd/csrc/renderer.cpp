// renderer.cpp -- RendererHIP (see renderer.h).  Host marshalling follows src/renderer.cpp:29-50 (init),
// :56-76 (commit), :78-145 (trace), :159-225 (grid upload), :227-242 (unit cube).
#include "renderer.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "vr_device.h"
#include "vr_math.h"

namespace vr {

mat3 Camera::view_inverse() const {
    const vec3 f = normalize(dir);
    const vec3 s = normalize(cross(f, up));
    const vec3 u = cross(s, f);
    mat3 r(0.0f);
    r.m[0] = s.x; r.m[1] = s.y; r.m[2] = s.z;
    r.m[3] = u.x; r.m[4] = u.y; r.m[5] = u.z;
    r.m[6] = -f.x; r.m[7] = -f.y; r.m[8] = -f.z;
    return r;
}

RendererHIP::~RendererHIP() {
    // samples recorded by trace() and never looked at are dropped with the framebuffer
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    for (hipEvent_t e : pt_events_) (void)hipEventDestroy(e);
}

void RendererHIP::init() {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        throw std::runtime_error("RendererHIP::init: no HIP device available (this renderer has no CPU path)");
    if (!volume) volume = std::make_shared<Volume>();
    if (!environment) {
        const float white[3] = { 1.f, 1.f, 1.f };             // renderer.cpp:36-38: 1x1 white background
        environment = std::make_shared<Environment>(white, 1, 1);
    }
    if (!ev0_) { VR_HIP(hipEventCreate(&ev0_)); VR_HIP(hipEventCreate(&ev1_)); }
    if (!status_) {
        status_ = make_device_buffer(16 * sizeof(uint32_t));         // [0] watchdog flag, [1..8] work-queue heads (one per XCD segment)
        VR_HIP(hipMemset(status_->get(), 0, 16 * sizeof(uint32_t)));
    }
    if (!color && resolution.x > 0 && resolution.y > 0) resize((uint32_t)resolution.x, (uint32_t)resolution.y);
}

void RendererHIP::resize(uint32_t w, uint32_t h) {
    if (w == 0 || h == 0) throw std::runtime_error("RendererHIP::resize: empty framebuffer");
    pending_n_ = 0; pending_ = LaunchInputs{};                   // samples recorded for the framebuffer that goes: nobody can see them any more
    resolution = { (int)w, (int)h };
    color = make_device_buffer((size_t)w * h * 4 * sizeof(float));
    display.reset();
    VR_HIP(hipMemset(color->get(), 0, color->size_bytes()));
    if (!tiles_host_.empty()) set_tiles(tiles_host_);
}

void RendererHIP::reset() { sample = 0; }

// ---------------------------------------------------------------------------------------------------
static void check_grid_bytes(size_t bytes, const char* what, const int32_t nb[3]);
constexpr size_t kBlockedMajorantBricks = (size_t)1 << 18;       // active bricks above which commit() picks the blocked majorant layout (512 KiB of fp16 level-0 cells: 1/8 of an XCD's L2)
static size_t brick_records(const BrickGridHIP& g);

void RendererHIP::commit() {
    flush_pending();                                              // recorded samples read the device grids that are about to be replaced
    rate_samples_per_ms_ = 0.0; rate_pending_samples_ = 0.0;      // a new volume: nothing measured yet
    density_grids.clear();
    emission_grids.clear();
    majorant_emission = 0.f;
    maj_key_ = MajKey{};
    std::cout << "Preparing brick grids for HIP..." << std::endl;
    std::vector<size_t> with_emission;                              // frames that have an emission grid (index into both vectors' tails)
    for (const auto& frame : volume->grids) {
        Volume::GridPtr density_grid = frame.at("density");       // throws std::out_of_range like the reference
        density_grids.push_back(grid_to_device(density_grid));
        Volume::GridPtr emission_grid;
        for (const char* name : { "flame", "flames", "temperature" }) {
            auto it = frame.find(name);
            if (it != frame.end()) { emission_grid = it->second; break; }
        }
        if (emission_grid) {
            emission_grids.push_back(grid_to_device(emission_grid));
            majorant_emission = std::max(majorant_emission, emission_grid->minorant_majorant().second);
            if (emission_grids.size() == density_grids.size()) with_emission.push_back(density_grids.size() - 1);
        }
    }
    // Second pass, after every grid the frames NEED is on the device (built inside the loop, the paired atlases of early frames could take the memory
    // a later frame's mandatory upload needs, and an animation that loaded before would fail commit()): for a frame whose density and emission grids are both in
    // brick form with the same brick layout, one paired atlas for the kernel compiled for that case (vr_scene.h).  The grids keep their own atlases too
    // (float-atlas decoder, the run-time variant); when a paired one does not fit, the run-time variant serves that frame.
    for (size_t f : with_emission) {
        BrickGridHIP& gd = density_grids[f];
        BrickGridHIP& ge = emission_grids[f];
        if (!(VR_PAIRED_ATLAS && VR_BRICK_HEADERS && gd.atlas && ge.atlas && !gd.dense && !ge.dense && gd.nb[0] == ge.nb[0] && gd.nb[1] == ge.nb[1] && gd.nb[2] == ge.nb[2])) continue;
        try {
            const size_t n_rec = brick_records(gd);
            check_grid_bytes(n_rec * kPairBlockBytes, "the paired density + emission atlas", gd.nb);
            auto paired = make_device_buffer(n_rec * kPairBlockBytes);
            launch_pair_atlas(gd.atlas->as<uint8_t>(), ge.atlas->as<uint8_t>(), paired->as<uint8_t>(), n_rec, stream);
            VR_HIP(hipGetLastError());
            VR_HIP(hipStreamSynchronize(stream));
            gd.atlas_paired = ge.atlas_paired = paired;
            // Layout of the density grid's majorant table for the kernel that serves this frame.  Levels 0-1 in 4x4x4-cell blocks of one cache line keep a DDA
            // walk's neighbouring cells in the line it has just fetched; that pays on large grids whose bricks are filled in bulk (BASELINE configs[4] as specified,
            // 350 000 active bricks of 2 M: +3 %) and costs 2 % where few bricks are active or the table is cache resident anyway (the 63 000-brick stand-in
            // grid, smoke.brick): profiles/r4d_blocked_majorants_nt_stores_unit_size.txt, profiles/r5_majorant_layout_per_grid.txt
            gd.maj_blocked = gd.n_active > kBlockedMajorantBricks;
        } catch (const std::exception& e) {
            (void)hipGetLastError();
            std::cerr << "volren_amd: no room for the paired density + emission atlas of frame " << f << " (" << e.what() << "): the run-time kernel variant serves it" << std::endl;
        }
    }
}

BrickGridHIP RendererHIP::grid_to_device(const Volume::GridPtr& grid) {
    if (auto f16 = std::dynamic_pointer_cast<DenseGridF16>(grid)) return dense_grid_to_device(f16);
    if (gpu_encoder)
        if (auto dense = std::dynamic_pointer_cast<DenseGrid>(grid)) return dense_to_bricks_on_device(dense);
    return brick_grid_to_device(Volume::to_brick_grid(grid));
}

// HBM a grid may take on the device before the upload is refused with a clear message instead of an out-of-memory error from deep inside
// (the brick-linear atlas is 512 bytes per brick of the grid's BOX, allocated or not: 1 GiB for 1024^3 voxels; the decoded float atlas of
// a transfer-function render is four times that).  Default: what the device reports free, less a reserve for framebuffers and pools.
static void check_grid_bytes(size_t bytes, const char* what, const int32_t nb[3]) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
    const size_t reserve = (size_t)1 << 30;
    if (bytes + reserve > free_b)
        throw std::runtime_error(std::string("grid upload: ") + what + " of a grid of " + std::to_string(nb[0]) + " x " + std::to_string(nb[1]) + " x " + std::to_string(nb[2]) +
                                 " bricks needs " + std::to_string(bytes >> 20) + " MiB of device memory, " + std::to_string(free_b >> 20) + " MiB are free");
}
// padded power-of-two extent of the majorant levels (vr_scene.h); brick records and atlas blocks have exact pitches
static void set_layout(BrickGridHIP& out) {
    for (int i = 0; i < 3; ++i) out.mshift[i] = std::max(3, ceil_log2((uint32_t)out.nb[i]));
    // (4e8 bricks: a tap addresses its cache line of the atlas by a 32-bit line index, 5 lines per brick -- 10 in a paired atlas --: vr_trace.h tap_load; such a
    // grid's atlas alone is 256 GB)
    if (out.mshift[0] + out.mshift[1] + out.mshift[2] > 30 || (uint64_t)out.nb[0] * out.nb[1] * out.nb[2] > 400000000ull)
        throw std::runtime_error("grid upload: more than 4e8 bricks");
    if ((uint64_t)out.nb[1] * out.nb[2] >= (1ull << 24) || out.nb[0] >= (1 << 24))
        throw std::runtime_error("grid upload: brick counts beyond the 24-bit index arithmetic of the kernels (n_bricks.y * n_bricks.z < 2^24)");
    const size_t cells = majorant_table_cells((uint32_t)(out.mshift[0] + out.mshift[1] + out.mshift[2]));
    out.majorant = make_device_buffer(cells * sizeof(float));
    out.majorant16 = make_device_buffer(cells * sizeof(uint16_t));
}
static size_t brick_records(const BrickGridHIP& g) { return (size_t)g.nb[0] * g.nb[1] * g.nb[2]; }

static void upload_range_words(BrickGridHIP& out, const uvec3 nb, const Buf3D<uint32_t>& range, const std::vector<Buf3D<uint32_t>>& mips) {
    if (mips.size() > 3) throw std::runtime_error("grid upload: at most 3 range mips are supported");
    std::vector<uint32_t> words(range.data);
    out.n_mips = (int)mips.size();
    for (int m = 0; m <= 3; ++m) out.mip_off[m] = 0;
    for (int m = 1; m <= out.n_mips; ++m) {
        const auto& mm = mips[m - 1];
        const uint32_t rnd = (1u << m) - 1u;
        if (mm.stride.x != ((nb.x + rnd) >> m) || mm.stride.y != ((nb.y + rnd) >> m) || mm.stride.z != ((nb.z + rnd) >> m))
            throw std::runtime_error("grid upload: range mip dimensions are not ceil(n_bricks / 2^m)");
        out.mip_off[m] = (int32_t)words.size();
        words.insert(words.end(), mm.data.begin(), mm.data.end());
    }
    out.n_cells = (int32_t)words.size();
    out.range_words = make_device_buffer(words.size() * sizeof(uint32_t));
    out.range_words->upload(words.data(), words.size() * sizeof(uint32_t));
    set_layout(out);
}

BrickGridHIP RendererHIP::dense_to_bricks_on_device(const std::shared_ptr<DenseGrid>& g) {
    BrickGridHIP out;
    auto up8 = [](uint32_t v) { return ((v + 7u) / 8u + 7u) / 8u * 8u; };
    const int32_t dim[3] = { (int32_t)g->dim.x, (int32_t)g->dim.y, (int32_t)g->dim.z };
    const int32_t nb[3] = { (int32_t)up8(g->dim.x), (int32_t)up8(g->dim.y), (int32_t)up8(g->dim.z) };
    const size_t n = (size_t)nb[0] * nb[1] * nb[2];
    for (int i = 0; i < 3; ++i) out.nb[i] = nb[i];
    out.transform = g->transform;
    DeviceBuffer dense(g->voxels.size() * sizeof(float)), flag(n * sizeof(uint32_t));
    dense.upload(g->voxels.data(), g->voxels.size() * sizeof(float), stream);
    // range words of all mips in one buffer (mip m has ceil(nb / 2^m) cells per axis)
    int32_t mdim[4][3];
    size_t total = 0;
    out.n_mips = 3;
    for (int m = 0; m <= 3; ++m) {
        for (int i = 0; i < 3; ++i) mdim[m][i] = (nb[i] + (1 << m) - 1) >> m;
        out.mip_off[m] = (int32_t)total;
        total += (size_t)mdim[m][0] * mdim[m][1] * mdim[m][2];
    }
    out.n_cells = (int32_t)total;
    out.range_words = make_device_buffer(total * sizeof(uint32_t));
    set_layout(out);
    uint32_t* words = out.range_words->as<uint32_t>();
    launch_encode_ranges(dense.as<float>(), dim, nb, words, flag.as<uint32_t>(), stream);
    VR_HIP(hipGetLastError());
    check_grid_bytes(brick_records(out) * (kBrickBlockBytes + sizeof(BrickRec) + 8), "the brick-linear atlas", out.nb);
    out.atlas = make_device_buffer(brick_records(out) * kBrickBlockBytes);   // brick-linear blocks (see brick_grid_to_device)
    VR_HIP(hipMemsetAsync(out.atlas->get(), 0, out.atlas->size_bytes(), stream));
    out.bricks = make_device_buffer(brick_records(out) * sizeof(BrickRec));
    VR_HIP(hipMemsetAsync(out.bricks->get(), 0, out.bricks->size_bytes(), stream));
    out.rng = make_device_buffer(brick_records(out) * 2 * sizeof(float));
    VR_HIP(hipMemsetAsync(out.rng->get(), 0, out.rng->size_bytes(), stream));
    launch_encode_bricks(dense.as<float>(), dim, nb, words, flag.as<uint32_t>(), out.bricks->as<BrickRec>(), out.rng->as<float>(), out.atlas->as<uint8_t>(), stream);
    for (int m = 1; m <= 3; ++m) launch_range_mip(words + out.mip_off[m - 1], mdim[m - 1], words + out.mip_off[m], mdim[m], stream);
    VR_HIP(hipGetLastError());
    VR_HIP(hipStreamSynchronize(stream));
    {   // bricks whose voxels matter (majorant layout choice, commit())
        std::vector<uint32_t> flags(n);
        flag.download(flags.data(), n * sizeof(uint32_t), stream);
        for (uint32_t f : flags) out.n_active += f != 0u;
    }
    return out;
}

static uint64_t fnv1a(const std::vector<uint8_t>& v) {
    uint64_t h = 1469598103934665603ull;
    for (uint8_t b : v) { h ^= b; h *= 1099511628211ull; }
    return h;
}
void RendererHIP::grid_checksums(const BrickGridHIP& g, uint64_t out[3]) const {
    const DeviceBufferPtr bufs[3] = { g.bricks, g.atlas, g.range_words };
    for (int i = 0; i < 3; ++i) {
        out[i] = 0;
        if (!bufs[i]) continue;
        std::vector<uint8_t> h(bufs[i]->size_bytes());
        bufs[i]->download(h.data(), h.size(), stream);
        out[i] = fnv1a(h);
    }
}

BrickGridHIP RendererHIP::dense_grid_to_device(const std::shared_ptr<DenseGridF16>& g) {
    BrickGridHIP out;
    const uvec3 nb = g->range.stride;
    out.nb[0] = (int)nb.x; out.nb[1] = (int)nb.y; out.nb[2] = (int)nb.z;
    out.dim[0] = (int)g->dim.x; out.dim[1] = (int)g->dim.y; out.dim[2] = (int)g->dim.z;
    if (g->dim.x > 65535u || g->dim.y > 65535u || g->dim.z > 65535u) throw std::runtime_error("dense_grid_to_device: more than 65535 voxels along an axis");
    out.transform = g->transform;
    // re-tile [z][y][x] into 4x4x4 blocks of 128 contiguous bytes (vr_scene.h), zero padded
    const uint32_t bx = (g->dim.x + 3u) / 4u, by = (g->dim.y + 3u) / 4u, bz = (g->dim.z + 3u) / 4u;
    if ((uint64_t)bx * by * bz > (1ull << 31)) throw std::runtime_error("dense_grid_to_device: more than 2^31 blocks");
    out.dblk[0] = (int)bx; out.dblk[1] = (int)by;
    std::vector<uint16_t> blocked((size_t)bx * by * bz * 64u, 0);
    for (uint32_t z = 0; z < g->dim.z; ++z)
        for (uint32_t y = 0; y < g->dim.y; ++y) {
            const uint16_t* row = &g->voxels[((size_t)z * g->dim.y + y) * g->dim.x];
            for (uint32_t x = 0; x < g->dim.x; ++x) blocked[dense_blocked_index(x, y, z, bx, by)] = row[x];
        }
    out.dense = make_device_buffer(blocked.size() * sizeof(uint16_t));
    out.dense->upload(blocked.data(), blocked.size() * sizeof(uint16_t));
    upload_range_words(out, nb, g->range, g->range_mipmaps);
    return out;
}

BrickGridHIP RendererHIP::brick_grid_to_device(const std::shared_ptr<BrickGrid>& g) {
    BrickGridHIP out;
    const uvec3 nb = g->n_bricks;
    const size_t n_bricks = (size_t)nb.x * nb.y * nb.z;
    if (n_bricks == 0) throw std::runtime_error("brick_grid_to_device: empty grid");
    out.nb[0] = (int)nb.x; out.nb[1] = (int)nb.y; out.nb[2] = (int)nb.z;
    out.transform = g->transform;
    // atlas: 3D texture of 8^3 blocks addressed through the indirection words -> brick-LINEAR blocks: the block of brick record i is bytes
    // [640 i, 640 i + 640), five cache lines of [rmin, rdiff | 120 voxels] (vr_scene.h), so that a tap fetches range and voxel from one line
    // instead of chasing the pointer (vr_trace.h tap_load).  A brick whose range is a single value keeps zero voxels (they never matter:
    // rmin + u * 0), and so does a pointer outside the atlas (GL: undefined fetch); every line of every brick carries the range.
    const uvec3 ad = g->atlas.stride;
    const uint32_t sx = ad.x / 8, sy = ad.y / 8, sz = ad.z / 8;
    upload_range_words(out, nb, g->range, g->range_mipmaps);        // also fixes the padded majorant layout (mshift)
    check_grid_bytes(brick_records(out) * (kBrickBlockBytes + sizeof(BrickRec) + 8), "the brick-linear atlas", out.nb);
    std::vector<BrickRec> recs(brick_records(out), BrickRec{ 0u, 0.f, 0.f, 0u });
    std::vector<uint8_t> atlas(recs.size() * (size_t)kBrickBlockBytes, 0);
    for (size_t i = 0; i < n_bricks; ++i) {
        const uint32_t ind = g->indirection.data[i], rg = g->range.data[i];
        const uint32_t px = ind >> 22, py = (ind >> 12) & 1023u, pz = (ind >> 2) & 1023u;
        const float lo = half2float(rg & 0xFFFFu), hi = half2float(rg >> 16);
        const size_t idx = i;                                        // record index = linear brick index (x fastest)
        BrickRec r;
        r.slot = (uint32_t)idx;
        r.rmin = lo;
        r.rdiff = hi - lo;
        r.range = rg;
        recs[idx] = r;
        uint8_t* dst = &atlas[idx * (size_t)kBrickBlockBytes];
        if (VR_BRICK_HEADERS)
            for (uint32_t l = 0; l < 5; ++l) { memcpy(dst + l * 128u, &r.rmin, 4); memcpy(dst + l * 128u + 4u, &r.rdiff, 4); }
        if (r.rdiff != 0.f && px < sx && py < sy && pz < sz) {
            ++out.n_active;
            for (uint32_t z = 0; z < 8; ++z)
                for (uint32_t y = 0; y < 8; ++y) {
                    const uint8_t* src = &g->atlas.data[g->atlas.index(px * 8, py * 8 + y, pz * 8 + z)];
                    for (uint32_t x = 0; x < 8; ++x) dst[brick_voxel_byte(z * 64 + y * 8 + x)] = src[x];
                }
        }
    }
    out.bricks = make_device_buffer(recs.size() * sizeof(BrickRec));
    out.bricks->upload(recs.data(), recs.size() * sizeof(BrickRec));
    std::vector<float> rng(recs.size() * 2);
    for (size_t i = 0; i < recs.size(); ++i) { rng[2 * i] = recs[i].rmin; rng[2 * i + 1] = recs[i].rdiff; }
    out.rng = make_device_buffer(rng.size() * sizeof(float));
    out.rng->upload(rng.data(), rng.size() * sizeof(float));
    out.atlas = make_device_buffer(atlas.size());
    out.atlas->upload(atlas.data(), atlas.size());
    return out;
}

void RendererHIP::scale_and_move_to_unit_cube() {
    // max AABB over the whole volume (animation); FLT_MIN start value is the reference's quirk (renderer.cpp:229)
    vec3 bb_min(FLT_MAX), bb_max(FLT_MIN);
    for (const auto& frame : volume->grids) {
        const auto grid = frame.at("density");
        const uvec3 e = grid->index_extent();
        bb_min = vmin(bb_min, transform_point(grid->transform, vec3(0.f, 0.f, 0.f)));
        bb_max = vmax(bb_max, transform_point(grid->transform, vec3((float)e.x, (float)e.y, (float)e.z)));
    }
    const vec3 extent = bb_max - bb_min;
    const float size = std::fmax(extent.x, std::fmax(extent.y, extent.z));
    if (size != 1.f) {
        volume->transform = scale_then_translate(1.f / size, -bb_min - extent * 0.5f);
        density_scale *= size;
    }
}

// ---------------------------------------------------------------------------------------------------
static void copy3(float* dst, vec3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

static GridView make_view(const BrickGridHIP& g, bool paired = false, bool maj_blocked = false) {
    GridView v;
    memset(&v, 0, sizeof v);                     // padding bytes too: launch inputs are compared byte by byte (LaunchInputs::same_launch_as)
    v.bricks = g.bricks ? g.bricks->as<BrickRec>() : nullptr;
    v.atlas = paired ? g.atlas_paired->as<uint8_t>() : (g.atlas ? g.atlas->as<uint8_t>() : nullptr);
    v.dense = g.dense ? g.dense->as<uint16_t>() : nullptr;
    for (int i = 0; i < 3; ++i) v.dim[i] = g.dim[i];
    for (int i = 0; i < 2; ++i) v.dblk[i] = g.dblk[i];
    v.majorant = g.majorant->as<float>();
    v.majorant16 = g.majorant16->as<uint16_t>();
    v.rng = g.rng ? g.rng->as<float>() : nullptr;
    v.atlas_f32 = g.atlas_f32 ? g.atlas_f32->as<float>() : nullptr;
    for (int i = 0; i < 3; ++i) v.nb[i] = g.nb[i];
    for (int i = 0; i < 3; ++i) { v.mshift[i] = g.mshift[i]; v.mlim[i] = (float)(8u << g.mshift[i]); }
    v.n_mips = g.n_mips;
    v.maj_blocked = maj_blocked ? 1 : 0;         // the layout the majorant table is (re)built in and the kernel variant reads (vr_kernels.hip pathtrace_variant)
    v.maj_outside = (int32_t)majorant_padded_cells((uint32_t)(g.mshift[0] + g.mshift[1] + g.mshift[2]));
    return v;
}

void RendererHIP::fill_params(SceneParams& P) {
    if (!volume || volume->grids.empty() || density_grids.empty())
        throw std::runtime_error("RendererHIP::trace: no volume committed");
    if (volume->grid_frame_counter >= density_grids.size())
        throw std::runtime_error("RendererHIP::trace: grid_frame_counter out of range (commit() after changing the volume)");
    memset(&P, 0, sizeof P);
    Uniforms& u = P.u;
    u.bounces = bounces;
    u.seed = seed;
    u.show_environment = show_environment ? 1 : 0;
    // camera
    copy3(u.cam_pos, camera.pos);
    u.cam_fov = camera.fov_degree;
    const mat3 ct = camera.view_inverse();
    memcpy(u.cam_transform, ct.m, sizeof ct.m);
    P.cam_z = -0.5f / tan_(0.5f * kPi * camera.fov_degree / 180.f);
    // volume
    const auto [bb_min, bb_max] = volume->AABB();
    const auto [mn, maj] = volume->minorant_majorant();
    copy3(u.vol_bb_min, bb_min + vol_clip_min * (bb_max - bb_min));
    copy3(u.vol_bb_max, bb_min + vol_clip_max * (bb_max - bb_min));
    u.vol_minorant = mn * density_scale;
    u.vol_majorant = maj * density_scale;
    u.vol_inv_majorant = 1.f / (maj * density_scale);
    copy3(u.vol_albedo, albedo);
    u.vol_phase_g = phase;
    u.vol_density_scale = density_scale;
    u.vol_emission_scale = emission_scale;
    u.vol_emission_norm = majorant_emission > 0.f ? 1.f / std::fmax(majorant_emission, 1e-4f) : 1.f;
    // density brick grid data
    const BrickGridHIP& density = density_grids[volume->grid_frame_counter];
    const mat4 dt = volume->transform * density.transform;
    const mat4 dti = inverse(dt);
    memcpy(u.vol_density_transform, dt.m, sizeof dt.m);
    memcpy(u.vol_density_inv_transform, dti.m, sizeof dti.m);
    P.density = make_view(density);
    // emission brick grid data
    if (volume->grid_frame_counter < emission_grids.size()) {
        const BrickGridHIP& emission = emission_grids[volume->grid_frame_counter];
        const mat4 et = volume->transform * emission.transform;
        const mat4 eti = inverse(et);
        memcpy(u.vol_emission_transform, et.m, sizeof et.m);
        memcpy(u.vol_emission_inv_transform, eti.m, sizeof eti.m);
        const mat4 efd = eti * dt;
        memcpy(P.emission_from_density, efd.m, sizeof efd.m);
        // the kernel compiled for two brick grids (DDA trackers) reads them from their paired atlas; every other kernel reads each grid's own
        // (an environment that fails the warp table's check is rendered by the run-time variant, from the grids' own atlases: vr_kernels.hip pathtrace_variant)
        const bool scale_ok = u.vol_density_scale >= 1.0f / 65536.0f && u.vol_density_scale <= 16777216.0f;      // (the same goes for a density scale the fixed kernels' march does not divide by)
        P.paired = (integrator == 0 && environment->cdf_div_safe && scale_ok && density.atlas_paired && density.atlas_paired == emission.atlas_paired) ? 1 : 0;
        // ... and is the one kernel compiled for both layouts of the majorant table's fine levels: blocked for the grids commit() marked (or as majorant_layout says)
        if (P.paired) P.density = make_view(density, true, majorant_layout < 0 ? density.maj_blocked : majorant_layout == 1);
        P.emission = make_view(emission, P.paired != 0);
        u.has_emission = 1;
    }
    // transfer function
    if (transferfunc) {
        u.use_tf = 1;
        u.tf_size = transferfunc->size();
        u.tf_window_left = transferfunc->window_left;
        u.tf_window_width = transferfunc->window_width;
        P.tf_lut = transferfunc->lut_ssbo->as<float>();
    }
    // environment
    const mat3 eit = inverse(environment->transform);
    memcpy(u.env_transform, environment->transform.m, sizeof eit.m);
    memcpy(u.env_inv_transform, eit.m, sizeof eit.m);
    u.env_strength = environment->strength;
    u.env_imp_inv_dim[0] = u.env_imp_inv_dim[1] = 1.f / (float)environment->dimension();
    u.env_imp_base_mip = (int)std::floor(std::log2((float)environment->dimension()));
    P.envmap = environment->envmap->as<float>();
    P.env_rgbe = environment->envmap_rgbe ? environment->envmap_rgbe->as<uint32_t>() : nullptr;
    P.env_w = environment->width; P.env_h = environment->height;
    P.env_avg_w = environment->avg_importance; P.env_avg_w_set = 1;
    P.impmap = environment->impmap->as<float>();
    P.imp_dim = (int)environment->dimension();
    P.env_cdf = environment->cdf->as<float>();
    P.env_div_safe = environment->cdf_div_safe ? 1 : 0;
    u.resolution[0] = resolution.x; u.resolution[1] = resolution.y;
    u.integrator = integrator;
}

void RendererHIP::update_majorants(const LaunchInputs& in, BrickGridHIP& g) {
    const MajKey& k = in.maj;
    if (k.density_scale == maj_key_.density_scale && k.tf_version == maj_key_.tf_version &&
        k.wl == maj_key_.wl && k.ww == maj_key_.ww && k.frame == maj_key_.frame && k.blocked == maj_key_.blocked)
        return;
    launch_majorants(in.P, g.range_words->as<uint32_t>(), g.nb, g.mip_off, g.n_mips, g.mshift, g.majorant->as<float>(), g.majorant16->as<uint16_t>(), in.stream);
    VR_HIP(hipGetLastError());
    maj_key_ = k;
}

void RendererHIP::set_tiles(const std::vector<int32_t>& tile_ids) {
    flush_pending();
    tiles_host_ = tile_ids;
    tiles_dev_.reset();
    if (tile_ids.empty()) return;
    if (resolution.x > 0) {
        const int n_all = ((resolution.x + 15) / 16) * ((resolution.y + 15) / 16);
        for (int32_t t : tile_ids)
            if (t < 0 || t >= n_all) throw std::runtime_error("set_tiles: tile id out of range");   // never launch out-of-bounds tiles
    }
    tiles_dev_ = make_device_buffer(tile_ids.size() * sizeof(int32_t));
    tiles_dev_->upload(tile_ids.data(), tile_ids.size() * sizeof(int32_t));
}

// The order in which a launch works through its tiles.  The persistent wavefronts pull work units from a queue (vr_pathtrace.h) whose positions enumerate the
// tile list front to back in 8 contiguous segments, one per XCD; when the queue runs empty every wavefront still has to finish the paths in its pool, and
// the launch ends with its deepest path (4-5 ms on the bench scene: 2 % of a frame on one GPU, 15 % of a rank's share on eight).  With the costly tiles --
// long chords through the volume's box -- at the FRONT of every segment and the tiles whose rays miss the box at its end, what is left at that point are
// camera rays that escape at once.  Cost estimate: the longest chord of five rays of the tile (corners and centre, no jitter) through the clipped box.
// Sorted (stably) inside each eighth of the list, so that an XCD keeps its band of tile rows.  Cached per (camera, box, frame size, tile set).
const int32_t* RendererHIP::tile_order(const SceneParams& P, int n_tiles) {
    const Uniforms& u = P.u;
    uint64_t key = 1469598103934665603ull;
    auto mix = [&key](const void* p, size_t nbytes) { const uint8_t* b = static_cast<const uint8_t*>(p); for (size_t i = 0; i < nbytes; ++i) { key ^= b[i]; key *= 1099511628211ull; } };
    mix(u.cam_pos, sizeof u.cam_pos); mix(u.cam_transform, sizeof u.cam_transform); mix(&P.cam_z, sizeof P.cam_z);
    mix(u.vol_bb_min, sizeof u.vol_bb_min); mix(u.vol_bb_max, sizeof u.vol_bb_max); mix(u.resolution, sizeof u.resolution); mix(&n_tiles, sizeof n_tiles);
    if (!tiles_host_.empty()) mix(tiles_host_.data(), tiles_host_.size() * sizeof(int32_t));
    key = key ? key : 1;
    if (order_dev_ && order_key_ == key && order_dev_->size_bytes() == (size_t)n_tiles * sizeof(int32_t)) return order_dev_->as<int32_t>();
    const int W = u.resolution[0], H = u.resolution[1], tiles_x = (W + 15) / 16;
    std::vector<int32_t> ids(tiles_host_);
    if (ids.empty()) { ids.resize((size_t)n_tiles); for (int i = 0; i < n_tiles; ++i) ids[(size_t)i] = i; }
    auto chord = [&](float px, float py) {
        float d[3] = { (px - 0.5f * (float)W) / (float)H, (py - 0.5f * (float)H) / (float)H, P.cam_z };
        float w[3];
        for (int i = 0; i < 3; ++i) w[i] = u.cam_transform[i] * d[0] + u.cam_transform[3 + i] * d[1] + u.cam_transform[6 + i] * d[2];      // direction up to its length: only t-ratios matter
        const float len = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        float t0 = 0.f, t1 = FLT_MAX;
        for (int i = 0; i < 3; ++i) {
            const float inv = 1.f / (w[i] / len);
            const float a = (u.vol_bb_min[i] - u.cam_pos[i]) * inv, b = (u.vol_bb_max[i] - u.cam_pos[i]) * inv;
            t0 = std::fmax(t0, std::fmin(a, b)); t1 = std::fmin(t1, std::fmax(a, b));
        }
        return t1 > t0 ? t1 - t0 : 0.f;
    };
    std::vector<float> cost(ids.size());
    for (size_t k = 0; k < ids.size(); ++k) {
        const float x0 = (float)((ids[k] % tiles_x) * 16), y0 = (float)((ids[k] / tiles_x) * 16);
        const float x1 = std::fmin(x0 + 16.f, (float)W), y1 = std::fmin(y0 + 16.f, (float)H);
        cost[k] = std::fmax(std::fmax(std::fmax(chord(x0, y0), chord(x1, y0)), std::fmax(chord(x0, y1), chord(x1, y1))), chord(0.5f * (x0 + x1), 0.5f * (y0 + y1)));
    }
    std::vector<int32_t> perm(ids.size());
    for (size_t k = 0; k < perm.size(); ++k) perm[k] = (int32_t)k;
    const size_t seg = (ids.size() + 7) / 8;                                  // the queue's segments are eighths of the unit list = of the tile list
    for (size_t b = 0; b < ids.size(); b += seg)
        std::stable_sort(perm.begin() + (ptrdiff_t)b, perm.begin() + (ptrdiff_t)std::min(ids.size(), b + seg), [&](int32_t x, int32_t y) { return cost[(size_t)x] > cost[(size_t)y]; });
    std::vector<int32_t> ordered(ids.size());
    for (size_t k = 0; k < ids.size(); ++k) ordered[k] = ids[(size_t)perm[k]];
    if (!order_dev_ || order_dev_->size_bytes() != ordered.size() * sizeof(int32_t)) order_dev_ = make_device_buffer(ordered.size() * sizeof(int32_t));
    order_dev_->upload(ordered.data(), ordered.size() * sizeof(int32_t), stream);
    order_key_ = key;
    return order_dev_->as<int32_t>();
}

bool RendererHIP::LaunchInputs::same_launch_as(const LaunchInputs& o) const {
    return memcmp(&P, &o.P, sizeof P) == 0 && frame == o.frame && maj.density_scale == o.maj.density_scale && maj.tf_version == o.maj.tf_version &&
           maj.wl == o.maj.wl && maj.ww == o.maj.ww && maj.blocked == o.maj.blocked && memcmp(tuning.thr, o.tuning.thr, sizeof tuning.thr) == 0 && tuning.stats == o.tuning.stats &&
           tuning.samples_per_unit == o.tuning.samples_per_unit && tuning.blocks_per_cu == o.tuning.blocks_per_cu && order_tiles == o.order_tiles &&
           launch_target_ms == o.launch_target_ms && fast_math == o.fast_math && sample_pool_bytes == o.sample_pool_bytes && stream == o.stream;
}

// What a launch reads from the renderer's public, freely mutable state -- taken when trace() / render() is CALLED (the reference issues its dispatch at that
// point: src/renderer.cpp:78-145), used when the samples are launched.
void RendererHIP::capture(LaunchInputs& in) {
    if (!color) throw std::runtime_error("RendererHIP::trace: no framebuffer (call resize first)");
    if (integrator == 2 && !transferfunc) throw std::runtime_error("RendererHIP::trace: integrator 2 (direct volume rendering) needs a transfer function");
    if (integrator < 0 || integrator > 3) throw std::runtime_error("RendererHIP::trace: unknown integrator");
    // the tolerance mode is offered where it stays within 1e-3 relative L2 of the reference's kernels; behind a transfer function it does
    // not (a dark image carried by a few bright pixels: 1.9e-3 at 64x48x1024 spp), so it is refused there rather than shipped
    if (fast_math && transferfunc) throw std::runtime_error("RendererHIP::trace: fast_math is not available while a transfer function is bound (it misses the 1e-3 bound there); set fast_math 0");
    if (!volume || volume->grids.empty() || density_grids.empty()) throw std::runtime_error("RendererHIP::trace: no volume committed");
    if (volume->grid_frame_counter >= density_grids.size()) throw std::runtime_error("RendererHIP::trace: grid_frame_counter out of range (commit() after changing the volume)");
    {   // transfer-function renders of brick grids read a decoded float atlas (vr_trace.h trilinear_load): build it on first use.
        // 4 bytes per voxel of every brick; when that does not fit, the byte atlas keeps serving (same values either way).
        BrickGridHIP& g = density_grids[volume->grid_frame_counter];
        const bool build = transferfunc && tf_float_atlas && !g.dense && g.atlas && g.rng && !g.atlas_f32 && !g.atlas_f32_failed;
        const bool drop = (!transferfunc || !tf_float_atlas) && (g.atlas_f32 || g.atlas_f32_failed);
        if (build || drop) flush_pending();                          // recorded samples may read the atlas that goes, or need the memory that comes
        if (build) {
            try {
                // decoded atlases of the other animation frames stay while they fit: a looping animation behind a transfer function (the reference cycles
                // its frames) decodes every frame once, not at every frame change (hipFree synchronises the device).  Budget: half of what the device
                // has free, and at the latest when the allocation fails, all others go (least recently decoded = lowest index first: no LRU bookkeeping)
                const size_t n_blocks = g.atlas->size_bytes() / kBrickBlockBytes, want = n_blocks * 512u * sizeof(float);
                auto drop_others = [&] { for (BrickGridHIP& other : density_grids) if (&other != &g) other.atlas_f32.reset(); };
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = ~(size_t)0; }
                if (want > free_b / 2) drop_others();
                try { g.atlas_f32 = make_device_buffer(want); }
                catch (const std::exception&) { (void)hipGetLastError(); drop_others(); g.atlas_f32 = make_device_buffer(want); }
                launch_decode_atlas(g.rng->as<float>(), g.atlas->as<uint8_t>(), g.atlas_f32->as<float>(), n_blocks, stream);
                VR_HIP(hipGetLastError());
            } catch (const std::exception& e) {
                (void)hipGetLastError();
                g.atlas_f32.reset();
                g.atlas_f32_failed = true;      // not retried per launch: a per-sample trace() loop would pay a failing multi-GB hipMalloc every call
                std::cerr << "volren_amd: no room for the decoded float atlas (" << (g.atlas->size_bytes() / kBrickBlockBytes * 2048u >> 20) << " MiB): transfer-function taps read the byte atlas (" << e.what() << ")" << std::endl;
            }
        }
        if (drop) { g.atlas_f32.reset(); g.atlas_f32_failed = false; }
    }
    fill_params(in.P);
    in.frame = volume->grid_frame_counter;
    in.maj = MajKey{};
    in.maj.density_scale = density_scale;
    in.maj.tf_version = transferfunc ? transferfunc->version : 0;
    in.maj.wl = transferfunc ? transferfunc->window_left : 0.f;
    in.maj.ww = transferfunc ? transferfunc->window_width : 0.f;
    in.maj.frame = in.frame;
    in.maj.blocked = in.P.density.maj_blocked;
    in.tuning = tuning;
    in.order_tiles = order_tiles; in.launch_target_ms = launch_target_ms; in.fast_math = fast_math ? 1 : 0;
    in.sample_pool_bytes = sample_pool_bytes;
    in.stream = stream;
    in.env = environment; in.tf = transferfunc;
    in.keep[0] = environment->envmap; in.keep[1] = environment->impmap; in.keep[2] = environment->cdf;
    in.keep[3] = transferfunc ? transferfunc->lut_ssbo : DeviceBufferPtr();
    in.keep[4] = environment->envmap_rgbe;
}

// samples of one sub-launch as the sample pool and the 32-bit item indices allow (before any sizing by time)
int RendererHIP::samples_per_launch(const LaunchInputs& in, int n_tiles) const {
    const size_t per_sample = pathtrace_pool_floats(in.tuning, n_tiles, 1) * sizeof(float);
    int per_launch = (int)std::max<size_t>(1, in.sample_pool_bytes / per_sample);
    // item indices inside a sub-launch are 32-bit (WorkUnit::base, C_ITEM): keep n_tiles * 256 * (per_launch rounded up to whole units of at most 8 samples)
    // below 2^32.  Saturating: for frames beyond 2^27 pixels the quotient is smaller than the margin (the unsigned difference wrapped and the clamp
    // stopped limiting anything); launch_pathtrace refuses a launch whose items do not fit.
    const size_t lim = (size_t)0xFFFFFFFFu / ((size_t)n_tiles * 256u);
    per_launch = (int)std::min<size_t>((size_t)per_launch, lim > 8 ? lim - 8 : 1);
    if (per_launch > 32) per_launch -= per_launch % 32;          // whole sample chunks (32 is a multiple of every unit size)
    return per_launch;
}

void RendererHIP::submit(const LaunchInputs& in, int first, int n) {
    if (n <= 0) return;
    const SceneParams& P = in.P;
    hipStream_t stream = in.stream;                              // (shadows the member: the launch goes where it was recorded for)
    const bool has_tf = P.u.use_tf != 0;
    update_majorants(in, density_grids.at(in.frame));
    const int tiles_x = (P.u.resolution[0] + 15) / 16, tiles_y = (P.u.resolution[1] + 15) / 16;
    const int n_tiles = tiles_dev_ ? (int)tiles_host_.size() : tiles_x * tiles_y;
    const bool ordered = in.order_tiles >= 2 || (in.order_tiles == 1 && tiles_dev_);
    const int32_t* tiles = ordered ? tile_order(P, n_tiles) : (tiles_dev_ ? tiles_dev_->as<int32_t>() : nullptr);
    // per-sample radiances live in a device pool; split the request so that one sub-launch fits the pool
    int per_launch = std::min(samples_per_launch(in, n_tiles), n);
    for (;;) {                                                   // a pool that does not fit the free HBM: halve the sub-launch, never fail for it
        const size_t need = pathtrace_pool_floats(in.tuning, n_tiles, per_launch) * sizeof(float);
        if (pool_ && pool_->size_bytes() >= need) break;
        if (pool_) { VR_HIP(hipStreamSynchronize(stream)); pool_.reset(); }
        try { pool_ = make_device_buffer(need); break; }
        catch (const std::exception&) {
            (void)hipGetLastError();
            if (per_launch <= 1) throw;
            per_launch = std::max(1, per_launch / 2);
            if (per_launch > 32) per_launch -= per_launch % 32;
        }
    }
    if (!workspace_) {
        workspace_ = make_device_buffer(pathtrace_workspace_floats() * sizeof(float));
        // test hook: no path may depend on what its cold line held before the path wrote it (tests/test_gpu_parity.py)
        if (const char* e = std::getenv("VR_TEST_POISON_WORKSPACE"); e && *e == '1') VR_HIP(hipMemsetAsync(workspace_->get(), 0xFF, workspace_->size_bytes(), stream));
    }
    // test hook, value 2: NaN patterns in the workspace AND the sample pool before EVERY submit -- a frame must not depend on what an earlier frame (of this or of
    // another mode's kernels) left in either (tests/tools_determinism.py, test_frames_are_reproducible_on_every_compiled_instance)
    if (const char* e = std::getenv("VR_TEST_POISON_WORKSPACE"); e && *e == '2') {
        VR_HIP(hipMemsetAsync(workspace_->get(), 0xFF, workspace_->size_bytes(), stream));
        VR_HIP(hipMemsetAsync(pool_->get(), 0xFF, pool_->size_bytes(), stream));
    }
    const int integ = P.u.integrator;
    const bool pt_kernel = !(integ == 3 || (integ == 2 && has_tf));      // launch_pathtrace records the events around the path-tracing kernel only
    // Launch sizing by time (round 4): no sub-launch is PLANNED to take longer than launch_target_ms.  The plan uses the rate of this renderer's last
    // finished sub-launch; when there is none for the current settings and the request is large, a short probe launch measures it first (one
    // synchronisation, once per change of settings).  A plan can still be off -- the rate is a property of the scene AND the view -- which costs nothing
    // but the bound: the kernel's watchdog watches progress, not duration (vr_pathtrace.h).
    const double px_samples = (double)n_tiles * 256.0;                          // samples per spp of this launch
    const uint64_t key = [&] {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&h](const void* p, size_t nbytes) { const uint8_t* b = static_cast<const uint8_t*>(p); for (size_t i = 0; i < nbytes; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
        const Uniforms& u = P.u;
        mix(&u.bounces, sizeof u.bounces); mix(u.vol_albedo, sizeof u.vol_albedo); mix(&u.vol_phase_g, sizeof u.vol_phase_g); mix(&u.vol_density_scale, sizeof u.vol_density_scale);
        mix(&u.use_tf, sizeof u.use_tf); mix(&u.tf_window_left, sizeof u.tf_window_left); mix(&u.tf_window_width, sizeof u.tf_window_width); mix(&u.integrator, sizeof u.integrator);
        mix(&u.has_emission, sizeof u.has_emission); mix(u.resolution, sizeof u.resolution); mix(u.vol_bb_min, sizeof u.vol_bb_min); mix(u.vol_bb_max, sizeof u.vol_bb_max);
        const size_t frame = in.frame; mix(&frame, sizeof frame); mix(&n_tiles, sizeof n_tiles);
        const uint64_t tfv = in.maj.tf_version; mix(&tfv, sizeof tfv);
        return h ? h : 1ull;
    }();
    auto cap_by_rate = [&](int planned) {
        if (in.launch_target_ms <= 0 || rate_samples_per_ms_ <= 0.0) return planned;
        const double cap = rate_samples_per_ms_ * (double)in.launch_target_ms / px_samples;
        int c = cap >= (double)planned ? planned : std::max(1, (int)cap);
        if (c > 32) c -= c % 32;
        return c;
    };
    int probe = 0;
    if (pt_kernel && in.launch_target_ms > 0) {
        harvest_rate(false);
        const bool known = rate_samples_per_ms_ > 0.0 && rate_key_ == key;
        if (!known && px_samples * (double)std::min(per_launch, n) > (double)(1u << 26)) {
            rate_samples_per_ms_ = 0.0;                                          // measured under other settings: not a basis for a plan
            probe = (int)std::min<double>(32.0, std::max(1.0, (double)(1u << 24) / px_samples));
            probe = std::min(probe, n);
        } else per_launch = cap_by_rate(per_launch);
    }
    VR_HIP(hipEventRecord(ev0_, stream));
    last_launches = 0;
    pt_events_used_ = 0;
    for (int done = 0; done < n;) {
        ++last_launches;
        const int m = probe > 0 ? probe : std::min(per_launch, n - done);
        hipEvent_t eb = nullptr, ee = nullptr;
        if (pt_kernel) {
            while (pt_events_.size() < pt_events_used_ + 2) { hipEvent_t e; VR_HIP(hipEventCreate(&e)); pt_events_.push_back(e); }
            eb = pt_events_[pt_events_used_]; ee = pt_events_[pt_events_used_ + 1];
            pt_events_used_ += 2;
        }
        launch_pathtrace(in.tuning, P, color->as<float>(), pool_->as<float>(), workspace_->as<float>(), status_->as<uint32_t>() + 1, tiles, n_tiles, first + 1 + done, m, status_->as<uint32_t>(), stream, in.fast_math != 0, eb, ee);
        VR_HIP(hipGetLastError());
        done += m;
        if (pt_kernel) { rate_pending_samples_ = px_samples * (double)m; rate_pending_key_ = key; }
        if (probe > 0) {                                                         // the probe: wait for it, then plan the rest
            probe = 0;
            harvest_rate(true);
            per_launch = cap_by_rate(per_launch);
        }
    }
    VR_HIP(hipEventRecord(ev1_, stream));
    timing_pending_ = true;
}

// rate of the last path-tracing sub-launch that was enqueued, if it has finished (wait: block until it has)
void RendererHIP::harvest_rate(bool wait) {
    if (rate_pending_samples_ <= 0.0 || pt_events_used_ < 2) return;
    hipEvent_t eb = pt_events_[pt_events_used_ - 2], ee = pt_events_[pt_events_used_ - 1];
    if (wait) VR_HIP(hipEventSynchronize(ee));
    else if (hipEventQuery(ee) != hipSuccess) { (void)hipGetLastError(); return; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, eb, ee) != hipSuccess || !(ms > 0.f)) { (void)hipGetLastError(); return; }
    rate_samples_per_ms_ = rate_pending_samples_ / (double)ms;
    rate_key_ = rate_pending_key_;
    rate_pending_samples_ = 0.0;
}

// One more sample (src/renderer.cpp:78-145).  The reference issues one dispatch per call; here a call records its launch inputs and consecutive calls that
// record the same bytes are launched together (renderer.h).  What a caller can observe is unchanged: `sample` advances by one per call, invalid
// state throws at the call, and every way of looking at the frame launches the recorded samples first -- with the values they were recorded with.
void RendererHIP::trace() {
    if (!coalesce_trace) { flush_pending(); LaunchInputs in; capture(in); submit(in, sample, 1); sample += 1; return; }
    LaunchInputs now;
    capture(now);
    if (pending_n_ > 0 && (sample != pending_first_ + pending_n_ || !now.same_launch_as(pending_))) flush_pending();
    if (pending_n_ == 0) {
        pending_ = std::move(now);
        pending_first_ = sample;
        const int n_tiles = tiles_dev_ ? (int)tiles_host_.size() : ((resolution.x + 15) / 16) * ((resolution.y + 15) / 16);
        pending_cap_ = samples_per_launch(pending_, n_tiles);      // a full sub-launch goes out at once: the GPU works while the caller keeps calling
    }
    ++pending_n_;
    ++sample;
    if (pending_n_ >= pending_cap_) flush_pending();
}

void RendererHIP::flush_pending() {
    if (pending_n_ <= 0) return;
    const int n = pending_n_, first = pending_first_;
    pending_n_ = 0;                                              // (first: submit may throw, and must not be retried with half of it enqueued)
    LaunchInputs in = std::move(pending_);
    pending_ = LaunchInputs{};
    submit(in, first, n);
}

void RendererHIP::render(int n) {
    flush_pending();
    if (n <= 0) n = sppx - sample;
    if (n <= 0) return;
    LaunchInputs in;
    capture(in);
    submit(in, sample, n);
    sample += n;
}

void RendererHIP::draw() {
    flush_pending();
    if (!color) return;
    if (!display || display->size_bytes() != color->size_bytes()) display = make_device_buffer(color->size_bytes());
    VR_HIP(hipMemcpyAsync(display->get(), color->get(), color->size_bytes(), hipMemcpyDeviceToDevice, stream));
    if (tonemapping) {
        launch_tonemap(display->as<float>(), resolution.x, resolution.y, tonemap_exposure, tonemap_gamma, stream);
        VR_HIP(hipGetLastError());
    }
}

double RendererHIP::last_kernel_ms() {
    flush_pending();
    if (timing_pending_) {
        VR_HIP(hipEventSynchronize(ev1_));
        float ms = 0.f;
        VR_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
        last_ms_ = (double)ms;
        last_pathtrace_ms_ = 0.0;
        for (size_t i = 0; i + 1 < pt_events_used_; i += 2)
            if (hipEventElapsedTime(&ms, pt_events_[i], pt_events_[i + 1]) == hipSuccess) last_pathtrace_ms_ += (double)ms;
        (void)hipGetLastError();
        timing_pending_ = false;
    }
    return last_ms_;
}

double RendererHIP::last_pathtrace_ms() { (void)last_kernel_ms(); return last_pathtrace_ms_; }

void RendererHIP::sched_stats(bool enable, unsigned long long out[32]) {
    flush_pending();
    if (out) {
        for (int i = 0; i < 32; ++i) out[i] = 0ull;
        if (stats_) { VR_HIP(hipStreamSynchronize(stream)); stats_->download(out, 32 * sizeof(unsigned long long), stream); }
    }
    if (enable) {
        if (!stats_) stats_ = make_device_buffer((32 + 3 * 8192) * sizeof(unsigned long long));      // 32 counters + (begin, queue empty, end) per wavefront (vr_pathtrace.h kStatsWaveBase)
        VR_HIP(hipMemsetAsync(stats_->get(), 0, stats_->size_bytes(), stream));
        tuning.stats = stats_->as<unsigned long long>();
    } else {
        tuning.stats = nullptr;
    }
}

void RendererHIP::wave_timeline(unsigned long long* out, size_t n_words) {
    if (!stats_) throw std::runtime_error("wave_timeline: statistics are not enabled");
    flush_pending();
    VR_HIP(hipStreamSynchronize(stream));
    const size_t have = stats_->size_bytes() / sizeof(unsigned long long) - 32;
    std::vector<unsigned long long> all(32 + have);
    stats_->download(all.data(), all.size() * sizeof(unsigned long long), stream);
    for (size_t i = 0; i < n_words; ++i) out[i] = i < have ? all[32 + i] : 0ull;
}

void RendererHIP::synchronize() { flush_pending(); VR_HIP(hipStreamSynchronize(stream)); }

void RendererHIP::download(float* rgba) {
    if (!color) throw std::runtime_error("RendererHIP::download: no framebuffer");
    flush_pending();
    color->download(rgba, color->size_bytes(), stream);
}
void RendererHIP::download_display(float* rgba) const {
    if (!display) throw std::runtime_error("RendererHIP::download_display: draw() first");
    display->download(rgba, display->size_bytes(), stream);
}

uint32_t RendererHIP::watchdog_status() {
    // bit 0: a wavefront gave up (iteration / shader-clock budget), bit 1: a path ended in an impossible state.
    // Read-and-clear, so that one bad launch does not poison the renderer.
    flush_pending();
    uint32_t s = 0;
    status_->download(&s, sizeof s, stream);
    if (s != 0) {
        VR_HIP(hipMemsetAsync(status_->get(), 0, sizeof(uint32_t), stream));
        VR_HIP(hipStreamSynchronize(stream));
    }
    return s;
}

}  // namespace vr
