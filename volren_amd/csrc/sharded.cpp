// sharded.cpp -- see sharded.h.
#include "sharded.h"

#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is opened at run time (no link-time dependency)

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <exception>
#include <mutex>
#include <set>
#include <thread>

#include "vr_device.h"

namespace vr {

std::vector<std::vector<int32_t>> tile_owner_lists(int width, int height, int n_parts) {
    if (width <= 0 || height <= 0 || n_parts <= 0) throw std::runtime_error("tile_owner_lists: bad arguments");
    const int tiles_x = (width + 15) / 16, tiles_y = (height + 15) / 16;
    std::vector<std::vector<int32_t>> lists((size_t)n_parts);
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) lists[(size_t)((tx + ty) % n_parts)].push_back(ty * tiles_x + tx);
    return lists;
}

// ---- librccl, opened on first use ------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
Rccl& rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        // VR_RCCL_LIBRARY names the library to open instead of the usual candidates (a site's own build; the tests use it to take RCCL away)
        const char* override_name = std::getenv("VR_RCCL_LIBRARY");
        std::string why;
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            if (override_name && *override_name) name = override_name;
            R.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (R.handle) break;
            const char* e = dlerror();                      // ONE call: it returns the message and clears it
            if (why.empty()) why = e ? e : "?";
            if (override_name && *override_name) break;
        }
        if (!R.handle) { R.error = std::string(override_name && *override_name ? override_name : "librccl.so.1") + " could not be opened: " + why; return; }
        auto sym = [&](const char* n) { void* p = dlsym(R.handle, n); if (!p && R.error.empty()) R.error = std::string("librccl lacks ") + n; return p; };
        R.CommInitAll = reinterpret_cast<decltype(R.CommInitAll)>(sym("ncclCommInitAll"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
        R.AllGather = reinterpret_cast<decltype(R.AllGather)>(sym("ncclAllGather"));
        R.Send = reinterpret_cast<decltype(R.Send)>(sym("ncclSend"));
        R.Recv = reinterpret_cast<decltype(R.Recv)>(sym("ncclRecv"));
        R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
        R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
        R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return R;
}
void rccl_check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string("RCCL error: ") + rccl().GetErrorString(r) + " in " + what);
}
// ncclGroupStart ... ncclGroupEnd, closed on every way out: a group left open would swallow the next RCCL call of this thread
struct RcclGroup {
    bool open = false;
    RcclGroup() { rccl_check(rccl().GroupStart(), "ncclGroupStart"); open = true; }
    void end() { open = false; rccl_check(rccl().GroupEnd(), "ncclGroupEnd"); }
    ~RcclGroup() { if (open) (void)rccl().GroupEnd(); }
};
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
ShardedRenderer::ShardedRenderer(const std::vector<RendererHIP*>& parts, const std::vector<int>& devices) : parts_(parts), devices_(devices) {
    if (parts_.empty() || parts_.size() != devices_.size()) throw std::runtime_error("ShardedRenderer: one device per part, at least one part");
    for (RendererHIP* p : parts_)
        if (!p) throw std::runtime_error("ShardedRenderer: null part");
    int n_dev = 0;
    VR_HIP(hipGetDeviceCount(&n_dev));
    for (int d : devices_)
        if (d < 0 || d >= n_dev) throw std::runtime_error("ShardedRenderer: device " + std::to_string(d) + " does not exist (" + std::to_string(n_dev) + " visible)");
    const std::set<int> distinct(devices_.begin(), devices_.end());
    const char* want = std::getenv("VR_SHARDED_TRANSPORT");
    const std::string forced = want ? want : "";
    if (!forced.empty() && forced != "rccl" && forced != "copy") throw std::runtime_error("VR_SHARDED_TRANSPORT must be rccl or copy");
    if (forced == "rccl" && distinct.size() != devices_.size()) throw std::runtime_error("ShardedRenderer: RCCL needs distinct devices (two parts share one)");
    // the collective of the rccl transport (SURVEY 8e names both): "gather" (default) = grouped ncclSend / ncclRecv to part 0, the only part that needs the whole frame;
    // VR_SHARDED_COLLECTIVE=allgather keeps round 4's grouped ncclAllGather (every part then holds -- and allocates -- all N tile buffers)
    const char* coll = std::getenv("VR_SHARDED_COLLECTIVE");
    collective_ = coll && *coll ? coll : "gather";
    if (collective_ != "gather" && collective_ != "allgather") throw std::runtime_error("VR_SHARDED_COLLECTIVE must be gather or allgather");
    if (parts_.size() == 1 && forced != "rccl") transport_ = "none";
    else if (forced == "copy" || distinct.size() != devices_.size()) transport_ = "copy";
    else transport_ = "rccl";
    buf_.resize(parts_.size());
    prev_streams_.resize(parts_.size(), nullptr);
    try {
    for (size_t i = 0; i < parts_.size(); ++i) {
        VR_HIP(hipSetDevice(devices_[i]));
        VR_HIP(hipStreamCreateWithFlags(&buf_[i].stream, hipStreamNonBlocking));
        VR_HIP(hipEventCreateWithFlags(&buf_[i].packed_ready, hipEventDisableTiming));
        parts_[i]->flush_pending();
        prev_streams_[i] = parts_[i]->stream;
        parts_[i]->stream = buf_[i].stream;
    }
    if (transport_ == "rccl") {
        // RCCL that cannot be opened or initialised is not fatal unless it was asked for by name: peer copies move the same buffers
        try {
            Rccl& R = rccl();
            if (!R.error.empty()) throw std::runtime_error(R.error);
            std::vector<ncclComm_t> comms(parts_.size());
            rccl_check(R.CommInitAll(comms.data(), (int)parts_.size(), devices_.data()), "ncclCommInitAll");
            for (ncclComm_t c : comms) comms_.push_back((void*)c);
        } catch (const std::exception& e) {
            if (forced == "rccl") throw std::runtime_error(std::string("ShardedRenderer: ") + e.what());
            std::cerr << "volren_amd: RCCL is not available (" << e.what() << "): the tile buffers are exchanged with peer copies" << std::endl;
            comms_.clear();
            transport_ = "copy";
        }
    }
    } catch (...) {                                   // the destructor does not run for a constructor that throws: give everything back here
        release();
        throw;
    }
}

// streams, events, buffers and communicators of this object; the parts get the streams back that they had before
void ShardedRenderer::release() {
    for (size_t i = 0; i < parts_.size(); ++i) {
        (void)hipSetDevice(devices_[i]);
        if (buf_[i].stream) (void)hipStreamSynchronize(buf_[i].stream);
    }
    for (void* c : comms_) (void)rccl().CommDestroy((ncclComm_t)c);
    comms_.clear();
    for (size_t i = 0; i < parts_.size(); ++i) {
        (void)hipSetDevice(devices_[i]);
        if (buf_[i].stream && parts_[i]->stream == buf_[i].stream) parts_[i]->stream = prev_streams_[i];
        buf_[i].pack_ids.reset(); buf_[i].packed.reset(); buf_[i].gathered.reset();
        if (i == 0) unpack_ids_.reset();
        if (buf_[i].packed_ready) (void)hipEventDestroy(buf_[i].packed_ready);
        if (buf_[i].stream) (void)hipStreamDestroy(buf_[i].stream);
        buf_[i].packed_ready = nullptr; buf_[i].stream = nullptr;
    }
}

ShardedRenderer::~ShardedRenderer() {
    for (size_t i = 0; i < parts_.size(); ++i) {
        (void)hipSetDevice(devices_[i]);
        try { parts_[i]->flush_pending(); } catch (...) { }       // recorded trace() calls go out on the stream that is about to be destroyed, or not at all
    }
    release();
}

void ShardedRenderer::setup(int width, int height) {
    width_ = width; height_ = height;
    const size_t n = parts_.size();
    if (transport_ == "none") {
        VR_HIP(hipSetDevice(devices_[0]));
        parts_[0]->set_tiles({});
        return;
    }
    const auto lists = tile_owner_lists(width, height, (int)n);
    n_max_ = 1;
    for (const auto& l : lists) n_max_ = std::max(n_max_, (int)l.size());
    const size_t packed_bytes = (size_t)n_max_ * 256u * 4u * sizeof(float);
    std::vector<int32_t> unpack;
    for (size_t i = 0; i < n; ++i) {
        VR_HIP(hipSetDevice(devices_[i]));
        parts_[i]->set_tiles(lists[i]);
        buf_[i].n_own = (int)lists[i].size();        // 0 (more parts than tile diagonals): the part renders nothing -- an empty list would mean the WHOLE frame to set_tiles
        // pack list: own tiles, padded by repeating the last one (any valid tile: its slot is ignored on unpack)
        std::vector<int32_t> pack(lists[i]);
        pack.resize((size_t)n_max_, lists[i].empty() ? 0 : lists[i].back());
        buf_[i].pack_ids = make_device_buffer(pack.size() * sizeof(int32_t));
        buf_[i].pack_ids->upload(pack.data(), pack.size() * sizeof(int32_t), buf_[i].stream);
        buf_[i].packed = make_device_buffer(packed_bytes);
        buf_[i].gathered = (i == 0 || (transport_ == "rccl" && collective_ == "allgather")) ? make_device_buffer(packed_bytes * n) : nullptr;
        unpack.insert(unpack.end(), lists[i].begin(), lists[i].end());
        unpack.resize((i + 1) * (size_t)n_max_, -1);
    }
    VR_HIP(hipSetDevice(devices_[0]));
    unpack_ids_ = make_device_buffer(unpack.size() * sizeof(int32_t));
    unpack_ids_->upload(unpack.data(), unpack.size() * sizeof(int32_t), buf_[0].stream);
}

void ShardedRenderer::reset() {
    for (RendererHIP* p : parts_) p->reset();
}

void ShardedRenderer::render(int spp) {
    RendererHIP& first = *parts_[0];
    for (RendererHIP* p : parts_)
        if (p->resolution.x != first.resolution.x || p->resolution.y != first.resolution.y || p->sample != first.sample)
            throw std::runtime_error("ShardedRenderer::render: the parts disagree on resolution or sample count");
    if (first.resolution.x != width_ || first.resolution.y != height_) setup(first.resolution.x, first.resolution.y);
    const int n_samples = spp <= 0 ? first.sppx - first.sample : spp;
    if (n_samples <= 0) return;
    const size_t n = parts_.size();
    const size_t count = (size_t)n_max_ * 256u * 4u;             // floats a part contributes
    // Every part: all samples of its tiles, then its compact tile buffer -- issued from one host thread per part.  A part's first render() on new settings
    // waits for its probe launch (RendererHIP::submit, launch_target_ms); issued one after the other the parts' first frames ran staggered (round 4).
    auto issue = [&](size_t i) {
        VR_HIP(hipSetDevice(devices_[i]));
        RendererHIP& p = *parts_[i];
        if (p.stream != buf_[i].stream) throw std::runtime_error("ShardedRenderer::render: a part's stream was changed behind the sharded renderer");
        if (transport_ == "none" || buf_[i].n_own > 0) p.render(n_samples);
        else { p.flush_pending(); p.sample += n_samples; }        // no tile of its own: nothing to render, the sample count keeps step
        if (transport_ == "none") return;
        launch_pack_tiles(p.color->as<float>(), width_, height_, buf_[i].pack_ids->as<int32_t>(), n_max_, buf_[i].packed->as<float>(), buf_[i].stream);
        VR_HIP(hipGetLastError());
    };
    if (n == 1) issue(0);
    else {
        std::vector<std::exception_ptr> errors(n);
        std::vector<std::thread> threads;
        for (size_t i = 1; i < n; ++i) threads.emplace_back([&, i] { try { issue(i); } catch (...) { errors[i] = std::current_exception(); } });
        try { issue(0); } catch (...) { errors[0] = std::current_exception(); }
        for (std::thread& t : threads) t.join();
        for (const std::exception_ptr& e : errors) if (e) std::rethrow_exception(e);
    }
    if (transport_ == "none") return;
    if (transport_ == "rccl" && collective_ == "allgather") {
        Rccl& R = rccl();
        RcclGroup group;
        for (size_t i = 0; i < n; ++i) {
            VR_HIP(hipSetDevice(devices_[i]));
            rccl_check(R.AllGather(buf_[i].packed->get(), buf_[i].gathered->get(), count, ncclFloat, (ncclComm_t)comms_[i], buf_[i].stream), "ncclAllGather");
        }
        group.end();
    } else if (transport_ == "rccl") {
        // gather to part 0: its own buffer moves on its stream, every other part sends its buffer and part 0 posts the matching receives -- ONE group, so the
        // N - 1 transfers over the N - 1 xGMI links into device 0 run concurrently.  Stream order does the rest: a part's send follows its pack, part 0's
        // receives follow its unpack of the frame before (which reads the buffer they fill) and precede this frame's.
        Rccl& R = rccl();
        float* gathered = buf_[0].gathered->as<float>();
        VR_HIP(hipSetDevice(devices_[0]));
        VR_HIP(hipMemcpyAsync(gathered, buf_[0].packed->get(), count * sizeof(float), hipMemcpyDeviceToDevice, buf_[0].stream));
        RcclGroup group;
        for (size_t i = 1; i < n; ++i) {
            VR_HIP(hipSetDevice(devices_[0]));
            rccl_check(R.Recv(gathered + i * count, count, ncclFloat, (int)i, (ncclComm_t)comms_[0], buf_[0].stream), "ncclRecv");
            VR_HIP(hipSetDevice(devices_[i]));
            rccl_check(R.Send(buf_[i].packed->get(), count, ncclFloat, 0, (ncclComm_t)comms_[i], buf_[i].stream), "ncclSend");
        }
        group.end();
    } else {
        // logical shards of one device (or VR_SHARDED_TRANSPORT=copy): every part copies its buffer into part 0's gathered buffer on its
        // own stream, and part 0's stream waits for all of them.  The copies of frame k+1 must not overtake part 0's unpack of frame k,
        // which reads that buffer: they wait for an event part 0 records after its pack -- in stream order after the unpack before it.
        VR_HIP(hipSetDevice(devices_[0]));
        VR_HIP(hipEventRecord(buf_[0].packed_ready, buf_[0].stream));       // part 0 has packed (and, in stream order, finished last frame's unpack)
        float* gathered = buf_[0].gathered->as<float>();
        for (size_t i = 0; i < n; ++i) {
            VR_HIP(hipSetDevice(devices_[i]));
            if (i > 0) VR_HIP(hipStreamWaitEvent(buf_[i].stream, buf_[0].packed_ready, 0));
            if (devices_[i] == devices_[0]) VR_HIP(hipMemcpyAsync(gathered + i * count, buf_[i].packed->get(), count * sizeof(float), hipMemcpyDeviceToDevice, buf_[i].stream));
            else VR_HIP(hipMemcpyPeerAsync(gathered + i * count, devices_[0], buf_[i].packed->get(), devices_[i], count * sizeof(float), buf_[i].stream));
            if (i > 0) VR_HIP(hipEventRecord(buf_[i].packed_ready, buf_[i].stream));
        }
        VR_HIP(hipSetDevice(devices_[0]));
        for (size_t i = 1; i < n; ++i) VR_HIP(hipStreamWaitEvent(buf_[0].stream, buf_[i].packed_ready, 0));
    }
    VR_HIP(hipSetDevice(devices_[0]));
    launch_unpack_tiles(buf_[0].gathered->as<float>(), unpack_ids_->as<int32_t>(), (int32_t)(n * (size_t)n_max_), first.color->as<float>(), width_, height_, buf_[0].stream);
    VR_HIP(hipGetLastError());
}

void ShardedRenderer::synchronize() {
    uint32_t tripped = 0;
    for (size_t i = 0; i < parts_.size(); ++i) {
        VR_HIP(hipSetDevice(devices_[i]));
        parts_[i]->synchronize();
        tripped |= parts_[i]->watchdog_status();
    }
    VR_HIP(hipSetDevice(devices_[0]));
    if (tripped) throw std::runtime_error("path-tracing kernel watchdog tripped: a launch did not finish within its budget");
}

}  // namespace vr
