// sharded.h -- ShardedRenderer: one process, N RendererHIP parts on N devices, one frame.
//
// No reference counterpart (the reference drives one GL context, src/main.cpp:524-557); this is SURVEY 8(e) inside the product
// rather than inside a benchmark harness: the scene is replicated on every device, the frame's 16x16 tiles (the reference's work
// group, shader/pathtracer_brick.glsl:3) are dealt diagonally, owner(tx, ty) = (tx + ty) mod N, every part renders all samples of
// its tiles with no communication on its own stream, and ONE gather per frame (RCCL over xGMI: grouped ncclSend / ncclRecv to part 0; round 4-5: a grouped
// ncclAllGather, still there behind VR_SHARDED_COLLECTIVE=allgather) of the compact
// per-part tile buffers puts the accumulated radiance together; part 0 scatters it back into its framebuffer, which then holds the
// whole frame -- bit-identical to a single-device render, because a pixel-sample depends on (seed, pixel, sample) only
// (pathtracer_brick.glsl:28-36).
//
// Transport: "rccl" when the parts sit on distinct devices (librccl.so.1 is opened at run time: the library has no link-time
// dependency on it), "copy" when two parts share a device (logical shards of one GPU: device-to-device copies ordered by events
// stand in for the collective -- RCCL refuses a communicator with duplicate devices) or when VR_SHARDED_TRANSPORT=copy asks for it
// (peer copies across devices).  VR_SHARDED_TRANSPORT=rccl forces the collective also for a single part.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "renderer.h"

namespace vr {

// raster tile ids (row 0 = bottom) owned by each of n parts: owner(tx, ty) = (tx + ty) mod n
std::vector<std::vector<int32_t>> tile_owner_lists(int width, int height, int n_parts);

struct ShardedRenderer {
    // parts[i] renders on devices[i] (repeats allowed); the renderers belong to the caller and must outlive this object.
    // Every part must have been init()-ed at the same resolution; scene state is the caller's to replicate (for_each).
    ShardedRenderer(const std::vector<RendererHIP*>& parts, const std::vector<int>& devices);
    ~ShardedRenderer();
    ShardedRenderer(const ShardedRenderer&) = delete;
    ShardedRenderer& operator=(const ShardedRenderer&) = delete;

    size_t n_parts() const { return parts_.size(); }
    RendererHIP& part(size_t i) { return *parts_[i]; }
    int device(size_t i) const { return devices_[i]; }
    // fn(part, index) with that part's device current: replicate scene calls
    template <class F> void for_each(F fn) {
        for (size_t i = 0; i < parts_.size(); ++i) { VR_HIP(hipSetDevice(devices_[i])); fn(*parts_[i], i); }
    }
    void reset();                         // every part: sample = 0
    // `spp` more samples per pixel (<= 0: up to sppx): every part renders its tiles (asynchronously, its own stream; issued from one host thread
    // per part), then the gather; returns when everything is enqueued -- the first frame on new settings after every part's probe launch has been
    // waited for (launch_target_ms; all parts at once, set the parts' launch_target_ms to 0 to have none).  The whole frame is in part(0).color once
    // synchronize() returns.  More parts than the frame has tile diagonals: the surplus parts own no tile and render nothing.
    void render(int spp = 0);
    void synchronize();                   // waits for all parts; throws if a kernel watchdog tripped
    const std::string& transport() const { return transport_; }      // "rccl" | "copy" | "none" (one part, nothing to exchange)
    const std::string& collective() const { return collective_; }    // of the rccl transport: "gather" (ncclSend / ncclRecv to part 0) | "allgather"

private:
    void setup(int width, int height);   // tile deal + buffers for the current resolution
    void release();                       // everything this object created; the parts get their previous streams back
    std::vector<RendererHIP*> parts_;
    std::vector<int> devices_;
    std::string transport_, collective_;
    int width_ = 0, height_ = 0, n_max_ = 0;
    struct PartBuffers {
        hipStream_t stream = nullptr;     // owned
        hipEvent_t packed_ready = nullptr;
        int n_own = 0;                    // tiles this part renders
        DeviceBufferPtr pack_ids, packed, gathered;      // gathered: on part 0 (every part under VR_SHARDED_COLLECTIVE=allgather)
    };
    std::vector<PartBuffers> buf_;
    DeviceBufferPtr unpack_ids_;          // on part 0's device: every part's tile ids in part order, -1 = padding
    std::vector<void*> comms_;            // ncclComm_t per part (rccl transport)
    std::vector<hipStream_t> prev_streams_;      // what the parts' `stream` fields held before this object took them
};

}  // namespace vr
