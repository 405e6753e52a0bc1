// devmem.h -- RAII device buffers + HIP error checking for the host classes.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>

namespace vr {

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) (void)hipGetLastError();      // the failed call is reported HERE: it must not stay behind as the thread's "last error" and fail the next launch check
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " in " + what + " (" + file + ":" + std::to_string(line) + ")");
}
#define VR_HIP(call) ::vr::hip_check((call), #call, __FILE__, __LINE__)

// a hipMalloc allocation; shared handles mirror cppgl's ref-counted GL objects (renderer.h:9-14, environment.h:22)
// Test hook: allocations larger than this fail as an exhausted device would, so that the out-of-memory paths can be exercised without
// taking the HBM away from other users of the GPU.  Initial value from VR_TEST_MAX_ALLOC_MB (MiB), read ONCE per process (unset, empty
// or not a number = no cap); changed at run time through vr_test_alloc_cap_mb() (include/volren_amd.h).
inline std::atomic<size_t>& test_alloc_cap() {
    static std::atomic<size_t> cap([] {
        const char* e = getenv("VR_TEST_MAX_ALLOC_MB");
        if (!e || !*e) return ~(size_t)0;
        char* end = nullptr;
        const unsigned long long mb = strtoull(e, &end, 10);
        return (end == e || *end != '\0') ? ~(size_t)0 : (size_t)mb << 20;
    }());
    return cap;
}

class DeviceBuffer {
public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t bytes) {
        if (!bytes) return;
        if (bytes > test_alloc_cap().load(std::memory_order_relaxed)) hip_check(hipErrorOutOfMemory, "hipMalloc (test allocation cap)", __FILE__, __LINE__);
        VR_HIP(hipMalloc(&ptr_, bytes));
        bytes_ = bytes;
    }
    ~DeviceBuffer() { if (ptr_) (void)hipFree(ptr_); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    void* get() const { return ptr_; }
    template <typename T> T* as() const { return static_cast<T*>(ptr_); }
    size_t size_bytes() const { return bytes_; }
    void upload(const void* src, size_t bytes, hipStream_t s = nullptr) {
        if (bytes > bytes_) throw std::runtime_error("DeviceBuffer::upload: too large");
        VR_HIP(hipMemcpyAsync(ptr_, src, bytes, hipMemcpyHostToDevice, s));
        VR_HIP(hipStreamSynchronize(s));
    }
    void download(void* dst, size_t bytes, hipStream_t s = nullptr) const {
        if (bytes > bytes_) throw std::runtime_error("DeviceBuffer::download: too large");
        VR_HIP(hipMemcpyAsync(dst, ptr_, bytes, hipMemcpyDeviceToHost, s));
        VR_HIP(hipStreamSynchronize(s));
    }
private:
    void* ptr_ = nullptr;
    size_t bytes_ = 0;
};
using DeviceBufferPtr = std::shared_ptr<DeviceBuffer>;
inline DeviceBufferPtr make_device_buffer(size_t bytes) { return std::make_shared<DeviceBuffer>(bytes); }

}  // namespace vr
