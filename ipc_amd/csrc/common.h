// Shared host-side helpers for libipcgpu (error handling, device buffers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace ipcgpu {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct ArgError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct StateError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            throw ::ipcgpu::HipError(std::string(#expr) + ": " + hipGetErrorString(e_) + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
    } while (0)

// RAII device array
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count)
    {
        if (count == n && p) return;
        release();
        if (count) HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
    }
    // grow-only variant for buffers whose size changes from call to call (hipMalloc / hipFree cost ~100 us each)
    void ensure(size_t count)
    {
        if (count > n || !p) alloc(count + count / 2 + 16);
    }
    void zeroN(size_t count, hipStream_t s)
    {
        if (count) HIP_CHECK(hipMemsetAsync(p, 0, count * sizeof(T), s));
    }
    void upload(const T* h, size_t count, hipStream_t s)
    {
        alloc(count);
        if (count) HIP_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void upload(const std::vector<T>& h, hipStream_t s) { upload(h.data(), h.size(), s); }
    void uploadGrow(const std::vector<T>& h, hipStream_t s) // capacity only grows; n stays the capacity
    {
        ensure(h.size());
        if (!h.empty()) HIP_CHECK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T* h, size_t count, hipStream_t s) const
    {
        if (count) HIP_CHECK(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    void zero(hipStream_t s)
    {
        if (n) HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
};

// pinned host scalar block for small readbacks
template <class T>
struct PinnedBuf {
    T* p = nullptr;
    T* dev = nullptr; // the same memory as seen by kernels: small results are written here directly instead of being copied
    size_t n = 0;
    ~PinnedBuf()
    {
        if (p) (void)hipHostFree(p);
    }
    void alloc(size_t count)
    {
        if (p) (void)hipHostFree(p);
        HIP_CHECK(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocMapped | hipHostMallocPortable));
        HIP_CHECK(hipHostGetDevicePointer((void**)&dev, p, 0));
        n = count;
    }
};

} // namespace ipcgpu
