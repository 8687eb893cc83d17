// icpx_host.h — host-side plumbing shared by the C-ABI translation units: error text, HIP
// checks, grow-only device buffers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/icpx/icpx.h"

namespace icpx {

std::string& last_error_ref();

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define ICPX_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return icpx::fail(e_ == hipErrorOutOfMemory ? ICPX_ERR_NOMEM : ICPX_ERR_HIP,       \
                              "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                              __LINE__);                                                       \
    } while (0)

#define ICPX_TRY(expr)              \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != ICPX_OK) return rc_; \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return ICPX_OK;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return fail(ICPX_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e));
        }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(ICPX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return ICPX_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

// true when `ptr` is device (or managed) memory as far as HIP knows
inline bool is_device_ptr(const void* ptr) {
    if (!ptr) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain host memory: not an error for us
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    int enter(int device) {
        if (device < 0) return ICPX_OK;
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return fail(ICPX_ERR_NO_DEVICE, "hipGetDevice failed: %s", hipGetErrorString(e));
        if (prev != device) {
            e = hipSetDevice(device);
            if (e != hipSuccess) return fail(ICPX_ERR_NO_DEVICE, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
            changed = true;
        }
        return ICPX_OK;
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

}  // namespace icpx
