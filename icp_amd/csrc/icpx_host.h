// icpx_host.h — host-side plumbing shared by the C-ABI translation units: error text, HIP
// checks, grow-only device buffers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
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

// Development knobs.  The ICPX_* environment variables are read ONCE, when the library is first used;
// afterwards only icpx_debug_knob() (icpx.h) changes them.  Product calls read these atomics, never
// the environment.
struct Knobs {
    std::atomic<int> prune{1};           // ICPX_PRUNE: 0 = batches use the brute-force filter instead of the grid-pruned search
    std::atomic<int> filter{1};          // ICPX_FILTER: 0 = plain f64 brute force (no f32 candidate filter)
    std::atomic<int> sched{-1};          // ICPX_SCHED: -1 auto, 0 never, 1 always (iteration-granular scheduler)
    std::atomic<int> sched_steps{0};     // ICPX_SCHED_STEPS: iterations per scheduler task, 0 = default
    std::atomic<int> prune_handles{0};   // ICPX_PRUNE_HANDLES: 1 = handles (single pairs) run the grid-pruned kernels of the batch path
    std::atomic<int> no_group_sums{0};   // ICPX_NO_GROUP_SUMS: 1 = multi-kernel path always sums in 4096-point blocks
    std::atomic<int> geom_qpt{0}, geom_threads{0};      // ICPX_GEOM="qpt,threads"
    std::atomic<double> fused_single_max{-1.0};         // ICPX_FUSED_SINGLE_MAX: largest n*m of a handle that stays fused; < 0 = default
    std::atomic<int> coop{1};            // ICPX_COOP: 0 = per-lane grid walk instead of the wave-cooperative one
    std::atomic<int> grid_coop{1};       // ICPX_GRID_COOP: 0 = HBM grid queries walk per lane even when they are in tile order; 2 = also block-cooperative inside converge
    std::atomic<int> poll_done{1};       // ICPX_POLL_DONE: 0 = the multi-kernel converge reads its done flag with a copy + stream synchronisation per chunk
    std::atomic<int> tile_wgs{0};        // ICPX_TILE_WGS: workgroups the tiled search aims at (0 = built-in 768)
    std::atomic<int> tile_min{0};        // ICPX_TILE_MIN: fewest targets per split of the tiled search (0 = built-in 512)
    std::atomic<int> grid_density{0};    // ICPX_GRID_DENSITY: targets per cell of the HBM grid in tenths (0 = built-in 20)
    std::atomic<int> zero_copy{1};       // ICPX_ZERO_COPY: 0 = host inputs of a fused pair are copied to the device (four staged copies)
    std::atomic<int> chunk_skip{1};      // ICPX_CHUNK_SKIP: 0 = trimmed ICP2's brute-force kernels search with the f32 filter
    std::atomic<int> defer_sparse{1};    // ICPX_DEFER_SPARSE: 0 = pruned kernels keep the pairs whose grid is a poor index
    std::atomic<int> grid_sd{1};         // ICPX_GRID_SD: 0 = no sorted-domain copy of the running cloud for the 3D grid search (LargeIcp::sd)
    std::atomic<int> kd_eager{-1};       // ICPX_KD_EAGER: -1 auto (the tree of a small target is built at begin() only for handles that met tie partners before), 0 never, 1 always for small targets
    std::atomic<int> tile_chunk{1};      // ICPX_TILE_CHUNK: 0 = the tiled search of the multi-kernel path keeps contiguous splits and scans every block
    std::atomic<int> grid_filter{1};     // ICPX_GRID_FILTER: 0 = seeded HBM grid queries of the 3D methods scan the 32-byte f64 records (no f32 filter records)
    std::atomic<int> grid_seed{1};       // ICPX_GRID_SEED: 0 = HBM grid queries of a converge never start from the previous step's neighbour
    std::atomic<int> fuse_prepass{1};    // ICPX_FUSE_PREPASS: 0 = the batch's pre-pass always runs as a launch of its own
};
Knobs& knobs();

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
