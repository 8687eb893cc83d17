// icpx_kd_build.h — the kd rule's tree (icpx_kd.h) of a cloud of any size, built on the device (multi-kernel path and
// the stand-alone query).  Same tree as kd_build_lds (icpx_kd_lds.h) and as the oracle's KdCanon: median split at
// (n - 1) / 2 of the subset ordered by (split coordinate, index), axis = depth % D, stored in order.
//
//   m <= kKdOneBlock:  one workgroup, everything in LDS (kd_build_lds).
//   larger clouds:     per axis a device-wide radix sort of (kd_ord_key, index) gives the D coordinate orders; the
//                      top levels are device-wide passes (position table of the split axis, then for every other axis a
//                      prefix count of "left of the median" / "is the median" and a stable scatter — what
//                      kd_level_passes does inside one workgroup); as soon as every segment holds <= kKdSegMax points,
//                      one workgroup per segment finishes its subtree in LDS.
#pragma once

#include "icpx_kd_lds.h"

namespace icpx {

constexpr int kKdOneBlock = 4096;   // clouds up to here: one workgroup, sorts included
constexpr int kKdSegMax = 6144;     // segment size from which a workgroup takes a subtree over
constexpr int kKdBuildThreads = 1024;

// LDS of kd_segment_kernel for a segment of s points: D orders + level-pass scratch (u16 / u32) + global ids (u32) +
// segment bounds (2 x u16)
inline __host__ __device__ size_t kd_segment_lds_bytes(int dim, int s) {
    const size_t ord = (2 * static_cast<size_t>(dim) * s + 15) & ~static_cast<size_t>(15);
    return ord + 8 * static_cast<size_t>(s) + 256 + 4 * static_cast<size_t>(s) + 4 * static_cast<size_t>(s);
}

template <int D>
__global__ void kd_keys_kernel(const double* __restrict__ pts, const int m, const int axis, unsigned long long* __restrict__ key,
                               uint32_t* __restrict__ val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    key[i] = kd_ord_key(pts[static_cast<int64_t>(i) * D + axis]);
    val[i] = static_cast<uint32_t>(i);
}

// posA[ord_a[p]] = p
__global__ void kd_pos_kernel(const uint32_t* __restrict__ ord_a, const int m, uint32_t* __restrict__ posA) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < m) posA[ord_a[p]] = static_cast<uint32_t>(p);
}

// the (left | median << 32) flag of position p of axis b's order at level k — the scan's input
struct KdFlag {
    const uint32_t* ord_b;
    const uint32_t* posA;
    int m, k;
    __host__ __device__ unsigned long long operator()(const uint32_t p) const {
        int lo = 0, hi = m;
        for (int l = 0; l < k; l++) {
            const int mid = lo + ((hi - lo - 1) >> 1);
            if (static_cast<int>(p) == mid) {
                lo = mid;
                hi = mid + 1;
                break;
            }
            if (static_cast<int>(p) < mid) hi = mid;
            else lo = mid + 1;
        }
        const int mid = lo + ((hi - lo - 1) >> 1);
        const int pa = static_cast<int>(posA[ord_b[p]]);
        return (pa < mid ? 1ull : 0ull) | (pa == mid ? (1ull << 32) : 0ull);
    }
};

// stable partition of axis b's order about every segment's median (E: exclusive prefix of KdFlag over all positions)
__global__ void kd_scatter_kernel(const uint32_t* __restrict__ ord_b, const uint32_t* __restrict__ posA,
                                  const unsigned long long* __restrict__ E, const int m, const int k, uint32_t* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    int lo, hi;
    kd_segment(p, m, k, lo, hi);
    const int mid = lo + ((hi - lo - 1) >> 1);
    const uint32_t t = ord_b[p];
    const int pa = static_cast<int>(posA[t]);
    const unsigned long long e = E[p] - E[lo];
    const int nl = static_cast<int>(e & 0xffffffffull), nm = static_cast<int>(e >> 32);
    const int np = pa < mid ? lo + nl : (pa == mid ? mid : mid + 1 + (p - lo - nl - nm));
    out[np] = t;
}

// node[p] = ord[p], pos[ord[p]] = p (the nodes of the top levels sit at their in-order positions in every order)
__global__ void kd_fill_kernel(const uint32_t* __restrict__ ord0, const int m, uint32_t* __restrict__ node, uint32_t* __restrict__ pos) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const uint32_t t = ord0[p];
    node[p] = t;
    pos[t] = static_cast<uint32_t>(p);
}

struct KdOrders {
    const uint32_t* axis[3];  // the coordinate orders (m entries each)
};

// One workgroup per segment of level K (blockIdx.x = the segment's path from the root, most significant bit first):
// the subtree of its <= kKdSegMax points, from the D orders' ranges [lo, hi).  loc: m words of scratch.
template <int D>
__global__ void __launch_bounds__(kKdBuildThreads)
kd_segment_kernel(const KdOrders ords, const int m, const int K, uint32_t* __restrict__ loc,
                  uint32_t* __restrict__ node, uint32_t* __restrict__ pos) {
    extern __shared__ __align__(16) char kd_seg_lds[];
    const int T = blockDim.x, tid = threadIdx.x;
    int lo = 0, hi = m;
    for (int l = 0; l < K; l++) {
        const int mid = lo + ((hi - lo - 1) >> 1);
        if (hi - lo <= 0) break;
        if ((blockIdx.x >> (K - 1 - l)) & 1) lo = mid + 1;
        else hi = mid;
    }
    const int s = hi - lo;
    if (s <= 0) return;  // (uniform)
    uint16_t* ord = reinterpret_cast<uint16_t*>(kd_seg_lds);
    char* u = kd_seg_lds + ((2 * static_cast<size_t>(D) * s + 15) & ~static_cast<size_t>(15));
    uint32_t* gid = reinterpret_cast<uint32_t*>(u + 8 * static_cast<size_t>(s) + 256);
    // local id of a point = its place in axis 0's order of the segment
    for (int p = tid; p < s; p += T) {
        const uint32_t g = ords.axis[0][lo + p];
        gid[p] = g;
        loc[g] = static_cast<uint32_t>(p);
        ord[p] = static_cast<uint16_t>(p);
    }
    __threadfence_block();
    __syncthreads();
    for (int b = 1; b < D; b++)
        for (int p = tid; p < s; p += T) ord[b * s + p] = static_cast<uint16_t>(loc[ords.axis[b][lo + p]]);
    __syncthreads();
    uint16_t* seg = reinterpret_cast<uint16_t*>(gid + s);
    const int fa = kd_level_passes<D>(ord, s, u, K, seg);
    for (int p = tid; p < s; p += T) {
        const uint32_t g = gid[ord[fa * s + p]];
        node[lo + p] = g;
        pos[g] = static_cast<uint32_t>(lo + p);
    }
}

// m <= kKdOneBlock: the whole build in one workgroup
template <int D>
__global__ void __launch_bounds__(kKdBuildThreads)
kd_small_kernel(const double* __restrict__ pts, const int m, uint32_t* __restrict__ node, uint32_t* __restrict__ pos) {
    extern __shared__ __align__(16) char kd_small_lds[];
    // (segment bounds behind the build's own scratch)
    kd_build_lds<D, uint32_t>(pts, m, kd_small_lds, node, pos, reinterpret_cast<uint16_t*>(kd_small_lds + ((kd_build_bytes(D, m) + 15) & ~size_t(15))));
}

}  // namespace icpx
