// icpx_nn.h — stand-alone correspondence kernels (device side).
//
//  * nn_brute_kernel: LDS-tiled brute force over (query block) x (target split); partial results
//    are merged by nn_merge_kernel.  Arithmetic and tie-break are those of
//    Trimmed::compute_matches (lib/icp/impl/trimmed.cpp:95-109).
//  * uniform-grid search (build + query) for large clouds: icpx_grid.h.
#pragma once

#include "icpx_filter.h"
#include "icpx_math.h"

namespace icpx {

constexpr int kNnTile = 1024;   // targets per LDS tile
constexpr int kNnThreads = 256;
constexpr int kNnQpt = 2;       // queries per lane

// grid = (query blocks, target splits).  q, t: AoS.  part_d2/part_idx: [splits][n].
template <int D>
__global__ void __launch_bounds__(kNnThreads)
nn_brute_kernel(const double* __restrict__ q, int64_t n, const double* __restrict__ t, int64_t m,
                int64_t split_len, double* __restrict__ part_d2, int32_t* __restrict__ part_idx) {
    __shared__ double tile[D][kNnTile];
    const int tid = threadIdx.x;
    const int64_t qbase = static_cast<int64_t>(blockIdx.x) * (kNnThreads * kNnQpt);
    const int64_t t0 = static_cast<int64_t>(blockIdx.y) * split_len;
    const int64_t t1 = t0 + split_len < m ? t0 + split_len : m;

    double qq[kNnQpt][D];
    double best[kNnQpt];
    int bidx[kNnQpt], bblk[kNnQpt];
#pragma unroll
    for (int k = 0; k < kNnQpt; k++) {
        int64_t i = qbase + tid + k * kNnThreads;
        best[k] = __builtin_inf();
        bidx[k] = 0;
        bblk[k] = 0;
#pragma unroll
        for (int d = 0; d < D; d++) qq[k][d] = i < n ? q[i * D + d] : 0.0;
    }
    for (int64_t base = t0; base < t1; base += kNnTile) {
        const int cnt = static_cast<int>(t1 - base < kNnTile ? t1 - base : kNnTile);
        const int cnt4 = (cnt + 3) & ~3;
        __syncthreads();
        for (int j = tid; j < cnt4; j += kNnThreads) {
#pragma unroll
            for (int d = 0; d < D; d++) tile[d][j] = j < cnt ? t[(base + j) * D + d] : __builtin_inf();
        }
        __syncthreads();
        // four targets at a time: block minimum vs running best (see nn_lds in icpx_pair.h)
        for (int jb = 0; jb < cnt4; jb += 4) {
            double tc[D][4];
#pragma unroll
            for (int d = 0; d < D; d++)
#pragma unroll
                for (int u = 0; u < 4; u++) tc[d][u] = tile[d][jb + u];
#pragma unroll
            for (int k = 0; k < kNnQpt; k++) {
                double dist[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    double df = tc[0][u] - qq[k][0];
                    double acc = df * df;
#pragma unroll
                    for (int d = 1; d < D; d++) {
                        df = tc[d][u] - qq[k][d];
                        acc = acc + df * df;
                    }
                    dist[u] = acc;
                }
                double mm = __builtin_fmin(__builtin_fmin(dist[0], dist[1]), __builtin_fmin(dist[2], dist[3]));
                bool lt = mm < best[k];
                best[k] = __builtin_fmin(best[k], mm);
                bblk[k] = lt ? static_cast<int>(base - t0) + jb : bblk[k];
            }
        }
    }
    // resolve the index inside the winning block of four (lowest j that equals the minimum)
#pragma unroll
    for (int k = 0; k < kNnQpt; k++) {
        int idx = 0;
        if (best[k] < __builtin_inf()) {
            for (int u = 3; u >= 0; u--) {
                int64_t j = t0 + bblk[k] + u;
                if (j < t1) {
                    double df = t[j * D + 0] - qq[k][0];
                    double acc = df * df;
#pragma unroll
                    for (int d = 1; d < D; d++) {
                        df = t[j * D + d] - qq[k][d];
                        acc = acc + df * df;
                    }
                    idx = acc == best[k] ? static_cast<int>(j) : idx;
                }
            }
        }
        bidx[k] = idx;
    }
#pragma unroll
    for (int k = 0; k < kNnQpt; k++) {
        int64_t i = qbase + tid + k * kNnThreads;
        if (i < n) {
            part_d2[static_cast<int64_t>(blockIdx.y) * n + i] = best[k];
            part_idx[static_cast<int64_t>(blockIdx.y) * n + i] = bidx[k];
        }
    }
}

// The same search through the f32 candidate filter with exact f64 refinement (icpx_filter.h):
// grid = (query tiles of 1024, target splits); every workgroup stages its split in LDS and runs the
// seedless two-minimum pass.  Results are those of nn_brute_kernel, bit for bit, at about a third
// of the VALU work.  LDS: (8 + 4) * D * scap bytes + kNnScratch doubles.
constexpr int kNnTileQpt = 4;
constexpr int kNnScratch = 96;
constexpr int kNnMaxSplit = 3072;
constexpr int kNnMinSplit = 256;

inline size_t nn_tile_lds_bytes(int dim, int scap) {
    return sizeof(double) * (static_cast<size_t>(dim) * scap + kNnScratch) + sizeof(float) * static_cast<size_t>(dim) * scap;
}

// TIES: the kd-tree rule's tie detection (icpx_kd.h); the split's flag travels in the sign bit of part_idx.
template <int D, bool TIES = false>
__global__ void __launch_bounds__(kNnThreads)
nn_tile_kernel(const double* __restrict__ q, int64_t n, const double* __restrict__ t, int64_t m, int split_len, int scap,
               double* __restrict__ part_d2, int32_t* __restrict__ part_idx) {
    extern __shared__ __attribute__((aligned(16))) double nn_tile_lds[];
    const int tid = threadIdx.x;
    const int64_t t0 = static_cast<int64_t>(blockIdx.y) * split_len;
    const int mt = static_cast<int>(m - t0 < split_len ? m - t0 : split_len);
    const int m4 = (mt + 3) & ~3;
    LdsView<D> L;
    double* p = nn_tile_lds;
    for (int d = 0; d < 3; d++) {
        L.t[d] = d < D ? p : nullptr;
        if (d < D) p += scap;
    }
    L.gs = p;
    p += kNnScratch;
    L.bc = nullptr;
    L.skey = nullptr;
    L.sval = nullptr;
    float* f = reinterpret_cast<float*>(p);
    for (int d = 0; d < 3; d++) L.tf[d] = d < D ? f + static_cast<size_t>(d) * scap : nullptr;
    L.j0 = static_cast<int>(t0);
    L.cloud = t;
    double qq[kNnTileQpt][D], best[kNnTileQpt];
    int bidx[kNnTileQpt];
    const int64_t qbase = static_cast<int64_t>(blockIdx.x) * (kNnThreads * kNnTileQpt);
#pragma unroll
    for (int k = 0; k < kNnTileQpt; k++) {
        const int64_t i = qbase + tid + k * kNnThreads;
#pragma unroll
        for (int d = 0; d < D; d++) qq[k][d] = i < n ? q[i * D + d] : 0.0;
    }
    const FilterFrame F = stage_targets<D>(t + t0 * D, mt, m4, kNnThreads, L, true);
    int tie[kNnTileQpt];
#pragma unroll
    for (int k = 0; k < kNnTileQpt; k++) tie[k] = 0;
    nn_twomin<D, kNnTileQpt, TIES>(qq, L, m4, F, best, bidx, tie);
#pragma unroll
    for (int k = 0; k < kNnTileQpt; k++) {
        const int64_t i = qbase + tid + k * kNnThreads;
        if (i < n) {
            // a split without any finite distance reports (inf, its first index), as nn_brute_kernel does
            const bool none = !(best[k] < __builtin_inf());
            part_d2[static_cast<int64_t>(blockIdx.y) * n + i] = none ? __builtin_inf() : best[k];
            part_idx[static_cast<int64_t>(blockIdx.y) * n + i] =
                none ? static_cast<int>(t0) : (bidx[k] | ((TIES && tie[k] != 0) ? static_cast<int>(0x80000000u) : 0));
        }
    }
}

// splits are ordered by target index and each partial already holds its lowest index, so a
// strict < in split order keeps the lexicographic (d2, index) minimum.
// kd_list / kd_count (null: none): queries with a tie partner of the kd rule — flagged by a split, or two splits'
// minima within the tie range of each other (cloud: the targets, dim D, to tell coincident targets apart)
__global__ void nn_merge_kernel(int64_t n, int splits, const double* __restrict__ part_d2,
                                const int32_t* __restrict__ part_idx, int64_t* __restrict__ idx,
                                double* __restrict__ d2, int D = 0, const double* __restrict__ cloud = nullptr,
                                uint32_t* __restrict__ kd_list = nullptr, int* __restrict__ kd_count = nullptr) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double best = part_d2[i];
    int32_t bi = part_idx[i];
    bool tie = bi < 0;
    bi &= 0x7fffffff;
    for (int s = 1; s < splits; s++) {
        double d = part_d2[static_cast<int64_t>(s) * n + i];
        int32_t j = part_idx[static_cast<int64_t>(s) * n + i];
        tie = tie || j < 0;
        j &= 0x7fffffff;
        if (kd_list && d < __builtin_inf() && kd_near(d, best)) {
            bool same = d == best;
            for (int e = 0; e < D && same; e++) same = cloud[static_cast<int64_t>(j) * D + e] == cloud[static_cast<int64_t>(bi) * D + e];
            tie = tie || !same;
        }
        if (d < best) {
            best = d;
            bi = j;
        }
    }
    if (idx) idx[i] = bi;
    if (d2) d2[i] = best;
    if (kd_list && tie && best < __builtin_inf()) kd_list[atomicAdd(kd_count, 1)] = static_cast<uint32_t>(i);
}

// Euclidean distances from squared ones (KdTree::search returns sqrt, kdtree.h:126)
__global__ void nn_sqrt_kernel(int64_t n, double* __restrict__ d) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) d[i] = __builtin_sqrt(d[i]);
}

// no targets at all: every query gets (index 0, +inf) — what the search loops leave when nothing is below +inf
__global__ void nn_no_targets_kernel(int64_t n, int64_t* __restrict__ idx, double* __restrict__ d2) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (idx) idx[i] = 0;
    if (d2) d2[i] = __builtin_inf();
}

}  // namespace icpx
