// icpx_bvh.h — exact nearest neighbour through a linear bounding-volume hierarchy (device side): the
// alternative to the uniform grid (icpx_grid.h) for clouds whose points fill their bounding box very unevenly.
// It takes the place of the reference's kd-tree (include/algo/kdtree.h:96-154: build by median splits, search
// by descent with pruning) with a structure that is built in parallel:
//   * targets sorted by their position along the Hilbert curve through the cloud's bounding box (radix sort),
//   * leaves = runs of kBvhLeaf consecutive sorted targets, stored as 32-byte records {x, y, z, index},
//   * an IMPLICIT complete binary tree over the leaves (heap numbering: node n has children 2n and 2n + 1, the
//     leaves are nodes P .. 2P - 1, P = leaf count rounded up to a power of two), one axis-aligned box per node,
//     f32 rounded OUTWARD (24 bytes per node); no pointers, no split search.
// A query descends nearer child first and skips a node when the squared distance to its box — computed in f64
// from the f32 box, scaled down by (1 - 2^-40) against its own roundings — is STRICTLY above the best squared
// distance so far.  Everything at most as far as the final best is therefore evaluated, with the reference's f64
// expression and the (d2, index) rule: the answer is the lexicographic minimum of (d2, target index) over ALL
// targets, bit for bit what the brute-force and grid searches return, whatever the tree looks like.
// The traversal keeps no stack of nodes: the tree is implicit, so the sibling of any node on the current path is
// arithmetic, and one bit per level remembers that the farther child is still to be entered.
#pragma once

#include "icpx_grid.h"

namespace icpx {

#ifndef ICPX_BVH_LEAF
#define ICPX_BVH_LEAF 8
#endif
constexpr int kBvhLeaf = ICPX_BVH_LEAF;  // targets per leaf (a multiple of 4)

constexpr int kBvhTopNodes = 1024;  // nodes 1 .. 1023 (ten levels) are copied to LDS by the query kernels

struct BvhView {
    const float* box;      // node n: box[6n .. 6n+5] = lo x y z, hi x y z (2D: z = 0); empty: lo = +inf, hi = -inf
    const GridRec* recs;   // targets in Morton order
    int64_t m;             // targets
    int P;                 // leaves, a power of two; 0 = no tree
};

struct BvhFrame {          // bounding box of the finite targets and the scale of the Morton grid
    double lo[3];
    double scale[3];       // Morton cells per unit length (0 on degenerate axes)
};

// bit spreading for the 63-bit (3D: 21 bits per axis) / 62-bit (2D: 31 bits per axis) keys
ICPX_DEV uint64_t bvh_spread3(uint64_t x) {  // 21 bits -> every third bit
    x &= 0x1fffffULL;
    x = (x | (x << 32)) & 0x1f00000000ffffULL;
    x = (x | (x << 16)) & 0x1f0000ff0000ffULL;
    x = (x | (x << 8)) & 0x100f00f00f00f00fULL;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ULL;
    x = (x | (x << 2)) & 0x1249249249249249ULL;
    return x;
}
ICPX_DEV uint64_t bvh_spread2(uint64_t x) {  // 31 bits -> every second bit
    x &= 0x7fffffffULL;
    x = (x | (x << 16)) & 0x0000ffff0000ffffULL;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffULL;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0fULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    x = (x | (x << 1)) & 0x5555555555555555ULL;
    return x;
}
// Sort key of a point: its position along the HILBERT curve through a (2^21)^3 (2D: (2^31)^2) grid over the frame
// (Skilling's transform, then the bits interleaved).  The resolution matters: with 10 bits per axis a cluster
// smaller than a thousandth of the box collapses into one key and its leaves become random subsets (two far-apart
// clusters of 100 000 points: 115 ms per step instead of a few).  Unlike the Z-order curve the Hilbert curve never jumps: any
// run of consecutive points along it is spatially compact, which is what lets the implicit tree below — whose
// nodes are runs of 2^k consecutive leaves, not Morton prefixes — have tight boxes (with Z-order keys a query
// visited ~10x as many nodes).  Any order gives correct results; the order only decides how much is pruned.
template <int D>
ICPX_DEV uint64_t bvh_morton(const BvhFrame& f, const double* p) {
    constexpr int B = D == 3 ? 21 : 31;
    const double top = static_cast<double>((1u << B) - 1u);
    uint32_t X[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        double x = (p[d] - f.lo[d]) * f.scale[d];
        x = __builtin_fmin(__builtin_fmax(x, 0.0), top);  // NaN -> 0
        X[d] = static_cast<uint32_t>(x);
    }
    const uint32_t M = 1u << (B - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
#pragma unroll
        for (int i = 0; i < D; i++) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const uint32_t t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
#pragma unroll
    for (int i = 1; i < D; i++) X[i] ^= X[i - 1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (X[D - 1] & Q) t ^= Q - 1;
#pragma unroll
    for (int i = 0; i < D; i++) X[i] ^= t;
    if (D == 3) return (bvh_spread3(X[0]) << 2) | (bvh_spread3(X[1]) << 1) | bvh_spread3(X[D - 1]);
    return (bvh_spread2(X[0]) << 1) | bvh_spread2(X[1]);
}

template <int D>
__global__ void bvh_keys_kernel(const double* __restrict__ p, int64_t n, BvhFrame f, uint64_t* __restrict__ keys,
                                uint32_t* __restrict__ iota) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v[D];
#pragma unroll
    for (int d = 0; d < D; d++) v[d] = p[i * D + d];
    keys[i] = bvh_morton<D>(f, v);
    iota[i] = static_cast<uint32_t>(i);
}

// records in Morton order
template <int D>
__global__ void bvh_gather_kernel(const double* __restrict__ t, int64_t m, const uint32_t* __restrict__ perm,
                                  GridRec* __restrict__ recs) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int64_t i = perm[j];
    GridRec r;
    r.x = t[i * D + 0];
    r.y = t[i * D + 1];
    r.z = D == 3 ? t[i * D + 2] : 0.0;
    r.idx = i;
    recs[j] = r;
}

// leaf boxes (nodes P .. 2P-1) over the finite points of each leaf, f32 rounded outward
template <int D>
__global__ void bvh_leaf_kernel(const GridRec* __restrict__ recs, int64_t m, int P, float* __restrict__ box) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= P) return;
    double lo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()};
    double hi[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
    for (int u = 0; u < kBvhLeaf; u++) {
        const int64_t j = static_cast<int64_t>(l) * kBvhLeaf + u;
        if (j >= m) break;
        const GridRec r = recs[j];
        const double v[3] = {r.x, r.y, r.z};
        bool fin = true;
#pragma unroll
        for (int d = 0; d < D; d++) fin = fin && finite_d(v[d]);
        if (fin) {  // a non-finite target never has a distance below +inf: it needs no box
#pragma unroll
            for (int d = 0; d < D; d++) {
                lo[d] = __builtin_fmin(lo[d], v[d]);
                hi[d] = __builtin_fmax(hi[d], v[d]);
            }
        }
    }
    float* b = box + static_cast<int64_t>(P + l) * 6;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const bool used = d < D && lo[d] <= hi[d];
        b[d] = used ? __double2float_rd(lo[d]) : (d < D ? __builtin_inff() : 0.0f);
        b[3 + d] = used ? __double2float_ru(hi[d]) : (d < D ? -__builtin_inff() : 0.0f);
    }
}

// one level of inner nodes: nodes [first, 2 * first)
__global__ void bvh_level_kernel(float* __restrict__ box, int first) {
    const int n = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= 2 * first) return;
    const float* a = box + static_cast<int64_t>(2 * n) * 6;
    const float* b = a + 6;
    float* o = box + static_cast<int64_t>(n) * 6;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        o[d] = fminf(a[d], b[d]);
        o[3 + d] = fmaxf(a[3 + d], b[3 + d]);
    }
}

// the top of the tree in one workgroup: levels `first`, first / 2, ... 1 (first <= 256)
__global__ void __launch_bounds__(256) bvh_top_kernel(float* __restrict__ box, int first) {
    for (int lvl = first; lvl >= 1; lvl >>= 1) {
        const int n = lvl + threadIdx.x;
        if (static_cast<int>(threadIdx.x) < lvl) {
            const float* a = box + static_cast<int64_t>(2 * n) * 6;
            const float* b = a + 6;
            float* o = box + static_cast<int64_t>(n) * 6;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                o[d] = fminf(a[d], b[d]);
                o[3 + d] = fmaxf(a[3 + d], b[3 + d]);
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

// squared distance from q to the box of node n (a lower bound of the squared distance to every target below n)
template <int D>
ICPX_DEV double bvh_box_d2(const BvhView& v, const float* top, int n, const double* q) {
    float b[6];
    if (top != nullptr && n < kBvhTopNodes) {
#pragma unroll
        for (int k = 0; k < 6; k++) b[k] = top[n * 6 + k];
    } else {
        const float* g = v.box + static_cast<int64_t>(n) * 6;
#pragma unroll
        for (int k = 0; k < 6; k++) b[k] = g[k];
    }
    double acc = 0.0;
#pragma unroll
    for (int d = 0; d < D; d++) {
        const double lo = static_cast<double>(b[d]), hi = static_cast<double>(b[3 + d]);
        double g = __builtin_fmax(lo - q[d], q[d] - hi);
        g = g > 0.0 ? g : 0.0;  // (inside the slab, or NaN: 0)
        acc = acc + g * g;
    }
    return acc;
}

// Workgroup-wide: the top of the tree into LDS (every query passes through it).  Contains a barrier.
ICPX_DEV void bvh_stage_top(const BvhView& v, float* lds_top) {
    const int nodes = 2 * v.P < kBvhTopNodes ? 2 * v.P : kBvhTopNodes;
    for (int i = threadIdx.x; i < nodes * 6; i += blockDim.x) lds_top[i] = v.box[i];
    __syncthreads();
}

// one query.  Returns (best d2, lowest index among equal d2); (inf, 0) when no target has a finite distance.
// TIES: as in icpx_grid.h (a node is skipped only when everything below it is farther than (1 + 2^-41) times the best)
template <int D, bool TIES = false>
ICPX_DEV void bvh_query(const BvhView& v, const float* top, const double* q, double& best_out, int& idx_out, int& visited,
                        int& tie, const double* __restrict__ cloud) {
    double best = __builtin_inf();
    int bidx = 0x7fffffff;
    bool qfin = v.P > 0 && v.m > 0;
#pragma unroll
    for (int d = 0; d < D; d++) qfin = qfin && finite_d(q[d]);
    if (qfin) {
        const double down = 0.99999999999909051;  // 1 - 2^-40
        // One move per trip, so that the lanes of a wave stay in step: expand an inner node (enter the nearer
        // child; bit `depth` of `pending` = the farther child is still to be entered), scan a leaf, or pop (the
        // deepest pending level: its second child is the sibling of the current path's node at that depth).
        int n = 1, depth = 0;
        unsigned pending = 0;
        bool pop = false;
        while (true) {
            if (pop) {
                if (pending == 0) break;
                const int d = 31 - __builtin_clz(pending);
                pending &= ~(1u << d);
                const int second = (n >> (depth - (d + 1))) ^ 1;
                n = second;
                depth = d + 1;
                pop = bvh_box_d2<D>(v, top, second, q) * down > best;  // nothing that near below it: next pending level
            } else if (n < v.P) {
                const double dl = bvh_box_d2<D>(v, top, 2 * n, q), dr = bvh_box_d2<D>(v, top, 2 * n + 1, q);
                const bool right = dr < dl;
                if ((right ? dr : dl) * down > best) {
                    pop = true;  // neither child can hold anything as near
                } else {
                    if (!((right ? dl : dr) * down > best)) pending |= 1u << depth;
                    n = 2 * n + (right ? 1 : 0);
                    depth++;
                }
            } else {
                const int64_t kb = static_cast<int64_t>(n - v.P) * kBvhLeaf;
                const int64_t ke = kb + kBvhLeaf < v.m ? kb + kBvhLeaf : v.m;
                for (int64_t k0 = kb; k0 < ke; k0 += 4) {
                    GridRec rec[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) rec[u] = v.recs[k0 + u < ke ? k0 + u : ke - 1];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        double df = rec[u].x - q[0];
                        double dist = df * df;
                        df = rec[u].y - q[1];
                        dist = dist + df * df;
                        if (D == 3) {
                            df = rec[u].z - q[2];
                            dist = dist + df * df;
                        }
                        const int j = static_cast<int>(rec[u].idx);
                        const double cc[3] = {rec[u].x, rec[u].y, rec[u].z};
                        nn_take<TIES, D>(k0 + u < ke, dist, j, cc, cloud, best, bidx, tie);
                    }
                }
                visited += ke > kb ? static_cast<int>(ke - kb) : 0;
                pop = true;
            }
        }
    }
    best_out = best;
    idx_out = bidx == 0x7fffffff ? 0 : bidx;
}

template <int D, bool TIES = false>
__global__ void __launch_bounds__(256)
bvh_nn_kernel(const double* __restrict__ q, int64_t n, BvhView v, int64_t* __restrict__ idx, double* __restrict__ d2,
              unsigned long long* __restrict__ visited_total, const uint32_t* __restrict__ perm,
              const double* __restrict__ cloud = nullptr, uint32_t* __restrict__ kd_list = nullptr,
              int* __restrict__ kd_count = nullptr) {
    __shared__ float top[kBvhTopNodes * 6];
    bvh_stage_top(v, top);
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    int visited = 0;
    const bool live = i < n;
    if (live && perm) i = perm[i];  // Morton order of the queries: neighbouring lanes take the same path
    if (live) {
        double qq[D];
#pragma unroll
        for (int d = 0; d < D; d++) qq[d] = q[i * D + d];
        double best;
        int bi;
        int tie = 0;
        bvh_query<D, TIES>(v, top, qq, best, bi, visited, tie, cloud);
        if (idx) idx[i] = bi;
        if (d2) d2[i] = best;
        if (TIES && tie) kd_list[atomicAdd(kd_count, 1)] = static_cast<uint32_t>(i);
    }
    if (visited_total) {
        int s = wave_sum_int(visited);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(visited_total, static_cast<unsigned long long>(s));
    }
}

}  // namespace icpx
