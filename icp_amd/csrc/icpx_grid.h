// icpx_grid.h — exact nearest neighbour on a uniform grid (device side), for clouds too large
// for the brute-force kernels (BASELINE config 4: 1 M x 1 M points).
//
// The target cloud is bucketed once per begin() (counting sort by cell) into 32-byte records
// {x, y, z, index}; the reference builds its kd-tree at the same point (vanilla_3d.cpp:75-80,
// include/algo/kdtree.h:42-52).  A query walks Chebyshev shells of cells around its home cell and
// stops as soon as its best squared distance is STRICTLY below a conservative lower bound on the
// squared distance (as the kernel itself would compute it) of every target in an unvisited cell.
// The answer is therefore exactly the lexicographic minimum of (d2, target index) over ALL
// targets — the same answer as the brute-force kernels and as Trimmed::compute_matches
// (trimmed.cpp:95-109) — whatever the cell size and whatever the order of points inside a cell.
#pragma once

#include "icpx_math.h"

namespace icpx {

struct GridParams {
    double lo[3];
    double hi[3];
    double h[3];      // cell edge per axis
    double inv_h[3];
    int g[3];         // cells per axis (g[2] == 1 in 2D)
    int rmax;         // max(g) - 1: shells needed to cover the grid from any cell
    int64_t m;
};

struct __attribute__((aligned(16))) GridRec {
    double x, y, z;
    int64_t idx;
};

// order-preserving map f64 -> u64 (for atomicMin / atomicMax on coordinates)
ICPX_DEV unsigned long long f64_key(double v) {
    unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
inline double f64_unkey(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
    double v;
    __builtin_memcpy(&v, &b, sizeof(v));
    return v;
}

ICPX_DEV bool finite_d(double v) { return dabs(v) <= 1.7976931348623157e308; }

// bounding box of the finite target points.  keys: [0..2] = min, [3..5] = max (as f64_key)
template <int D>
__global__ void grid_bbox_kernel(const double* __restrict__ t, int64_t m, unsigned long long* keys) {
    unsigned long long mn[D], mx[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        mn[d] = ~0ULL;
        mx[d] = 0ULL;
    }
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < m;
         j += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        double p[D];
        bool fin = true;
#pragma unroll
        for (int d = 0; d < D; d++) {
            p[d] = t[j * D + d];
            fin = fin && finite_d(p[d]);
        }
        if (fin) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                unsigned long long k = f64_key(p[d]);
                mn[d] = k < mn[d] ? k : mn[d];
                mx[d] = k > mx[d] ? k : mx[d];
            }
        }
    }
    __shared__ unsigned long long smn[D][16], smx[D][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
#pragma unroll
    for (int d = 0; d < D; d++) {
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            unsigned long long o = __shfl_xor(mn[d], s, 64);
            mn[d] = o < mn[d] ? o : mn[d];
            o = __shfl_xor(mx[d], s, 64);
            mx[d] = o > mx[d] ? o : mx[d];
        }
        if (lane == 0) {
            smn[d][wave] = mn[d];
            smx[d][wave] = mx[d];
        }
    }
    __syncthreads();
    if (threadIdx.x < D) {
        const int d = threadIdx.x;
        unsigned long long a = smn[d][0], b = smx[d][0];
        for (int w = 1; w < W; w++) {
            a = smn[d][w] < a ? smn[d][w] : a;
            b = smx[d][w] > b ? smx[d][w] : b;
        }
        atomicMin(&keys[d], a);
        atomicMax(&keys[3 + d], b);
    }
}

template <int D>
ICPX_DEV int cell_coord(const GridParams& gp, double v, int d) {
    double f = (v - gp.lo[d]) * gp.inv_h[d];
    int c = (f >= 0.0) ? (f < 2.0e9 ? static_cast<int>(f) : gp.g[d] - 1) : 0;  // NaN -> 0
    return c < gp.g[d] ? c : gp.g[d] - 1;
}

template <int D>
ICPX_DEV int cell_of(const GridParams& gp, const double* p) {
    int cx = cell_coord<D>(gp, p[0], 0);
    int cy = cell_coord<D>(gp, p[1], 1);
    int cz = D == 3 ? cell_coord<D>(gp, p[2], 2) : 0;
    return (cz * gp.g[1] + cy) * gp.g[0] + cx;
}

template <int D>
__global__ void grid_count_kernel(const double* __restrict__ t, GridParams gp, int* __restrict__ cell_id,
                                  int* __restrict__ counts, int* __restrict__ occupied) {
    int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= gp.m) return;
    double p[D];
#pragma unroll
    for (int d = 0; d < D; d++) p[d] = t[j * D + d];
    int c = cell_of<D>(gp, p);
    cell_id[j] = c;
    if (atomicAdd(&counts[c], 1) == 0) atomicAdd(occupied, 1);
}

// exclusive scan of counts[0..ncell) into start[0..ncell] (start[ncell] = total) in three steps:
// per-block totals (kScanBlk cells per block), one-block scan of the totals, per-block local scan.
constexpr int kScanBlk = 4096;

__global__ void __launch_bounds__(256) grid_scan_totals_kernel(const int* __restrict__ counts, int ncell,
                                                                int* __restrict__ totals) {
    __shared__ int ws[4];
    const int base = blockIdx.x * kScanBlk;
    int s = 0;
    for (int i = threadIdx.x; i < kScanBlk; i += 256) {
        int c = base + i;
        s += c < ncell ? counts[c] : 0;
    }
    s = wave_sum_int(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) totals[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// in-place exclusive scan of totals[0..nb); totals[nb] = grand total.  One block of 1024.
__global__ void __launch_bounds__(1024) grid_scan_top_kernel(int* __restrict__ totals, int nb) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int chunk = (nb + 1023) / 1024;
    const int b = tid * chunk;
    const int e = b + chunk < nb ? b + chunk : nb;
    int s = 0;
    for (int i = b; i < e; i++) s += totals[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid == 0 ? 0 : part[tid - 1];
    for (int i = b; i < e; i++) {
        int c = totals[i];
        totals[i] = run;
        run += c;
    }
    if (tid == 1023) totals[nb] = part[1023];
}

__global__ void __launch_bounds__(256) grid_scan_local_kernel(const int* __restrict__ counts, int ncell,
                                                               const int* __restrict__ totals,
                                                               int* __restrict__ start) {
    __shared__ int part[256];
    constexpr int PER = kScanBlk / 256;  // consecutive cells per thread
    const int tid = threadIdx.x;
    const int base = blockIdx.x * kScanBlk + tid * PER;
    int v[PER];
    int s = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        int c = base + i;
        v[i] = c < ncell ? counts[c] : 0;
        s += v[i];
    }
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int x = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    int run = totals[blockIdx.x] + (tid == 0 ? 0 : part[tid - 1]);
#pragma unroll
    for (int i = 0; i < PER; i++) {
        int c = base + i;
        if (c < ncell) start[c] = run;
        run += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 255) start[ncell] = totals[gridDim.x];
}

template <int D>
__global__ void grid_scatter_kernel(const double* __restrict__ t, int64_t m, const int* __restrict__ cell_id,
                                    const int* __restrict__ start, int* __restrict__ cursor,
                                    GridRec* __restrict__ recs) {
    int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= m) return;
    int c = cell_id[j];
    int pos = start[c] + atomicAdd(&cursor[c], 1);
    GridRec r;
    r.x = t[j * D + 0];
    r.y = t[j * D + 1];
    r.z = D == 3 ? t[j * D + 2] : 0.0;
    r.idx = j;
    recs[pos] = r;
}

// one query against the grid.  Returns (best d2, lowest index among equal d2); (inf, 0) when no
// target has a finite distance.
template <int D>
ICPX_DEV void grid_query(const GridParams& gp, const GridRec* __restrict__ recs,
                         const int* __restrict__ start, const double* q, double& best_out,
                         int& idx_out, int& visited) {
    double best = __builtin_inf();
    int bidx = 0x7fffffff;
    bool qfin = true;
#pragma unroll
    for (int d = 0; d < D; d++) qfin = qfin && finite_d(q[d]);
    if (qfin && gp.m > 0) {
        int c[3] = {0, 0, 0};
#pragma unroll
        for (int d = 0; d < D; d++) c[d] = cell_coord<D>(gp, q[d], d);
        // slack on a face position: covers the rounding of the cell assignment and of the face
        double slack[3];
#pragma unroll
        for (int d = 0; d < D; d++)
            slack[d] = (dabs(gp.lo[d]) + dabs(gp.hi[d]) + dabs(q[d])) * 2.8421709430404007e-14;  // 2^-45
        const int gx = gp.g[0], gy = gp.g[1], gz = D == 3 ? gp.g[2] : 1;
        for (int r = 0; r <= gp.rmax; r++) {
            const int zlo = D == 3 ? (c[2] - r > 0 ? c[2] - r : 0) : 0;
            const int zhi = D == 3 ? (c[2] + r < gz - 1 ? c[2] + r : gz - 1) : 0;
            const int ylo = c[1] - r > 0 ? c[1] - r : 0;
            const int yhi = c[1] + r < gy - 1 ? c[1] + r : gy - 1;
            const int xlo = c[0] - r > 0 ? c[0] - r : 0;
            const int xhi = c[0] + r < gx - 1 ? c[0] + r : gx - 1;
            for (int z = zlo; z <= zhi; z++) {
                const bool zshell = D == 3 && (z == c[2] - r || z == c[2] + r);
                for (int y = ylo; y <= yhi; y++) {
                    const bool yshell = (y == c[1] - r || y == c[1] + r);
                    const int row = (z * gy + y) * gx;
                    // segments of this row that belong to shell r
                    int seg_b[2], seg_e[2], nseg;
                    if (zshell || yshell) {
                        seg_b[0] = start[row + xlo];
                        seg_e[0] = start[row + xhi + 1];
                        nseg = 1;
                    } else {
                        nseg = 0;
                        if (c[0] - r >= 0) {
                            seg_b[nseg] = start[row + c[0] - r];
                            seg_e[nseg] = start[row + c[0] - r + 1];
                            nseg++;
                        }
                        if (r > 0 && c[0] + r <= gx - 1) {
                            seg_b[nseg] = start[row + c[0] + r];
                            seg_e[nseg] = start[row + c[0] + r + 1];
                            nseg++;
                        }
                    }
                    for (int s = 0; s < nseg; s++) {
                        // 4 records in flight per step: the walk is latency-bound, not bandwidth-bound
                        for (int k0 = seg_b[s]; k0 < seg_e[s]; k0 += 4) {
                            GridRec rec[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                int k = k0 + u < seg_e[s] ? k0 + u : seg_e[s] - 1;
                                rec[u] = recs[k];
                            }
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
                                int j = static_cast<int>(rec[u].idx);
                                bool take = (k0 + u < seg_e[s]) &&
                                            (dist < best || (dist == best && j < bidx && bidx != 0x7fffffff));
                                best = take ? dist : best;
                                bidx = take ? j : bidx;
                            }
                            visited += seg_e[s] - k0 < 4 ? seg_e[s] - k0 : 4;
                        }
                    }
                }
            }
            // lower bound on the distance to anything outside the visited block
            double lb = __builtin_inf();
            bool all_covered = true;
#pragma unroll
            for (int d = 0; d < D; d++) {
                int kl = c[d] - r;           // first visited cell coordinate on this axis
                int kh = c[d] + r + 1;       // one past the last visited
                if (kl > 0) {
                    all_covered = false;
                    double face = gp.lo[d] + static_cast<double>(kl) * gp.h[d];
                    double gap = (q[d] - face) - slack[d];
                    gap = gap > 0.0 ? gap : 0.0;
                    lb = gap < lb ? gap : lb;
                }
                if (kh < gp.g[d]) {
                    all_covered = false;
                    double face = gp.lo[d] + static_cast<double>(kh) * gp.h[d];
                    double gap = (face - q[d]) - slack[d];
                    gap = gap > 0.0 ? gap : 0.0;
                    lb = gap < lb ? gap : lb;
                }
            }
            if (all_covered) break;
            double lb2 = lb * lb * 0.99999999999909051;  // (1 - 2^-40)
            if (best < lb2) break;
        }
    }
    best_out = best;
    idx_out = bidx == 0x7fffffff ? 0 : bidx;
}

template <int D>
__global__ void __launch_bounds__(256)
grid_nn_kernel(const double* __restrict__ q, int64_t n, GridParams gp, const GridRec* __restrict__ recs,
               const int* __restrict__ start, int64_t* __restrict__ idx, double* __restrict__ d2,
               unsigned long long* __restrict__ visited_total, const uint32_t* __restrict__ perm) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    int visited = 0;
    const bool live = i < n;
    // queries are visited in cell order (perm, grid_sort_queries) so that neighbouring lanes walk the
    // same cells: 1.6x on 1 M random queries
    if (live && perm) i = perm[i];
    if (live) {
        double qq[D];
#pragma unroll
        for (int d = 0; d < D; d++) qq[d] = q[i * D + d];
        double best;
        int bi;
        grid_query<D>(gp, recs, start, qq, best, bi, visited);
        if (idx) idx[i] = bi;
        if (d2) d2[i] = best;
    }
    if (visited_total) {
        int v = wave_sum_int(visited);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(visited_total, static_cast<unsigned long long>(v));
    }
}

// cell of every query (clamped into the grid) as the sort key of grid_sort_queries
template <int D>
__global__ void grid_query_keys_kernel(const double* __restrict__ q, int64_t n, GridParams gp,
                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ iota) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p[D];
#pragma unroll
    for (int d = 0; d < D; d++) p[d] = q[i * D + d];
    keys[i] = static_cast<uint32_t>(cell_of<D>(gp, p));
    iota[i] = static_cast<uint32_t>(i);
}

}  // namespace icpx
