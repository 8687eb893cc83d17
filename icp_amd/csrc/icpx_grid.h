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

#include "icpx_kd.h"
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

// Opt-in f32 correspondence mode (ICPX_NN_GRID_F32; SURVEY 8 f3 — NOT reference arithmetic, never the default):
// 16-byte records with the coordinates relative to the grid's lower corner, rounded to f32.  The searches below are
// templates over the record type; with GridRec32 a candidate's squared distance is evaluated in f32 (FMA) against the
// query moved by the same corner and rounded to f32, then widened — every bound, cell range and stopping test stays
// the f64 code on the f64 query.  Half the record bytes and about a third of the arithmetic per candidate; the
// returned target is the nearest one up to that rounding (two targets whose distances differ by less than ~2^-21 of
// the cloud's extent may swap), so index parity with the reference is given up — which is why it is opt-in.
struct __attribute__((aligned(16))) GridRec32 {
    float x, y, z;
    int idx;
};

template <typename Rec>
struct RecTraits;
template <>
struct RecTraits<GridRec> {
    using coord = double;
    static constexpr bool f32 = false;
    static ICPX_DEV const double* query(const double* q, const float*) { return q; }
};
template <>
struct RecTraits<GridRec32> {
    using coord = float;
    static constexpr bool f32 = true;
    static ICPX_DEV const float* query(const double*, const float* q32) { return q32; }
};

// the squared distance as the search computes it: the reference's f64 expression (trimmed.cpp:101, kdtree.h:124) ...
template <int D>
ICPX_DEV double d2_eval(const double x, const double y, const double z, const double* q) {
    double df = x - q[0];
    double dist = df * df;
    df = y - q[1];
    dist = dist + df * df;
    if (D == 3) {
        df = z - q[2];
        dist = dist + df * df;
    }
    return dist;
}
// ... or, in the f32 mode, three f32 subtractions and FMAs
template <int D>
ICPX_DEV double d2_eval(const float x, const float y, const float z, const float* q) {
    float df = x - q[0];
    float dist = df * df;
    df = y - q[1];
    dist = __builtin_fmaf(df, df, dist);
    if (D == 3) {
        df = z - q[2];
        dist = __builtin_fmaf(df, df, dist);
    }
    return static_cast<double>(dist);
}

// the query as the f32 records see it
template <int D>
ICPX_DEV void query_f32(const double* lo, const double* q, float* q32) {
#pragma unroll
    for (int d = 0; d < 3; d++) q32[d] = d < D ? static_cast<float>(q[d] - lo[d]) : 0.0f;
}

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
#pragma unroll 4
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

// An upper bound of sqrt(x) for x >= 0 (inf for inf), from the f32 square root: radii that only size a cell range need
// no f64 square root (a sequence of ~25 instructions; the seeded walk took ten of them per query).  The f32 conversion,
// v_sqrt_f32 (1 ulp) and the conversion back err by less than 2^-22 relative in all; arguments below the f32 normal
// range (where the conversion flushes or loses bits) get the bound sqrt(FLT_MIN).
ICPX_DEV double sqrt_up(double x) {
    const float xf = static_cast<float>(x);
    const double r = static_cast<double>(__builtin_sqrtf(xf)) * 1.0000009536743164;  // (1 + 2^-20)
    return x < 1.1754943508222875e-38 ? 1.0842021724855044e-19 : r;
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
                                  int* __restrict__ counts) {
    int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= gp.m) return;
    double p[D];
#pragma unroll
    for (int d = 0; d < D; d++) p[d] = t[j * D + d];
    int c = cell_of<D>(gp, p);
    cell_id[j] = c;
    // result unused: the atomic does not wait for its return value (a million returning atomics on random
    // cells were 0.19 ms of the 0.33 ms build)
    __hip_atomic_fetch_add(&counts[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// number of cells that hold at least one target
__global__ void __launch_bounds__(256) grid_occupied_kernel(const int* __restrict__ counts, int ncell, int* __restrict__ occupied) {
    int nz = 0;
    for (int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; c < ncell;
         c += static_cast<int64_t>(gridDim.x) * blockDim.x)
        nz += counts[c] != 0 ? 1 : 0;
    __shared__ int part[4];
    nz = wave_sum_int(nz);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = nz;
    __syncthreads();
    if (threadIdx.x == 0) {
        nz = part[0] + part[1] + part[2] + part[3];
        if (nz) atomicAdd(occupied, nz);
    }
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

template <int D, typename Rec = GridRec>
__global__ void grid_scatter_kernel(const double* __restrict__ t, int64_t m, const int* __restrict__ cell_id,
                                    const int* __restrict__ start, int* __restrict__ cursor,
                                    Rec* __restrict__ recs, const GridParams gp = GridParams{},
                                    GridRec32* __restrict__ filt = nullptr) {
    int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= m) return;
    int c = cell_id[j];
    int pos = start[c] + atomicAdd(&cursor[c], 1);
    if (filt) {
        // the same record once more as a 16-byte f32 FILTER record (grid_scan_rows_filtered), at the same position
        GridRec32 f;
        f.x = static_cast<float>(t[j * D + 0] - gp.lo[0]);
        f.y = static_cast<float>(t[j * D + 1] - gp.lo[1]);
        f.z = D == 3 ? static_cast<float>(t[j * D + 2] - gp.lo[2]) : 0.0f;
        f.idx = static_cast<int>(j);
        filt[pos] = f;
    }
    Rec r;
    if constexpr (RecTraits<Rec>::f32) {
        r.x = static_cast<float>(t[j * D + 0] - gp.lo[0]);
        r.y = static_cast<float>(t[j * D + 1] - gp.lo[1]);
        r.z = D == 3 ? static_cast<float>(t[j * D + 2] - gp.lo[2]) : 0.0f;
        r.idx = static_cast<int>(j);
    } else {
        r.x = t[j * D + 0];
        r.y = t[j * D + 1];
        r.z = D == 3 ? t[j * D + 2] : 0.0;
        r.idx = j;
    }
    recs[pos] = r;
}

// The records of NR rows of cells, [kb[t], ke[t]) each, against one query, as ONE sequence, four records per step:
// record r of the sequence is record r - pre[t] of the row t with pre[t] <= r < pre[t + 1].  A lane's records are
// spread over several short rows, and row by row every row costs the wave a memory round trip of its own (config 4
// converge 16.3 -> 14.3 ms).  (best, bidx): running (d2, index) minimum; bidx = 0x7fffffff while there is none.
// TIES (all searches of this file): tie |= another, non-coincident target lies within kKdNearUlps of the minimum
// (icpx_kd.h; cloud = the targets in cloud order).  The stopping tests leave unvisited only what is farther than
// (1 + 2^-41) times the best, so every tie partner is among the evaluated records.
template <int D, int NR, bool TIES = false, typename Rec = GridRec, int STEP = 4>
ICPX_DEV void grid_scan_rows(const Rec* __restrict__ recs, const double* q, const int (&kb)[NR], const int (&ke)[NR],
                             double& best, int& bidx, int& visited, int& tie, const double* __restrict__ cloud,
                             const float* q32 = nullptr) {
    const auto* qc = RecTraits<Rec>::query(q, q32);
    // records in flight per lane (8: more registers, fewer waves, no gain — measured, f64 and f32 records; STEP = 1:
    // the rarely taken fallbacks of a kernel that must stay small, lg_nn_kernel<..., LEAN>)
    constexpr int kStep = STEP;
    int pre[NR], off[NR];
    int total = 0;
#pragma unroll
    for (int t = 0; t < NR; t++) {
        pre[t] = total;
        off[t] = kb[t] - total;
        total += ke[t] - kb[t];
    }
    for (int r0 = 0; r0 < total; r0 += kStep) {
        Rec rec[kStep];
#pragma unroll
        for (int u = 0; u < kStep; u++) {
            const int r = r0 + u < total ? r0 + u : total - 1;
            int o = off[0];
#pragma unroll
            for (int t = 1; t < NR; t++) o = r >= pre[t] ? off[t] : o;
            rec[u] = recs[r + o];
        }
#pragma unroll
        for (int u = 0; u < kStep; u++) {
            const double dist = d2_eval<D>(rec[u].x, rec[u].y, rec[u].z, qc);
            const int j = static_cast<int>(rec[u].idx);
            const double cc[3] = {static_cast<double>(rec[u].x), static_cast<double>(rec[u].y), static_cast<double>(rec[u].z)};
            nn_take<TIES, D>(r0 + u < total, dist, j, cc, cloud, best, bidx, tie);
        }
    }
    visited += total;
}

// grid_scan_rows for a query that HAS a candidate (best finite), through 16-byte f32 FILTER records — same answer, bit
// for bit, half the bytes and a third of the arithmetic per scanned record.  A record is the target moved by the
// grid's lower corner and rounded to f32; its f32 distance to the query (moved and rounded the same way) is off by at
// most E = 2^-21 M in length (M = the larger of the cloud's extent and the query's offset from the corner: three
// roundings of 2^-24 (ext + M) per axis, times sqrt 3) and by 3 * 2^-24 relative in the squared sum.  So EVERY target
// whose exact f64 d2 is <= best has an f32 value <= thr = (sqrt(best) + E)^2 (1 + 2^-21), rounded up: records above
// thr can neither improve the minimum nor tie with it (a tie partner lies within kKdNearUlps of the minimum, far
// inside the margin).  The others — the answer, near misses, nothing else late in a converge — are re-evaluated
// from the ORIGINAL cloud with the reference's f64 expression and the (d2, index) / kd tie rules of nn_take, in
// scan order like grid_scan_rows.  The current best itself is skipped by index (its exact d2 is `best`).
// Overflowing or non-finite f32 values compare like the f64 ones would (inf passes only an infinite threshold,
// NaN never passes and could never be taken).  Measured: config-4 converge lg_nn_kernel 144 -> ... us per iteration.
template <int D, int NR, bool TIES>
ICPX_DEV void grid_scan_rows_filtered(const GridRec32* __restrict__ filt, const GridParams& gp, const double* q,
                                      const int (&kb)[NR], const int (&ke)[NR], double& best, int& bidx, int& visited,
                                      int& tie, const double* __restrict__ cloud) {
    // (measured and dropped, all bit-identical: survivors parked per lane and evaluated together with their gathers
    // in flight at once — 9.67 against 9.56 ms for config 4's converge; the winner's coordinates handed to the caller
    // instead of its gather of cloud[bidx] — six more live registers, 10.6 ms; eight records in flight — 10.9 ms:
    // the kernel sits at its 80-register budget and every added live value spills)
    constexpr int kStep = 4;
    float q32[3];
    query_f32<D>(gp.lo, q, q32);
    double M = 0.0;
#pragma unroll
    for (int d = 0; d < D; d++) M = dmax(M, dmax(gp.hi[d] - gp.lo[d], dabs(q[d] - gp.lo[d])));
    const double E = M * 4.76837158203125e-07;  // 2^-21
    auto thr_of = [&](const double b) -> float {
        const double s = sqrt_up(b) + E;
        return static_cast<float>(s * s * 1.000000476837158203125) * 1.0000002384185791f;  // (1 + 2^-21), then (1 + 2^-22): rounded up
    };
    float thr = thr_of(best);
    int pre[NR], off[NR];
    int total = 0;
#pragma unroll
    for (int t = 0; t < NR; t++) {
        pre[t] = total;
        off[t] = kb[t] - total;
        total += ke[t] - kb[t];
    }
    for (int r0 = 0; r0 < total; r0 += kStep) {
        GridRec32 rec[kStep];
#pragma unroll
        for (int u = 0; u < kStep; u++) {
            const int r = r0 + u < total ? r0 + u : total - 1;
            int o = off[0];
#pragma unroll
            for (int t = 1; t < NR; t++) o = r >= pre[t] ? off[t] : o;
            rec[u] = filt[r + o];
        }
        bool pass[kStep], any = false;
#pragma unroll
        for (int u = 0; u < kStep; u++) {
            float df = rec[u].x - q32[0];
            float v = df * df;
            df = rec[u].y - q32[1];
            v = __builtin_fmaf(df, df, v);
            if (D == 3) {
                df = rec[u].z - q32[2];
                v = __builtin_fmaf(df, df, v);
            }
            pass[u] = r0 + u < total && v <= thr && rec[u].idx != bidx;
            any = any || pass[u];
        }
        if (any) {
#pragma unroll
            for (int u = 0; u < kStep; u++) {
                if (pass[u]) {
                    const int j = rec[u].idx;
                    double cc[3] = {0.0, 0.0, 0.0};
#pragma unroll
                    for (int d = 0; d < D; d++) cc[d] = cloud[static_cast<int64_t>(j) * D + d];
                    const double dist = d2_eval<D>(cc[0], cc[1], cc[2], q);
                    const double before = best;
                    nn_take<TIES, D>(true, dist, j, cc, cloud, best, bidx, tie);
                    if (best < before) thr = thr_of(best);
                }
            }
        }
    }
    visited += total;
}

// Continuation of a query that has a finite candidate (best, bidx) and has not met the shells' stopping test:
// instead of more shells (whose cell count grows with the cube of the distance — queries outside the target's
// box, sparse regions), only the rows of cells that the BALL of the current best can reach are visited, and in
// each row only the cells within the remaining radius.  Lower bounds per axis come from the inner faces of the
// slabs exactly as in the shells' stopping test (same slack), so a row or cell is skipped only when everything in
// it is strictly farther than the current best; cell ranges use cell_coord, the (monotone) function that placed
// the targets.  best only shrinks, so everything at most as far as the final best is evaluated.
template <int D, bool TIES = false, typename Rec = GridRec, int STEP = 4>
ICPX_DEV void grid_ball_scan(const GridParams& gp, const Rec* __restrict__ recs, const int* __restrict__ start,
                             const double* q, const int (&c)[3], double& best, int& bidx, int& visited, int& tie,
                             const double* __restrict__ cloud, const float* q32 = nullptr) {
    const auto* qc = RecTraits<Rec>::query(q, q32);
    const double down = 0.99999999999909051;  // 1 - 2^-40
    double slack[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int d = 0; d < D; d++) slack[d] = (dabs(gp.lo[d]) + dabs(gp.hi[d]) + dabs(q[d])) * 2.8421709430404007e-14;  // 2^-45
    // lower bound of |t - q| along axis d for a target in slab k (0 for the query's own slab)
    auto gap_of = [&](int d, int k) -> double {
        double g = 0.0;
        if (k > c[d]) g = (gp.lo[d] + static_cast<double>(k) * gp.h[d]) - q[d] - slack[d];
        else if (k < c[d]) g = q[d] - (gp.lo[d] + static_cast<double>(k + 1) * gp.h[d]) - slack[d];
        return g > 0.0 ? g : 0.0;
    };
    // cells of axis d within `rad` of the query (rad: a distance that bounds |t - q| on this axis)
    auto range_of = [&](int d, double rad, int& k0, int& k1) {
        const double r = rad * (1.0 + 1.7763568394002505e-15) + 1e-150;  // (1 + 2^-49)
        const double w = (dabs(q[d]) + r) * 4.4408920985006262e-16;      // 2^-51
        k0 = cell_coord<D>(gp, q[d] - r - w, d);
        k1 = cell_coord<D>(gp, q[d] + r + w, d);
    };
    int z0 = 0, z1 = 0;
    if (D == 3) range_of(2, sqrt_up(best), z0, z1);
    for (int z = z0; z <= z1; z++) {
        const double gz = D == 3 ? gap_of(2, z) : 0.0;
        if (gz * gz * down > best) continue;
        int y0, y1;
        range_of(1, sqrt_up(best), y0, y1);
        for (int y = y0; y <= y1; y++) {
            const double gy = gap_of(1, y);
            const double far2 = (gz * gz + gy * gy) * down;
            if (far2 > best) continue;
            int x0, x1;
            range_of(0, sqrt_up(best - far2 + best * 9.0949470177292824e-13), x0, x1);  // + 2^-40 best: the subtraction's rounding
            const int row = (z * gp.g[1] + y) * gp.g[0];
            const int kb = start[row + x0], ke = start[row + x1 + 1];
            for (int k0 = kb; k0 < ke; k0 += STEP) {
                Rec rec[STEP];
#pragma unroll
                for (int u = 0; u < STEP; u++) rec[u] = recs[k0 + u < ke ? k0 + u : ke - 1];
#pragma unroll
                for (int u = 0; u < STEP; u++) {
                    const double dist = d2_eval<D>(rec[u].x, rec[u].y, rec[u].z, qc);
                    const int j = static_cast<int>(rec[u].idx);
                    const double cc[3] = {static_cast<double>(rec[u].x), static_cast<double>(rec[u].y), static_cast<double>(rec[u].z)};
                    nn_take<TIES, D>(k0 + u < ke, dist, j, cc, cloud, best, bidx, tie);
                }
            }
            visited += ke - kb;
        }
    }
}

// A query that starts from a candidate — its neighbour of the previous correspondence step, (best, bidx) with best
// finite and the query finite.  The answer lies in the ball of that candidate, which late in a converge is a
// fraction of the 3^D cells grid_query opens: rows of the ball's cell box (at most 3 x 3, else grid_ball_scan) are
// tested and bounded exactly as in grid_ball_scan, but all their bounds are fetched before the first record (the
// walk is a chain of memory latencies).  The candidate itself is among the visited targets, so the result is the
// (d2, index) minimum over the whole cloud: what grid_query returns.
template <int D, bool TIES = false, typename Rec = GridRec, int STEP = 4>
ICPX_DEV void grid_query_seeded(const GridParams& gp, const Rec* __restrict__ recs, const int* __restrict__ start,
                                const double* q, double& best, int& bidx, int& visited, int& tie,
                                const double* __restrict__ cloud, const float* q32 = nullptr,
                                const GridRec32* __restrict__ filt = nullptr) {
    int c[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < D; d++) c[d] = cell_coord<D>(gp, q[d], d);
    const double down = 0.99999999999909051;  // 1 - 2^-40
    double slack[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int d = 0; d < D; d++) slack[d] = (dabs(gp.lo[d]) + dabs(gp.hi[d]) + dabs(q[d])) * 2.8421709430404007e-14;  // 2^-45
    auto gap_of = [&](int d, int k) -> double {
        double g = 0.0;
        if (k > c[d]) g = (gp.lo[d] + static_cast<double>(k) * gp.h[d]) - q[d] - slack[d];
        else if (k < c[d]) g = q[d] - (gp.lo[d] + static_cast<double>(k + 1) * gp.h[d]) - slack[d];
        return g > 0.0 ? g : 0.0;
    };
    auto range_of = [&](int d, double rad, int& k0, int& k1) {
        const double r = rad * (1.0 + 1.7763568394002505e-15) + 1e-150;  // (1 + 2^-49)
        const double w = (dabs(q[d]) + r) * 4.4408920985006262e-16;      // 2^-51
        k0 = cell_coord<D>(gp, q[d] - r - w, d);
        k1 = cell_coord<D>(gp, q[d] + r + w, d);
    };
    const double rad = sqrt_up(best);
    int z0 = 0, z1 = 0, y0, y1;
    if (D == 3) range_of(2, rad, z0, z1);
    range_of(1, rad, y0, y1);
    if (z1 - z0 > 2 || y1 - y0 > 2) {
        grid_ball_scan<D, TIES, Rec, STEP>(gp, recs, start, q, c, best, bidx, visited, tie, cloud, q32);
        return;
    }
    constexpr int NR = D == 3 ? 9 : 3;
    const double best_in = best;
    int kb[NR], ke[NR];
#pragma unroll
    for (int t = 0; t < NR; t++) {
        const int y = y0 + t % 3, z = z0 + t / 3;
        bool in = y <= y1 && z <= z1;
        const double gz = D == 3 ? gap_of(2, in ? z : c[2]) : 0.0, gy = gap_of(1, in ? y : c[1]);
        const double far2 = (gz * gz + gy * gy) * down;
        in = in && !(far2 > best_in);
        int x0 = 0, x1 = 0;
        range_of(0, sqrt_up(in ? best_in - far2 + best_in * 9.0949470177292824e-13 : 0.0), x0, x1);  // as grid_ball_scan
        const int row = in ? (z * gp.g[1] + y) * gp.g[0] : 0;
        kb[t] = start[row + x0];
        ke[t] = in ? start[row + x1 + 1] : kb[t];
    }
    if constexpr (!RecTraits<Rec>::f32) {
        // exact search through the f32 filter records (grid_scan_rows_filtered): same answer
        if (filt != nullptr) {
            grid_scan_rows_filtered<D, NR, TIES>(filt, gp, q, kb, ke, best, bidx, visited, tie, cloud);
            return;
        }
    }
    grid_scan_rows<D, NR, TIES, Rec, STEP>(recs, q, kb, ke, best, bidx, visited, tie, cloud, q32);
}

// one query against the grid.  Returns (best d2, lowest index among equal d2); (inf, 0) when no
// target has a finite distance.
// (r0, best0, bidx0): the shells below r0 have been visited already, with this result (grid_block_query)
template <int D, bool TIES = false, typename Rec = GridRec, int STEP = 4>
ICPX_DEV void grid_query(const GridParams& gp, const Rec* __restrict__ recs,
                         const int* __restrict__ start, const double* q, double& best_out,
                         int& idx_out, int& visited, int& tie, const double* __restrict__ cloud, const int r0 = 0,
                         const double best0 = __builtin_inf(), const int bidx0 = 0x7fffffff, const float* q32 = nullptr) {
    const auto* qc = RecTraits<Rec>::query(q, q32);
    double best = best0;
    int bidx = bidx0;
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
        int r_begin = r0;
        if (r0 == 0 && gp.rmax >= 1) {
            // Shells 0 and 1 together — the 3^D cells around the query, almost always all it needs: the bounds of their
            // (up to 9) rows are fetched FIRST, as independent loads, and only then the records row by row; shell by
            // shell every row costs two dependent round trips before its first record arrives (the walk is a chain of
            // L2 latencies: 261 -> ... us on config 4).  Same cells, same (d2, index) rule, the stopping test of shell 1.
            constexpr int NR = D == 3 ? 9 : 3;
            const int xlo = c[0] - 1 > 0 ? c[0] - 1 : 0, xhi = c[0] + 1 < gx - 1 ? c[0] + 1 : gx - 1;
            int kb[NR], ke[NR];
#pragma unroll
            for (int t = 0; t < NR; t++) {
                const int y = c[1] - 1 + t % 3, z = D == 3 ? c[2] - 1 + t / 3 : 0;
                const bool in = y >= 0 && y < gy && z >= 0 && z < gz;
                const int row = in ? (z * gy + y) * gx : 0;
                kb[t] = start[row + xlo];
                ke[t] = in ? start[row + xhi + 1] : kb[t];
            }
            grid_scan_rows<D, NR, TIES, Rec, STEP>(recs, q, kb, ke, best, bidx, visited, tie, cloud, q32);
            // the stopping test after shell 1 (as at the end of the loop below)
            double lb = __builtin_inf();
            bool all_covered = true;
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int kl = c[d] - 1, kh = c[d] + 2;
                if (kl > 0) {
                    all_covered = false;
                    double gap = (q[d] - (gp.lo[d] + static_cast<double>(kl) * gp.h[d])) - slack[d];
                    gap = gap > 0.0 ? gap : 0.0;
                    lb = gap < lb ? gap : lb;
                }
                if (kh < gp.g[d]) {
                    all_covered = false;
                    double gap = ((gp.lo[d] + static_cast<double>(kh) * gp.h[d]) - q[d]) - slack[d];
                    gap = gap > 0.0 ? gap : 0.0;
                    lb = gap < lb ? gap : lb;
                }
            }
            r_begin = (all_covered || best < lb * lb * 0.99999999999909051) ? gp.rmax + 1 : 2;
        }
        for (int r = r_begin; r <= gp.rmax; r++) {
            if (r >= 2 && best < __builtin_inf()) {
                // shells 0 and 1 did not settle it: the rest through the ball of the candidate
                grid_ball_scan<D, TIES, Rec, STEP>(gp, recs, start, q, c, best, bidx, visited, tie, cloud, q32);
                break;
            }
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
                        for (int k0 = seg_b[s]; k0 < seg_e[s]; k0 += STEP) {
                            Rec rec[STEP];
#pragma unroll
                            for (int u = 0; u < STEP; u++) {
                                int k = k0 + u < seg_e[s] ? k0 + u : seg_e[s] - 1;
                                rec[u] = recs[k];
                            }
#pragma unroll
                            for (int u = 0; u < STEP; u++) {
                                const double dist = d2_eval<D>(rec[u].x, rec[u].y, rec[u].z, qc);
                                const int j = static_cast<int>(rec[u].idx);
                                const double cc[3] = {static_cast<double>(rec[u].x), static_cast<double>(rec[u].y), static_cast<double>(rec[u].z)};
                                nn_take<TIES, D>(k0 + u < seg_e[s], dist, j, cc, cloud, best, bidx, tie);
                            }
                            visited += seg_e[s] - k0 < STEP ? seg_e[s] - k0 : STEP;
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

// ---------------------------------------------------------------------------------------------
// Block-cooperative query.  The queries are sorted by TILE (grid_query_keys_kernel: tiles of 4 x 4 (x 2) cells,
// x fastest); a workgroup owns a block of four tiles along x — 16 x 4 (x 2) cells, ~256 queries at two targets
// per cell — and its queries are one contiguous range of the sorted order (qstart, grid_block_starts_kernel).
// The cells any of them needs for shells 0 and 1 are the block widened by one cell: those records are staged ONCE
// in LDS with coalesced loads (SoA, plus the box's slice of the cell start table rebased to LDS positions: 3.4
// records fetched per query instead of 52), and every lane scans its own 3^D cells from there.  Then the lane stops
// by the same test as grid_query (best < lower bound of everything outside its visited block) or goes on with
// shell 2 in global memory.  Same (d2, index) minimum: the rule does not depend on the order of evaluation.
// Queries that have left their block since the sort (the running cloud of a converge moves) and blocks whose
// records do not fit take grid_query from shell 0.
// ---------------------------------------------------------------------------------------------
constexpr int kGridBlkRecs = 1152;     // records staged per workgroup (a full block holds 18 x 6 x 4 cells)
constexpr int kGridBlkRows = 24;
constexpr int kGridBlkSlice = kGridBlkRows * 19;
constexpr size_t kGridBlkBytes = sizeof(double) * 3 * kGridBlkRecs + sizeof(int) * (kGridBlkRecs + kGridBlkSlice + 3 * kGridBlkRows + 8);

// tiles per axis; x padded to whole blocks of four tiles
inline __host__ __device__ void grid_tiles(const GridParams& gp, int dim, int (&nt)[3]) {
    nt[0] = (((gp.g[0] + 3) >> 2) + 3) & ~3;
    nt[1] = (gp.g[1] + 3) >> 2;
    nt[2] = dim == 3 ? (gp.g[2] + 1) >> 1 : 1;
}

// qstart[b] = first sorted query of block b (b = 0 .. nblocks; qstart[nblocks] = n), from the sorted keys
__global__ void grid_block_starts_kernel(const uint32_t* __restrict__ keys_sorted, int64_t n, int nblocks, int* __restrict__ qstart) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cur = static_cast<int>(keys_sorted[i] >> 7);  // key = tile << 5 | cell in tile; block = tile >> 2
    const int prev = i > 0 ? static_cast<int>(keys_sorted[i - 1] >> 7) : -1;
    for (int b = prev + 1; b <= cur; b++) qstart[b] = static_cast<int>(i);
    if (i == n - 1)
        for (int b = cur + 1; b <= nblocks; b++) qstart[b] = static_cast<int>(n);
}

// Workgroup-wide: stages the block's box.  Returns false (uniformly) when it does not fit.  b0 / b1: the box.
template <int D, typename Rec = GridRec>
ICPX_DEV bool grid_block_stage(const GridParams& gp, const Rec* __restrict__ recs, const int* __restrict__ start, const int blk,
                               char* lds, int (&b0)[3], int (&b1)[3]) {
    using C = typename RecTraits<Rec>::coord;
    const int tid = threadIdx.x, T = blockDim.x;
    int nt[3];
    grid_tiles(gp, D, nt);
    const int nbx = nt[0] >> 2;
    const int bx = blk % nbx, ty = (blk / nbx) % nt[1], tz = blk / (nbx * nt[1]);
    b0[0] = max(16 * bx - 1, 0);
    b1[0] = min(16 * bx + 16, gp.g[0] - 1);
    b0[1] = max(4 * ty - 1, 0);
    b1[1] = min(4 * ty + 4, gp.g[1] - 1);
    b0[2] = D == 3 ? max(2 * tz - 1, 0) : 0;
    b1[2] = D == 3 ? min(2 * tz + 2, gp.g[2] - 1) : 0;
    const int nx = b1[0] - b0[0] + 1, ny = b1[1] - b0[1] + 1, nz = b1[2] - b0[2] + 1;
    const int nrows = ny * nz;
    C* sx = reinterpret_cast<C*>(lds);
    int* sidx = reinterpret_cast<int*>(sx + 3 * kGridBlkRecs);
    int* slice = sidx + kGridBlkRecs;
    int* rowoff = slice + kGridBlkSlice;
    int* rowbeg = rowoff + kGridBlkRows;
    int* rowcell = rowbeg + kGridBlkRows;
    int* misc = rowcell + kGridBlkRows;   // [0] total records
    if (nx < 1 || nrows < 1 || nrows > kGridBlkRows || nx + 1 > 19) return false;  // (uniform)
    if (tid < 64) {  // wave 0: the rows of the box and their prefix sum
        int len = 0, beg = 0, cell0 = 0;
        if (tid < nrows) {
            const int z = b0[2] + tid / ny, y = b0[1] + tid % ny;
            cell0 = (z * gp.g[1] + y) * gp.g[0] + b0[0];
            beg = start[cell0];
            len = start[cell0 + nx] - beg;
        }
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            incl += tid >= o ? up : 0;
        }
        if (tid < nrows) {
            rowoff[tid] = incl - len;
            rowbeg[tid] = beg;
            rowcell[tid] = cell0;
        }
        if (tid == 63) misc[0] = incl;
    }
    __syncthreads();
    const int total = misc[0];
    if (total > kGridBlkRecs) return false;
    for (int i = tid; i < nrows * (nx + 1); i += T) {
        const int r = i / (nx + 1), k = i - r * (nx + 1);
        slice[i] = start[rowcell[r] + k] - rowbeg[r] + rowoff[r];
    }
    for (int i = tid; i < total; i += T) {
        int r = 0;  // the last row whose offset is <= i
#pragma unroll
        for (int step = 16; step >= 1; step >>= 1) {
            const int t = r + step;
            r = (t < nrows && rowoff[t] <= i) ? t : r;
        }
        const Rec rec = recs[rowbeg[r] + (i - rowoff[r])];
        sx[i] = rec.x;
        sx[kGridBlkRecs + i] = rec.y;
        if (D == 3) sx[2 * kGridBlkRecs + i] = rec.z;
        sidx[i] = static_cast<int>(rec.idx);
    }
    __syncthreads();
    return true;
}

// One query of a staged block (staged: grid_block_stage returned true).  Per lane; no barriers.
template <int D, bool TIES = false, typename Rec = GridRec>
ICPX_DEV void grid_block_query(const GridParams& gp, const Rec* __restrict__ recs, const int* __restrict__ start,
                               const double* q, const bool staged, const int (&b0)[3], const int (&b1)[3], const char* lds,
                               double& best_out, int& idx_out, int& visited, int& tie, const double* __restrict__ cloud,
                               const float* q32 = nullptr) {
    using C = typename RecTraits<Rec>::coord;
    const auto* qc = RecTraits<Rec>::query(q, q32);
    bool qfin = gp.m > 0;
#pragma unroll
    for (int d = 0; d < D; d++) qfin = qfin && finite_d(q[d]);
    if (!qfin) {
        best_out = __builtin_inf();
        idx_out = 0;
        return;
    }
    int c[3] = {0, 0, 0};
    bool inside = staged;
#pragma unroll
    for (int d = 0; d < D; d++) {
        c[d] = cell_coord<D>(gp, q[d], d);
        inside = inside && max(c[d] - 1, 0) >= b0[d] && min(c[d] + 1, gp.g[d] - 1) <= b1[d];
    }
    if (!inside) {
        grid_query<D, TIES, Rec>(gp, recs, start, q, best_out, idx_out, visited, tie, cloud, 0, __builtin_inf(), 0x7fffffff, q32);
        return;
    }
    const C* sx = reinterpret_cast<const C*>(lds);
    const int* sidx = reinterpret_cast<const int*>(sx + 3 * kGridBlkRecs);
    const int* slice = sidx + kGridBlkRecs;
    const int nx = b1[0] - b0[0] + 1, ny = b1[1] - b0[1] + 1;
    double best = __builtin_inf();
    int bidx = 0x7fffffff;
    const int xs = max(c[0] - 1, 0) - b0[0], xe = min(c[0] + 1, gp.g[0] - 1) - b0[0] + 1;
    for (int z = (D == 3 ? max(c[2] - 1, 0) : 0); z <= (D == 3 ? min(c[2] + 1, gp.g[2] - 1) : 0); z++)
        for (int y = max(c[1] - 1, 0); y <= min(c[1] + 1, gp.g[1] - 1); y++) {
            const int rr = (z - b0[2]) * ny + (y - b0[1]);
            const int s = slice[rr * (nx + 1) + xs], e = slice[rr * (nx + 1) + xe];
            for (int k = s; k < e; k++) {
                const C tx = sx[k], ty = sx[kGridBlkRecs + k], tz = D == 3 ? sx[2 * kGridBlkRecs + k] : C(0);
                const double dist = d2_eval<D>(tx, ty, tz, qc);
                const int j = sidx[k];
                const double cc[3] = {static_cast<double>(tx), static_cast<double>(ty), static_cast<double>(tz)};
                nn_take<TIES, D>(true, dist, j, cc, cloud, best, bidx, tie);
            }
            visited += e - s;
        }
    // the stopping test of grid_query after shell 1
    double lb = __builtin_inf();
    bool all_covered = true;
#pragma unroll
    for (int d = 0; d < D; d++) {
        const double slack = (dabs(gp.lo[d]) + dabs(gp.hi[d]) + dabs(q[d])) * 2.8421709430404007e-14;  // 2^-45
        const int kl = c[d] - 1, kh = c[d] + 2;
        if (kl > 0) {
            all_covered = false;
            double gap = (q[d] - (gp.lo[d] + static_cast<double>(kl) * gp.h[d])) - slack;
            gap = gap > 0.0 ? gap : 0.0;
            lb = gap < lb ? gap : lb;
        }
        if (kh < gp.g[d]) {
            all_covered = false;
            double gap = ((gp.lo[d] + static_cast<double>(kh) * gp.h[d]) - q[d]) - slack;
            gap = gap > 0.0 ? gap : 0.0;
            lb = gap < lb ? gap : lb;
        }
    }
    if (all_covered || best < lb * lb * 0.99999999999909051) {  // (1 - 2^-40)
        best_out = best;
        idx_out = bidx == 0x7fffffff ? 0 : bidx;
    } else {
        grid_query<D, TIES, Rec>(gp, recs, start, q, best_out, idx_out, visited, tie, cloud, 2, best, bidx, q32);
    }
}

// stand-alone query, one workgroup per block of tiles
// kd_list / kd_count (TIES): the queries with a tie partner are appended (icpx_kd.h), for kd_resolve_kernel
ICPX_DEV void kd_append(const int tie, const int64_t i, uint32_t* __restrict__ kd_list, int* __restrict__ kd_count) {
    if (tie) kd_list[atomicAdd(kd_count, 1)] = static_cast<uint32_t>(i);
}

template <int D, bool TIES = false, typename Rec = GridRec>
__global__ void __launch_bounds__(256)
grid_nn_block_kernel(const double* __restrict__ q, GridParams gp, const Rec* __restrict__ recs,
                     const int* __restrict__ start, int64_t* __restrict__ idx, double* __restrict__ d2,
                     unsigned long long* __restrict__ visited_total, const uint32_t* __restrict__ perm,
                     const int* __restrict__ qstart, const double* __restrict__ cloud = nullptr,
                     uint32_t* __restrict__ kd_list = nullptr, int* __restrict__ kd_count = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char grid_lds[];
    const int qs = qstart[blockIdx.x], qe = qstart[blockIdx.x + 1];
    if (qs >= qe) return;  // (uniform)
    int b0[3], b1[3];
    const bool staged = grid_block_stage<D, Rec>(gp, recs, start, blockIdx.x, grid_lds, b0, b1);
    int visited = 0;
    for (int k = qs + threadIdx.x; k < qe; k += blockDim.x) {
        const int64_t i = perm[k];
        double qq[D];
#pragma unroll
        for (int d = 0; d < D; d++) qq[d] = q[i * D + d];
        double best;
        int bi;
        int tie = 0;
        float q32[3];
        if (RecTraits<Rec>::f32) query_f32<D>(gp.lo, qq, q32);
        grid_block_query<D, TIES, Rec>(gp, recs, start, qq, staged, b0, b1, grid_lds, best, bi, visited, tie, cloud, q32);
        if (idx) idx[i] = bi;
        if (d2) d2[i] = best;
        if (TIES) kd_append(tie, i, kd_list, kd_count);
    }
    if (visited_total) {
        int v = wave_sum_int(visited);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(visited_total, static_cast<unsigned long long>(v));
    }
}

template <int D, bool TIES = false, typename Rec = GridRec>
__global__ void __launch_bounds__(256)
grid_nn_kernel(const double* __restrict__ q, int64_t n, GridParams gp, const Rec* __restrict__ recs,
               const int* __restrict__ start, int64_t* __restrict__ idx, double* __restrict__ d2,
               unsigned long long* __restrict__ visited_total, const uint32_t* __restrict__ perm,
               const double* __restrict__ cloud = nullptr, uint32_t* __restrict__ kd_list = nullptr,
               int* __restrict__ kd_count = nullptr) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    int visited = 0;
    const bool live = i < n;
    // queries are visited in tile order (perm, grid_sort_queries) so that neighbouring lanes walk the same cells
    if (live && perm) i = perm[i];
    if (live) {
        double qq[D];
#pragma unroll
        for (int d = 0; d < D; d++) qq[d] = q[i * D + d];
        double best;
        int bi;
        int tie = 0;
        float q32[3];
        if (RecTraits<Rec>::f32) query_f32<D>(gp.lo, qq, q32);
        grid_query<D, TIES, Rec>(gp, recs, start, qq, best, bi, visited, tie, cloud, 0, __builtin_inf(), 0x7fffffff, q32);
        if (idx) idx[i] = bi;
        if (d2) d2[i] = best;
        if (TIES) kd_append(tie, i, kd_list, kd_count);
    }
    if (visited_total) {
        int v = wave_sum_int(visited);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(visited_total, static_cast<unsigned long long>(v));
    }
}

// Sort key of grid_sort_queries: the query's cell (clamped into the grid) in TILE order — tiles of 4 x 4 (x 2)
// cells, x fastest over the tiles (the tile count along x padded to a multiple of four: grid_tiles) and inside a
// tile — so that consecutive queries cover a compact block of cells instead of a run along x, and four tiles
// along x form the block of one workgroup of the block-cooperative query
template <int D>
__global__ void grid_query_keys_kernel(const double* __restrict__ q, int64_t n, GridParams gp,
                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ iota) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p[D];
#pragma unroll
    for (int d = 0; d < D; d++) p[d] = q[i * D + d];
    const int cx = cell_coord<D>(gp, p[0], 0), cy = cell_coord<D>(gp, p[1], 1), cz = D == 3 ? cell_coord<D>(gp, p[2], 2) : 0;
    int nt[3];
    grid_tiles(gp, D, nt);
    const uint32_t tile = static_cast<uint32_t>(((cz >> 1) * nt[1] + (cy >> 2)) * nt[0] + (cx >> 2));
    keys[i] = (tile << 5) | static_cast<uint32_t>((((cz & 1) * 4) + (cy & 3)) * 4 + (cx & 3));
    iota[i] = static_cast<uint32_t>(i);
}

}  // namespace icpx
