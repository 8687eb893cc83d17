// icpx_prune.h — grid-pruned exact nearest-neighbour search for the fused pair kernel (batches).
//
// The brute-force search of icpx_filter.h looks at every (query, target) pair.  Here a pre-pass
// (pair_perm_kernel, icpx_pair.h) bins every target cloud into a small uniform grid (<= 4096 cells,
// a handful of targets per cell) and sorts it by cell; the fused kernel stages the cloud into LDS in
// that order together with the cell start table.  A query then only looks at the cells that can
// hold a target within its current bound:
//   seeded   bound = exact d2 to the previous neighbour (seed_bound)
//   seedless bound = best of the 3^D cells around the query (complete if that ball stays inside them)
// and walks those cells two targets per step, every lane on its own.
//
// Exactness.  cell_of() is a monotone function of the coordinate, evaluated by the same code for
// targets (pre-pass) and queries (here).  A target with exact d2 <= bound differs from the query
// by at most R = sqrt(bound) on every axis, so its cell index lies in [cell_of(q - R'), cell_of(q +
// R')] for any R' >= R that survives the roundings of q -/+ R' (grid_range).  Every target of those
// cells is evaluated with the reference's f64 expression and compared by (d2, index in the CLOUD)
// — the permutation table settles exact ties — so the result is the lexicographic minimum over
// ALL targets, bit for bit the one the brute-force search gives.  Clouds or queries with
// non-finite or huge (>= 1e150) coordinates take an exact scan of everything (exact_scan_p).
#pragma once

#include "icpx_filter.h"

namespace icpx {

constexpr int kGridMaxCells = 4096;
constexpr int kGridTable = kGridMaxCells + 8;  // u16 entries reserved per cloud (cells + 1, padded)

struct PairGrid {        // per target cloud, written by the pre-pass
    double lo[3];
    double scale[3];     // cells per unit length
    int G;               // cells per axis
    int bad;             // a coordinate is not finite or >= 1e150: no pruning
};

struct GridView {
    const uint16_t* cstart;  // LDS: first slot of cell c, c = 0 .. G^D (last = m)
    const uint16_t* tperm;   // global: cloud index of LDS slot j
    PairGrid g;
    int m;
};

ICPX_DEV int cell_of(double v, double lo, double scale, int G) {
    double x = (v - lo) * scale;
    x = __builtin_fmin(__builtin_fmax(x, 0.0), static_cast<double>(G - 1));  // NaN -> 0
    return static_cast<int>(x);
}

#ifndef ICPX_GRID_DENSITY
#define ICPX_GRID_DENSITY 1
#endif
// cells per axis for a cloud of m points: about ICPX_GRID_DENSITY targets per cell, at most
// kGridMaxCells cells
inline __host__ __device__ int grid_cells_per_axis(int dim, int m) {
    int G = 1;
    if (dim == 3) {
        while (G < 16 && (G + 1) * (G + 1) * (G + 1) * ICPX_GRID_DENSITY <= m) G++;
    } else {
        while (G < 64 && (G + 1) * (G + 1) * ICPX_GRID_DENSITY <= m) G++;
    }
    return G;
}

// Exact scan of all targets: same outcome as nn_lds on the unpermuted cloud, including its
// conventions for non-finite distances (nothing below +inf -> cloud index 0).
template <int D>
ICPX_DEV void exact_scan_p(const double (&q)[D], const LdsView<D>& L, const GridView& P, double& best, int& bslot) {
    double b = __builtin_inf();
    int bs = 0, bo = 0x7fffffff, zslot = 0;
    for (int j = 0; j < P.m; j++) {
        double df = L.t[0][j] - q[0];
        double acc = df * df;
#pragma unroll
        for (int d = 1; d < D; d++) {
            df = L.t[d][j] - q[d];
            acc = acc + df * df;
        }
        const int o = P.tperm[j];
        zslot = o == 0 ? j : zslot;
        const bool better = acc < b || (acc == b && acc < __builtin_inf() && o < bo);
        b = better ? acc : b;
        bs = better ? j : bs;
        bo = better ? o : bo;
    }
    if (!(b < __builtin_inf())) bs = zslot;
    best = b;
    bslot = bs;
}

// exact (d2, cloud index) update with the target in slot j
template <int D>
ICPX_DEV void grid_eval(const double (&q)[D], const bool act, const int j, const LdsView<D>& L, const GridView& P,
                        double& best, int& bslot) {
    double df = L.t[0][j] - q[0];
    double acc = df * df;
#pragma unroll
    for (int d = 1; d < D; d++) {
        df = L.t[d][j] - q[d];
        acc = acc + df * df;
    }
    bool better = act && acc < best;
    const bool tie = act && acc == best && j != bslot && bslot >= 0;
    if (__any(tie)) {
        if (tie) better = P.tperm[j] < P.tperm[bslot];
    }
    best = better ? acc : best;
    bslot = better ? j : bslot;
}

// One query per lane walks the cells [c0, c1] (inclusive, per axis).  The cells of one x-row are
// neighbours in the sorted order (x runs fastest in the cell id), so a row is ONE run of slots;
// two targets per step.  (best, bslot): running lexicographic minimum of (d2, cloud index);
// bslot < 0 = none yet.
template <int D>
ICPX_DEV void grid_walk(const double (&q)[D], bool act, const int (&c0)[D], const int (&c1)[D], const LdsView<D>& L,
                        const GridView& P, double& best, int& bslot) {
    const int G = P.g.G;
    int cy = c0[1], cz = D == 3 ? c0[D - 1] : 0;
    int j = 0, e = 0;
    bool fresh = true;  // the row (cy, cz) has not been opened yet
    bool done = !act;
#ifdef ICPX_STATS
    unsigned st_steps = 0, st_evals = 0;
#endif
    while (__any(!done)) {
#ifdef ICPX_STATS
        st_steps++;
        st_evals += __popcll(__ballot(!done && j < e));
#endif
        if (!done && j >= e) {
            if (!fresh) {
                cy++;
                if (cy > c1[1]) {
                    if (D == 3) {
                        cy = c0[1];
                        cz++;
                        if (cz > c1[D - 1]) done = true;
                    } else {
                        done = true;
                    }
                }
            }
            fresh = false;
            if (!done) {
                const int row = (cz * G + cy) * G;
                j = P.cstart[row + c0[0]];
                e = P.cstart[row + c1[0] + 1];
            }
        }
        const bool go = !done && j < e;
        if (__any(go)) {
            const bool two = go && j + 1 < e;
            const int ja = go ? j : 0, jb = two ? j + 1 : ja;
            grid_eval<D>(q, go, ja, L, P, best, bslot);
            grid_eval<D>(q, two, jb, L, P, best, bslot);
            j += two ? 2 : (go ? 1 : 0);
        }
    }
#ifdef ICPX_STATS
    if ((threadIdx.x >> 6) == 0) {  // wave 0 of every workgroup
        ICPX_STAT(0, 1);
        ICPX_STAT(12, st_steps);
        ICPX_STAT(1, st_evals);
    }
#endif
}

// cell range that holds every target within sqrt(bound) of q on each axis
template <int D>
ICPX_DEV void grid_range(const double (&q)[D], double bound, const PairGrid& g, int (&c0)[D], int (&c1)[D]) {
    // A target whose COMPUTED d2 is <= bound has fl(dx * dx) <= bound for dx = fl(t - q) on every axis
    // (the sum only adds non-negative terms and rounding is monotone), hence |t - q| <= sqrt(bound) *
    // (1 + 2^-51) + 2^-537 (the second term: squares that underflow).  R' covers that: the correctly
    // rounded root is >= (1 - 2^-53) of the true one, times (1 + 2^-50), plus 1e-150
    const double R = __builtin_sqrt(bound) * (1.0 + 1.0 / 1125899906842624.0) + 1e-150;
#pragma unroll
    for (int d = 0; d < D; d++) {
        // q -/+ R' rounds by at most 2^-53 (|q| + R'): widen by 2^-51 of that
        const double pad = (__builtin_fabs(q[d]) + R) * (1.0 / 2251799813685248.0) + 1e-300;
        c0[d] = cell_of(q[d] - R - pad, g.lo[d], g.scale[d], g.G);
        c1[d] = cell_of(q[d] + R + pad, g.lo[d], g.scale[d], g.G);
    }
}

// Nearest neighbour of one query per lane.  seeded: (best, bslot) = exact d2 and slot of some
// target on entry.  Out: lexicographic minimum of (d2, cloud index) over all targets, as its slot.
template <int D>
ICPX_DEV void nn_grid_slot(const double (&q)[D], const bool v, const bool seeded, const LdsView<D>& L,
                           const GridView& P, double& best, int& bslot) {
    bool fin = P.g.bad == 0;
#pragma unroll
    for (int d = 0; d < D; d++) fin = fin && (__builtin_fabs(q[d]) < 1e150);
    if (__any(v && !fin)) {
        double lb;
        int ls;
        exact_scan_p<D>(q, L, P, lb, ls);
        bool take = true;
        if (seeded) take = lb < best || (lb == best && P.tperm[ls] < P.tperm[bslot]);
        best = take ? lb : best;
        bslot = take ? ls : bslot;
        return;
    }
    int c0[D], c1[D];
    bool need = v;
    if (!seeded) {
        best = __builtin_inf();
        bslot = -1;
        // first bound: the best of the 3^D cells around the query.  With about one target per cell
        // the ball of that bound rarely leaves this box, and then the search is already complete
        int b0[D], b1[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int cq = cell_of(q[d], P.g.lo[d], P.g.scale[d], P.g.G);
            b0[d] = cq > 0 ? cq - 1 : 0;
            b1[d] = cq < P.g.G - 1 ? cq + 1 : P.g.G - 1;
        }
        grid_walk<D>(q, v, b0, b1, L, P, best, bslot);
        grid_range<D>(q, best, P.g, c0, c1);  // best = +inf (nothing found): every cell
        bool inside = true;
#pragma unroll
        for (int d = 0; d < D; d++) inside = inside && c0[d] >= b0[d] && c1[d] <= b1[d];
        need = v && !inside;
    } else {
        grid_range<D>(q, best, P.g, c0, c1);
    }
    if (__any(need)) grid_walk<D>(q, need, c0, c1, L, P, best, bslot);
}

}  // namespace icpx
