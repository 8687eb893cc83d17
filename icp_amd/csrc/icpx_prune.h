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
    int bad;             // bit 0: a coordinate is not finite or >= 1e150: no pruning; bit 1: occupied cells hold many
                         // points each (prepare_target_grid): the brute-force kernels suit this cloud better
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
// near-tie test of the kd rule between the candidate in slot j (d2 = acc) and the best so far (icpx_kd.h): another
// target within kKdNearUlps of it that does not coincide with it
template <int D>
ICPX_DEV bool slot_tie(const bool act, const double acc, const int j, const double best, const int bslot, const LdsView<D>& L) {
    bool near = act && bslot >= 0 && j != bslot && acc < __builtin_inf() && kd_near(acc, best);
    if (near && acc == best) {
        bool same = true;
#pragma unroll
        for (int d = 0; d < D; d++) same = same && L.t[d][j] == L.t[d][bslot];
        near = !same;
    }
    return near;
}

template <int D, bool TIES = false>
ICPX_DEV void exact_scan_p(const double (&q)[D], const LdsView<D>& L, const GridView& P, double& best, int& bslot, int* tie = nullptr) {
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
        if (TIES) *tie |= slot_tie<D>(bo != 0x7fffffff, acc, j, b, bs, L) ? 1 : 0;
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
template <int D, bool TIES = false>
ICPX_DEV void grid_eval(const double (&q)[D], const bool act, const int j, const LdsView<D>& L, const GridView& P,
                        double& best, int& bslot, int* tie = nullptr) {
    double df = L.t[0][j] - q[0];
    double acc = df * df;
#pragma unroll
    for (int d = 1; d < D; d++) {
        df = L.t[d][j] - q[d];
        acc = acc + df * df;
    }
    if (TIES) *tie |= slot_tie<D>(act, acc, j, best, bslot, L) ? 1 : 0;
    bool better = act && acc < best;
    const bool eq = act && acc == best && j != bslot && bslot >= 0;
    if (__any(eq)) {
        if (eq) better = P.tperm[j] < P.tperm[bslot];
    }
    best = better ? acc : best;
    bslot = better ? j : bslot;
}

// One query per lane walks the cells [c0, c1] (inclusive, per axis).  The cells of one x-row are
// neighbours in the sorted order (x runs fastest in the cell id), so a row is ONE run of slots;
// two targets per step.  (best, bslot): running lexicographic minimum of (d2, cloud index);
// bslot < 0 = none yet.
template <int D, bool TIES = false>
ICPX_DEV void grid_walk(const double (&q)[D], bool act, const int (&c0)[D], const int (&c1)[D], const LdsView<D>& L,
                        const GridView& P, double& best, int& bslot, int* tie = nullptr) {
    const int G = P.g.G;
    int cy = c0[1], cz = D == 3 ? c0[D - 1] : 0;
    int j = 0, e = 0;
    bool fresh = true;  // the row (cy, cz) has not been opened yet
    bool done = !act;
#ifdef ICPX_STATS
    unsigned st_steps = 0, st_evals = 0, st_real = 0, st_issued = 0;
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
#ifdef ICPX_STATS
            st_real += __popcll(__ballot(go)) + __popcll(__ballot(two));
            st_issued += 128;
#endif
            const int ja = go ? j : 0, jb = two ? j + 1 : ja;
            grid_eval<D, TIES>(q, go, ja, L, P, best, bslot, tie);
            grid_eval<D, TIES>(q, two, jb, L, P, best, bslot, tie);
            j += two ? 2 : (go ? 1 : 0);
        }
    }
#ifdef ICPX_STATS
    if ((threadIdx.x >> 6) == 0) {  // wave 0 of every workgroup
        ICPX_STAT(0, 1);
        ICPX_STAT(12, st_steps);
        ICPX_STAT(1, st_evals);
    }
    ICPX_STAT(26, st_real);    // every wave: distance evaluations of real candidates ...
    ICPX_STAT(27, st_issued);  // ... and the lane slots issued for them
#endif
}

// cell range that holds every target within sqrt(bound) of q on each axis
template <int D>
ICPX_DEV void grid_range(const double (&q)[D], double bound, const PairGrid& g, int (&c0)[D], int (&c1)[D]) {
    // A target whose COMPUTED d2 is <= bound has fl(dx * dx) <= bound for dx = fl(t - q) on every axis
    // (the sum only adds non-negative terms and rounding is monotone), hence |t - q| <= sqrt(bound) *
    // (1 + 2^-51) + 2^-537 (the second term: squares that underflow).  R' covers that: the correctly
    // rounded root is >= (1 - 2^-53) of the true one, times (1 + 2^-50), plus 1e-150
    // (2^-48 instead of the 2^-50 that would do: the kd rule's tie partners, up to kKdNearUlps above the bound, are
    // inside the box too)
    const double R = __builtin_sqrt(bound) * (1.0 + 1.0 / 281474976710656.0) + 1e-150;
#pragma unroll
    for (int d = 0; d < D; d++) {
        // q -/+ R' rounds by at most 2^-53 (|q| + R'): widen by 2^-51 of that
        const double pad = (__builtin_fabs(q[d]) + R) * (1.0 / 2251799813685248.0) + 1e-300;
        c0[d] = cell_of(q[d] - R - pad, g.lo[d], g.scale[d], g.G);
        c1[d] = cell_of(q[d] + R + pad, g.lo[d], g.scale[d], g.G);
    }
}

// grid_range for the cooperative search, fewer instructions: the same box or a slightly larger one.  In cell
// units u(v) = (v - lo) * scale a target within R' of q has u(t) >= u(q) - R' * scale; cell_of's own roundings
// move fl(u) by less than 2^-50 (|u| + 1), which the margin (1e-12 relative, 1e-9 cells absolute) covers, and
// the clamp to [0, G - 1] is monotone.
template <int D>
ICPX_DEV void grid_range_cells(const double (&q)[D], double bound, const PairGrid& g, int (&c0)[D], int (&c1)[D]) {
    const double R = __builtin_sqrt(bound) * (1.0 + 1.0 / 281474976710656.0) + 1e-150;  // as in grid_range
    const double gm = static_cast<double>(g.G - 1);
#pragma unroll
    for (int d = 0; d < D; d++) {
        const double x = (q[d] - g.lo[d]) * g.scale[d];
        const double w = (R + (__builtin_fabs(q[d]) + R) * (1.0 / 2251799813685248.0)) * g.scale[d];
        const double pad = (__builtin_fabs(x) + w) * 1e-12 + 1e-9;
        const double a = __builtin_fmin(__builtin_fmax(x - w - pad, 0.0), gm);  // NaN -> 0
        const double b = __builtin_fmin(__builtin_fmax(x + w + pad, 0.0), gm);
        c0[d] = static_cast<int>(a);
        c1[d] = static_cast<int>(b);
    }
}

// Seed certificate.  hc (pair_perm_kernel) is a lower bound of the distance from the seed s to every OTHER target
// t of the cloud.  If 2.000001 * |q - s| < hc then |q - t| >= |s - t| - |q - s| > 1.0000009 |q - s| for every
// t != s: the computed d2(q, t) exceeds the computed d2(q, s) (their roundings are 1e-15 relative) — the seed is
// the unique nearest target and no search is needed.  hc = 0 (duplicated or degenerate targets) never certifies.
ICPX_DEV bool seed_certified(double d2_seed, double hc) {
    return hc > 0.0 && 2.000001 * __builtin_sqrt(d2_seed) < hc;
}

// Nearest neighbour of one query per lane.  seeded: (best, bslot) = exact d2 and slot of some
// target on entry.  Out: lexicographic minimum of (d2, cloud index) over all targets, as its slot.
template <int D, bool TIES = false>
ICPX_DEV void nn_grid_slot(const double (&q)[D], const bool v, const bool seeded, const LdsView<D>& L,
                           const GridView& P, double& best, int& bslot, int* tie = nullptr) {
    bool fin = (P.g.bad & 1) == 0;
#pragma unroll
    for (int d = 0; d < D; d++) fin = fin && (__builtin_fabs(q[d]) < 1e150);
    if (__any(v && !fin)) {
        double lb;
        int ls;
        exact_scan_p<D, TIES>(q, L, P, lb, ls, tie);  // (every target, the seed among them)
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
        grid_walk<D, TIES>(q, v, b0, b1, L, P, best, bslot, tie);
        grid_range<D>(q, best, P.g, c0, c1);  // best = +inf (nothing found): every cell
        bool inside = true;
#pragma unroll
        for (int d = 0; d < D; d++) inside = inside && c0[d] >= b0[d] && c1[d] <= b1[d];
        need = v && !inside;
    } else {
        grid_range<D>(q, best, P.g, c0, c1);
    }
    if (__any(need)) grid_walk<D, TIES>(q, need, c0, c1, L, P, best, bslot, tie);
}

// ---------------------------------------------------------------------------------------------
// Wave-cooperative search (default; ICPX_COOP=0 selects the per-lane walk above).
//
// The per-lane walk is a pointer chase in which every lane waits for the slowest lane of its wave
// at two levels (rows of the box, targets of a row): about 30 % of the lanes do useful work per step.
// Here the 64 queries of a wave (one slot of every lane) are searched TOGETHER:
//   1. every lane turns its bound into a cell box (grid_range) and reads the start / end slot of the box's
//      rows (at most 3 x 3 rows; lanes with larger boxes keep the per-lane walk): branch-free, clamped reads;
//   2. the candidates (query lane, target slot) of all lanes go into ONE flat list in LDS, compacted across the
//      lanes by ballot / mbcnt, (row, position in row)-major;
//   3. the flat list is evaluated 64 candidates at a time — every lane busy, the loads of a step independent
//      of the step before — with the reference's f64 expression; the squared distances go to the query's
//      accumulator by a 64-bit LDS atomic minimum (non-negative doubles order like their bit patterns);
//   4. the candidates that attain the minimum report their slot (atomic min and max); where the two differ —
//      an exact tie between two targets — the tied candidates compete by (cloud index, slot) through tperm.
// The result is the lexicographic minimum of (d2, cloud index) over the targets of the box: what grid_walk
// returns, bit for bit.  Per-wave LDS staging: CoopStage.  Wave-slots with more candidates than the list
// holds are searched in lane groups (halves, quarters, ...).
// ---------------------------------------------------------------------------------------------
#ifndef ICPX_LIST_STRIDED
#define ICPX_LIST_STRIDED 1
#endif
#ifndef ICPX_RESOLVE_MASKED
#define ICPX_RESOLVE_MASKED 0
#endif
#ifndef ICPX_LIST_FIRST_3D
#define ICPX_LIST_FIRST_3D 2
#endif
#ifndef ICPX_COOP_BLOCK
#define ICPX_COOP_BLOCK 2
#endif
constexpr int kCoopCap = 512;      // candidates of one pass that fit the flat list (8 steps of 64)
constexpr int kCoopSteps = kCoopCap / 64;
constexpr int kCoopBlock = ICPX_COOP_BLOCK;      // steps whose loads are issued together

inline __host__ __device__ size_t coop_wave_bytes(int dim) {
    return static_cast<size_t>(64) * 8 * dim + 64 * 8 + 64 * 4 + 64 * 4 + static_cast<size_t>(kCoopCap + 64) * 4 + 64 * 4;
}

template <int D>
struct CoopStage {
    double* q;                  // [D][64] the wave's queries
    unsigned long long* accd;   // [64] running minimum of d2 (bit pattern) per query
    unsigned* acci;             // [64] lowest slot at the minimum; (cloud index << 16 | slot) after a tie
    unsigned* accx;             // [64] highest slot at the minimum; 0xffffffff marks a tied query
    unsigned* list;             // [kCoopCap + 64] (query lane << 16 | target slot); the last 64: per-lane dump
    unsigned* acct;             // [64] kd rule: non-zero = a non-coincident target within kKdNearUlps of the minimum
};

template <int D>
ICPX_DEV CoopStage<D> coop_carve(char* base) {  // base: this wave's staging area
    CoopStage<D> S;
    S.q = reinterpret_cast<double*>(base);
    S.accd = reinterpret_cast<unsigned long long*>(base + 64 * 8 * D);
    S.acci = reinterpret_cast<unsigned*>(base + 64 * 8 * D + 64 * 8);
    S.accx = S.acci + 64;
    S.list = S.accx + 64;
    S.acct = S.list + kCoopCap + 64;
    return S;
}

// sum over the wave, in a scalar
ICPX_DEV int wave_total_int(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);  // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);  // row_ror:8
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

// inclusive prefix sum over the lanes of a wave (DPP: row scan, then the row totals)
ICPX_DEV int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
    int t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1, out-of-row lanes read 0
    v += t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += t;
    // v = inclusive scan inside each row of 16 lanes; add the totals of the rows before
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    return v + (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
}

// One pass: the lanes in `grp` (their candidates fit the list: the caller has counted) put their candidates
// into the list, the whole wave evaluates it.  rs / rl: first slot and length of the box's rows (rl = 0 for the
// rows a lane does not have); nrows: rows any lane of the wave has.  Out, for the lanes of grp: (rb, rsl) = the
// box's lexicographic minimum of (d2, cloud index) and its slot, (+inf, -1) for an empty box.
// TIES (icpx_kd.h): the candidates within kKdNearUlps of the minimum report their slots too (instead of those AT the
// minimum only), so `tied` also covers the kd rule's near ties; *tie |= the query has such a candidate that does not
// coincide with the others.  The (d2, cloud index) answer is unchanged.
template <int D, int NR, bool TIES = false>
ICPX_DEV void coop_pass(const double (&q)[D], const bool grp, const int (&rs)[NR], const int (&rl)[NR], const int nrows,
                        const LdsView<D>& L, const GridView& P, const CoopStage<D>& S, double& rb, int& rsl, int* tie = nullptr) {
    const int lane = threadIdx.x & 63;
    ICPX_CYC_BEGIN();
#pragma unroll
    for (int d = 0; d < D; d++) S.q[d * 64 + lane] = q[d];
    S.accd[lane] = 0x7ff0000000000000ull;  // +inf
    S.acci[lane] = 0xffffffffu;
    S.accx[lane] = 0u;
    if (TIES) S.acct[lane] = 0u;
    // ---- the flat list, query-major: a lane's candidates follow those of the lanes before it ----
    int total = 0;
#pragma unroll
    for (int r = 0; r < NR; r++) total += grp ? rl[r] : 0;
    const int incl = wave_incl_scan(total);
    const int base = __builtin_amdgcn_readlane(incl, 63);  // uniform: candidates of this pass
    const unsigned tag = static_cast<unsigned>(lane) << 16;
    const int dump = kCoopCap + lane;
    int pos = incl - total;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        if (r % 3 < (nrows & 3) && r / 3 < (nrows >> 2)) {  // uniform: some lane of the wave has this row
            const int len = grp ? rl[r] : 0;
            // the first few of a row without a branch (a row rarely holds more), the rest in a loop.  Measured: two for
            // the 9 rows of a 3D box (3D 256 x 8: 0.686 -> 0.668 ms), four for the 3 rows of a 2D box
            constexpr int kFirst = D == 3 ? ICPX_LIST_FIRST_3D : 4;
#pragma unroll
            for (int u = 0; u < kFirst; u++) S.list[u < len ? pos + u : dump] = tag | static_cast<unsigned>(rs[r] + u);
            if (__any(len > kFirst)) {
                for (int u = kFirst; __any(u < len); u++) S.list[u < len ? pos + u : dump] = tag | static_cast<unsigned>(rs[r] + u);
            }
            pos += len;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ICPX_CYC(19);  // list
#ifdef ICPX_CYCONLY
    if (threadIdx.x == 0) {
        atomicAdd(&g_icpx_stats[6], 1ull);
        atomicAdd(&g_icpx_stats[7], static_cast<unsigned long long>(base));
    }
#endif
    if (threadIdx.x == 0) ICPX_STAT(22, base);
    ICPX_STAT(26, base);
    ICPX_STAT(27, 64 * kCoopBlock * ((base + 64 * kCoopBlock - 1) / (64 * kCoopBlock)));
    ICPX_STAT(28, 1);  // cooperative passes (wave level)
    // ---- evaluate the flat list: 64 candidates per step, kCoopBlock steps per block (loads first, then arithmetic) ----
    unsigned long long dk[kCoopSteps];
    unsigned ent[kCoopSteps];
#pragma unroll
    for (int c = 0; c < kCoopSteps; c++) {
        dk[c] = ~0ull;
        ent[c] = tag;
    }
    const int nsteps = (base + 63) >> 6;  // uniform
#pragma unroll
    for (int cb = 0; cb < kCoopSteps; cb += kCoopBlock) {
        if (cb * 64 < base) {  // uniform
            unsigned e[kCoopBlock];
            bool on[kCoopBlock];
#pragma unroll
            for (int u = 0; u < kCoopBlock; u++) {
#if ICPX_LIST_STRIDED
                // lane l takes entries l * nsteps + step: a query's candidates are neighbours in the list, and this way
                // they meet in different steps instead of hitting the query's accumulator in one LDS atomic instruction
                const int i = lane * nsteps + (cb + u);
                on[u] = (cb + u) < nsteps && i < base;
#else
                const int i = (cb + u) * 64 + lane;
                on[u] = i < base;
#endif
                e[u] = S.list[on[u] ? i : dump];
            }
            double tq[kCoopBlock][D], tt[kCoopBlock][D];
#pragma unroll
            for (int u = 0; u < kCoopBlock; u++) {
                e[u] = on[u] ? e[u] : tag;  // lanes past the end: slot 0 against their own query, never counted
                const int ql = static_cast<int>(e[u] >> 16), j = static_cast<int>(e[u] & 0xffffu);
#pragma unroll
                for (int d = 0; d < D; d++) {
                    tq[u][d] = S.q[d * 64 + ql];
                    tt[u][d] = L.t[d][j];
                }
            }
#pragma unroll
            for (int u = 0; u < kCoopBlock; u++) {
                double df = tt[u][0] - tq[u][0];
                double acc = df * df;
#pragma unroll
                for (int d = 1; d < D; d++) {
                    df = tt[u][d] - tq[u][d];
                    acc = acc + df * df;
                }
                const unsigned long long key = on[u] ? static_cast<unsigned long long>(__double_as_longlong(acc)) : ~0ull;
                ent[cb + u] = e[u];
                dk[cb + u] = key;
                __hip_atomic_fetch_min(&S.accd[e[u] >> 16], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ICPX_CYC(20);  // evaluation
    // ---- slot of the minimum: the candidates at the minimum report their slot; the others aim at their own
    //      lane's words with values that change nothing ----
#pragma unroll
    for (int cb = 0; cb < kCoopSteps; cb += kCoopBlock) {
        if (cb * 64 < base) {
            unsigned long long mn[kCoopBlock];
#pragma unroll
            for (int u = 0; u < kCoopBlock; u++) mn[u] = S.accd[ent[cb + u] >> 16];
#pragma unroll
            for (int u = 0; u < kCoopBlock; u++) {
                const unsigned ql = ent[cb + u] >> 16, j = ent[cb + u] & 0xffffu;
                const bool win = dk[cb + u] != ~0ull && (TIES ? (dk[cb + u] - mn[u] <= kKdNearUlps && dk[cb + u] < 0x7ff0000000000000ull)
                                                                 : dk[cb + u] == mn[u]);
#if ICPX_RESOLVE_MASKED
                if (win) {
                    __hip_atomic_fetch_min(&S.acci[ql], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_max(&S.accx[ql], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#else
                const unsigned w = win ? ql : static_cast<unsigned>(lane);
                __hip_atomic_fetch_min(&S.acci[w], win ? j : 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&S.accx[w], win ? j : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool mine = grp && total > 0;
    const bool tied = mine && S.acci[lane] != S.accx[lane];
    if (__any(tied)) {
        if (TIES) {
            // kd rule: does any candidate in the tie range differ in coordinates from the lowest such slot?
#pragma unroll
            for (int c = 0; c < kCoopSteps; c++) {
                if (c * 64 < base) {
                    const unsigned ql = ent[c] >> 16, j = ent[c] & 0xffffu;
                    const unsigned lo_slot = S.acci[ql], hi_slot = S.accx[ql];
                    if (dk[c] != ~0ull && dk[c] < 0x7ff0000000000000ull && dk[c] - S.accd[ql] <= kKdNearUlps && lo_slot != hi_slot && j != lo_slot) {
                        bool same = true;
#pragma unroll
                        for (int d = 0; d < D; d++) same = same && L.t[d][j] == L.t[d][lo_slot];
                        if (!same) S.acct[ql] = 1u;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // exact ties between different targets: the candidates at the minimum compete by cloud index
        if (tied) {
            S.acci[lane] = 0xffffffffu;
            S.accx[lane] = 0xffffffffu;  // marks the query as tied
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < kCoopSteps; c++) {
            if (c * 64 < base) {
                const unsigned ql = ent[c] >> 16, j = ent[c] & 0xffffu;
                if (dk[c] != ~0ull && dk[c] == S.accd[ql] && S.accx[ql] == 0xffffffffu) {
                    const unsigned key = (static_cast<unsigned>(P.tperm[j]) << 16) | j;
                    __hip_atomic_fetch_min(&S.acci[ql], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    ICPX_CYC(21);  // resolution
    if (grp) {
        if (total > 0) {
            rb = __longlong_as_double(static_cast<long long>(S.accd[lane]));
            rsl = static_cast<int>(S.acci[lane] & 0xffffu);
            if (TIES) *tie |= S.acct[lane] != 0u ? 1 : 0;
        } else {
            rb = __builtin_inf();
            rsl = -1;
        }
    }
    __builtin_amdgcn_wave_barrier();  // the staging area is reused by the next pass
}

// Lexicographic minimum of (d2, cloud index) over the targets of the cell box [c0, c1] of every lane with `act`
// whose box has at most 3 x 3 rows: (rb, rsl), (+inf, -1) for an empty box.  Returns (per lane) whether the
// lane's box was taken; the others (`big`) are the caller's business.
template <int D, bool TIES = false>
ICPX_DEV bool coop_box(const double (&q)[D], const bool act, const int (&c0)[D], const int (&c1)[D],
                       const LdsView<D>& L, const GridView& P, const CoopStage<D>& S, double& rb, int& rsl, int* tie = nullptr) {
    const int lane = threadIdx.x & 63;
    const int G = P.g.G;
    constexpr int NR = D == 3 ? 9 : 3;
    ICPX_CYC_BEGIN();
    const int ny = c1[1] - c0[1] + 1;
    const int nz = D == 3 ? c1[D - 1] - c0[D - 1] + 1 : 1;
    const bool mine = act && ny <= 3 && nz <= 3;
    // rows the wave needs at all (uniform): boxes are 1 - 2 cells wide once the clouds are close
    const int nyw = __any(mine && ny > 2) ? 3 : (__any(mine && ny > 1) ? 2 : 1);
    const int nzw = D == 3 ? (__any(mine && nz > 2) ? 3 : (__any(mine && nz > 1) ? 2 : 1)) : 1;
    const int nrows = nyw | (nzw << 2);  // handed on to coop_pass in one scalar
    // ---- rows of the box: row r = (iy, iz) = (r % 3, r / 3), a fixed pattern.  All reads are issued; those of
    //      rows a lane does not have land on other words of the workgroup's LDS and are discarded ----
    const int row0 = ((D == 3 ? c0[D - 1] * G : 0) + c0[1]) * G + c0[0];
    const int width = c1[0] - c0[0] + 1;
    int rs[NR], re[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int iy = r % 3, iz = r / 3;  // compile-time
        const int at = row0 + (iy + iz * G) * G;  // (iy + iz * G) * G is uniform
        rs[r] = P.cstart[at];
        re[r] = P.cstart[at + width];
    }
    int rl[NR];
    int total = 0;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int iy = r % 3, iz = r / 3;
        rl[r] = (mine && iy < ny && iz < nz) ? re[r] - rs[r] : 0;
        total += rl[r];
    }
    const int wtot = wave_total_int(total);
    ICPX_CYC(17);  // row bounds
    if (threadIdx.x == 0) ICPX_STAT(16, 1);
    if (mine) {
        rb = __builtin_inf();
        rsl = -1;
    }
    if (wtot == 0) return mine;
    // lane groups: 1, 2, 4 ... 64 groups of consecutive lanes, so that a group's candidates fit the list
    int shift = 6;  // lanes per group = 1 << shift
    int gmax = wtot;
    while (gmax > kCoopCap && shift > 0) {
        shift--;
        int gsum = total;
        for (int o = 1; o < (1 << shift); o <<= 1) gsum += __shfl_xor(gsum, o);
        gmax = 0;
        for (int g = 0; g < 64; g += 1 << shift) gmax = max(gmax, __shfl(gsum, g));
        gmax = __builtin_amdgcn_readfirstlane(gmax);
    }
    if (gmax > kCoopCap) return false;  // one query with more candidates than the list holds: nobody was taken
    for (int g = 0; g < 64; g += 1 << shift) {
        const bool grp = mine && (lane >> shift) == (g >> shift);
        coop_pass<D, NR, TIES>(q, grp, rs, rl, nrows, L, P, S, rb, rsl, tie);
    }
    return mine;
}

// Large boxes (more than 3 x 3 rows, an overfull list, no bound yet = every cell): one such query at a time, ALL
// lanes of the wave scan for it.  A single lane walking its box alone is a chain of dependent LDS round trips — a
// query with no target in its 3^D cells walked all m targets that way (130 us for 1024 targets, the whole
// workgroup waiting: the tail of the 2D setup search) — 64 lanes take 64 targets per step.  They scan the RUN of
// slots from the box's first cell to its last (cells are in x-fastest order): a superset of the box's targets,
// which is as good (everything at most as far as the bound is inside the box, hence inside the run).
// (best, bslot) of the lanes in `walk`: running lexicographic minimum of (d2, cloud index), bslot < 0 = none yet.
template <int D, bool TIES = false>
ICPX_DEV void coop_run_scan(const double (&q)[D], const bool walk, const int (&w0)[D], const int (&w1)[D],
                            const LdsView<D>& L, const GridView& P, double& best, int& bslot, int* tie = nullptr) {
    const int lane = threadIdx.x & 63;
    const int G = P.g.G;
    int first = 0, last = 0;
#pragma unroll
    for (int d = D - 1; d >= 0; d--) {
        first = first * G + w0[d];
        last = last * G + w1[d];
    }
    unsigned long long todo = __ballot(walk);
    while (todo) {
        const int src = __ffsll(static_cast<long long>(todo)) - 1;
        todo &= todo - 1;
        double qq[D];
#pragma unroll
        for (int d = 0; d < D; d++) qq[d] = __shfl(q[d], src);
        const int kb = P.cstart[__shfl(first, src)], ke = P.cstart[__shfl(last, src) + 1];
        double mb = __builtin_inf();
        int ms = -1;
        int lt = 0;
        for (int j = kb + lane; j < ke; j += 64) {
            double df = L.t[0][j] - qq[0];
            double acc = df * df;
#pragma unroll
            for (int d = 1; d < D; d++) {
                df = L.t[d][j] - qq[d];
                acc = acc + df * df;
            }
            if (TIES) lt |= slot_tie<D>(true, acc, j, mb, ms, L) ? 1 : 0;
            bool better = acc < mb;
            if (acc == mb && ms >= 0) better = P.tperm[j] < P.tperm[ms];
            mb = better ? acc : mb;
            ms = better ? j : ms;
        }
        // lexicographic minimum over the lanes
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double ob = __shfl_xor(mb, o);
            const int os = __shfl_xor(ms, o);
            if (TIES) lt |= slot_tie<D>(os >= 0, ob, os, mb, ms, L) ? 1 : 0;
            bool take = os >= 0 && (ms < 0 || ob < mb);
            const bool tie = os >= 0 && ms >= 0 && ob == mb && os != ms;
            if (__any(tie)) {
                if (tie) take = P.tperm[os] < P.tperm[ms];
            }
            mb = take ? ob : mb;
            ms = take ? os : ms;
        }
        if (TIES) lt = __any(lt != 0) ? 1 : 0;
        if (lane == src && ms >= 0) {
            if (TIES) *tie |= lt | (slot_tie<D>(true, mb, ms, best, bslot, L) ? 1 : 0);
            bool take = bslot < 0 || mb < best;
            if (bslot >= 0 && mb == best && ms != bslot) take = P.tperm[ms] < P.tperm[bslot];
            best = take ? mb : best;
            bslot = take ? ms : bslot;
        }
    }
}

template <int D>
ICPX_DEV bool box_inside(const int (&a0)[D], const int (&a1)[D], const int (&b0)[D], const int (&b1)[D]) {
    bool in = true;
#pragma unroll
    for (int d = 0; d < D; d++) in = in && a0[d] >= b0[d] && a1[d] <= b1[d];
    return in;
}

// The search of one slot: same contract as nn_grid_slot.  Every lane walks through stages — 0: the 2^D cells
// nearest to the query (a first bound: any evaluated target gives a valid one), 1: the 3^D cells around it (when
// stage 0 found nothing or left a box too large for the list), 2: the box of its bound (complete: that box holds
// every target at most that far, the bound's own target included) — one cooperative pass per trip, lanes in
// different stages side by side.  A seeded lane whose bound gives a small box starts at stage 2: one trip.
// TIES: *tie |= a non-coincident target lies within kKdNearUlps of the answer (icpx_kd.h).  The pass that settles a
// query evaluates every target that close (its box is that of the bound), so it also sees the tie partners.
template <int D, bool TIES = false>
ICPX_DEV void nn_coop_slot(const double (&q)[D], const bool v, const bool seeded, const LdsView<D>& L,
                           const GridView& P, const CoopStage<D>& S, double& best, int& bslot, int* tie = nullptr) {
    bool fin = (P.g.bad & 1) == 0;
#pragma unroll
    for (int d = 0; d < D; d++) fin = fin && (__builtin_fabs(q[d]) < 1e150);
    if (__any(v && !fin)) {  // non-finite or huge coordinates in this wave: the per-lane path knows them
        nn_grid_slot<D, TIES>(q, v, seeded, L, P, best, bslot, tie);
        return;
    }
    const int G = P.g.G;
#ifdef ICPX_CYCONLY
    unsigned long long slot_t0_ = __builtin_amdgcn_s_memtime();
#endif
    int r0[D], r1[D];  // box of the current bound
    int stage = 0;
    if (seeded) {
        grid_range_cells<D>(q, best, P.g, r0, r1);
        bool small = r1[1] - r0[1] < 3;
        if (D == 3) small = small && r1[D - 1] - r0[D - 1] < 3;
        stage = small ? 2 : 0;
    } else {
        best = __builtin_inf();
        bslot = -1;
#pragma unroll
        for (int d = 0; d < D; d++) r0[d] = r1[d] = 0;
    }
    if (!v) stage = 3;
    for (int trip = 0; trip < 4 && __any(stage < 3); trip++) {
        int c0[D], c1[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            c0[d] = r0[d];
            c1[d] = r1[d];
        }
        if (__any(stage < 2)) {
            // local blocks: the query's cell and which half of it the query is in
#pragma unroll
            for (int d = 0; d < D; d++) {
                double x = (q[d] - P.g.lo[d]) * P.g.scale[d];
                x = __builtin_fmin(__builtin_fmax(x, 0.0), static_cast<double>(G - 1));  // NaN -> 0
                const int cq = static_cast<int>(x);
                const bool upper = x - static_cast<double>(cq) > 0.5;
                const int lo2 = upper ? cq : (cq > 0 ? cq - 1 : 0);
                const int hi2 = upper ? (cq < G - 1 ? cq + 1 : G - 1) : cq;
                const int lo3 = cq > 0 ? cq - 1 : 0, hi3 = cq < G - 1 ? cq + 1 : G - 1;
                c0[d] = stage == 0 ? lo2 : (stage == 1 ? lo3 : r0[d]);
                c1[d] = stage == 0 ? hi2 : (stage == 1 ? hi3 : r1[d]);
            }
        }
        const bool act = stage < 3;
        double rb = __builtin_inf();
        int rsl = -1;
        const bool taken = coop_box<D, TIES>(q, act, c0, c1, L, P, S, rb, rsl, tie);
        // boxes the list did not take (more than 3 x 3 rows, or an overfull list): the per-lane walk, final
        // stage only (the local blocks always fit), from the lane's own candidate
        const bool walk = act && !taken;
        if (__any(walk)) {
            if (threadIdx.x == 0) ICPX_STAT(23, 1);
            ICPX_CYC_BEGIN();
            int w0[D], w1[D];
#pragma unroll
            for (int d = 0; d < D; d++) {
                w0[d] = r0[d];
                w1[d] = r1[d];
            }
            if (stage < 2) grid_range<D>(q, best, P.g, w0, w1);  // (a local block that overflowed the list)
            coop_run_scan<D, TIES>(q, walk, w0, w1, L, P, best, bslot, tie);
            ICPX_CYC(24);  // the large boxes, one query at a time, the wave scanning together
        }
        if (act && taken) {
            if (stage == 2) {
                best = rb;
                bslot = rsl;
                stage = 3;
            } else {
                // a local block: its minimum joins the bound; ties go to the block (it is the lexicographic minimum there)
                if (TIES) *tie |= slot_tie<D>(rsl >= 0, rb, rsl, best, bslot, L) ? 1 : 0;
                const bool take = rsl >= 0 && !(rb > best);
                best = take ? rb : best;
                bslot = take ? rsl : bslot;
                grid_range_cells<D>(q, best, P.g, r0, r1);  // best = +inf (nothing yet): every cell
                bool small = r1[1] - r0[1] < 3;
                if (D == 3) small = small && r1[D - 1] - r0[D - 1] < 3;
                if (take && box_inside<D>(r0, r1, c0, c1)) stage = 3;      // everything that close was in the block
                else if (small) stage = 2;
                else stage = stage == 0 ? 1 : 2;                           // try the larger block once, then walk
            }
        } else if (walk) {
            stage = 3;
        }
#ifdef ICPX_CYCONLY
        if (threadIdx.x == 0) atomicAdd(&g_icpx_stats[5], 1ull);  // trips
#endif
    }
#ifdef ICPX_CYCONLY
    if (threadIdx.x == 0) {
        atomicAdd(&g_icpx_stats[3], __builtin_amdgcn_s_memtime() - slot_t0_);
        atomicAdd(&g_icpx_stats[4], 1ull);
    }
#endif
}

}  // namespace icpx
