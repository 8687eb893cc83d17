// icpx_kd.h — the reference kd-tree's tie rule on top of the exact searches (device side).
//
// The reference's kd-tree methods (Vanilla 2D: vanilla.cpp:68-88; Vanilla3d / Trimmed3d: vanilla_3d.cpp:30-45,
// trimmed_3d.cpp:36-51) take the neighbour from KdTree::search (include/algo/kdtree.h:79-89,132-154), which compares
// sqrt(d2) with strict < and keeps the FIRST node it visits, children near side first.  Pruning never hides that node
// (DESIGN.md section 3), so its answer is
//     among the targets whose sqrt(d2) equals the minimal sqrt(d2), bit for bit, the one that comes first in the
//     near-child-first pre-order of the tree for this query.
// Every search of this library finds the (d2, index) minimum; almost always that target is alone at its sqrt(d2) and
// both rules name it.  The searches therefore only DETECT the other case — another target within kKdNearUlps of the
// best d2 (kd_near; at most four consecutive doubles share one square root) — and the rare flagged queries are
// resolved here: tied set by sqrt(d2), pre-order key of every member (kd_key), smallest key wins, then the lowest
// index among the targets coincident with the winner (which of those the reference returns is its std::nth_element's
// choice).
//
// The tree is the median-split tree of kdtree.h:96-116 — node of the range [lo, hi) = element (n - 1) / 2 of the
// subset ordered by the split coordinate, axis = depth % dim — stored IN ORDER: node[p] is the target at in-order
// position p, pos[t] its inverse; the node of [lo, hi) sits at lo + (hi - lo - 1) / 2.  Equal coordinates are ordered
// by index (the reference leaves them to nth_element: implementation-defined there).
// oracle/icp_oracle.cpp (KdCanon) states the same rule on the CPU, in two formulations.
#pragma once

#include "icpx_math.h"

namespace icpx {

constexpr unsigned long long kKdNearUlps = 4;

// |bits(a) - bits(b)| <= kKdNearUlps for squared distances (non-negative: bit patterns order like the values)
ICPX_DEV bool kd_near(double a, double b) {
    const long long x = __double_as_longlong(a), y = __double_as_longlong(b);
    const unsigned long long d = x > y ? static_cast<unsigned long long>(x - y) : static_cast<unsigned long long>(y - x);
    return d <= kKdNearUlps;
}

// order-preserving f64 -> u64 (a total order; the tree's coordinate order)
ICPX_DEV unsigned long long kd_ord_key(double v) {
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}

template <typename I>
struct KdView {
    const I* node;      // [m] target at in-order position p
    const I* pos;       // [m] in-order position of target t
    const double* pts;  // the target cloud (AoS, cloud order, global memory)
    int m;
};

// Pre-order key of target t for query q: two bits per level from the root — 0 this node, 1 the child on the query's
// side (kdtree.h:146: left iff q[axis] < node[axis]), 2 the other child.  Smaller key = visited earlier.
// The path from the root to t's node depends on t's position alone (which side of every median it lies on), so the
// medians of up to eight levels are known before anything is loaded: their nodes and split coordinates are fetched
// as two rounds of independent loads instead of two dependent loads per level (the resolve kernel of a cloud with many
// ties was a chain of those: 28 round trips per key).
template <int D, typename I>
ICPX_DEV unsigned long long kd_key(const KdView<I>& kd, const double (&q)[D], const int t) {
    const int P = static_cast<int>(kd.pos[t]);
    int lo = 0, hi = kd.m, axis = 0;
    unsigned long long k = 0;
    bool fin = false;
    for (int depth = 0; depth < 32 && !fin; depth += 8) {
        int mids[8], axes[8];
        bool on[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int mid = lo + ((hi - lo - 1) >> 1);
            fin = fin || P == mid || depth + u >= 31;
            on[u] = !fin;
            mids[u] = mid;
            axes[u] = axis;
            if (on[u]) {
                if (P < mid) hi = mid;
                else lo = mid + 1;
                axis = axis + 1 == D ? 0 : axis + 1;
            }
        }
        int64_t nd[8];
#pragma unroll
        for (int u = 0; u < 8; u++) nd[u] = on[u] ? static_cast<int64_t>(kd.node[mids[u]]) : 0;
        double c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = on[u] ? kd.pts[nd[u] * D + axes[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            double qa = q[0];
#pragma unroll
            for (int d = 1; d < D; d++) qa = axes[u] == d ? q[d] : qa;
            const bool near_left = qa < c[u], go_left = P < mids[u];
            if (on[u]) k |= static_cast<unsigned long long>(go_left == near_left ? 1 : 2) << (60 - 2 * (depth + u));
        }
    }
    return k;
}

// Is the candidate (dist, j; coordinates c[0..D)) a tie partner of the current best (best, bidx)?  Another target
// within kKdNearUlps of it — unless the two coincide (cloud: the targets in cloud order, to look the best's coordinates
// up; a rare branch; null: not checked).  bidx == 0x7fffffff: no best yet.  No false negatives over a whole search:
// whichever of two such targets is evaluated second sees the first (or something closer still) as the best.
template <int D>
ICPX_DEV bool kd_tie_check(const bool valid, const double dist, const int j, const double* c,
                           const double* __restrict__ cloud, const double best, const int bidx) {
    // (a cheap necessary condition first — 4 ulps are well inside 2^-40 relative; NaN and inf fail it — so that the
    // common far candidate costs three instructions here)
    bool near = valid && dabs(dist - best) <= best * 9.0949470177292824e-13;
    near = near && bidx != 0x7fffffff && j != bidx && dist < __builtin_inf() && kd_near(dist, best);
    if (near && dist == best && cloud != nullptr) {
        bool same = true;
#pragma unroll
        for (int d = 0; d < D; d++) same = same && cloud[static_cast<int64_t>(bidx) * D + d] == c[d];
        near = !same;
    }
    return near;
}

// (best, bidx) <- (d2, index) minimum with the candidate; KD: tie |= kd_tie_check.
// (tie is an int on purpose: carried as a bool through the irregular loop of the BVH walk it was miscompiled — a lane
// mask merged across divergent control flow — and flags went missing; as a vector register it is plain data)
template <bool KD, int D>
ICPX_DEV void nn_take(const bool valid, const double dist, const int j, const double* c, const double* __restrict__ cloud,
                      double& best, int& bidx, int& tie) {
    const bool take = valid && (dist < best || (dist == best && j < bidx && bidx != 0x7fffffff));
    if (KD) tie |= kd_tie_check<D>(valid, dist, j, c, cloud, best, bidx) ? 1 : 0;
    best = take ? dist : best;
    bidx = take ? j : bidx;
}

// Resolution of flagged queries, one workgroup per query, scanning the whole cloud (multi-kernel path and the
// stand-alone query: flagged queries are rare, and this needs nothing from the engine that found the minimum).
// list[0..count): query indices.  MODE as lg_nn_kernel (0: queries = cur, 1: src, 2 / 3: transform * src); 4: xf * cur.
template <int D>
ICPX_DEV void kd_query_of(const int mode, const double* __restrict__ src, const double* __restrict__ cur, const double* xf,
                          const int64_t i, double (&q)[D]) {
    if (mode == 0) {
#pragma unroll
        for (int d = 0; d < D; d++) q[d] = cur[i * D + d];
    } else if (mode == 1) {
#pragma unroll
        for (int d = 0; d < D; d++) q[d] = src[i * D + d];
    } else {
        // 2 / 3: xf * src; 4: xf * cur (the tiled 3D search: xf = the step, the cloud not yet stepped)
        Xf<D> X;
#pragma unroll
        for (int e = 0; e < (D + 1) * (D + 1); e++) X.m[e] = xf[e];
        const XfRows<D> xr = xf_rows<D>(X);
        double p[D];
#pragma unroll
        for (int d = 0; d < D; d++) p[d] = mode == 4 ? cur[i * D + d] : src[i * D + d];
        xf_apply<D>(xr, p, q);
    }
}

constexpr int kKdResolveThreads = 256;
constexpr int kKdResolveBatch = 8;  // flagged queries a workgroup takes through the cloud together

// count: device word (number of flagged queries); nb_idx: in = the (d2, index) minimum's target, out = the rule's;
// nb_pt (may be null): out = that target's coordinates (at pt_perm[i] when given), pt_sorted (may be null): a second
// copy at i.  A workgroup takes kKdResolveBatch flagged queries through the cloud together.  (queen.ply flags 2 989 of
// 10 720 queries at its first step, nearly all of them for coincident duplicates of the winner that the in-tile
// detection cannot tell from a tie: 84 us.  Measured and dropped: eight targets' loads in flight per trip (112 us: the
// unrolled body with a copy of kd_key behind every test outgrows the instruction cache) and kd_key as a real call (154).)
template <int D>
__global__ void __launch_bounds__(kKdResolveThreads)
kd_resolve_kernel(const int* __restrict__ count, const uint32_t* __restrict__ list, const int mode,
                  const double* __restrict__ src, const double* __restrict__ cur, const double* __restrict__ xf,
                  const KdView<uint32_t> kd, int64_t* __restrict__ nb_idx, double* __restrict__ nb_pt,
                  const uint32_t* __restrict__ pt_perm, double* __restrict__ pt_sorted) {
    constexpr int W = kKdResolveThreads / 64, B = kKdResolveBatch;
    __shared__ unsigned long long s_key[B][W];
    __shared__ int s_idx[B][W];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = *count;
    for (int e0 = blockIdx.x * B; e0 < total; e0 += gridDim.x * B) {
        const int nq = total - e0 < B ? total - e0 : B;  // (uniform)
        double q[B][D], best[B], s[B];
        int64_t qi[B];
#pragma unroll
        for (int b = 0; b < B; b++) {
            qi[b] = list[e0 + (b < nq ? b : 0)];
            kd_query_of<D>(mode, src, cur, xf, qi[b], q[b]);
            // the minimal d2 (the engines' answer, recomputed: nb_idx names a target that attains it)
            const int64_t b0 = nb_idx[qi[b]];
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < D; d++) {
                const double df = kd.pts[b0 * D + d] - q[b][d];
                acc = d == 0 ? df * df : acc + df * df;
            }
            // (nothing finite, or a slot beyond the batch's end: NaN compares false everywhere below)
            best[b] = (b < nq && acc < __builtin_inf()) ? acc : __builtin_nan("");
            s[b] = __builtin_sqrt(best[b]);
        }
        unsigned long long mk[B];
        int mi[B];
#pragma unroll
        for (int b = 0; b < B; b++) {
            mk[b] = ~0ull;
            mi[b] = 0x7fffffff;
        }
        for (int j = tid; j < kd.m; j += kKdResolveThreads) {
            double t[D];
#pragma unroll
            for (int d = 0; d < D; d++) t[d] = kd.pts[static_cast<int64_t>(j) * D + d];
#pragma unroll
            for (int b = 0; b < B; b++) {
                double df = t[0] - q[b][0];
                double acc = df * df;
#pragma unroll
                for (int d = 1; d < D; d++) {
                    df = t[d] - q[b][d];
                    acc = acc + df * df;
                }
                if (dabs(acc - best[b]) <= best[b] * 9.0949470177292824e-13 && kd_near(acc, best[b]) && __builtin_sqrt(acc) == s[b]) {
                    const unsigned long long k = kd_key<D, uint32_t>(kd, q[b], j);
                    if (k < mk[b]) {
                        mk[b] = k;
                        mi[b] = j;
                    }
                }
            }
        }
        __syncthreads();  // (the previous batch's reads of the shared words are done)
#pragma unroll
        for (int b = 0; b < B; b++) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const unsigned long long ok = __shfl_xor(mk[b], o);
                const int oi = __shfl_xor(mi[b], o);
                const bool take = ok < mk[b];
                mk[b] = take ? ok : mk[b];
                mi[b] = take ? oi : mi[b];
            }
            if (lane == 0) {
                s_key[b][wave] = mk[b];
                s_idx[b][wave] = mi[b];
            }
        }
        __syncthreads();
        double w[B][D];
#pragma unroll
        for (int b = 0; b < B; b++) {
            for (int ww = 0; ww < W; ww++) {
                const bool take = s_key[b][ww] < mk[b];
                mk[b] = take ? s_key[b][ww] : mk[b];
                mi[b] = take ? s_idx[b][ww] : mi[b];
            }
            // (mi == 0x7fffffff: no finite minimum, or beyond the batch: nothing to do for this slot)
            const int64_t wj = mi[b] == 0x7fffffff ? 0 : mi[b];
#pragma unroll
            for (int d = 0; d < D; d++) w[b][d] = kd.pts[wj * D + d];
        }
        // the lowest index among the targets coincident with each winner
        int lowest[B];
#pragma unroll
        for (int b = 0; b < B; b++) lowest[b] = mi[b];
        for (int j = tid; j < kd.m; j += kKdResolveThreads) {
            double t[D];
#pragma unroll
            for (int d = 0; d < D; d++) t[d] = kd.pts[static_cast<int64_t>(j) * D + d];
#pragma unroll
            for (int b = 0; b < B; b++) {
                bool same = true;
#pragma unroll
                for (int d = 0; d < D; d++) same = same && t[d] == w[b][d];
                lowest[b] = (same && j < lowest[b]) ? j : lowest[b];
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < B; b++) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) lowest[b] = min(lowest[b], __shfl_xor(lowest[b], o));
            if (lane == 0) s_idx[b][wave] = lowest[b];
        }
        __syncthreads();
        if (tid < nq) {
            // (one thread per query of the batch; runtime index into the register arrays through a small switch)
            int low = 0x7fffffff;
            int64_t i = 0;
#pragma unroll
            for (int b = 0; b < B; b++)
                if (b == tid) {
                    for (int ww = 0; ww < W; ww++) low = min(low, s_idx[b][ww]);
                    i = qi[b];
                }
            if (low != 0x7fffffff) {
                nb_idx[i] = low;
                if (nb_pt) {  // (the multi-kernel path keeps the matched target's coordinates beside its index)
                    const int64_t at = pt_perm ? static_cast<int64_t>(pt_perm[i]) : i;  // (sorted domain: i is a position of the query order)
#pragma unroll
                    for (int d = 0; d < D; d++) {
                        const double v = kd.pts[static_cast<int64_t>(low) * D + d];
                        nb_pt[at * D + d] = v;
                        if (pt_sorted) pt_sorted[i * D + d] = v;  // (sorted domain: the copy in query order)
                    }
                }
            }
        }
    }
}

}  // namespace icpx
