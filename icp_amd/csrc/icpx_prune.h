// icpx_prune.h — spatially pruned nearest-neighbour search for the fused pair kernel.
//
// The brute-force search of icpx_filter.h looks at every (query, target) pair.  Here both clouds
// of a pair are put into Morton order once (pair_perm_kernel, a pre-pass): the LDS target tile is
// staged in that order with an f32 bounding box per 16 targets, and the queries are handed to the
// lanes in that order too, so the 64 queries of a wave slot lie close together.  A wave slot then
//   1. reduces its queries' bounding box and its largest search radius (from the seeds: the
//      previous neighbours),
//   2. tests every target block's box against them — one block per lane, 64 blocks per step —
//   3. and runs the f32 candidate filter + exact f64 refinement of icpx_filter.h over the
//      surviving blocks only.
// The result is still the lexicographic minimum of (d2, target index in the CLOUD) over ALL
// targets, bit for bit: a pruned block cannot hold a target with d2 <= the query's bound (see
// prune_test), exact evaluations use the reference's expression, and exact ties are decided on the
// cloud index through the permutation table.  ANY permutation gives the same answer; the Morton
// order only decides how much is pruned.
#pragma once

#include "icpx_filter.h"

namespace icpx {

constexpr int kPruneBlk = 16;  // targets per bounding box

struct PruneView {
    float* lo[3];            // LDS: box minima per block, f32 filter frame
    float* hi[3];
    int nb;                  // blocks
    const uint16_t* tperm;   // global: cloud index of LDS slot j
    int m;
};

// (B + E)^2 rounded up: the reach of a query whose exact bound is d2max, in the f32 frame.  Every
// target with exact d2 <= d2max lies within sqrt(reach) of the query's f32 image (icpx_filter.h).
ICPX_DEV float filter_reach(double d2max, float e_dist) {
    float d2f = static_cast<float>(d2max) * 1.0000005f + 1e-30f;
    float s = __builtin_sqrtf(d2f) * 1.000002f + e_dist;
    return s * s * 1.000004f;
}

// exact (d2, cloud index) update of one query against the four targets of block `blk` (slots);
// lanes with blk < 0 idle.  Ties on d2 are rare: the permutation table is read only then.
template <int D>
ICPX_DEV void exact_block_p(const double (&q)[D], const LdsView<D>& L, int blk, const uint16_t* __restrict__ tperm,
                            double& best, int& bslot) {
    const bool act = blk >= 0;
    const int jb = act ? blk : 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        double df = L.t[0][jb + u] - q[0];
        double acc = df * df;
#pragma unroll
        for (int d = 1; d < D; d++) {
            df = L.t[d][jb + u] - q[d];
            acc = acc + df * df;
        }
        const int slot = jb + u;
        bool better = act && acc < best;
        const bool tie = act && acc == best && acc < __builtin_inf() && slot != bslot;
        if (__any(tie)) {
            if (tie) better = tperm[slot] < tperm[bslot];
        }
        best = better ? acc : best;
        bslot = better ? slot : bslot;
    }
}

// Exact scan of all targets (waves whose coordinates the filter cannot take): same outcome as
// nn_lds on the unpermuted cloud, including its conventions for non-finite distances (nothing
// below +inf -> cloud index 0).
template <int D>
ICPX_DEV void exact_scan_p(const double (&q)[D], const LdsView<D>& L, const PruneView& P, double& best, int& bslot) {
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

template <int D, int QPT>
ICPX_DEV void nn_fallback_p(const double (&q)[QPT][D], const LdsView<D>& L, const PruneView& P, bool seeded,
                            double (&best)[QPT], int (&bslot)[QPT]) {
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        double lb;
        int ls;
        exact_scan_p<D>(q[k], L, P, lb, ls);
        bool take = true;
        if (seeded) take = lb < best[k] || (lb == best[k] && P.tperm[ls] < P.tperm[bslot[k]]);
        best[k] = take ? lb : best[k];
        bslot[k] = take ? ls : bslot[k];
    }
}

ICPX_DEV float wave_min_f32(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = __builtin_fminf(v, __shfl_xor(v, o));
    return v;
}
ICPX_DEV float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Seeded, pruned search of ONE query per lane (the 64 queries of a wave: consecutive in the sorted
// order) against the sorted LDS tile.  In: (best, bslot) = exact d2 and slot of some target.  Out:
// the lexicographic minimum of (d2, cloud index) over all targets; bslot is its LDS slot.
//
// prune test of a block: with lo/hi the box of the wave's f32 queries and reach = max over its
// lanes of filter_reach, the squared distance between the two boxes, computed in f32 (<= 4
// roundings, each relatively 2^-24, hence the factor), is a lower bound of |t32 - q32|^2 for every
// target of the block and every query of the wave; above `reach` no target of the block is within
// any query's bound.
template <int D>
ICPX_DEV void nn_pruned_slot(const double (&q)[D], const bool v, const LdsView<D>& L, const PruneView& P,
                             const FilterFrame& F, double& best, int& bslot) {
    const double (&q1)[1][D] = reinterpret_cast<const double (&)[1][D]>(q);
    float q2[1][D], qn[1], e_dist, e_val;
    const bool ok = filter_queries<D, 1>(q1, F, q2, qn, e_dist, e_val);
    if (__any(!ok && v)) {
        double lb;
        int ls;
        exact_scan_p<D>(q, L, P, lb, ls);
        const bool take = lb < best || (lb == best && P.tperm[ls] < P.tperm[bslot]);
        best = take ? lb : best;
        bslot = take ? ls : bslot;
        return;
    }
    const int lane = threadIdx.x & 63;
#ifdef ICPX_STATS
    unsigned st_blocks = 0, st_flush = 0;  // one atomic each at the end: contended atomics would distort the timers
#endif
    float thr = v ? filter_threshold(best, qn[0], e_dist, e_val) : -__builtin_inff();
    int pend = -1;
    const float reach = wave_max_f32(v ? filter_reach(best, e_dist) : -__builtin_inff());
    float qlo[D], qhi[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const float c = -0.5f * q2[0][d];
        qlo[d] = wave_min_f32(v ? c : __builtin_inff());
        qhi[d] = wave_max_f32(v ? c : -__builtin_inff());
    }
    for (int g0 = 0; g0 < P.nb; g0 += 64) {
        const int b = g0 + lane;
        bool pass = false;
        if (b < P.nb) {
            float lb = 0.0f;
#pragma unroll
            for (int d = 0; d < D; d++) {
                float gap = __builtin_fmaxf(__builtin_fmaxf(P.lo[d][b] - qhi[d], qlo[d] - P.hi[d][b]), 0.0f);
                lb = __builtin_fmaf(gap, gap, lb);
            }
            pass = lb * 0.999999f <= reach;
        }
        unsigned long long mask = __ballot(pass);
#ifdef ICPX_STATS
        st_blocks += __popcll(mask);
#endif
        while (mask) {
            const int jb0 = (g0 + __ffsll(static_cast<long long>(mask)) - 1) * kPruneBlk;
            mask &= mask - 1;
            float mn[kPruneBlk / 4];
#pragma unroll
            for (int u = 0; u < kPruneBlk / 4; u++) {
                f32x4 tc[D], tn;
                load_filter_block<D>(L, jb0 + 4 * u, tc, tn);
                mn[u] = block_filter_min<D>(tc, tn, q2[0]);
            }
#pragma unroll
            for (int u = 0; u < kPruneBlk / 4; u++) {
                const bool cand = !(mn[u] > thr);
                if (__any(cand)) {
                    // one pending block per lane; a lane's second candidate flushes wave-wide
                    if (__any(cand && pend >= 0)) {
#ifdef ICPX_STATS
                        st_flush++;
#endif
                        const double before = best;
                        exact_block_p<D>(q, L, pend, P.tperm, best, bslot);
                        pend = -1;
                        if (best < before) thr = filter_threshold(best, qn[0], e_dist, e_val);
                    }
                    pend = cand ? jb0 + 4 * u : pend;
                }
            }
        }
    }
    if (__any(pend >= 0)) exact_block_p<D>(q, L, pend, P.tperm, best, bslot);
#ifdef ICPX_STATS
    if ((clock64() & 63) == 0) {  // sampled 1 in 64
        ICPX_STAT(0, 64);
        ICPX_STAT(1, 64 * st_flush);
        ICPX_STAT(12, 64 * st_blocks);
    }
#endif
}

// wave-cooperative rescan of single queries over the permuted tile (see coop_rescan)
template <int D>
ICPX_DEV void coop_rescan_p(bool flag, const double (&q)[D], const float (&q2)[D], float thr, const LdsView<D>& L,
                            int m4, const uint16_t* __restrict__ tperm, double& best, int& bslot) {
    const int lane = threadIdx.x & 63;
    unsigned long long mask = __ballot(flag);
    while (mask) {
        const int src = __ffsll(static_cast<long long>(mask)) - 1;
        mask &= mask - 1;
        double qb[D];
        float q2b[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            qb[d] = __shfl(q[d], src);
            q2b[d] = __shfl(q2[d], src);
        }
        const float thr_b = __shfl(thr, src);
        double bb = __builtin_inf();
        int bs = 0x7fffffff;
        for (int jb = lane * 4; jb < m4; jb += 256) {
            f32x4 tc[D], tn;
            load_filter_block<D>(L, jb, tc, tn);
            if (!(block_filter_min<D>(tc, tn, q2b) > thr_b)) exact_block_p<D>(qb, L, jb, tperm, bb, bs);
        }
        int bo = bb < __builtin_inf() ? static_cast<int>(tperm[bs]) : 0x7fffffff;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            double ob = __shfl_xor(bb, o);
            int oo = __shfl_xor(bo, o);
            int os = __shfl_xor(bs, o);
            bool take = ob < bb || (ob == bb && oo < bo);
            bb = take ? ob : bb;
            bo = take ? oo : bo;
            bs = take ? os : bs;
        }
        if (lane == src && bb < __builtin_inf()) {
            best = bb;
            bslot = bs;
        }
    }
}

// seedless search over the permuted tile: nn_twomin with cloud-index tie-breaks
template <int D, int QPT>
ICPX_DEV void nn_twomin_p(const double (&q)[QPT][D], const LdsView<D>& L, int m4, const PruneView& P,
                          const FilterFrame& F, double (&best)[QPT], int (&bslot)[QPT]) {
    float q2[QPT][D], qn[QPT], e_dist, e_val;
    const bool ok = filter_queries<D, QPT>(q, F, q2, qn, e_dist, e_val);
    if (__any(!ok)) {
        nn_fallback_p<D, QPT>(q, L, P, false, best, bslot);
        return;
    }
    float b1[QPT], b2[QPT];
    int blk[QPT];
    ICPX_STAT(2, 1);
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        b1[k] = __builtin_inff();
        b2[k] = __builtin_inff();
        blk[k] = 0;
    }
    for (int jb = 0; jb < m4; jb += 4) {
        f32x4 tc[D], tn;
        load_filter_block<D>(L, jb, tc, tn);
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            float mn = block_filter_min<D>(tc, tn, q2[k]);
            bool lt = mn < b1[k];
            b2[k] = raw_min_f32(b2[k], raw_max_f32(b1[k], mn));
            b1[k] = raw_min_f32(b1[k], mn);
            blk[k] = lt ? jb : blk[k];
        }
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        best[k] = __builtin_inf();
        bslot[k] = 0x7fffffff;
        exact_block_p<D>(q[k], L, blk[k], P.tperm, best[k], bslot[k]);
        const float thr = filter_threshold(best[k], qn[k], e_dist, e_val);
        const bool again = !(b2[k] > thr);
        ICPX_STAT(3, __popcll(__ballot(again)));
        if (__any(again)) {
            ICPX_STAT(4, 1);
            coop_rescan_p<D>(again, q[k], q2[k], thr, L, m4, P.tperm, best[k], bslot[k]);
        }
    }
}

// ---- pre-pass: Morton order of one cloud ------------------------------------------------------
ICPX_DEV uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
    v &= 0x3ff;
    v = (v | (v << 16)) & 0x030000ff;
    v = (v | (v << 8)) & 0x0300f00f;
    v = (v | (v << 4)) & 0x030c30c3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}
ICPX_DEV uint32_t spread2(uint32_t v) {  // 16 bits -> every second bit
    v &= 0xffff;
    v = (v | (v << 8)) & 0x00ff00ff;
    v = (v | (v << 4)) & 0x0f0f0f0f;
    v = (v | (v << 2)) & 0x33333333;
    v = (v | (v << 1)) & 0x55555555;
    return v;
}

}  // namespace icpx
