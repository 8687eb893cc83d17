// icpx_filter.h — nearest-neighbour search of register-resident queries against a target tile in
// LDS: the exact f64 search (nn_lds) and the f32 candidate filter with exact f64 refinement
// (nn_filtered, nn_twomin).  Used by the fused pair kernel (icpx_pair.h) and by the tiled
// correspondence kernel of the multi-kernel path (icpx_large.hip).
//
// Target indices: LDS slot j stands for target j0 + j of the cloud (j0 = 0 in the fused kernel, the
// tile origin in the tiled kernel); all (d2, index) comparisons use the cloud index.
#pragma once

#include "icpx_kd.h"
#include "icpx_math.h"

namespace icpx {

template <int D>
struct LdsView {
    double* t[3];    // target SoA
    double* gs;      // kGsValues x kGsStride
    double* bc;      // broadcast
    uint64_t* skey;  // sort keys
    uint32_t* sval;  // sort values
    float* tf[3];    // f32 copy of the target SoA (candidate filter), null when not reserved
    int j0;          // cloud index of LDS slot 0
    const double* cloud = nullptr;  // KD searches: the whole target cloud (AoS, global), to tell coincident targets from ties
    // Interleaved tiles (tiled search of the multi-kernel path): the tile holds every cstride/32-th chunk of 32 consecutive
    // targets, so slot j stands for target j0 + (j / 32) * cstride + j % 32 (ascending in j).  0: the tile is contiguous.
    int cstride = 0;
};
// cloud index of LDS slot j
template <int D>
ICPX_DEV int lds_cloud_index(const LdsView<D>& L, const int j) {
    return L.cstride ? L.j0 + (j >> 5) * L.cstride + (j & 31) : L.j0 + j;
}

// brute-force NN of QPT register-resident queries against the LDS target cloud.
// d2 = ((dx*dx) + dy*dy) + dz*dz with dx = target - query (trimmed.cpp:101), strict <,
// ascending j  =>  lexicographic min of (d2, j).
//
// Targets are taken four at a time: the block minimum (v_min_f64 tree) is compared with the
// running best by strict <, so `bblk` ends as the FIRST block that attains the global minimum —
// exactly the block the one-by-one loop of the reference would have stopped improving in.  The
// index inside that block is resolved afterwards by recomputing its four distances (same
// operations, same bits) and taking the lowest j that equals the minimum.  This costs 9.5 (3D) /
// 6.5 (2D) VALU instructions per pair instead of 12 / 9 with a per-target compare-and-select.
// `m4` is m rounded up to a multiple of 4; LDS slots [m, m4) hold +inf (never selected).
// KD (icpx_kd.h): tie[k] reports whether another target lies within kKdNearUlps of the minimum — the second smallest
// block minimum is carried along (two more instructions per block), the winning block is counted at the end.
template <int D, int QPT, bool KD = false>
ICPX_DEV void nn_lds(const double (&q)[QPT][D], const LdsView<D>& L, int m4, double (&best)[QPT],
                     int (&bidx)[QPT], int* tie = nullptr) {
    int bblk[QPT];
    double b2[KD ? QPT : 1];
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        best[k] = __builtin_inf();
        bblk[k] = 0;
        if (KD) b2[k] = __builtin_inf();
    }
    for (int jb = 0; jb < m4; jb += 4) {
        double tc[D][4];
#pragma unroll
        for (int d = 0; d < D; d++)
#pragma unroll
            for (int u = 0; u < 4; u++) tc[d][u] = L.t[d][jb + u];
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            double dist[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                double df = tc[0][u] - q[k][0];
                double acc = df * df;
#pragma unroll
                for (int d = 1; d < D; d++) {
                    df = tc[d][u] - q[k][d];
                    acc = acc + df * df;
                }
                dist[u] = acc;
            }
            double mm = __builtin_fmin(__builtin_fmin(dist[0], dist[1]), __builtin_fmin(dist[2], dist[3]));
            bool lt = mm < best[k];
            if (KD) b2[k] = __builtin_fmin(b2[k], __builtin_fmax(best[k], mm));  // second smallest block minimum
            best[k] = __builtin_fmin(best[k], mm);
            bblk[k] = lt ? jb : bblk[k];
        }
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        int idx = 0, near = 0;
        if (best[k] < __builtin_inf()) {
#pragma unroll
            for (int u = 3; u >= 0; u--) {
                double df = L.t[0][bblk[k] + u] - q[k][0];
                double acc = df * df;
#pragma unroll
                for (int d = 1; d < D; d++) {
                    df = L.t[d][bblk[k] + u] - q[k][d];
                    acc = acc + df * df;
                }
                idx = acc == best[k] ? bblk[k] + u : idx;
                if (KD) near += kd_near(acc, best[k]) ? 1 : 0;
            }
        }
        bidx[k] = idx;
        if (KD) tie[k] = (best[k] < __builtin_inf() && (near > 1 || kd_near(b2[k], best[k]))) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Chunk-skipping exact search (trimmed ICP2 in the brute-force kernels).  The targets stay in cloud order; every
// kChunk consecutive ones carry their bounding box.  One query slot at a time, the wave walks the chunks in ascending
// order and evaluates a chunk — with nn_lds's block code, for all lanes — only if SOME lane's query can have a target
// there that is not farther than its bound: the smaller of its running minimum and the exact distance to its seed
// (the previous neighbour).  Exactness: for a target t of the chunk and every axis |t - q| >= g = max(lo - q,
// q - hi, 0) in real numbers; rounded differences, their rounded squares and the rounded sums of non-negative terms
// are monotone, so the computed d2 = ((dx dx) + dy dy) + dz dz is >= lb = ((gx gx) + gy gy) + gz gz computed the
// same way.  A chunk is skipped only if lb > bound, strictly: every target with d2 <= bound — the seed itself, equal
// distances with lower indices — is evaluated, in ascending order with strict <, which is nn_lds's rule; the seed only
// bounds, it is never taken as the answer without having been scanned.  NaN bounds or coordinates never skip.
// What it buys depends on coherence: the points of a lidar scan come in scan order, so the 64 queries of a wave-slot
// are neighbours and so are the 32 targets of a chunk — ex_data/scan1: 4 of 30 chunks evaluated per wave-slot.  Clouds
// in random order evaluate every chunk (the cost of nn_lds plus the tests).
// ---------------------------------------------------------------------------------------------
constexpr int kChunk = 32;

// boxes of the staged cloud: cb[(c * 2 + 0) * D + d] = min, cb[(c * 2 + 1) * D + d] = max over targets [c kChunk,
// (c + 1) kChunk) below m (NaN coordinates are ignored; an empty or all-NaN chunk gets +inf / -inf and is never needed).
// Workgroup-wide; ends with a barrier.
template <int D>
ICPX_DEV void chunk_boxes(const LdsView<D>& L, const int m, double* __restrict__ cb) {
    const int nch = (m + kChunk - 1) / kChunk;
    for (int c = threadIdx.x; c < nch; c += blockDim.x) {
        double lo[D], hi[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            lo[d] = __builtin_inf();
            hi[d] = -__builtin_inf();
        }
        const int j1 = min((c + 1) * kChunk, m);
        for (int j = c * kChunk; j < j1; j++)
#pragma unroll
            for (int d = 0; d < D; d++) {
                const double v = L.t[d][j];
                lo[d] = __builtin_fmin(lo[d], v);
                hi[d] = __builtin_fmax(hi[d], v);
            }
#pragma unroll
        for (int d = 0; d < D; d++) {
            cb[(c * 2 + 0) * D + d] = lo[d];
            cb[(c * 2 + 1) * D + d] = hi[d];
        }
    }
    __syncthreads();
}

// chunk_boxes with eight lanes per chunk (four targets each, then a butterfly over the eight): the tiled search computes
// the boxes of its tile in every launch, and one thread walking 32 targets is a chain of a hundred LDS round trips.
// blockDim.x must be a multiple of 64.  Same boxes.  Workgroup-wide; ends with a barrier.
// Returns (uniformly) whether the chunks are COMPACT: the mean, over chunks and over the axes along which the tile
// extends, of (chunk extent / tile extent) is below kChunkCompact.  32 consecutive points of a cloud in random order
// span 94 % of the cloud along every axis — their boxes exclude nothing, and the tests in front of every chunk cost
// 5 - 16 % (measured: 5 000^2 ... 30 000^2 uniform clouds); mesh vertices and scans give 0.1 - 0.4 (queen.ply: 0.32).
// cb needs room for one more chunk (the verdict's word).
constexpr double kChunkCompact = 0.6;
template <int D>
ICPX_DEV bool chunk_boxes_wide(const LdsView<D>& L, const int m, double* __restrict__ cb) {
    const int nch = (m + kChunk - 1) / kChunk;
    const int sub = threadIdx.x & 7;
    for (int c0 = 0; c0 < nch; c0 += static_cast<int>(blockDim.x >> 3)) {
        const int c = c0 + static_cast<int>(threadIdx.x >> 3);
        double lo[D], hi[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            lo[d] = __builtin_inf();
            hi[d] = -__builtin_inf();
        }
        if (c < nch) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = c * kChunk + sub * 4 + u;
                if (j < m) {
#pragma unroll
                    for (int d = 0; d < D; d++) {
                        const double v = L.t[d][j];
                        lo[d] = __builtin_fmin(lo[d], v);
                        hi[d] = __builtin_fmax(hi[d], v);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 1; o <= 4; o <<= 1) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                lo[d] = __builtin_fmin(lo[d], __shfl_xor(lo[d], o));
                hi[d] = __builtin_fmax(hi[d], __shfl_xor(hi[d], o));
            }
        }
        if (c < nch && sub == 0) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                cb[(c * 2 + 0) * D + d] = lo[d];
                cb[(c * 2 + 1) * D + d] = hi[d];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // wave 0: the tile's box, then the mean extent ratio
        const int lane = threadIdx.x;
        double tlo[D], thi[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            tlo[d] = __builtin_inf();
            thi[d] = -__builtin_inf();
        }
        for (int c = lane; c < nch; c += 64)
#pragma unroll
            for (int d = 0; d < D; d++) {
                tlo[d] = __builtin_fmin(tlo[d], cb[(c * 2 + 0) * D + d]);
                thi[d] = __builtin_fmax(thi[d], cb[(c * 2 + 1) * D + d]);
            }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
            for (int d = 0; d < D; d++) {
                tlo[d] = __builtin_fmin(tlo[d], __shfl_xor(tlo[d], o));
                thi[d] = __builtin_fmax(thi[d], __shfl_xor(thi[d], o));
            }
        double sum = 0.0, cnt = 0.0;
        for (int c = lane; c < nch; c += 64)
#pragma unroll
            for (int d = 0; d < D; d++) {
                const double te = thi[d] - tlo[d], ce = cb[(c * 2 + 1) * D + d] - cb[(c * 2 + 0) * D + d];
                if (te > 0.0 && ce >= 0.0) {  // (empty or all-NaN chunks: -inf extents)
                    sum += ce / te;
                    cnt += 1.0;
                }
            }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            sum += __shfl_xor(sum, o);
            cnt += __shfl_xor(cnt, o);
        }
        if (lane == 0) cb[2 * D * nch] = (cnt > 0.0 && sum < kChunkCompact * cnt) ? 1.0 : 0.0;
    }
    __syncthreads();
    return cb[2 * D * nch] != 0.0;
}

// bound[k]: exact d2 of slot k's query to some target of the cloud (+inf: none).  Out: what nn_lds returns.
template <int D, int QPT>
ICPX_DEV void nn_lds_chunked(const double (&q)[QPT][D], const bool (&valid)[QPT], const LdsView<D>& L, const int m,
                             const int m4, const double* __restrict__ cb, const double (&bound)[QPT],
                             double (&best)[QPT], int (&bidx)[QPT]) {
    const int nch = (m + kChunk - 1) / kChunk;
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        double bst = __builtin_inf();
        int bblk = 0;
        double bnd = bound[k];
        if (__any(valid[k] && !(bnd < __builtin_inf()))) {
            // no seed (a pair's first search): every lane looks through the chunk whose box is nearest to its query —
            // its own addresses, no broadcast — and takes the smallest distance there as its bound.  Without it the
            // scan below meets the query's neighbourhood only when it gets there, half the cloud later on average.
            double lbmin = __builtin_inf();
            int cmin = 0;
            for (int c = 0; c < nch; c++) {
                double lb = 0.0;
#pragma unroll
                for (int d = 0; d < D; d++) {
                    const double lo = cb[(c * 2 + 0) * D + d], hi = cb[(c * 2 + 1) * D + d];
                    const double g = __builtin_fmax(__builtin_fmax(lo - q[k][d], q[k][d] - hi), 0.0);
                    lb = d == 0 ? g * g : lb + g * g;
                }
                const bool lt = lb < lbmin;
                lbmin = lt ? lb : lbmin;
                cmin = lt ? c : cmin;
            }
            double pb_ = __builtin_inf();
            const int j1 = min((cmin + 1) * kChunk, m);
            for (int j = cmin * kChunk; j < j1; j++) {
                double df = L.t[0][j] - q[k][0];
                double acc = df * df;
#pragma unroll
                for (int d = 1; d < D; d++) {
                    df = L.t[d][j] - q[k][d];
                    acc = acc + df * df;
                }
                pb_ = __builtin_fmin(pb_, acc);
            }
            bnd = __builtin_fmin(bnd, pb_);  // (NaN bound: the probe's)
        }
        for (int c = 0; c < nch; c++) {
            double lb = 0.0;
#pragma unroll
            for (int d = 0; d < D; d++) {
                const double lo = cb[(c * 2 + 0) * D + d], hi = cb[(c * 2 + 1) * D + d];
                const double g = __builtin_fmax(__builtin_fmax(lo - q[k][d], q[k][d] - hi), 0.0);
                lb = d == 0 ? g * g : lb + g * g;
            }
            const double lim = __builtin_fmin(bst, bnd);  // (a NaN bound leaves the running minimum)
            if (!__any(valid[k] && !(lb > lim))) continue;
            const int j1 = min((c + 1) * kChunk, m4);
            for (int jb = c * kChunk; jb < j1; jb += 4) {
                double dist[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    double df = L.t[0][jb + u] - q[k][0];
                    double acc = df * df;
#pragma unroll
                    for (int d = 1; d < D; d++) {
                        df = L.t[d][jb + u] - q[k][d];
                        acc = acc + df * df;
                    }
                    dist[u] = acc;
                }
                const double mm = __builtin_fmin(__builtin_fmin(dist[0], dist[1]), __builtin_fmin(dist[2], dist[3]));
                const bool lt = mm < bst;
                bst = __builtin_fmin(bst, mm);
                bblk = lt ? jb : bblk;
            }
        }
        int idx = 0;
        if (bst < __builtin_inf()) {
#pragma unroll
            for (int u = 3; u >= 0; u--) {
                double df = L.t[0][bblk + u] - q[k][0];
                double acc = df * df;
#pragma unroll
                for (int d = 1; d < D; d++) {
                    df = L.t[d][bblk + u] - q[k][d];
                    acc = acc + df * df;
                }
                idx = acc == bst ? bblk + u : idx;
            }
        }
        best[k] = bst;
        bidx[k] = idx;
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// development statistics (build with -DICPX_STATS; read with icpx_debug_stats):
//   0 seeded scans (wave)  1 exact flushes (wave)  2 two-min scans (wave)  3 near-tie lanes  4 slot rescans (wave)
#ifdef ICPX_STATS
__device__ unsigned long long g_icpx_stats[32];
#ifdef ICPX_CYCONLY
#define ICPX_STAT(slot_, amount_) do { } while (0)
#else
#define ICPX_STAT(slot_, amount_)                                                                    \
    do {                                                                                             \
        unsigned long long stat_v_ = (unsigned long long)(amount_);                                  \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_icpx_stats[slot_], stat_v_);                       \
    } while (0)
#endif
// cycle stamps of wave 0 (s_memtime): ICPX_CYC_BEGIN(); ... ICPX_CYC(slot); ... ICPX_CYC(slot2);
#define ICPX_CYC_BEGIN() unsigned long long cyc_t0_ = __builtin_amdgcn_s_memtime()
#define ICPX_CYC(slot_)                                                                              \
    do {                                                                                             \
        unsigned long long cyc_t1_ = __builtin_amdgcn_s_memtime();                                   \
        if (threadIdx.x == 0) atomicAdd(&g_icpx_stats[slot_], cyc_t1_ - cyc_t0_);                    \
        cyc_t0_ = cyc_t1_;                                                                           \
    } while (0)
#else
#define ICPX_STAT(slot_, amount_) do { } while (0)
#define ICPX_CYC_BEGIN() do { } while (0)
#define ICPX_CYC(slot_) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// f32 candidate filter in front of the exact f64 search.
//
// Measured on MI355X (tools/micro/rate_probe.hip): a wave64 v_fma_f64 / v_add_f64 / v_pk_*_f32
// issues every ~5 cycles per SIMD, a plain v_fma_f32 / v_min_f32 every ~2.8.  The exact search
// costs ~10 f64 instructions per (query, target); the filter below costs 3 plain f32 FMAs.
//
// Filter value of target j for query q (both moved by the same offset c0, the centre of the
// target's bounding box, then rounded to f32: t32, q32):
//     v_j = |t32_j|^2 - 2 t32_j . q32          ( = |t32_j - q32|^2 - |q32|^2 )
// computed as fma(tz, -2qz, fma(ty, -2qy, fma(tx, -2qx, |t32|^2))), |t32|^2 once per target and wave.
// Error budget, u = 2^-24, Mt / Mq = max |coordinate| of the moved targets / of the moved query:
//   * rounding the moved coordinates to f32 displaces t32 - q32 from the true difference vector
//     by at most sqrt(D) u (Mt + Mq) <= E = 2^-22 (Mt + Mq)                 (distance domain)
//   * the f32 arithmetic of v_j is off by at most 6 u (|t32| + |q32|)^2 <= E2 = 24 u (Mt + Mq)^2
// so every target with exact distance d_j <= B satisfies  v_j <= (B + E)^2 - |q32|^2 + E2.
// filter_threshold returns an upper bound of that right-hand side for B^2 = d2max; a block of
// four targets whose smallest v is above it cannot contain a target with d2 <= d2max — ties
// included — whatever the rounding did.  Candidate blocks are re-evaluated exactly (exact_block),
// with the reference's expression on the original coordinates and the (d2, j) rule, so the result
// is the one nn_lds gives, bit for bit.
// ---------------------------------------------------------------------------------------------
struct FilterFrame {
    double c0[3];      // common offset: centre of the target bounding box
    double t_absmax;   // max |t - c0| over the target cloud; +inf when a coordinate is not finite / >= 1e18
};

ICPX_DEV float filter_threshold(double d2max, float qn_lo, float e_dist, float e_val) {
    float d2f = static_cast<float>(d2max) * 1.0000005f + 1e-30f;   // >= d2max
    float s = __builtin_sqrtf(d2f) * 1.000002f + e_dist;           // >= sqrt(d2max) + E
    float a = s * s * 1.000004f;                                    // >= (B + E)^2
    return (a - qn_lo) + (e_val + (a + qn_lo) * 2.4e-7f + 1e-30f);  // f32 roundings of this line: <= 2^-22 (a + qn)
}

// exact (d2, j) update of one query against the four targets of block `blk` (lanes with blk < 0
// idle); j counts in the cloud (L.j0 + slot)
// TIES (icpx_kd.h): *tie becomes true when another, non-coincident target lies within kKdNearUlps of the best
template <int D, bool TIES = false>
ICPX_DEV void exact_block(const double (&q)[D], const LdsView<D>& L, int blk, double& best, int& bidx, int* tie = nullptr) {
    const bool act = blk >= 0;
    const int jb = act ? blk : 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        double c[D];
#pragma unroll
        for (int d = 0; d < D; d++) c[d] = L.t[d][jb + u];
        double df = c[0] - q[0];
        double acc = df * df;
#pragma unroll
        for (int d = 1; d < D; d++) {
            df = c[d] - q[d];
            acc = acc + df * df;
        }
        const int j = lds_cloud_index<D>(L, jb + u);
        if (TIES) *tie |= kd_tie_check<D>(act, acc, j, c, L.cloud, best, bidx) ? 1 : 0;
        bool better = act && (acc < best || (acc == best && j < bidx));
        best = better ? acc : best;
        bidx = better ? j : bidx;
    }
}

ICPX_DEV float raw_min_f32(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
ICPX_DEV float raw_max_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// per-lane filter operands of the QPT queries; returns false when this lane cannot use the filter
template <int D, int QPT>
ICPX_DEV bool filter_queries(const double (&q)[QPT][D], const FilterFrame& F, float (&q2)[QPT][D], float (&qn)[QPT],
                             float& e_dist, float& e_val) {
    bool ok = F.t_absmax < 1e18;
    double mq = 0.0;
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        double nn = 0.0;
#pragma unroll
        for (int d = 0; d < D; d++) {
            double c = q[k][d] - F.c0[d];
            ok = ok && (__builtin_fabs(c) < 1e18);
            mq = __builtin_fmax(mq, __builtin_fabs(c));
            float f = static_cast<float>(c);
            q2[k][d] = -2.0f * f;
            nn = nn + static_cast<double>(f) * static_cast<double>(f);
        }
        qn[k] = static_cast<float>(nn * (1.0 - 1.0 / 4194304.0));  // <= |q32|^2
    }
    const double mm = F.t_absmax + mq;
    e_dist = static_cast<float>(mm * (1.001 / 4194304.0));           // 2^-22 (Mt + Mq)
    e_val = static_cast<float>(mm * mm * (24.3 / 16777216.0));       // 24 u (Mt + Mq)^2
    return ok;
}

// smallest filter value of the four targets of a block for one query
template <int D>
ICPX_DEV float block_filter_min(const f32x4 (&tc)[D], const f32x4& tn, const float (&q2)[D]) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        float a = __builtin_fmaf(tc[0][u], q2[0], tn[u]);
#pragma unroll
        for (int d = 1; d < D; d++) a = __builtin_fmaf(tc[d][u], q2[d], a);
        v[u] = a;
    }
    return __builtin_fminf(__builtin_fminf(v[0], v[1]), __builtin_fminf(v[2], v[3]));
}

template <int D>
ICPX_DEV void load_filter_block(const LdsView<D>& L, int jb, f32x4 (&tc)[D], f32x4& tn) {
#pragma unroll
    for (int d = 0; d < D; d++) tc[d] = *reinterpret_cast<const f32x4*>(L.tf[d] + jb);
#pragma unroll
    for (int u = 0; u < 4; u++) {
        float a = tc[0][u] * tc[0][u];
#pragma unroll
        for (int d = 1; d < D; d++) a = __builtin_fmaf(tc[d][u], tc[d][u], a);
        tn[u] = a;
    }
}

// Stages a target tile (m points, AoS in global memory) into LDS: f64 SoA padded with +inf to m4,
// and — for the filter — its f32 copy moved to the centre of the tile's bounding box (padding
// 1e18f: large, finite, never a candidate) together with the frame the error budget needs.
// Workgroup-wide (contains barriers); L.gs serves as reduction scratch (>= 72 doubles).
template <int D>
ICPX_DEV FilterFrame stage_targets(const double* __restrict__ dst, const int m, const int m4, const int T,
                                   const LdsView<D>& L, const bool filter,
                                   const uint16_t* __restrict__ perm = nullptr) {
    const int tid = threadIdx.x;
    double lo[D], hi[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        lo[d] = __builtin_inf();
        hi[d] = -__builtin_inf();
    }
    bool t_bad = false;
    for (int j = tid; j < m4; j += T) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            // perm: LDS slot j holds target perm[j] of the cloud (icpx_prune.h); interleaved tiles (L.cstride): slot j
            // holds target (j / 32) * cstride + j % 32 counted from dst
            const int src = perm ? static_cast<int>(perm[j]) : (L.cstride ? (j >> 5) * L.cstride + (j & 31) : j);
            double v = j < m ? dst[static_cast<int64_t>(src) * D + d] : __builtin_inf();
            L.t[d][j] = v;
            if (filter && j < m) {
                t_bad = t_bad || !(__builtin_fabs(v) < 1e300);
                lo[d] = __builtin_fmin(lo[d], v);
                hi[d] = __builtin_fmax(hi[d], v);
            }
        }
    }
    // filter frame: bounding-box centre, f32 copy of the moved targets, max |moved coordinate|
    FilterFrame F;
#pragma unroll
    for (int d = 0; d < 3; d++) F.c0[d] = 0.0;
    F.t_absmax = __builtin_inf();
    if (filter) {
        const int lane = tid & 63, wave = tid >> 6, W = T >> 6;
#pragma unroll
        for (int d = 0; d < D; d++) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                lo[d] = __builtin_fmin(lo[d], __shfl_xor(lo[d], o));
                hi[d] = __builtin_fmax(hi[d], __shfl_xor(hi[d], o));
            }
            if (lane == 0) {
                L.gs[wave * 8 + d] = lo[d];
                L.gs[wave * 8 + 4 + d] = hi[d];
            }
        }
        const bool wave_bad = __any(t_bad);
        if (lane == 0) L.gs[wave * 8 + 3] = wave_bad ? 1.0 : 0.0;
        __syncthreads();
        bool any_bad = false;
#pragma unroll
        for (int d = 0; d < D; d++) {
            double l = __builtin_inf(), h = -__builtin_inf();
            for (int w = 0; w < W; w++) {
                l = __builtin_fmin(l, L.gs[w * 8 + d]);
                h = __builtin_fmax(h, L.gs[w * 8 + 4 + d]);
            }
            F.c0[d] = m > 0 ? 0.5 * l + 0.5 * h : 0.0;
        }
        for (int w = 0; w < W; w++) any_bad = any_bad || L.gs[w * 8 + 3] != 0.0;
        double amax = 0.0;
        for (int j = tid; j < m4; j += T) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                double c = L.t[d][j] - F.c0[d];
                bool real = j < m;
                // padding: a large finite value, never a candidate and never NaN in the filter
                L.tf[d][j] = real ? static_cast<float>(c) : 1e18f;
                if (real) amax = __builtin_fmax(amax, __builtin_fabs(c) < 1e18 ? __builtin_fabs(c) : __builtin_inf());
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) amax = __builtin_fmax(amax, __shfl_xor(amax, o));
        __syncthreads();  // everyone has read the bounding box before gs is reused
        if (lane == 0) L.gs[wave] = amax;
        __syncthreads();
        amax = 0.0;
        for (int w = 0; w < W; w++) amax = __builtin_fmax(amax, L.gs[w]);
        F.t_absmax = any_bad ? __builtin_inf() : amax;
        // the barrier after the state block below separates these reads from the next use of gs
    }
    __syncthreads();  // staged data and frame visible; gs free again
    return F;
}

// exact distance of every query to its seed target (slot index in this tile): the upper bound the
// seeded search starts from
template <int D, int QPT>
ICPX_DEV void seed_bound(const double (&q)[QPT][D], const LdsView<D>& L, int m, double (&best)[QPT], int (&bidx)[QPT]) {
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        int j = bidx[k];
        j = j < 0 ? 0 : (j >= m ? m - 1 : j);
        double df = L.t[0][j] - q[k][0];
        double acc = df * df;
#pragma unroll
        for (int d = 1; d < D; d++) {
            df = L.t[d][j] - q[k][d];
            acc = acc + df * df;
        }
        best[k] = acc;
        bidx[k] = L.j0 + j;
    }
}

// Seeded search.  In: (best, bidx) = exact d2 and cloud index of SOME target of the cloud (the
// previous iteration's neighbour) — an upper bound of the answer.  Out: the lexicographic minimum
// of that pair and of all targets of this tile.  ALL targets are scanned through the filter, four
// at a time; candidate blocks are evaluated exactly and every improvement tightens the
// threshold.  One pending block per query slot is kept in a register; a second candidate for the
// same slot flushes the first (wave-wide).  Waves with unusable coordinates take nn_lds.
// TIES: tie[k] |= another non-coincident target within kKdNearUlps of slot k's minimum (icpx_kd.h)
// CHUNKED (tiled search of the multi-kernel path): cb holds the bounding boxes of every kChunk consecutive slots
// (chunk_boxes_wide); a chunk's eight blocks go through the filter only if SOME lane's query can have a target there
// within (1 + 2^-40) of its running minimum — lb, the squared distance to the box computed like a d2 (monotone
// roundings: no target of the chunk has a smaller computed d2, see nn_lds_chunked), compared strictly.  A skipped chunk
// holds nothing that could improve a minimum or tie with it (a tie partner lies within kKdNearUlps), so the results
// and the tie flags are those of the full scan.  Pays when queries and targets come in a coherent order (mesh
// vertices, scans) AND the tiles are interleaved (LdsView::cstride): with contiguous tiles 70 % of queen.ply's chunks
// are skipped and the launch is as long as before — as long as the few workgroups whose tile lies in their queries'
// slab; interleaved, every workgroup holds its share of every slab.
template <int D, int QPT, bool TIES = false, bool CHUNKED = false>
ICPX_DEV void nn_filtered(const double (&q)[QPT][D], const LdsView<D>& L, int m4, const FilterFrame& F,
                          double (&best)[QPT], int (&bidx)[QPT], int* tie = nullptr, const double* __restrict__ cb = nullptr) {
    float q2[QPT][D], qn[QPT], e_dist, e_val;
    const bool ok = filter_queries<D, QPT>(q, F, q2, qn, e_dist, e_val);
    if (__any(!ok)) {
        double lb[QPT];
        int li[QPT];
        int lt[TIES ? QPT : 1];
        nn_lds<D, QPT, TIES>(q, L, m4, lb, li, TIES ? lt : nullptr);
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            const int j = lds_cloud_index<D>(L, li[k]);
            const bool take = lb[k] < best[k] || (lb[k] == best[k] && j < bidx[k]);
            // (coincident targets are not told apart here: a flag too many only costs the resolution's time)
            if (TIES) tie[k] |= lt[k] | ((j != bidx[k] && lb[k] < __builtin_inf() && kd_near(lb[k], best[k])) ? 1 : 0);
            best[k] = take ? lb[k] : best[k];
            bidx[k] = take ? j : bidx[k];
        }
        return;
    }
    float thr[QPT];
    int pend[QPT];
    ICPX_STAT(0, 1);
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        thr[k] = filter_threshold(best[k], qn[k], e_dist, e_val);
        pend[k] = -1;
    }
    constexpr int G = QPT >= 2 ? 2 : 1;  // slots per branch: wider basic blocks schedule better
    constexpr int kOuter = CHUNKED ? kChunk : 4;
    for (int jc = 0; jc < m4; jc += kOuter) {
        if (CHUNKED) {
            const int c = jc / kChunk;
            bool need = false;
#pragma unroll
            for (int k = 0; k < QPT; k++) {
                double lb = 0.0;
#pragma unroll
                for (int d = 0; d < D; d++) {
                    const double lo = cb[(c * 2 + 0) * D + d], hi = cb[(c * 2 + 1) * D + d];
                    const double g = __builtin_fmax(__builtin_fmax(lo - q[k][d], q[k][d] - hi), 0.0);
                    lb = d == 0 ? g * g : lb + g * g;
                }
                need = need || !(lb > best[k] * 1.0000000000009095);  // (1 + 2^-40); NaN never skips
            }
            if (!__any(need)) continue;
        }
        const int jend = CHUNKED ? (jc + kChunk < m4 ? jc + kChunk : m4) : jc + 4;
      for (int jb = jc; jb < jend; jb += 4) {
        f32x4 tc[D], tn;
        load_filter_block<D>(L, jb, tc, tn);
#pragma unroll
        for (int k0 = 0; k0 < QPT; k0 += G) {
            bool cand[G];
            bool any = false;
#pragma unroll
            for (int g = 0; g < G; g++) {
                cand[g] = !(block_filter_min<D>(tc, tn, q2[k0 + g]) > thr[k0 + g]);
                any = any || cand[g];
            }
            if (__any(any)) {
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int k = k0 + g;
                    if (__any(cand[g] && pend[k] >= 0)) {
                        ICPX_STAT(1, 1);
                        double before = best[k];
                        exact_block<D, TIES>(q[k], L, pend[k], best[k], bidx[k], TIES ? &tie[k] : nullptr);
                        pend[k] = -1;
                        if (best[k] < before) thr[k] = filter_threshold(best[k], qn[k], e_dist, e_val);
                    }
                    pend[k] = cand[g] ? jb : pend[k];
                }
            }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < QPT; k++)
        if (__any(pend[k] >= 0)) exact_block<D, TIES>(q[k], L, pend[k], best[k], bidx[k], TIES ? &tie[k] : nullptr);
}

// Wave-cooperative rescan of single queries: for every lane whose flag is set, the wave takes
// that lane's query (broadcast), each lane filters the blocks lane, lane + 64, ... and evaluates
// its candidate blocks exactly; a butterfly over (d2, j) leaves the lexicographic minimum, which
// the owning lane keeps.  1024 blocks cost 16 steps instead of a 1024-step scan with one useful
// lane.
template <int D, bool TIES = false>
ICPX_DEV void coop_rescan(bool flag, const double (&q)[D], const float (&q2)[D], float thr, const LdsView<D>& L,
                          int m4, double& best, int& bidx, int* tie = nullptr) {
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
        int bi = 0x7fffffff;
        int lt = 0;
        for (int jb = lane * 4; jb < m4; jb += 256) {
            f32x4 tc[D], tn;
            load_filter_block<D>(L, jb, tc, tn);
            if (!(block_filter_min<D>(tc, tn, q2b) > thr_b)) exact_block<D, TIES>(qb, L, jb, bb, bi, TIES ? &lt : nullptr);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            double ob = __shfl_xor(bb, o);
            int oi = __shfl_xor(bi, o);
            bool take = ob < bb || (ob == bb && oi < bi);
            // two lanes' minima within the tie range: a tie unless the two targets coincide (a cloud with duplicate
            // points — queen.ply has 1 956 — would flag every query whose neighbour has a twin)
            if (TIES) {
                bool near = oi != 0x7fffffff && bi != 0x7fffffff && oi != bi && ob < __builtin_inf() && kd_near(ob, bb);
                if (near && ob == bb && L.cloud != nullptr) {
                    bool same = true;
#pragma unroll
                    for (int d = 0; d < D; d++) same = same && L.cloud[static_cast<int64_t>(oi) * D + d] == L.cloud[static_cast<int64_t>(bi) * D + d];
                    near = !same;
                }
                lt |= near ? 1 : 0;
            }
            bb = take ? ob : bb;
            bi = take ? oi : bi;
        }
        if (TIES) lt = __any(lt != 0) ? 1 : 0;
        if (lane == src && bb < __builtin_inf()) {
            best = bb;
            bidx = bi;
            if (TIES) *tie = lt;  // the rescan saw every target that close: its verdict replaces the winning block's
        }
    }
}

// Seedless search (first search of a pair / of a tile).  Out: the lexicographic minimum over the
// targets of this tile, index in the cloud.  ONE filter pass keeps, per query, the smallest
// block value b1 with its block and the second smallest b2 over all other blocks:
//     b2' = min(b2, max(b1, v)),  b1' = min(b1, v).
// The winning block is evaluated exactly; if b2 lies above the threshold of that exact best, no
// other block can hold a target with d2 <= best and the search is complete.  Otherwise — exact
// ties and near-ties inside the error budget — the query is rescanned cooperatively (coop_rescan).
template <int D, int QPT, bool TIES = false>
ICPX_DEV void nn_twomin(const double (&q)[QPT][D], const LdsView<D>& L, int m4, const FilterFrame& F,
                        double (&best)[QPT], int (&bidx)[QPT], int* tie = nullptr) {
    float q2[QPT][D], qn[QPT], e_dist, e_val;
    const bool ok = filter_queries<D, QPT>(q, F, q2, qn, e_dist, e_val);
    if (__any(!ok)) {
        nn_lds<D, QPT, TIES>(q, L, m4, best, bidx, tie);
#pragma unroll
        for (int k = 0; k < QPT; k++) bidx[k] = lds_cloud_index<D>(L, bidx[k]);
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
            // plain v_max / v_min: the values are never NaN here, and the builtins would re-quiet the
            // loop-carried operands on every step (two extra instructions per slot and block)
            b2[k] = raw_min_f32(b2[k], raw_max_f32(b1[k], mn));
            b1[k] = raw_min_f32(b1[k], mn);
            blk[k] = lt ? jb : blk[k];
        }
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        best[k] = __builtin_inf();
        bidx[k] = 0x7fffffff;
        if (TIES) tie[k] = 0;
        exact_block<D, TIES>(q[k], L, blk[k], best[k], bidx[k], TIES ? &tie[k] : nullptr);
        // near-ties (about 0.15 % of the queries on uniform clouds): cooperative rescan
        const float thr = filter_threshold(best[k], qn[k], e_dist, e_val);
        const bool again = !(b2[k] > thr);
        ICPX_STAT(3, __popcll(__ballot(again)));
        if (__any(again)) {
            ICPX_STAT(4, 1);
            coop_rescan<D, TIES>(again, q[k], q2[k], thr, L, m4, best[k], bidx[k], TIES ? &tie[k] : nullptr);
        }
    }
}

}  // namespace icpx
