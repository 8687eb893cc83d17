// icpx_pair.h — the fused, device-resident ICP loop for scan-sized clouds: ONE workgroup runs a
// whole ICPDriver::converge() for one scan pair.
//
// Per iteration, without leaving the kernel (reference: lib/icp/impl/{vanilla,trimmed,vanilla_3d,
// trimmed_3d}.cpp iterate(); include/icp/driver.h:51-72,154-222):
//   transform apply  ->  brute-force NN against the target cloud staged in LDS (SoA, broadcast
//   reads; each lane owns up to QPT queries in registers)  ->  selection (2D trimmed: bitonic
//   sort of (cost, point) keys in LDS; 3D trimmed: distance threshold)  ->  centroid and
//   cross-covariance sums in the TREE64 order (wave butterfly, then one butterfly over the <= 64
//   group sums)  ->  single-lane Jacobi SVD / Kabsch  ->  transform composition, cost,
//   should_terminate.
//
// Limits of this kernel: n <= 4096 source points (<= 64 TREE64 groups) and a target cloud that
// fits LDS (checked on the host, icpx_capi.hip).  Larger clouds take the multi-kernel path.
#pragma once

#include "icpx_math.h"

namespace icpx {

constexpr int kPairMaxN = 4096;
constexpr int kGsStride = 64;   // level-1 group sums per value
constexpr int kGsValues = 10;   // up to 9 covariance terms (+ spare)
constexpr int kBcDoubles = 48;  // broadcast area

struct PairBatch {
    int dim;
    int method;
    const double* src;         // concatenated source clouds (AoS)
    const int64_t* src_off;    // n_pairs+1 offsets in points (device)
    const double* dst;
    const int64_t* dst_off;
    const double* guesses;     // n_pairs x (D+1)^2 or nullptr
    double* cur;               // 3D running cloud c, same offsets as src (persistent)
    int64_t* m_point;          // matches / neighbours, same offsets as src; may be nullptr
    int64_t* m_pair;
    double* m_cost;
    int32_t* pair_tmp;         // n_total scratch (pair by point), nullptr iff m_point is
    PairState* states;         // n_pairs
    icpx_state* out_states;    // n_pairs, device staging (may be nullptr)
    icpx_criteria crit;
    double overlap_rate;
    double max_distance;
    double clock_to_s;         // seconds per wall_clock64() tick
    double host_elapsed_s;     // time already spent on the host side of converge()
    uint64_t max_steps;        // iterations this launch may run
    int32_t do_setup;          // run begin()/setup() first
    int32_t check;             // evaluate should_terminate (converge) or not (iterate)
    int32_t lds_target_cap;    // padded target capacity the launch reserved in LDS
    int32_t lds_sort_cap;      // npad reserved for the sort area (2D trimmed)
    // iteration-granular scheduling (batches whose pairs outnumber the resident workgroups only
    // slightly): workgroups are persistent and pick (pair, next `sched_steps` iterations) tasks
    int32_t* sched_state;      // n_pairs words: (iterations done << 2) | {0 fresh, 1 ready, 2 running, 3 done}; null = off
    int32_t* sched_done;       // number of pairs in state 3
    int32_t n_pairs;
    int32_t sched_steps;       // iterations per task
    // packed-f32 candidate filter in front of the exact f64 search (nn_filtered below)
    int32_t use_filter;        // the launch reserved the f32 target copy in LDS
    int32_t* seeds;            // n_total: previous nearest neighbour by point, carried between launches / tasks; may be nullptr
};

template <int D>
struct LdsView {
    double* t[3];    // target SoA
    double* gs;      // kGsValues x kGsStride
    double* bc;      // broadcast
    uint64_t* skey;  // sort keys
    uint32_t* sval;  // sort values
    float* tf[3];    // f32 copy of the target SoA (candidate filter), null when not reserved
};

inline __host__ __device__ size_t pair_lds_filter_offset(int dim, int tcap, int scap) {
    size_t b = sizeof(double) * (static_cast<size_t>(dim) * tcap + kGsValues * kGsStride + kBcDoubles) +
               static_cast<size_t>(scap) * 12;
    return (b + 15) & ~static_cast<size_t>(15);
}

template <int D>
ICPX_DEV LdsView<D> carve(double* base, int tcap, int scap, int filter) {
    LdsView<D> v;
    double* p = base;
    for (int d = 0; d < D; d++) {
        v.t[d] = p;
        p += tcap;
    }
    if (D == 2) v.t[2] = nullptr;
    v.gs = p;
    p += kGsValues * kGsStride;
    v.bc = p;
    p += kBcDoubles;
    v.skey = reinterpret_cast<uint64_t*>(p);
    v.sval = reinterpret_cast<uint32_t*>(v.skey + scap);
    float* f = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + pair_lds_filter_offset(D, tcap, scap));
    for (int d = 0; d < 3; d++) v.tf[d] = (filter && d < D) ? f + static_cast<size_t>(d) * tcap : nullptr;
    return v;
}

inline size_t pair_lds_bytes(int dim, int tcap, int scap, bool filter) {
    return pair_lds_filter_offset(dim, tcap, scap) + (filter ? sizeof(float) * static_cast<size_t>(dim) * tcap : 0);
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
template <int D, int QPT>
ICPX_DEV void nn_lds(const double (&q)[QPT][D], const LdsView<D>& L, int m4, double (&best)[QPT],
                     int (&bidx)[QPT]) {
    int bblk[QPT];
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        best[k] = __builtin_inf();
        bblk[k] = 0;
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
            best[k] = __builtin_fmin(best[k], mm);
            bblk[k] = lt ? jb : bblk[k];
        }
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        int idx = 0;
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
            }
        }
        bidx[k] = idx;
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// development statistics (build with -DICPX_STATS; read with icpx_debug_stats):
//   0 seeded scans (wave)  1 exact flushes (wave)  2 two-min scans (wave)  3 near-tie lanes  4 slot rescans (wave)
#ifdef ICPX_STATS
__device__ unsigned long long g_icpx_stats[8];
#define ICPX_STAT(slot_, amount_)                                                                    \
    do {                                                                                             \
        unsigned long long stat_v_ = (unsigned long long)(amount_);                                  \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_icpx_stats[slot_], stat_v_);                       \
    } while (0)
#else
#define ICPX_STAT(slot_, amount_) do { } while (0)
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

// exact (d2, j) update of one query against the four targets of block `blk` (lanes with blk < 0 idle)
template <int D>
ICPX_DEV void exact_block(const double (&q)[D], const LdsView<D>& L, int blk, double& best, int& bidx) {
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
        bool better = act && (acc < best || (acc == best && jb + u < bidx));
        best = better ? acc : best;
        bidx = better ? jb + u : bidx;
    }
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

// Seeded search: bidx holds the previous iteration's neighbours.  The exact distance to the seed
// bounds the answer; ALL targets are scanned through the filter, four at a time; candidate
// blocks are evaluated exactly and every improvement tightens the threshold.  One pending block
// per query slot is kept in a register; a second candidate for the same slot flushes the first
// (wave-wide).  Waves with unusable coordinates take nn_lds.
template <int D, int QPT>
ICPX_DEV void nn_filtered(const double (&q)[QPT][D], const LdsView<D>& L, int m, int m4, const FilterFrame& F,
                          double (&best)[QPT], int (&bidx)[QPT]) {
    float q2[QPT][D], qn[QPT], e_dist, e_val;
    const bool ok = filter_queries<D, QPT>(q, F, q2, qn, e_dist, e_val);
    if (__any(!ok)) {
        nn_lds<D, QPT>(q, L, m4, best, bidx);
        return;
    }
    float thr[QPT];
    int pend[QPT];
    ICPX_STAT(0, 1);
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
        bidx[k] = j;
        thr[k] = filter_threshold(acc, qn[k], e_dist, e_val);
        pend[k] = -1;
    }
    constexpr int G = QPT >= 2 ? 2 : 1;  // slots per branch: wider basic blocks schedule better
    for (int jb = 0; jb < m4; jb += 4) {
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
                        exact_block<D>(q[k], L, pend[k], best[k], bidx[k]);
                        pend[k] = -1;
                        if (best[k] < before) thr[k] = filter_threshold(best[k], qn[k], e_dist, e_val);
                    }
                    pend[k] = cand[g] ? jb : pend[k];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < QPT; k++)
        if (__any(pend[k] >= 0)) exact_block<D>(q[k], L, pend[k], best[k], bidx[k]);
}

// Wave-cooperative rescan of single queries: for every lane whose flag is set, the wave takes
// that lane's query (broadcast), each lane filters the blocks lane, lane + 64, ... and evaluates
// its candidate blocks exactly; a butterfly over (d2, j) leaves the lexicographic minimum, which
// the owning lane keeps.  1024 blocks cost 16 steps instead of a 1024-step scan with one useful
// lane.
template <int D>
ICPX_DEV void coop_rescan(bool flag, const double (&q)[D], const float (&q2)[D], float thr, const LdsView<D>& L,
                          int m4, double& best, int& bidx) {
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
        for (int jb = lane * 4; jb < m4; jb += 256) {
            f32x4 tc[D], tn;
            load_filter_block<D>(L, jb, tc, tn);
            if (!(block_filter_min<D>(tc, tn, q2b) > thr_b)) exact_block<D>(qb, L, jb, bb, bi);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            double ob = __shfl_xor(bb, o);
            int oi = __shfl_xor(bi, o);
            bool take = ob < bb || (ob == bb && oi < bi);
            bb = take ? ob : bb;
            bi = take ? oi : bi;
        }
        if (lane == src && bb < __builtin_inf()) {
            best = bb;
            bidx = bi;
        }
    }
}

// Seedless search (first search of a pair): ONE filter pass keeps, per query, the smallest
// block value b1 with its block and the second smallest b2 over all other blocks:
//     b2' = min(b2, max(b1, v)),  b1' = min(b1, v).
// The winning block is evaluated exactly; if b2 lies above the threshold of that exact best, no
// other block can hold a target with d2 <= best and the search is complete.  Otherwise — exact
// ties and near-ties inside the error budget — the query is rescanned cooperatively (coop_rescan).
template <int D, int QPT>
ICPX_DEV void nn_twomin(const double (&q)[QPT][D], const LdsView<D>& L, int m, int m4, const FilterFrame& F,
                        double (&best)[QPT], int (&bidx)[QPT]) {
    float q2[QPT][D], qn[QPT], e_dist, e_val;
    const bool ok = filter_queries<D, QPT>(q, F, q2, qn, e_dist, e_val);
    if (__any(!ok)) {
        nn_lds<D, QPT>(q, L, m4, best, bidx);
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
            b2[k] = __builtin_fminf(b2[k], __builtin_fmaxf(b1[k], mn));
            b1[k] = __builtin_fminf(b1[k], mn);
            blk[k] = lt ? jb : blk[k];
        }
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        best[k] = __builtin_inf();
        bidx[k] = 0;
        exact_block<D>(q[k], L, blk[k], best[k], bidx[k]);
        // near-ties (about 0.15 % of the queries on uniform clouds): cooperative rescan
        const float thr = filter_threshold(best[k], qn[k], e_dist, e_val);
        const bool again = !(b2[k] > thr);
        ICPX_STAT(3, __popcll(__ballot(again)));
        if (__any(again)) {
            ICPX_STAT(4, 1);
            coop_rescan<D>(again, q[k], q2[k], thr, L, m4, best[k], bidx[k]);
        }
    }
}

// TREE64 sum of NV values per point.  f(k, vals) fills the NV contributions of point tid + k*T
// (0.0 where the point does not exist or is not selected).  Every thread receives the totals.
template <int NV, int QPT, typename F>
ICPX_DEV void block_sum(F&& f, int n, int T, double* gs, double (&out)[NV]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = T >> 6;
    const int G = (n + 63) >> 6;
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        int g = k * W + wave;
        if (g < G) {
            double vals[NV];
            f(k, vals);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                double s = wave_sum(vals[v]);
                if (lane == 0) gs[v * kGsStride + g] = s;
            }
        }
    }
    __syncthreads();
    if (G <= 1) {
#pragma unroll
        for (int v = 0; v < NV; v++) out[v] = G == 1 ? gs[v * kGsStride] : 0.0;
    } else {
#pragma unroll
        for (int v = 0; v < NV; v++) {
            double x = lane < G ? gs[v * kGsStride + lane] : 0.0;
            out[v] = wave_sum(x);
        }
    }
    __syncthreads();
}

template <int QPT>
ICPX_DEV int block_count(const int (&flags)[QPT], int T, double* gs_as_scratch) {
    int* cnt = reinterpret_cast<int*>(gs_as_scratch);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = T >> 6;
    int c = 0;
#pragma unroll
    for (int k = 0; k < QPT; k++) c += flags[k];
    c = wave_sum_int(c);
    if (lane == 0) cnt[wave] = c;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < W; w++) tot += cnt[w];
    __syncthreads();
    return tot;
}

// in-LDS bitonic sort of npad (power of two) (key, val) pairs, ascending lexicographic
ICPX_DEV void bitonic_sort(uint64_t* key, uint32_t* val, int npad, int T) {
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npad; i += T) {
                int ixj = i ^ j;
                if (ixj > i) {
                    uint64_t ka = key[i], kb = key[ixj];
                    uint32_t va = val[i], vb = val[ixj];
                    bool a_gt_b = ka > kb || (ka == kb && va > vb);
                    bool up = (i & k) == 0;
                    if (a_gt_b == up) {
                        key[i] = kb;
                        key[ixj] = ka;
                        val[i] = vb;
                        val[ixj] = va;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// k-th smallest of the (key, point) pairs held QPT per thread, rank 1-based, lexicographic order:
// MSB-first radix select with 8-bit digits over the 64 key bits (LDS histogram, wave-0 prefix
// scan), then — only when the rank falls strictly inside a run of equal keys — over the point
// index.  Replaces the full sort in Trimmed::iterate (trimmed.cpp:55-58): only the threshold is
// needed to mark the kept matches.  hist: >= 264 ints of LDS.
template <int QPT>
ICPX_DEV void radix_select(const uint64_t (&key)[QPT], const bool (&valid)[QPT], int T, int rank, int* hist,
                           uint64_t& kth, uint32_t& vth) {
    const int tid = threadIdx.x, lane = tid & 63;
    bool act[QPT];
#pragma unroll
    for (int k = 0; k < QPT; k++) act[k] = valid[k];
    uint64_t prefix = 0;
    uint32_t vprefix = 0;
    int remaining = rank;
    for (int pass = 0; pass < 10; pass++) {
        const bool on_key = pass < 8;
        const int shift = on_key ? 56 - 8 * pass : 8 * (9 - pass);
        for (int i = tid; i < 256; i += T) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            const uint32_t point = static_cast<uint32_t>(tid + k * T);
            const int d = on_key ? static_cast<int>((key[k] >> shift) & 255) : static_cast<int>((point >> shift) & 255);
            // the leading digits of positive doubles of one cloud mostly coincide: one add per wave then
            const unsigned long long am = __ballot(act[k]);
            if (am) {
                const int d0 = __shfl(d, __ffsll(static_cast<long long>(am)) - 1);
                if (__all(!act[k] || d == d0)) {
                    if (lane == 0) atomicAdd(&hist[d0], __popcll(am));
                } else if (act[k]) {
                    atomicAdd(&hist[d], 1);
                }
            }
        }
        __syncthreads();
        if (tid < 64) {
            int c[4], s = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                c[u] = hist[4 * lane + u];
                s += c[u];
            }
            int incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                int up = __shfl_up(incl, o);
                incl += lane >= o ? up : 0;
            }
            int excl = incl - s;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (excl < remaining && remaining <= excl + c[u]) {
                    hist[256] = 4 * lane + u;
                    hist[257] = remaining - excl;
                    hist[258] = c[u];
                }
                excl += c[u];
            }
        }
        __syncthreads();
        const int d = hist[256];
        remaining = hist[257];
        const int bucket = hist[258];
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            const uint32_t point = static_cast<uint32_t>(tid + k * T);
            const int dk = on_key ? static_cast<int>((key[k] >> shift) & 255) : static_cast<int>((point >> shift) & 255);
            act[k] = act[k] && dk == d;
        }
        if (on_key) prefix |= static_cast<uint64_t>(d) << shift;
        else vprefix |= static_cast<uint32_t>(d) << shift;
        if (pass == 7 && remaining == bucket) {
            // the whole run of equal keys is kept
            vprefix = 0xffffffffu;
            break;
        }
    }
    kth = prefix;
    vth = vprefix;
    __syncthreads();  // hist may be reused by the caller
}

// One visit of pair p: optional begin()/setup(), then up to max_steps iterations of the converge
// loop.  Returns 1 when should_terminate fired (pair finished), 2 when it paused at max_steps.
template <int D, int METHOD, int QPT>
ICPX_DEV int pair_body(const PairBatch& pb, const int p, const int do_setup, const uint64_t max_steps,
                       double* lds_raw) {
    constexpr int L1 = D + 1;
    constexpr int XN = L1 * L1;
    const int T = blockDim.x;
    const int tid = threadIdx.x;
    const int64_t s0 = pb.src_off[p], d0 = pb.dst_off[p];
    const int n = static_cast<int>(pb.src_off[p + 1] - s0);
    const int m = static_cast<int>(pb.dst_off[p + 1] - d0);
    const double* __restrict__ src = pb.src + s0 * D;
    const double* __restrict__ dst = pb.dst + d0 * D;
    PairState* st = pb.states + p;
    const bool filter = pb.use_filter != 0 && !(D == 2 && METHOD == ICPX_VANILLA);
    LdsView<D> L = carve<D>(lds_raw, pb.lds_target_cap, pb.lds_sort_cap, filter);
    const bool empty = (n == 0 || m == 0);

    // ---- stage the target cloud into LDS as SoA ----
    const int m4 = (m + 3) & ~3;
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
            double v = j < m ? dst[static_cast<int64_t>(j) * D + d] : __builtin_inf();
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
    // ---- per-thread point slots ----
    double a[QPT][D];    // 2D: source points; 3D: running cloud c
    double cost[QPT];    // match cost (2D) / neighbour distance (3D)
    int pair[QPT];
    bool valid[QPT];
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        int i = tid + k * T;
        valid[k] = i < n;
        cost[k] = 0.0;
        pair[k] = 0;
#pragma unroll
        for (int d = 0; d < D; d++) {
            double v = 0.0;
            if (valid[k]) {
                if (D == 3 && !do_setup) v = pb.cur[(s0 + i) * D + d];
                else v = src[static_cast<int64_t>(i) * D + d];
            }
            a[k][d] = v;
        }
    }

    // ---- state ----
    if (tid == 0) {
        if (do_setup) {
            // ICP::begin, icp.h:120-133
            for (int e = 0; e < XN; e++) {
                double g = pb.guesses ? pb.guesses[static_cast<int64_t>(p) * XN + e]
                                      : ((e % (L1 + 1)) == 0 ? 1.0 : 0.0);
                st->transform[e] = g;
                st->last_transform[e] = g;
            }
            st->iteration_count = 0;
            st->have_last = 0;
            st->done = 0;
            st->last_cost = 0.0;
            st->start_clock = wall_clock64();
        }
        for (int e = 0; e < XN; e++) L.bc[e] = st->transform[e];
    }
    __syncthreads();
    // seeds for the filtered search: the previous nearest neighbours.  Trimmed ICP2 has them from
    // setup on; ICP3 after its first iterate; a resumed launch reloads them.
    bool have_seed = false;
    if (filter && !do_setup && pb.seeds && !empty && (D == 2 || st->iteration_count > 0)) {
#pragma unroll
        for (int k = 0; k < QPT; k++) pair[k] = valid[k] ? pb.seeds[s0 + tid + k * T] : 0;
        have_seed = true;
    }

    bool sorted_matches = false;

    // current transform in registers (uniform)
    Xf<D> X;
#pragma unroll
    for (int e = 0; e < XN; e++) X.m[e] = L.bc[e];

    double state_cost = 0.0;
    if (do_setup) {
        // ---- method setup(): initial matches (2D) ----
        if (D == 2 && !empty) {
            double q[QPT][D];
            if (METHOD == ICPX_VANILLA) {
                // vanilla.cpp:80 — the query is the UNTRANSFORMED source point
#pragma unroll
                for (int k = 0; k < QPT; k++)
#pragma unroll
                    for (int d = 0; d < D; d++) q[k][d] = a[k][d];
            } else {
                XfRows<D> xr = xf_rows<D>(X);
#pragma unroll
                for (int k = 0; k < QPT; k++) xf_apply<D>(xr, a[k], q[k]);
            }
            double best[QPT];
            if (filter) nn_twomin<D, QPT>(q, L, m, m4, F, best, pair);
            else nn_lds<D, QPT>(q, L, m4, best, pair);
            have_seed = filter;
#pragma unroll
            for (int k = 0; k < QPT; k++) {
                if (METHOD == ICPX_VANILLA) cost[k] = __builtin_sqrt(__builtin_sqrt(best[k]));  // vanilla.cpp:84 on kdtree.h:126
                else cost[k] = best[k];                                                        // trimmed.cpp:104
                if (!valid[k]) cost[k] = 0.0;
            }
        }
        // ICP::calculate_cost, icp.h:149-155
        double tot[1];
        block_sum<1, QPT>([&](int k, double (&v)[1]) { v[0] = (D == 2 && valid[k]) ? cost[k] : 0.0; },
                          n, T, L.gs, tot);
        state_cost = __builtin_sqrt(tot[0] / static_cast<double>(n));
        if (tid == 0) st->cost = state_cost;
    } else {
        state_cost = st->cost;
        if (D == 2 && METHOD == ICPX_VANILLA && pb.m_pair) {
            // resumed launch: matches never change (vanilla.cpp:80), reload them
#pragma unroll
            for (int k = 0; k < QPT; k++)
                if (valid[k]) {
                    pair[k] = static_cast<int>(pb.m_pair[s0 + tid + k * T]);
                    cost[k] = pb.m_cost[s0 + tid + k * T];
                }
        }
    }

    // ---- ICPDriver::converge loop, driver.h:63-69 ----
    uint64_t steps = 0;
    int exit_code = 0;
    while (true) {
        if (tid == 0) {
            int stop = 0;
            if (pb.check) {
                double el = pb.host_elapsed_s +
                            static_cast<double>(wall_clock64() - st->start_clock) * pb.clock_to_s;
                if (should_terminate<D>(pb.crit, *st, el)) {
                    stop = 1;
                    st->done = 1;
                }
            }
            if (!stop && steps >= max_steps) stop = 2;
            if (!stop) {
                // last_state = state
                for (int e = 0; e < XN; e++) st->last_transform[e] = st->transform[e];
                st->last_cost = st->cost;
                st->have_last = 1;
            }
            reinterpret_cast<int*>(L.bc + 40)[0] = stop;
        }
        __syncthreads();
        int stop = reinterpret_cast<int*>(L.bc + 40)[0];
        __syncthreads();
        exit_code = stop;
        if (stop) break;
        steps++;

        if (!empty) {
            // ================= iterate() =================
            double cur[QPT][D];
            int sel[QPT];
            int k_sel = n;
            if (D == 2) {
                XfRows<D> xr = xf_rows<D>(X);
#pragma unroll
                for (int k = 0; k < QPT; k++) xf_apply<D>(xr, a[k], cur[k]);  // a_current = transform * a
                if (METHOD == ICPX_TRIMMED) {
                    double best[QPT];
                    if (have_seed) nn_filtered<D, QPT>(cur, L, m, m4, F, best, pair);
                    else if (filter) nn_twomin<D, QPT>(cur, L, m, m4, F, best, pair);
                    else nn_lds<D, QPT>(cur, L, m4, best, pair);
                    have_seed = filter;
#pragma unroll
                    for (int k = 0; k < QPT; k++) cost[k] = valid[k] ? best[k] : 0.0;
                    // trimming step, trimmed.cpp:55-58: in the order (cost, point) keep the first new_n.
                    // Only the new_n-th pair is needed to mark them (radix_select); the sorted order
                    // the reference leaves in `matches` is produced once, at publish time.
                    sorted_matches = true;
                    long long new_n = static_cast<long long>(pb.overlap_rate * static_cast<double>(n));
                    if (new_n < 1) new_n = 1;
                    if (new_n > n) new_n = n;
                    k_sel = static_cast<int>(new_n);
                    uint64_t skeys[QPT];
#pragma unroll
                    for (int k = 0; k < QPT; k++) skeys[k] = static_cast<uint64_t>(__double_as_longlong(cost[k]));
                    uint64_t kth;
                    uint32_t vth;
                    radix_select<QPT>(skeys, valid, T, k_sel, reinterpret_cast<int*>(L.gs), kth, vth);
#pragma unroll
                    for (int k = 0; k < QPT; k++) {
                        uint64_t kk = static_cast<uint64_t>(__double_as_longlong(cost[k]));
                        uint32_t vv = static_cast<uint32_t>(tid + k * T);
                        sel[k] = valid[k] && (kk < kth || (kk == kth && vv <= vth));
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < QPT; k++) sel[k] = valid[k];
                }
            } else {
                // 3D: nearest_neighbor(c), vanilla_3d.cpp:30-45
#pragma unroll
                for (int k = 0; k < QPT; k++)
#pragma unroll
                    for (int d = 0; d < D; d++) cur[k][d] = a[k][d];
                double best[QPT];
                if (have_seed) nn_filtered<D, QPT>(cur, L, m, m4, F, best, pair);
                else if (filter) nn_twomin<D, QPT>(cur, L, m, m4, F, best, pair);
                else nn_lds<D, QPT>(cur, L, m4, best, pair);
                have_seed = filter;
#pragma unroll
                for (int k = 0; k < QPT; k++) {
                    cost[k] = __builtin_sqrt(__builtin_sqrt(best[k]));  // vanilla_3d.cpp:41 on kdtree.h:126
                    sel[k] = valid[k] && (METHOD == ICPX_VANILLA || cost[k] <= pb.max_distance);  // trimmed_3d.cpp:98
                }
                if (METHOD == ICPX_TRIMMED) k_sel = block_count<QPT>(sel, T, L.gs);
            }

            // matched target coordinates
            double bp[QPT][D];
#pragma unroll
            for (int k = 0; k < QPT; k++)
#pragma unroll
                for (int d = 0; d < D; d++) bp[k][d] = sel[k] ? L.t[d][pair[k]] : 0.0;

            // pass 1: centroids (+ cost sum for 2D)
            constexpr int NV1 = 2 * D + 1;
            double s1[NV1];
            block_sum<NV1, QPT>(
                [&](int k, double (&v)[NV1]) {
#pragma unroll
                    for (int d = 0; d < D; d++) {
                        v[d] = sel[k] ? cur[k][d] : 0.0;
                        v[D + d] = bp[k][d];
                    }
                    v[2 * D] = (D == 2 && valid[k]) ? cost[k] : 0.0;
                },
                n, T, L.gs, s1);
            double ca[D], cb[D];
            const double kd = static_cast<double>(k_sel);
#pragma unroll
            for (int d = 0; d < D; d++) {
                ca[d] = s1[d] / kd;
                cb[d] = s1[D + d] / kd;
            }

            // pass 2: cross-covariance N = sum (a - ca)(b - cb)^T
            double s2[D * D];
            block_sum<D * D, QPT>(
                [&](int k, double (&v)[D * D]) {
#pragma unroll
                    for (int r = 0; r < D; r++)
#pragma unroll
                        for (int c = 0; c < D; c++)
                            v[c * D + r] = sel[k] ? (cur[k][r] - ca[r]) * (bp[k][c] - cb[c]) : 0.0;
                },
                n, T, L.gs, s2);

            if (tid == 0) {
                Mat<D> N;
#pragma unroll
                for (int e = 0; e < D * D; e++) N.a[e] = s2[e];
                Xf<D> step = kabsch<D>(N, ca, cb);
                Xf<D> nx = xf_mul<D>(step, X);
                for (int e = 0; e < XN; e++) {
                    L.bc[e] = nx.m[e];
                    L.bc[16 + e] = step.m[e];
                    st->transform[e] = nx.m[e];
                }
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < XN; e++) X.m[e] = L.bc[e];
            if (D == 3) {
                // c = T * c, vanilla_3d.cpp:91
                Xf<D> S;
#pragma unroll
                for (int e = 0; e < XN; e++) S.m[e] = L.bc[16 + e];
                XfRows<D> sr = xf_rows<D>(S);
#pragma unroll
                for (int k = 0; k < QPT; k++) xf_apply<D>(sr, cur[k], a[k]);
            }
            if (D == 2 && METHOD == ICPX_TRIMMED) state_cost = __builtin_sqrt(s1[2 * D] / static_cast<double>(n));
            __syncthreads();
        }
        if (tid == 0) {
            st->iteration_count += 1;
            st->cost = state_cost;
        }
    }

    // ---- publish ----
    if (filter && pb.seeds) {
#pragma unroll
        for (int k = 0; k < QPT; k++)
            if (valid[k]) pb.seeds[s0 + tid + k * T] = pair[k];
    }
    if (D == 3 && pb.cur) {
#pragma unroll
        for (int k = 0; k < QPT; k++)
            if (valid[k])
#pragma unroll
                for (int d = 0; d < D; d++) pb.cur[(s0 + tid + k * T) * D + d] = a[k][d];
    }
    if (pb.m_point) {
        if (D == 3) {
            // Neighbors of the last iterate (debug view); ICP3 matches stay zero (Q4)
#pragma unroll
            for (int k = 0; k < QPT; k++)
                if (valid[k]) {
                    pb.m_pair[s0 + tid + k * T] = pair[k];
                    pb.m_cost[s0 + tid + k * T] = cost[k];
                    pb.m_point[s0 + tid + k * T] = tid + k * T;
                }
        } else if (!sorted_matches) {
#pragma unroll
            for (int k = 0; k < QPT; k++)
                if (valid[k]) {
                    pb.m_point[s0 + tid + k * T] = tid + k * T;
                    pb.m_pair[s0 + tid + k * T] = pair[k];
                    pb.m_cost[s0 + tid + k * T] = cost[k];
                }
        } else {
            // Trimmed leaves matches sorted by cost (trimmed.cpp:55-56): sort (cost, point) now
            const int npad = pb.lds_sort_cap;
            for (int i = tid; i < npad; i += T) {
                L.skey[i] = ~0ULL;
                L.sval[i] = static_cast<uint32_t>(i);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < QPT; k++)
                if (valid[k]) {
                    L.skey[tid + k * T] = static_cast<uint64_t>(__double_as_longlong(cost[k]));
                    pb.pair_tmp[s0 + tid + k * T] = pair[k];
                }
            __syncthreads();
            bitonic_sort(L.skey, L.sval, npad, T);
            for (int r = tid; r < n; r += T) {
                uint32_t pt = L.sval[r];
                pb.m_point[s0 + r] = pt;
                pb.m_pair[s0 + r] = pb.pair_tmp[s0 + pt];
                pb.m_cost[s0 + r] = __longlong_as_double(static_cast<long long>(L.skey[r]));
            }
        }
    }
    if (tid == 0 && pb.out_states) {
        icpx_state* o = pb.out_states + p;
        o->cost = st->cost;
        o->iteration_count = st->iteration_count;
        for (int e = 0; e < 16; e++) o->transform[e] = e < XN ? st->transform[e] : 0.0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's stores are out before the hand-off
    __syncthreads();                                   // and LDS may be restaged by the next task
    return exit_code;
}

template <int D, int METHOD, int QPT, int TMAX>
__global__ void __launch_bounds__(TMAX) pair_kernel(const PairBatch pb) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    if (pb.sched_state == nullptr) {
        pair_body<D, METHOD, QPT>(pb, blockIdx.x, pb.do_setup, pb.max_steps, lds_raw);
        return;
    }
    // ---- persistent workgroup: iteration-granular tasks ----
    // Hand-off of a pair between workgroups (possibly on different XCDs) follows the agent-scope
    // release / acquire recipe: every wave drains its stores and meets at the barrier (end of
    // pair_body), lane 0 releases, then publishes the state word; the taker wins the word with a
    // CAS, acquires, and only then lets the workgroup load.
    __shared__ int s_key, s_pick;
    const int tid = threadIdx.x;
    const int P = pb.n_pairs;
    const int home = static_cast<int>(static_cast<long long>(blockIdx.x) * P / gridDim.x);
    for (int guard = 0; guard < (1 << 30); guard++) {
        if (tid == 0) s_key = 0x7fffffff;
        __syncthreads();
        int best = 0x7fffffff;
        for (int q = tid; q < P; q += blockDim.x) {
            int w = __hip_atomic_load(&pb.sched_state[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int st = w & 3;
            if (st <= 1) {
                // fewest iterations first (longest expected remaining chain); among equals the pair
                // nearest after this workgroup's home index, so that workgroups that look at the
                // same moment go for different pairs instead of all racing for the lowest one
                int it = w >> 2;
                int rot = q - home;
                rot += rot < 0 ? P : 0;
                int key = ((it < 2047 ? it : 2047) << 20) | rot;
                best = key < best ? key : best;
            }
        }
        if (best != 0x7fffffff) atomicMin(&s_key, best);
        __syncthreads();
        if (tid == 0) {
            int pick = -2;  // retry
            if (s_key != 0x7fffffff) {
                int q = (s_key & 0xfffff) + home;
                q -= q >= P ? P : 0;
                int w = __hip_atomic_load(&pb.sched_state[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((w & 3) <= 1 && atomicCAS(&pb.sched_state[q], w, (w & ~3) | 2) == w) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    pick = (q << 1) | ((w & 3) == 0 ? 1 : 0);  // bit 0: fresh -> run setup
                }
            } else if (__hip_atomic_load(pb.sched_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= P) {
                pick = -1;  // everything finished
            } else {
                __builtin_amdgcn_s_sleep(64);
            }
            s_pick = pick;
        }
        __syncthreads();
        const int pick = s_pick;
        if (pick == -1) break;
        if (pick == -2) continue;
        const int q = pick >> 1;
        const int code = pair_body<D, METHOD, QPT>(pb, q, pick & 1, static_cast<uint64_t>(pb.sched_steps), lds_raw);
        // pair_body ended with a barrier behind every wave's stores
        if (tid == 0) {
            const int iters = static_cast<int>(pb.states[q].iteration_count);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&pb.sched_state[q], (iters << 2) | (code == 1 ? 3 : 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            if (code == 1) atomicAdd(pb.sched_done, 1);
        }
    }
}

}  // namespace icpx
