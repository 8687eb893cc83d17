// icpx_pair.h — the fused, device-resident ICP loop for scan-sized clouds: ONE workgroup runs a
// whole ICPDriver::converge() for one scan pair.
//
// Per iteration, without leaving the kernel (reference: lib/icp/impl/{vanilla,trimmed,vanilla_3d,
// trimmed_3d}.cpp iterate(); include/icp/driver.h:51-72,154-222):
//   transform apply  ->  exact NN against the target cloud staged in LDS (SoA; each lane owns up
//   to QPT queries in registers): f32 candidate filter + exact f64 refinement over all targets
//   (icpx_filter.h), or — PRUNE kernels, batches — a walk of the cells of a per-pair grid that can
//   hold a closer target (icpx_prune.h)  ->  selection (2D trimmed: radix select of the
//   overlap_rate * n-th (cost, point) key; 3D trimmed: distance threshold)  ->  centroid and
//   cross-covariance sums in the TREE64 order (wave butterfly, then one butterfly over the <= 64
//   group sums)  ->  single-lane Jacobi SVD / Kabsch  ->  transform composition, cost,
//   should_terminate.
//
// Limits of this kernel: n <= 4096 source points (<= 64 TREE64 groups) and a target cloud that
// fits LDS (checked on the host, icpx_capi.hip).  Larger clouds take the multi-kernel path.
#pragma once

#ifndef ICPX_SELECT_BITS
#define ICPX_SELECT_BITS 10
#endif
#ifndef ICPX_3D_SMALL_WAVES
#define ICPX_3D_SMALL_WAVES 3
#endif

#include "icpx_filter.h"
#include "icpx_kd_lds.h"
#include "icpx_math.h"
#include "icpx_prune.h"

namespace icpx {

constexpr int kPairMaxN = 4096;
constexpr size_t kPairMaxLds = 160 * 1024 - 256;  // dynamic LDS budget (the kernel keeps a few static words)

// launch geometry of the fused kernel, chosen by the host (icpx_capi.hip: fused_fits)
struct PairLaunch {
    int qpt, threads;
    size_t lds;
    int cus;      // compute units of the device
};
constexpr int kGsStride = 64;   // level-1 group sums per value
constexpr int kGsValues = 10;   // up to 9 covariance terms (+ spare)
constexpr int kBcDoubles = 96;  // broadcast area: transform, step, flags; [48, 86): the visit's PairState (pruned kernels)

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
    int32_t coop;              // PRUNE kernels: wave-cooperative search (icpx_prune.h) instead of the per-lane walk
    int32_t* seeds;            // n_total: previous nearest neighbour by point, carried between launches / tasks; may be nullptr
    // grid-pruned search (icpx_prune.h; PRUNE kernels): per target cloud its order by grid cell
    // (pair_perm_kernel); `seeds` then holds LDS slots (that order) instead of cloud indices
    uint16_t* tperm;           // m_total: target at LDS slot j of its pair
    PairGrid* grids;           // n_pairs: grid of every target cloud
    uint16_t* cstart;          // n_pairs x kGridTable: cell start tables
    float* hcert;              // m_total: certificate radius of the target at LDS slot j of its pair; null = none
    int32_t prepare;           // the setup visit of a pair computes the four above itself (prepare_target_grid);
                               // 0: pair_perm_kernel has run before this launch
    // the kd-tree tie rule (icpx_kd.h): per target cloud its tree in in-order layout, built on a pair's first tie
    uint16_t* kdnode;          // m_total
    uint16_t* kdpos;           // m_total
    int32_t* kd_paused;        // one word: pairs that a kernel without the rule's resolution paused at a tie
                               // (PairState::kd_flags bit 1); the KD kernel launched behind it takes them over
    int32_t kd_takeover;       // this launch is that KD kernel: only paused pairs run, from where they stopped
                               // (and pairs deferred by the pruned kernel, bit 2 of kd_flags: from the beginning)
    int32_t defer_sparse;      // pruned kernels: hand pairs whose grid is a poor index (PairGrid::bad bit 1) to the kernel launched behind
    int32_t lazy_sort;         // handles, trimmed ICP2: the matches are published in point order and marked (PairState::kd_flags
                               // bit 4); icpx_matches sorts them by (cost, point) when they are asked for (pair_sort_matches_kernel)
    int32_t chunk_skip;        // trimmed ICP2, brute-force kernels: chunk-skipping exact search (nn_lds_chunked, icpx_filter.h);
                               // the launch reserved pair_lds_bytes_chunk (the boxes take the f32 copy's place)
};

inline __host__ __device__ size_t pair_lds_filter_offset(int dim, int tcap, int scap) {
    size_t b = sizeof(double) * (static_cast<size_t>(dim) * tcap + kGsValues * kGsStride + kBcDoubles) +
               static_cast<size_t>(scap) * 12;
    return (b + 15) & ~static_cast<size_t>(15);
}

template <int D>
ICPX_DEV LdsView<D> carve(double* base, int tcap, int scap, int filter) {
    LdsView<D> v;
    // group sums and broadcast area first: everything behind them is the target cloud and search scratch, which the
    // tree build of the kd rule (kd_build_lds) may take over as one piece (the cloud is restaged afterwards)
    double* p = base;
    v.gs = p;
    p += kGsValues * kGsStride;
    v.bc = p;
    p += kBcDoubles;
    for (int d = 0; d < D; d++) {
        v.t[d] = p;
        p += tcap;
    }
    if (D == 2) v.t[2] = nullptr;
    v.skey = reinterpret_cast<uint64_t*>(p);
    v.sval = reinterpret_cast<uint32_t*>(v.skey + scap);
    float* f = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + pair_lds_filter_offset(D, tcap, scap));
    for (int d = 0; d < 3; d++) v.tf[d] = (filter && d < D) ? f + static_cast<size_t>(d) * tcap : nullptr;
    v.j0 = 0;
    return v;
}

// prune (grid-pruned search, icpx_prune.h): no f32 copy; the cell start table takes its place, followed by
// the per-wave staging of the cooperative search
inline size_t pair_lds_bytes(int dim, int tcap, int scap, bool filter, bool prune = false, int threads = 0) {
    if (prune)
        return pair_lds_filter_offset(dim, tcap, scap) + sizeof(uint16_t) * kGridTable +
               static_cast<size_t>(threads / 64) * coop_wave_bytes(dim);
    return pair_lds_filter_offset(dim, tcap, scap) + (filter ? sizeof(float) * static_cast<size_t>(dim) * tcap : 0);
}
// chunk-skipping search: the chunks' boxes behind the sort area
inline size_t pair_lds_bytes_chunk(int dim, int tcap, int scap) {
    return pair_lds_filter_offset(dim, tcap, scap) + sizeof(double) * 2 * dim * static_cast<size_t>((tcap + kChunk - 1) / kChunk);
}
// KD kernels: at least the tree build's scratch behind the group sums and the broadcast area
inline size_t pair_lds_bytes_kd(int dim, int tcap, int scap, int64_t m) {
    const size_t plain = pair_lds_bytes(dim, tcap, scap, false);
    const size_t build = sizeof(double) * (kGsValues * kGsStride + kBcDoubles) + kd_build_bytes(dim, m);
    return plain > build ? plain : build;
}

// TREE64 sum of NV values per point.  f(k, v) is contribution v of point tid + k*T (exactly +0.0 where the
// point does not exist or is not selected).  Every thread receives the totals.
// Level 1 (per group of 64 consecutive points = one wave at one slot k): the QPT slots of a value go through
// the butterfly TOGETHER — at xor 32, 16 and 8 two slots share one exchange (xchg_sum: the lanes with the bit
// clear carry on with one slot, the others with the other), so a lane ends the three levels with ONE slot's
// partial, k = 4 b3 + 2 b4 + b5 (QPT 8), and the remaining levels run once per value instead of once per
// (value, slot).  Same partners and same additions as a butterfly per (value, slot): same bits.
template <int NV, int QPT, typename F>
ICPX_DEV void block_sum(F&& f, int n, int T, double* gs, double (&out)[NV]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = T >> 6;
    const int G = (n + 63) >> 6;
#pragma unroll
    for (int v = 0; v < NV; v++) {
        double s;
        int k;  // the slot whose group sum this lane ends up with
        if constexpr (QPT == 8) {
            const double y0 = xchg_sum<32>(f(0, v), f(1, v)), y1 = xchg_sum<32>(f(2, v), f(3, v));
            const double y2 = xchg_sum<32>(f(4, v), f(5, v)), y3 = xchg_sum<32>(f(6, v), f(7, v));
            const double z0 = xchg_sum<16>(y0, y1), z1 = xchg_sum<16>(y2, y3);
            s = xchg_sum<8>(z0, z1);
            k = ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 5) & 1);
        } else if constexpr (QPT == 4) {
            const double y0 = xchg_sum<32>(f(0, v), f(1, v)), y1 = xchg_sum<32>(f(2, v), f(3, v));
            s = xchg_sum<16>(y0, y1);
            s = s + dpp_xor<8>(s);
            k = ((lane >> 4) & 1) * 2 + ((lane >> 5) & 1);
        } else if constexpr (QPT == 2) {
            s = xchg_sum<32>(f(0, v), f(1, v));
            s = s + swap_xor<16>(s);
            s = s + dpp_xor<8>(s);
            k = (lane >> 5) & 1;
        } else {
            static_assert(QPT == 1, "block_sum: QPT must be 1, 2, 4 or 8");
            s = f(0, v);
            s = s + swap_xor<32>(s);
            s = s + swap_xor<16>(s);
            s = s + dpp_xor<8>(s);
            k = 0;
        }
        s = s + dpp_xor<4>(s);
        s = s + dpp_xor<2>(s);
        s = s + dpp_xor<1>(s);
        const int g = k * W + wave;
        if ((lane & 7) == 0 && g < G) gs[v * kGsStride + g] = s;
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

// AND / OR over the wave, in scalars (DPP: quads, rows, then the four row results)
ICPX_DEV uint32_t wave_and_u32(uint32_t v) {
    int x = static_cast<int>(v);
    x &= __builtin_amdgcn_update_dpp(-1, x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    x &= __builtin_amdgcn_update_dpp(-1, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    x &= __builtin_amdgcn_update_dpp(-1, x, 0x124, 0xf, 0xf, false);  // row_ror:4
    x &= __builtin_amdgcn_update_dpp(-1, x, 0x128, 0xf, 0xf, false);  // row_ror:8
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(x, 0) & __builtin_amdgcn_readlane(x, 16) &
                                 __builtin_amdgcn_readlane(x, 32) & __builtin_amdgcn_readlane(x, 48));
}
ICPX_DEV uint32_t wave_or_u32(uint32_t v) {
    int x = static_cast<int>(v);
    x |= __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(x, 0) | __builtin_amdgcn_readlane(x, 16) |
                                 __builtin_amdgcn_readlane(x, 32) | __builtin_amdgcn_readlane(x, 48));
}

// The rank-th smallest (1-based) of the (key, point) pairs held QPT per thread, lexicographic order; out: its key
// and its point, so that the pairs <= (kth, vth) are exactly the first `rank` of the sorted sequence.  Replaces
// the full sort in Trimmed::iterate (trimmed.cpp:55-58): only the threshold is needed to mark the kept matches.
//   1. the bits the valid keys do not share (AND / OR over the workgroup): the digits above them are skipped;
//   2. MSB-first radix passes with ICPX_SELECT_BITS-bit digits from the first differing bit down (LDS histogram, wave-0 scan;
//      after the key's bits, the point's), until at most 64 pairs are left in the rank's bucket;
//   3. those pairs go into a list and every wave ranks them directly.
// Positive doubles of one cloud share their leading digits, and 8 bits divide ~1000 pairs into buckets of a few:
// typically two passes.  Workgroup-wide (barriers); hist: >= 2^ICPX_SELECT_BITS + 232 ints of LDS (kGsValues * kGsStride
// doubles = 1280 ints are there); points must be below 65536.
template <int QPT>
ICPX_DEV void radix_select(const uint64_t (&key)[QPT], const bool (&valid)[QPT], int T, int rank, int* hist,
                           uint64_t& kth, uint32_t& vth) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = T >> 6;
    constexpr int kBits = ICPX_SELECT_BITS, kBins = 1 << kBits, kPer = kBins / 64;
    int* res = hist + kBins;                                             // [0] digit, [1] rank inside, [2] bucket size, [4] list length
    uint64_t* lkey = reinterpret_cast<uint64_t*>(hist + kBins + 8);     // 64 keys
    uint32_t* lpt = reinterpret_cast<uint32_t*>(hist + kBins + 136);    // 64 points
    uint32_t* wred = reinterpret_cast<uint32_t*>(hist + kBins + 200);   // per wave: and_lo, and_hi, or_lo, or_hi
    bool act[QPT];
    uint32_t alo = 0xffffffffu, ahi = 0xffffffffu, olo = 0, ohi = 0;
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        act[k] = valid[k];
        const uint32_t lo = static_cast<uint32_t>(key[k]), hi = static_cast<uint32_t>(key[k] >> 32);
        alo &= act[k] ? lo : 0xffffffffu;
        ahi &= act[k] ? hi : 0xffffffffu;
        olo |= act[k] ? lo : 0u;
        ohi |= act[k] ? hi : 0u;
    }
    alo = wave_and_u32(alo);
    ahi = wave_and_u32(ahi);
    olo = wave_or_u32(olo);
    ohi = wave_or_u32(ohi);
    if (lane == 0) {
        wred[wave * 4 + 0] = alo;
        wred[wave * 4 + 1] = ahi;
        wred[wave * 4 + 2] = olo;
        wred[wave * 4 + 3] = ohi;
    }
    for (int i = tid; i < kBins; i += T) hist[i] = 0;
    if (tid == 0) res[4] = 0;
    __syncthreads();
    for (int w = 0; w < W; w++) {
        alo &= wred[w * 4 + 0];
        ahi &= wred[w * 4 + 1];
        olo |= wred[w * 4 + 2];
        ohi |= wred[w * 4 + 3];
    }
    const uint64_t differ = (static_cast<uint64_t>(ahi ^ ohi) << 32) | (alo ^ olo);  // (no valid key: all ones; rank is 0 then)
    // digit windows: `shift` = low bit of the window, in the key while on_key, then in the point
    bool on_key = differ != 0;
    int shift = on_key ? max(63 - __builtin_clzll(differ) - (kBits - 1), 0) : 8;
    int remaining = rank;
    for (int pass = 0; pass < 12; pass++) {
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            const uint32_t point = static_cast<uint32_t>(tid + k * T);
            const int d = on_key ? static_cast<int>((key[k] >> shift) & (kBins - 1)) : static_cast<int>((point >> shift) & 255);
            if (act[k]) atomicAdd(&hist[d], 1);
        }
        __syncthreads();
        if (tid < 64) {
            int c[kPer], s = 0;
#pragma unroll
            for (int u = 0; u < kPer; u++) {
                c[u] = hist[kPer * lane + u];
                hist[kPer * lane + u] = 0;  // for the next pass
                s += c[u];
            }
            const int incl = wave_incl_scan(s);
            int excl = incl - s;
#pragma unroll
            for (int u = 0; u < kPer; u++) {
                if (excl < remaining && remaining <= excl + c[u]) {
                    res[0] = kPer * lane + u;
                    res[1] = remaining - excl;
                    res[2] = c[u];
                }
                excl += c[u];
            }
        }
        __syncthreads();
        const int d = res[0];
        remaining = res[1];
        const int bucket = res[2];
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            const uint32_t point = static_cast<uint32_t>(tid + k * T);
            const int dk = on_key ? static_cast<int>((key[k] >> shift) & (kBins - 1)) : static_cast<int>((point >> shift) & 255);
            act[k] = act[k] && dk == d;
        }
        if (bucket <= 64) break;
        // (points are unique: after the point's two windows the bucket holds one pair)
        if (on_key) {
            if (shift == 0) {
                on_key = false;
                shift = 8;
            } else {
                shift = max(shift - kBits, 0);
            }
        } else {
            shift = 0;
        }
    }
    // ---- the bucket's pairs, ranked directly ----
#pragma unroll
    for (int k = 0; k < QPT; k++)
        if (act[k]) {
            const int pos = atomicAdd(&res[4], 1);
            if (pos < 64) {
                lkey[pos] = key[k];
                lpt[pos] = static_cast<uint32_t>(tid + k * T);
            }
        }
    __syncthreads();
    const int cnt = min(res[4], 64);
    const uint64_t mk = lane < cnt ? lkey[lane] : ~0ull;
    const uint32_t mp = lane < cnt ? lpt[lane] : 0xffffffffu;
    int below = 0;
    for (int j = 0; j < cnt; j++) {
        const uint64_t kj = lkey[j];
        const uint32_t pj = lpt[j];
        below += (kj < mk || (kj == mk && pj < mp)) ? 1 : 0;
    }
    const unsigned long long hit = __ballot(lane < cnt && below == remaining - 1);
    const int src = hit ? __ffsll(static_cast<long long>(hit)) - 1 : 0;
    kth = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mk >> 32), src))) << 32) |
          static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mk & 0xffffffffu), src));
    vth = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mp), src));
    __syncthreads();  // hist may be reused by the caller
}

// Pre-pass of the pruned search for ONE target cloud, by one workgroup: its grid (returned to every thread),
// tperm (order by cell, a counting sort), the cell start table and the certificate radii.  Target order, table
// and grid parameters are what the search's exactness argument rests on (icpx_prune.h).
//   pts / cnt       the cloud (AoS, global), cnt >= 1
//   out             tperm of the cloud (global)
//   table           the cloud's cell start table (global, kGridTable entries); table_lds: a copy in LDS, or null
//   hc              certificate radii by slot (global), or null: none
//   red             64 doubles of LDS scratch;  hist: 2 * (G^D + 4) ints of LDS scratch
//   ts / ts_stride  LDS, D rows of ts_stride doubles: receives the cloud in cell order (SoA), padded with +inf up
//                   to pad_to; null (only when hc is null): not staged
// Called from pair_perm_kernel (separate launch) and from the pair kernels' setup visit (icpx_pair_pruned.h), where
// ts is the kernel's own target area.  Workgroup-wide (barriers); ends with a barrier.
constexpr double kGridSigmas = 3.0;  // (a uniform cloud's box is mean +- 1.73 sigma: unchanged)
// The grid is a poor index of a cloud when sum(cell count^2) / points — the points a query that falls among the targets
// finds in its own cell — passes max(kSparseFloor, points * kSparseNum / kSparseDen): a uniform cloud has 2, the
// reference's lidar scans 32..61 at 660..960 points, a sphere's surface 4, a circle 10 (measured at 1024 points: the
// pruned kernels win 3x at 10 and lose 2.4x at 23; the brute-force kernels' work grows with the point count)
constexpr int kSparseFloor = 8, kSparseNum = 1, kSparseDen = 64;
constexpr int kSparseMinPoints = 128;

template <int D>
ICPX_DEV PairGrid prepare_target_grid(const double* __restrict__ pts, const int cnt, uint16_t* __restrict__ out,
                                      uint16_t* __restrict__ table, uint16_t* table_lds, float* __restrict__ hc, double* red,
                                      int* hist, double* ts, const int ts_stride, const int pad_to) {
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = T >> 6;
    double lo[D], hi[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        lo[d] = __builtin_inf();
        hi[d] = -__builtin_inf();
    }
    bool bad = false;
    // (first and second moments beside the bounding box: the grid covers mean +- kGridSigmas standard deviations when
    // that is inside the box — a scan with a few far returns would otherwise spend its cells on empty space; the
    // points beyond it fall into the border cells, cell_of clamps.  Any box gives the same search results.)
    double s1[D], s2[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        s1[d] = 0.0;
        s2[d] = 0.0;
    }
    for (int i = tid; i < cnt; i += T)
#pragma unroll
        for (int d = 0; d < D; d++) {
            double v = pts[static_cast<int64_t>(i) * D + d];
            if (__builtin_fabs(v) < 1e150) {
                lo[d] = __builtin_fmin(lo[d], v);
                hi[d] = __builtin_fmax(hi[d], v);
                s1[d] += v;
                s2[d] += v * v;
            } else {
                bad = true;
            }
        }
#pragma unroll
    for (int d = 0; d < D; d++) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo[d] = __builtin_fmin(lo[d], __shfl_xor(lo[d], o));
            hi[d] = __builtin_fmax(hi[d], __shfl_xor(hi[d], o));
            s1[d] += __shfl_xor(s1[d], o);
            s2[d] += __shfl_xor(s2[d], o);
        }
        if (lane == 0) {
            red[wave * 8 + d] = lo[d];
            red[wave * 8 + 4 + d] = hi[d];
        }
    }
    const bool wave_bad = __any(bad);
    if (lane == 0) red[wave * 8 + 3] = wave_bad ? 1.0 : 0.0;
    __syncthreads();
    const int G = grid_cells_per_axis(D, cnt);
    double scale[D];
    bool any_bad = false;
    for (int w = 0; w < W; w++) any_bad = any_bad || red[w * 8 + 3] != 0.0;
#pragma unroll
    for (int d = 0; d < D; d++) {
        double l = __builtin_inf(), h = -__builtin_inf();
        for (int w = 0; w < W; w++) {
            l = __builtin_fmin(l, red[w * 8 + d]);
            h = __builtin_fmax(h, red[w * 8 + 4 + d]);
        }
        if (!(l <= h)) {  // no usable coordinate on this axis
            l = 0.0;
            h = 0.0;
        }
        lo[d] = l;
        hi[d] = h;
    }
    // the moments through the same scratch words (everyone has read the box)
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            red[wave * 8 + d] = s1[d];
            red[wave * 8 + 4 + d] = s2[d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < D; d++) {
        double l = lo[d], h = hi[d];
        if (!any_bad && cnt > 1) {
            double a = 0.0, b = 0.0;
            for (int w = 0; w < W; w++) {
                a += red[w * 8 + d];
                b += red[w * 8 + 4 + d];
            }
            const double mean = a / static_cast<double>(cnt);
            const double var = b / static_cast<double>(cnt) - mean * mean;
            const double sd = var > 0.0 ? __builtin_sqrt(var) : 0.0;
            const double l2 = __builtin_fmax(l, mean - kGridSigmas * sd), h2 = __builtin_fmin(h, mean + kGridSigmas * sd);
            if (l2 < h2 && (h2 - l2) < 1e300) {
                l = l2;
                h = h2;
            }
        }
        lo[d] = l;
        scale[d] = h > l ? static_cast<double>(G) / (h - l) : 0.0;
        if (!(scale[d] < 1e300)) scale[d] = 0.0;
    }
    PairGrid g{};
#pragma unroll
    for (int d = 0; d < D; d++) {
        g.lo[d] = lo[d];
        g.scale[d] = scale[d];
    }
    g.G = G;
    g.bad = any_bad ? 1 : 0;
    int ncell = 1;
    for (int d = 0; d < D; d++) ncell *= G;
    int* cursor = hist + ncell + 4;
    // cell id, x fastest — the function the search evaluates for the queries
    auto cell_id = [&](int i) -> int {
        int c = 0;
        for (int d = D - 1; d >= 0; d--) c = c * G + cell_of(pts[static_cast<int64_t>(i) * D + d], lo[d], scale[d], G);
        return c;
    };
    for (int c = tid; c <= ncell + 1; c += T) hist[c] = 0;  // (hist[ncell + 1]: sum of squared cell counts)
    __syncthreads();  // (also: every thread has read `red`)
    for (int i = tid; i < cnt; i += T) atomicAdd(&hist[cell_id(i)], 1);
    __syncthreads();
    // exclusive scan of the cell counts: per-thread chunk, wave scan, wave totals
    const int chunk = (ncell + T - 1) / T;
    int local = 0, occ = 0;
    for (int u = 0; u < chunk; u++) {
        const int c = tid * chunk + u;
        if (c < ncell) {
            local += hist[c];
            occ += hist[c] * hist[c];
        }
    }
    occ = wave_sum_int(occ);
    if (lane == 0 && occ) atomicAdd(&hist[ncell + 1], occ);
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int up = __shfl_up(incl, o);
        incl += lane >= o ? up : 0;
    }
    int* wtot = reinterpret_cast<int*>(red);  // the bounding-box values have been consumed above
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = incl - local;
    for (int w = 0; w < wave; w++) base += wtot[w];
    for (int u = 0; u < chunk; u++) {
        const int c = tid * chunk + u;
        if (c < ncell) {
            cursor[c] = base;
            table[c] = static_cast<uint16_t>(base);
            if (table_lds) table_lds[c] = static_cast<uint16_t>(base);
            base += hist[c];
        }
    }
    if (tid == 0) {
        table[ncell] = static_cast<uint16_t>(cnt);
        if (table_lds) table_lds[ncell] = static_cast<uint16_t>(cnt);
    }
    __syncthreads();
    // a cloud with many points in few cells — a lidar scan: dense returns near the sensor, points on curves, most
    // cells empty — is searched faster by the brute-force filter kernels (a batch of the reference's scan pairs:
    // 5.5x): bit 1 of `bad` tells the pruned kernel to hand the pair over when it can (icpx_pair_pruned.h)
    {
        const long long per_query = static_cast<long long>(cnt) * kSparseNum / kSparseDen;
        const long long limit = per_query > kSparseFloor ? per_query : kSparseFloor;
        if (cnt >= kSparseMinPoints && static_cast<long long>(hist[ncell + 1]) > limit * cnt) g.bad |= 2;
    }
    // the order inside a cell is whatever the atomics give: any order is a valid permutation.  The cloud goes
    // into LDS in that order at once
    for (int i = tid; i < cnt; i += T) {
        const int j = atomicAdd(&cursor[cell_id(i)], 1);
        out[j] = static_cast<uint16_t>(i);
        if (ts) {
#pragma unroll
            for (int d = 0; d < D; d++) ts[d * ts_stride + j] = pts[static_cast<int64_t>(i) * D + d];
        }
    }
    if (ts) {
        for (int j = cnt + tid; j < pad_to; j += T)
#pragma unroll
            for (int d = 0; d < D; d++) ts[d * ts_stride + j] = __builtin_inf();
    }
    __syncthreads();  // cursor[c] = end of cell c now; ts complete
    if (hc == nullptr) return g;
    // ---- certificate radii (icpx_prune.h, "seed certificate"): hc[j] <= distance from the target in slot j to
    //      every other target of the cloud.  Exact distances to the targets of the 2^D cells nearest to it, and
    //      the distance to the faces of that block behind which there are cells (everything outside the block
    //      is at least that far) ----
    double inv_scale[D];  // (a rounding away from 1 / scale: covered by the 1e-6 below)
#pragma unroll
    for (int d = 0; d < D; d++) inv_scale[d] = scale[d] > 0.0 ? 1.0 / scale[d] : 0.0;
    for (int j = tid; j < cnt; j += T) {
        double t[D], h2 = __builtin_inf(), face = __builtin_inf();
        int b0[D], b1[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            t[d] = ts[d * ts_stride + j];
            double x = (t[d] - lo[d]) * scale[d];  // cell_of's own value
            x = __builtin_fmin(__builtin_fmax(x, 0.0), static_cast<double>(G - 1));
            const int cq = static_cast<int>(x);
            const bool upper = x - static_cast<double>(cq) > 0.5;
            b0[d] = upper ? cq : (cq > 0 ? cq - 1 : 0);
            b1[d] = upper ? (cq < G - 1 ? cq + 1 : G - 1) : cq;
            // a target in a cell beyond the block has fl(u) >= b1 + 1 (or < b0): at least this far along the axis,
            // with 1e-9 cells (+ 1e-9 relative) against the roundings of u
            const double slack = 1e-9 * (__builtin_fabs(x) + 1.0);
            if (scale[d] > 0.0) {
                if (b1[d] < G - 1) face = __builtin_fmin(face, __builtin_fmax(static_cast<double>(b1[d] + 1) - x - slack, 0.0) * inv_scale[d]);
                if (b0[d] > 0) face = __builtin_fmin(face, __builtin_fmax(x - static_cast<double>(b0[d]) - slack, 0.0) * inv_scale[d]);
            }
        }
        // the block's (up to 2 x 2) rows as ONE sequence of slots: all row bounds first, then two slots per step
        // (row by row, every row is a chain of LDS round trips that the whole wave waits for)
        constexpr int NRC = D == 3 ? 4 : 2;
        int pre[NRC], off[NRC], total = 0;
#pragma unroll
        for (int r = 0; r < NRC; r++) {
            const int cy = b0[1] + (r & 1), cz = D == 3 ? b0[D - 1] + (r >> 1) : 0;
            const bool in = cy <= b1[1] && (D == 2 || cz <= b1[D - 1]);
            const int row = in ? (cz * G + cy) * G : 0;
            const int s = cursor[row + b0[0]] - hist[row + b0[0]], e = cursor[row + b1[0]];
            pre[r] = total;
            off[r] = s - total;
            total += in ? e - s : 0;
        }
        for (int r0 = 0; r0 < total; r0 += 2) {
            int jj[2];
            double acc[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int r = r0 + u < total ? r0 + u : total - 1;
                int o = off[0];
#pragma unroll
                for (int q = 1; q < NRC; q++) o = r >= pre[q] ? off[q] : o;
                jj[u] = r + o;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                acc[u] = 0.0;
#pragma unroll
                for (int d = 0; d < D; d++) {
                    const double df = ts[d * ts_stride + jj[u]] - t[d];
                    acc[u] = acc[u] + df * df;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (jj[u] != j) h2 = __builtin_fmin(h2, acc[u]);
        }
        double h = __builtin_fmin(__builtin_sqrt(h2), face) * (1.0 - 1e-6);  // below the true value, f32 rounding included
        if (!(h > 1e-30) || any_bad) h = 0.0;                                 // duplicates, tiny or unusable: never certify
        hc[j] = h < 1e30 ? static_cast<float>(h) : 1e30f;
    }
    __syncthreads();  // the scratch areas may be reused
    return g;
}

// LDS scratch of prepare_target_grid beside the staged cloud: histogram and cursors of the largest grid
inline __host__ __device__ size_t prepare_hist_bytes(int dim, int64_t m_max) {
    const int G = grid_cells_per_axis(dim, static_cast<int>(m_max > 0 ? (m_max < 65536 ? m_max : 65535) : 1));
    int ncell = 1;
    for (int d = 0; d < dim; d++) ncell *= G;
    return 2 * sizeof(int) * static_cast<size_t>(ncell + 4);
}

// The pre-pass as a launch of its own, one workgroup per target cloud of a batch (handles; batches whose
// workgroups cannot hold the scratch).  Dynamic LDS: pair_perm_lds_bytes(D, largest cloud; 0 = no certificates).
inline size_t pair_perm_lds_bytes(int dim, int64_t m_max) {
    return 512 + 8 * static_cast<size_t>(kGridMaxCells + 4) + sizeof(double) * dim * static_cast<size_t>(m_max > 0 ? m_max : 1);
}

template <int D>
__global__ void __launch_bounds__(512)
pair_perm_kernel(const double* __restrict__ dst, const int64_t* __restrict__ dst_off, uint16_t* __restrict__ tperm,
                 PairGrid* __restrict__ grids, uint16_t* __restrict__ cstart, float* __restrict__ hcert) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    const int64_t o0 = dst_off[blockIdx.x];
    const int cnt = static_cast<int>(dst_off[blockIdx.x + 1] - o0);
    const int T = blockDim.x, tid = threadIdx.x;
    uint16_t* __restrict__ table = cstart + static_cast<int64_t>(blockIdx.x) * kGridTable;
    if (cnt <= 0) {
        if (tid == 0) {
            PairGrid g{};
            g.G = 1;
            grids[blockIdx.x] = g;
        }
        for (int c = tid; c < kGridTable; c += T) table[c] = 0;
        return;
    }
    // LDS: 64 doubles of reduction scratch, the counting sort's histogram and cursors, the cloud in cell order
    double* red = lds_raw;
    int* hist = reinterpret_cast<int*>(red + 64);
    double* ts = hcert ? reinterpret_cast<double*>(hist + 2 * (kGridMaxCells + 4)) : nullptr;
    const PairGrid g = prepare_target_grid<D>(dst + o0 * D, cnt, tperm + o0, table, nullptr, hcert ? hcert + o0 : nullptr, red,
                                              hist, ts, cnt, cnt);
    if (tid == 0) grids[blockIdx.x] = g;
}

// Correspondence step of the pruned mode: the wave searches its queries one slot (64 queries) at a time.
// pair[k]: in = seed slot (seeded), out = LDS slot of the nearest target; best[k] = its d2.
// The slot loop is a RUNTIME loop over register vectors (the slot index is wave-uniform, so q[k] is a
// v_movrel, not scratch memory): one copy of the search code per kernel instead of one per slot — the
// unrolled form was several hundred KB of instructions.
// Per-slot registers of a lane.  The slot loops that must exist once per kernel (the search) run at runtime
// over slot 0 and ROTATE the arrays by one slot per trip (a handful of moves), so every index stays a
// compile-time constant: no scratch memory, no register tuples for movrel-indexing.
template <int N>
struct SlotVec {
    double d[N];
    ICPX_DEV void rotate() {
        const double t = d[0];
#pragma unroll
        for (int i = 0; i + 1 < N; i++) d[i] = d[i + 1];
        d[N - 1] = t;
    }
};
template <int N>
struct SlotVecI {
    int d[N];
    ICPX_DEV void rotate() {
        const int t = d[0];
#pragma unroll
        for (int i = 0; i + 1 < N; i++) d[i] = d[i + 1];
        d[N - 1] = t;
    }
};

// OR over the workgroup through one LDS word that thread 0 cleared at least one barrier ago (the pair bodies clear
// it at the top of every trip of the converge loop).  (HIP's own barrier-with-OR reserves 256 bytes of static LDS.)
ICPX_DEV bool block_any(const bool v, int* word) {
    if (v) *word = 1;
    __syncthreads();
    return *word != 0;
}

// One visit of pair p: optional begin()/setup(), then up to max_steps iterations of the converge
// loop.  Returns 1 when should_terminate fired (pair finished), 2 when it paused at max_steps.
// (The brute-force body: handles and small clouds.  Batches run pair_body_pruned, icpx_pair_pruned.h.)
// The kd-tree methods (vanilla 2D, both 3D) follow the reference tree's tie rule (icpx_kd.h):
//   KD kernels  search with the plain f64 scan (nn_lds<KD>), build the pair's tree on its first flagged query and
//               resolve flagged queries in place: every 2D vanilla kernel (it searches once per begin), and the 3D
//               take-over kernel;
//   3D kernels without it only detect: a pair with a flagged query stops BEFORE that iteration changes anything
//               (PairState::kd_flags bit 1, exit as if finished) and the KD kernel launched behind takes it over.
template <int D, int METHOD, int QPT, bool KD>
ICPX_DEV int pair_body(const PairBatch& pb, const int p, const int do_setup, const uint64_t max_steps,
                       double* lds_raw) {
    constexpr bool kKdRule = D == 3 || METHOD == ICPX_VANILLA;  // methods whose neighbour comes from KdTree::search
    static_assert(!KD || kKdRule, "KD kernels exist for the kd-tree methods only");
    constexpr bool TIES = kKdRule && !KD;                        // detect only
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
#ifdef ICPX_STATS
    uint64_t vs0 = wall_clock64();
#define ICPX_VSTAMP(slot_)                                                                             \
    do {                                                                                               \
        __syncthreads();                                                                               \
        uint64_t vs1 = wall_clock64();                                                                 \
        if (threadIdx.x == 0) atomicAdd(&g_icpx_stats[slot_], static_cast<unsigned long long>(vs1 - vs0)); \
        vs0 = vs1;                                                                                     \
    } while (0)
#else
#define ICPX_VSTAMP(slot_) do { } while (0)
#endif
    constexpr bool kChunkable = D == 2 && METHOD == ICPX_TRIMMED && !KD;
    const bool chunked = kChunkable && pb.chunk_skip != 0;
    const bool filter = !KD && !chunked && pb.use_filter != 0 && !(D == 2 && METHOD == ICPX_VANILLA);
    LdsView<D> L = carve<D>(lds_raw, pb.lds_target_cap, pb.lds_sort_cap, filter);
    L.cloud = dst;
    const bool empty = (n == 0 || m == 0);

    // ---- stage the target cloud into LDS as SoA (+ f32 copy about the bounding-box centre) ----
    const int m4 = (m + 3) & ~3;
    const FilterFrame F = stage_targets<D>(dst, m, m4, T, L, filter);
    double* const cbox = reinterpret_cast<double*>(reinterpret_cast<char*>(lds_raw) +
                                                   pair_lds_filter_offset(D, pb.lds_target_cap, pb.lds_sort_cap));
    if constexpr (kChunkable) {
        if (chunked) chunk_boxes<D>(L, m, cbox);
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
            st->kd_flags = 0;
        }
        for (int e = 0; e < XN; e++) L.bc[e] = st->transform[e];
        reinterpret_cast<int*>(L.bc + 41)[0] = do_setup ? 0 : static_cast<int>(st->kd_flags);
        reinterpret_cast<int*>(L.bc + 42)[0] = 0;  // block_any's word
    }
    __syncthreads();
    int* const any_word = reinterpret_cast<int*>(L.bc + 42);
    // kd rule: is the pair's tree built (bit 0)?  A take-over visit (bit 1 was set by the kernel that stopped at a
    // tie) resumes INSIDE the loop: that kernel has already evaluated should_terminate and copied state -> last_state
    bool kd_ready = (reinterpret_cast<int*>(L.bc + 41)[0] & 1) != 0;
    bool kd_resume = KD && pb.kd_takeover != 0 && !do_setup;
    // KD kernels: flagged queries of one search are resolved here, by the whole workgroup (barriers)
    auto kd_fix = [&](const double (&q)[QPT][D], const double (&best)[QPT], int (&pair_)[QPT], const int (&tie)[QPT],
                      const bool (&valid_)[QPT]) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < QPT; k++) any = any || (tie[k] != 0 && valid_[k]);
        if (!block_any(any, any_word)) return;
        if (!kd_ready) {
            kd_build_lds<D, uint16_t>(dst, m, reinterpret_cast<char*>(L.t[0]), pb.kdnode + d0, pb.kdpos + d0);
            (void)stage_targets<D>(dst, m, (m + 3) & ~3, T, L, false);  // the build used the cloud's place in LDS
            kd_ready = true;
            if (tid == 0) st->kd_flags |= 1;
        }
        KdView<uint16_t> kd;
        kd.node = pb.kdnode + d0;
        kd.pos = pb.kdpos + d0;
        kd.pts = dst;
        kd.m = m;
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            int sl = pair_[k];
            kd_resolve_wave<D, uint16_t>(tie[k] != 0 && valid_[k], q[k], best[k], L, m, nullptr, kd, sl);
            pair_[k] = sl;
        }
    };
    // seeds for the filtered search: the previous nearest neighbours.  Trimmed ICP2 has them from
    // setup on; ICP3 after its first iterate; a resumed launch reloads them.
    bool have_seed = false;
    if ((filter || chunked) && !do_setup && pb.seeds && !empty && (D == 2 || st->iteration_count > 0)) {
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
    ICPX_VSTAMP(12);  // staging, boxes, state
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
            if constexpr (KD) {
                int tie[QPT];
                nn_lds<D, QPT, true>(q, L, m4, best, pair, tie);
                kd_fix(q, best, pair, tie, valid);
            } else {
                bool searched = false;
                if constexpr (kChunkable) {
                    if (chunked) {
                        double bound[QPT];
#pragma unroll
                        for (int k = 0; k < QPT; k++) bound[k] = __builtin_inf();
                        nn_lds_chunked<D, QPT>(q, valid, L, m, m4, cbox, bound, best, pair);
                        searched = true;
                    }
                }
                if (searched) {
                } else if (filter) nn_twomin<D, QPT>(q, L, m4, F, best, pair);
                else nn_lds<D, QPT>(q, L, m4, best, pair);
            }
            have_seed = filter || chunked;
#pragma unroll
            for (int k = 0; k < QPT; k++) {
                if (METHOD == ICPX_VANILLA) cost[k] = __builtin_sqrt(__builtin_sqrt(best[k]));  // vanilla.cpp:84 on kdtree.h:126
                else cost[k] = best[k];                                                        // trimmed.cpp:104
                if (!valid[k]) cost[k] = 0.0;
            }
        }
        // ICP::calculate_cost, icp.h:149-155
        double tot[1];
        block_sum<1, QPT>([&](int k, int) -> double { return (D == 2 && valid[k]) ? cost[k] : 0.0; },
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

    ICPX_VSTAMP(13);  // setup search + cost
    // ---- ICPDriver::converge loop, driver.h:63-69 ----
    uint64_t steps = 0;
    int exit_code = 0;
    while (true) {
        if (tid == 0) {
            int stop = 0;
            if (pb.check && !kd_resume) {
                double el = pb.host_elapsed_s +
                            static_cast<double>(wall_clock64() - st->start_clock) * pb.clock_to_s;
                if (should_terminate<D>(pb.crit, *st, el)) {
                    stop = 1;
                    st->done = 1;
                }
            }
            if (!stop && steps >= max_steps) stop = 2;
            if (!stop && !kd_resume) {
                // last_state = state
                for (int e = 0; e < XN; e++) st->last_transform[e] = st->transform[e];
                st->last_cost = st->cost;
                st->have_last = 1;
            }
            reinterpret_cast<int*>(L.bc + 40)[0] = stop;
            *any_word = 0;
        }
        __syncthreads();
        int stop = reinterpret_cast<int*>(L.bc + 40)[0];
        __syncthreads();
        exit_code = stop;
        if (stop) break;
        steps++;
        kd_resume = false;

        if (!empty) {
            // ================= iterate() =================
#ifdef ICPX_STATS
            uint64_t ph0 = wall_clock64();
#define ICPX_PHASE(slot_)                                                                              \
    do {                                                                                               \
        __syncthreads(); /* stats builds only: phases end when the slowest wave gets there */         \
        uint64_t ph1 = wall_clock64();                                                                 \
        if (tid == 0) atomicAdd(&g_icpx_stats[slot_], static_cast<unsigned long long>(ph1 - ph0));     \
        ph0 = ph1;                                                                                     \
    } while (0)
#else
#define ICPX_PHASE(slot_) do { } while (0)
#endif
            double cur[QPT][D];
            int sel[QPT];
            int k_sel = n;
            if (D == 2) {
                XfRows<D> xr = xf_rows<D>(X);
#pragma unroll
                for (int k = 0; k < QPT; k++) xf_apply<D>(xr, a[k], cur[k]);  // a_current = transform * a
                if (METHOD == ICPX_TRIMMED) {
                    double best[QPT];
                    bool searched = false;
                    if constexpr (kChunkable) {
                        if (chunked) {
                            double bound[QPT];
                            if (have_seed) {
                                seed_bound<D, QPT>(cur, L, m, bound, pair);  // (the seed only bounds: the scan finds it again)
                            } else {
#pragma unroll
                                for (int k = 0; k < QPT; k++) bound[k] = __builtin_inf();
                            }
                            nn_lds_chunked<D, QPT>(cur, valid, L, m, m4, cbox, bound, best, pair);
                            searched = true;
                        }
                    }
                    if (searched) {
                    } else if (have_seed) {
                        seed_bound<D, QPT>(cur, L, m, best, pair);
                        nn_filtered<D, QPT>(cur, L, m4, F, best, pair);
                    } else if (filter) nn_twomin<D, QPT>(cur, L, m4, F, best, pair);
                    else nn_lds<D, QPT>(cur, L, m4, best, pair);
                    have_seed = filter || chunked;
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
                    ICPX_PHASE(8);   // correspondence search
                    radix_select<QPT>(skeys, valid, T, k_sel, reinterpret_cast<int*>(L.gs), kth, vth);
                    ICPX_PHASE(9);   // selection
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
                double best[QPT];
#pragma unroll
                for (int k = 0; k < QPT; k++)
#pragma unroll
                    for (int d = 0; d < D; d++) cur[k][d] = a[k][d];
                int tie[QPT];
#pragma unroll
                for (int k = 0; k < QPT; k++) tie[k] = 0;
                if constexpr (KD) {
                    nn_lds<D, QPT, true>(cur, L, m4, best, pair, tie);
                    kd_fix(cur, best, pair, tie, valid);
                } else {
                    if (have_seed) {
                        seed_bound<D, QPT>(cur, L, m, best, pair);
                        nn_filtered<D, QPT, TIES>(cur, L, m4, F, best, pair, tie);
                    } else if (filter) nn_twomin<D, QPT, TIES>(cur, L, m4, F, best, pair, tie);
                    else nn_lds<D, QPT, TIES>(cur, L, m4, best, pair, tie);
                    if (TIES) {
                        bool any = false;
#pragma unroll
                        for (int k = 0; k < QPT; k++) any = any || (tie[k] != 0 && valid[k]);
                        if (block_any(any, any_word)) {
                            // a flagged query: this kernel cannot resolve it.  Nothing of this iteration has been
                            // stored yet; hand the pair over as it stood before it
                            if (tid == 0) {
                                st->kd_flags |= 2;
                                atomicAdd(pb.kd_paused, 1);
                            }
                            exit_code = 1;
                            break;
                        }
                    }
                }
                have_seed = filter;
#pragma unroll
                for (int k = 0; k < QPT; k++) {
                    cost[k] = __builtin_sqrt(__builtin_sqrt(best[k]));  // vanilla_3d.cpp:41 on kdtree.h:126
                    sel[k] = valid[k] && (METHOD == ICPX_VANILLA || cost[k] <= pb.max_distance);  // trimmed_3d.cpp:98
                }
                if (METHOD == ICPX_TRIMMED) k_sel = block_count<QPT>(sel, T, L.gs);
            }

            if (!(D == 2 && METHOD == ICPX_TRIMMED)) ICPX_PHASE(8);
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
                [&](int k, int v) -> double {
                    if (v < D) return sel[k] ? cur[k][v] : 0.0;
                    if (v < 2 * D) return bp[k][v - D];
                    return (D == 2 && valid[k]) ? cost[k] : 0.0;
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
                [&](int k, int v) -> double {
                    const int r = v % D, c = v / D;
                    return sel[k] ? (cur[k][r] - ca[r]) * (bp[k][c] - cb[c]) : 0.0;
                },
                n, T, L.gs, s2);

            ICPX_PHASE(10);  // sums
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
            ICPX_PHASE(11);  // Kabsch + broadcast
        }
        if (tid == 0) {
            st->iteration_count += 1;
            st->cost = state_cost;
        }
    }

    ICPX_VSTAMP(14);  // loop
    // ---- publish ----
    if ((filter || chunked) && pb.seeds) {
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
        } else if (!sorted_matches || pb.lazy_sort != 0) {
            // (lazy_sort: trimmed.cpp:55-56 leaves the matches sorted by cost; a converge() call rarely reads them, so the
            // sort — 55 barrier-separated bitonic stages, 24 us for 1000 points — waits until icpx_matches asks)
#pragma unroll
            for (int k = 0; k < QPT; k++)
                if (valid[k]) {
                    pb.m_point[s0 + tid + k * T] = tid + k * T;
                    pb.m_pair[s0 + tid + k * T] = pair[k];
                    pb.m_cost[s0 + tid + k * T] = cost[k];
                    if (sorted_matches) pb.pair_tmp[s0 + tid + k * T] = pair[k];
                }
            if (tid == 0) st->kd_flags = sorted_matches ? (st->kd_flags | 16) : (st->kd_flags & ~static_cast<uint64_t>(16));
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
    ICPX_VSTAMP(15);  // publish
    return exit_code;
}

// Small geometries are asked for 3-4 waves per SIMD (<= 168 / 128 VGPRs): their serial sections
// (radix select, single-lane SVD) only overlap when several workgroups share a CU.  Measured on
// 2048 pairs of 1024 x 1024: 2D trimmed, 20 iterations 7.99 -> 6.13 ms; 3D vanilla, 10 iterations 8.52 -> 4.43 ms.
#ifndef ICPX_2D_Q4_WAVES
#define ICPX_2D_Q4_WAVES 2
#endif
// (the pruned 2D kernel of 4 points per thread spills ~180 registers under a 168-register budget and is bound by
// the latency of its iteration chain, not by throughput: two workgroups per compute unit without spills)
constexpr int pair_min_waves(int d, int qpt, int tmax, bool prune = false) {
    return tmax > 256 ? 1 : (d == 2 ? (qpt <= 2 ? 4 : (qpt <= 4 ? (prune ? ICPX_2D_Q4_WAVES : 3) : 1)) : (qpt <= 4 ? ICPX_3D_SMALL_WAVES : 1));
}

template <int D, int METHOD, int QPT>
ICPX_DEV int pair_body_pruned(const PairBatch& pb, const int p, const int do_setup, const uint64_t max_steps,
                              double* lds_raw);  // icpx_pair_pruned.h

template <int D, int METHOD, int QPT, bool PRUNE, bool KD>
ICPX_DEV int pair_visit(const PairBatch& pb, const int p, const int do_setup, const uint64_t max_steps, double* lds_raw) {
    if constexpr (PRUNE) return pair_body_pruned<D, METHOD, QPT>(pb, p, do_setup, max_steps, lds_raw);
    else return pair_body<D, METHOD, QPT, KD>(pb, p, do_setup, max_steps, lds_raw);
}

template <int D, int METHOD, int QPT, int TMAX, bool PRUNE, bool KD>
__global__ void __launch_bounds__(TMAX, pair_min_waves(D, QPT, TMAX, PRUNE)) pair_kernel(const PairBatch pb) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    if (pb.kd_takeover) {
        // the kernel behind a launch that only detects the kd rule's ties (KD kernels: workgroup b looks after pair b if
        // that launch stopped it at a tie, from where it stopped) and / or hands over pairs its grid does not suit (bit 2
        // of kd_flags: from the beginning, icpx_pair_pruned.h)
        if (__hip_atomic_load(pb.kd_paused, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
        const int flags = static_cast<int>(pb.states[blockIdx.x].kd_flags);
        __syncthreads();
        const bool fresh = (flags & 4) != 0, paused = KD && (flags & 2) != 0;
        if (!fresh && !paused) return;
        if (threadIdx.x == 0) pb.states[blockIdx.x].kd_flags &= ~static_cast<uint64_t>(6);
        (void)pair_visit<D, METHOD, QPT, PRUNE, KD>(pb, blockIdx.x, fresh ? 1 : 0, pb.max_steps, lds_raw);
        return;
    }
    // Two ways of handing out pairs, ONE call site of the pair's code (it is large):
    //   plain      workgroup b runs pair b from begin to end;
    //   scheduled  persistent workgroups take iteration-granular tasks (pair, next iterations).
    // Hand-off of a pair between workgroups (possibly on different XCDs) follows the agent-scope
    // release / acquire recipe: every wave drains its stores and meets at the barrier (end of the
    // pair's visit), lane 0 releases, then publishes the state word; the taker wins the word with a
    // CAS, acquires, and only then lets the workgroup load.
    __shared__ int s_key, s_pick, s_ready;
    const bool sched = pb.sched_state != nullptr;
    const int tid = threadIdx.x;
    const int P = pb.n_pairs;
    const int home = sched ? static_cast<int>(static_cast<long long>(blockIdx.x) * P / gridDim.x) : 0;
#ifdef ICPX_STATS
    const uint64_t stat_alive0 = wall_clock64();
#endif
    for (int guard = 0; guard < (1 << 30); guard++) {
        int q = blockIdx.x, setup = pb.do_setup;
        uint64_t task_steps = pb.max_steps;
        if (sched) {
            if (tid == 0) {
                s_key = 0x7fffffff;
                s_ready = 0;
            }
            __syncthreads();
            int best = 0x7fffffff;
            int ready = 0;
            for (int c = tid; c < P; c += blockDim.x) {
                int w = __hip_atomic_load(&pb.sched_state[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int st = w & 3;
                if (st <= 1) {
                    // fewest iterations first (longest expected remaining chain); among equals the pair
                    // nearest after this workgroup's home index, so that workgroups that look at the
                    // same moment go for different pairs instead of all racing for the lowest one
                    int it = w >> 2;
                    int rot = c - home;
                    rot += rot < 0 ? P : 0;
                    int key = ((it < 2047 ? it : 2047) << 20) | rot;
                    best = key < best ? key : best;
                    ready++;
                }
            }
            if (best != 0x7fffffff) {
                atomicMin(&s_key, best);
                atomicAdd(&s_ready, ready);
            }
            __syncthreads();
            if (tid == 0) {
                int pick = -2;  // retry
                if (s_key != 0x7fffffff) {
                    int c = (s_key & 0xfffff) + home;
                    c -= c >= P ? P : 0;
                    int w = __hip_atomic_load(&pb.sched_state[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((w & 3) <= 1 && atomicCAS(&pb.sched_state[c], w, (w & ~3) | 2) == w) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        pick = (c << 1) | ((w & 3) == 0 ? 1 : 0);  // bit 0: fresh -> run setup
                    }
                } else {
                    // Nothing fresh or ready: every unfinished pair is running in some workgroup, and that workgroup
                    // looks again right after it releases the pair (it finds it, or whatever else is ready then) — so
                    // this one is not needed any more.  It used to poll until the whole batch was done: a 3D workgroup
                    // owns its compute unit, and whatever waited on another stream (the 2D take-over launch of a bench
                    // step, the next batch) could not start before the last pair had finished.
                    pick = -1;
                }
                s_pick = pick;
            }
            __syncthreads();
            const int pick = s_pick;
            if (pick == -1) break;
            if (pick == -2) continue;
            q = pick >> 1;
            setup = pick & 1;
            // a task carries sched_steps iterations (default 16: a pair is set up and converged in one visit, with no
            // write-back of its running cloud in between); shorter tasks in the tail were tried for balance and lost
            // (512 3D pairs: 1.52 -> 1.58 ms per step — every hand-off is a write-back, a release and a restaging)
            task_steps = static_cast<uint64_t>(pb.sched_steps);
        }
#ifdef ICPX_STATS
        const uint64_t stat_t0 = wall_clock64();
#endif
        const int code = pair_visit<D, METHOD, QPT, PRUNE, KD>(pb, q, setup, task_steps, lds_raw);
        if (!sched) break;
#ifdef ICPX_STATS
        if (tid == 0) {
            atomicAdd(&g_icpx_stats[5], static_cast<unsigned long long>(wall_clock64() - stat_t0));  // busy ticks
            atomicAdd(&g_icpx_stats[6], 1ULL);                                                         // tasks
        }
#endif
        // the visit ended with a barrier behind every wave's stores
        if (tid == 0) {
            const int iters = static_cast<int>(pb.states[q].iteration_count);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&pb.sched_state[q], (iters << 2) | (code == 1 ? 3 : 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            if (code == 1) atomicAdd(pb.sched_done, 1);
        }
    }
#ifdef ICPX_STATS
    if (sched && tid == 0) atomicAdd(&g_icpx_stats[7], static_cast<unsigned long long>(wall_clock64() - stat_alive0));  // alive ticks
#endif
}

}  // namespace icpx
