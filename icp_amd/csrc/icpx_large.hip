// icpx_large.hip — the multi-kernel ICP path for clouds beyond the fused kernel's limits, and the
// host side of the uniform-grid nearest-neighbour search.
//
// One iteration = 5 launches chained on the stream, no host round trip:
//   lg_nn      c = step*c (3D, vanilla_3d.cpp:91), exact grid NN of every source point
//              (vanilla_3d.cpp:30-45), neighbour distance / selection flag (trimmed_3d.cpp:98);
//              trimmed ICP2: NN of transform*a with squared-distance cost (trimmed.cpp:95-109),
//              then a stable radix sort of (cost, point) and the best-overlap_rate selection
//              (trimmed.cpp:55-62)
//   lg_sum1    TREE64 levels 1-2 of the centroid sums (geo.h:33-37 on the selected pairs)
//   lg_fin1    remaining levels, centroids
//   lg_sum2    TREE64 levels 1-2 of the cross-covariance (vanilla_3d.cpp:51)
//   lg_fin2    remaining levels, Kabsch SVD, transform = step*transform (vanilla_3d.cpp:53-66,93),
//              ICPDriver::should_terminate (driver.h:154-222) -> device-side done flag
// Every kernel returns at once when the done flag is set, so the host can enqueue iterations in
// chunks and poll the flag only between chunks.

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <thread>
#include <unordered_map>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "icpx_filter.h"
#include "icpx_bvh.h"
#include "icpx_grid.h"
#include "icpx_kd.h"
#include "icpx_kd_build.h"
#include "icpx_large.h"
#include "icpx_math.h"

namespace icpx {

constexpr int kBlk = 1024;          // threads per reduction block
constexpr int kSlots = 4;           // points per thread -> 4096 points (64 TREE64 groups) per block
constexpr int kBlkPts = kBlk * kSlots;
constexpr int64_t kLargeMaxN = int64_t(1) << 24;  // <= 4096 block partials -> at most 4 tree levels
// up to here (<= 4096 TREE64 groups) the sum kernels write level-1 group sums from many small blocks
// and the finishing block walks the remaining levels: a few thousand points cannot fill 4096-point blocks
constexpr int64_t kGroupSumMaxN = int64_t(1) << 18;
constexpr int kGroupBlk = 1024;     // threads per block in that mode: one group per wave (a full block: its last one finishes the sum)

struct GridIndex {
    GridParams gp{};
    int dim = 0;
    int ncell = 0;
    bool f32 = false;  // set before grid_build: `recs` holds GridRec32 (opt-in f32 correspondence mode, icpx_grid.h)
    // set before grid_build: the build also writes `filt`, the records once more as 16-byte f32 FILTER records at the
    // same positions — the seeded queries of a converge scan those and re-evaluate the survivors from the cloud in f64
    // (grid_scan_rows_filtered: exact, bit-identical)
    bool want_filt = false;
    DevBuf filt;
    DevBuf recs, start, cell_id, counts, keys, totals;
    // query order by cell (grid_sort_queries); q_perm_n == 0: none
    DevBuf q_keys, q_keys_out, q_iota, q_perm, q_tmp, q_start;
    int64_t q_blocks = 0;  // workgroups of the block-cooperative query (q_start: their ranges of q_perm)
    double occupied_frac = 0.0;  // share of the cells that hold a target (grid_build)
    double per_occupied = 0.0;   // targets per occupied cell after the refinement (grid_build)
    int64_t q_perm_n = 0;
    // grid_build(defer = true): the occupied-cell count of the first round is read back by grid_settle, after the
    // caller has enqueued the work that follows the build — the build's second host round trip leaves the critical path
    struct Build {
        const double* d_t = nullptr;
        int64_t m = 0, max_cells = 0;
        double ext[3] = {0, 0, 0}, h = 1.0;
        int nz = 0;
    } b;
    bool refined_last = false;    // the last build needed a finer round: the next one does not defer (it would build twice)
    int* occ_host = nullptr;      // pinned
    unsigned long long* keys_host = nullptr;  // pinned: the bounding box keys' read-back (a pageable destination is staged by the runtime)
    hipEvent_t occ_ev = nullptr;
    bool occ_pending = false;
};

struct BvhIndex {
    int dim = 0;
    int64_t m = 0;
    int P = 0;             // leaves (power of two)
    BvhFrame frame{};      // Morton grid over the targets' bounding box (also orders the queries)
    DevBuf recs, box, keys, keys_out, iota, perm, tmp, bbox;
    DevBuf q_perm;         // Morton order of the queries (bvh_sort_queries); q_perm_n == 0: none
    int64_t q_perm_n = 0;
    BvhView view() const {
        BvhView v;
        v.box = box.as<float>();
        v.recs = recs.as<GridRec>();
        v.m = m;
        v.P = P;
        return v;
    }
};

struct LargeDev {
    PairState st;
    double step[16];
    double ca[3], cb[3];
    double ksel;
    double cost_sum;
    int nn_valid;  // nb_idx holds the previous correspondence step's neighbours (seeds of the tiled search); 2: seeds of lg_tile_seed_kernel
    int seed_bad;  // ... and some query found no seed there: that search runs unseeded
    // FeatureAware: maxima as f64_key bit patterns (atomicMax), centroids of the two feature passes
    unsigned long long fmax_key, dmax_key;
    double fcent[2];
    int with_cost;    // the matches carry a cost that ICP::calculate_cost sums (2D; 3D with ICPX_FLAG_FILL_MATCHES)
    int update_cost;  // that cost changes in every iterate (not for reference 2D vanilla, Q1)
    // kd-tree tie rule (icpx_kd.h): the correspondence kernels append the queries that have a tie partner
    // (the four words every per-iteration kernel tests first sit in ONE aligned 16-byte word: lg_head fetches them with one load)
    alignas(16) int kd_count;  // entries of the list (cleared by the kernel that ends an iteration)
    int kd_ready;     // the target's tree is on the device: kd_resolve_kernel follows every correspondence step
    int kd_stall;     // flagged queries but no tree yet: every kernel returns at once until the host has built it
    int done_m;       // mirror of st.done (written wherever st.done is)
    int tick1, tick2; // workgroups of lg_sum1 / lg_sum2 that have written their partials (the last one finishes the sums)
    int _pad2;
};

// The head of a per-iteration kernel.  `lg_skip(dev) -> return; lg_kd_gate(dev) -> return; transform ...` compiled to
// a chain of loads with a wait and a branch behind each — five or six dependent round trips to words the previous
// kernel has just written (they come from memory: another XCD's L2), 4 - 6 us of an 11 us kernel.  The four flags are
// one 16-byte word, fetched once; what else the kernel needs from LargeDev is loaded before the first branch.
struct LgHead {
    int kd_count, kd_ready, kd_stall, done;
};
ICPX_DEV LgHead lg_head(const LargeDev* dev) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 w = *reinterpret_cast<const i32x4*>(&dev->kd_count);
    LgHead h;
    h.kd_count = w[0];
    h.kd_ready = w[1];
    h.kd_stall = w[2];
    h.done = w[3];
    return h;
}
// every kernel of an iteration behind the correspondence stage begins with this
ICPX_DEV bool lg_skip(const LgHead& h, int respect_done) { return (respect_done && h.done) || h.kd_stall != 0; }
// ... and the first of them (lg_sum1) turns "flagged queries and no tree" into the stall, telling the host
ICPX_DEV bool lg_kd_gate(const LgHead& h, LargeDev* dev, int* hflags) {
    if (h.kd_count == 0 || h.kd_ready != 0) return false;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        dev->kd_stall = 1;
        if (hflags) __hip_atomic_store(&hflags[2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return true;
}

struct LargeIcp {
    int dim = 0, method = 0;
    icpx_params prm{};
    double clock_to_s = 1e-8;
    hipStream_t stream = nullptr;
    int64_t n = 0, m = 0;
    int nblk = 0;         // partials per summed term: 4096-point blocks, or 64-point groups when `groups`
    bool groups = false;  // n <= kGroupSumMaxN
    GridIndex* grid = nullptr;
    BvhIndex* bvh = nullptr;   // built instead of / after the grid: nn = bvh, or nn = auto with a degenerate grid
    bool use_bvh = false;
    int tile_qpt = 4;        // queries per thread of the tiled search (large_begin)
    bool nn_seeded = false;  // nb_idx holds the neighbours of an earlier correspondence step of this begin()
    DevBuf src, dst, cur, guess, nb_idx, nb_cost, sel, part, cntp, dev;
    DevBuf nb_pt;  // the matched target's coordinates per source point (written with nb_idx): the sums stream them
    // Sorted domain (3D grid search over >= kGridSortMinQueries points): the correspondence kernel visits the points in
    // the cell order of begin() (q_perm).  It keeps its own copy of the running cloud and the previous neighbours IN
    // THAT ORDER (cur_s, idx_s: streamed) and scatters only the matched target's coordinates to the point's place
    // (nb_pt); the sums stream the cloud and nb_pt in point order (TREE64 is defined on point indices), apply the step
    // to `cur` themselves and recompute the selection from the two points.  nb_idx / nb_cost / sel in point order are
    // written on demand (sd_unpack).
    bool sd = false, sd_dirty = false;
    DevBuf cur_s, idx_s, pt_s;  // pt_s: the previous neighbour's coordinates in query order (the seed without a gather)
    // trimmed ICP2: sort of (cost bits, point) -> matches order of trimmed.cpp:55-56
    DevBuf sort_keys, sort_iota, sort_order, sort_tmp;
    size_t sort_tmp_bytes = 0;
    int64_t new_n = 0;
    bool sorted = false;  // an iterate has run since begin: matches are in sorted order (Q9)
    // tiled brute-force correspondence (mid-size clouds): (query tiles) x (target splits)
    bool tile = false;
    int splits = 0, split_len = 0;
    bool tile_interleaved = false;  // split s holds the chunks s, s + splits, ... of 32 targets (lg_tile_nn_kernel)
    bool tile_first = false;        // no tiled search since begin(): lg_tile_seed_kernel goes first
    DevBuf tile_prep;               // per split: filter frame, chunk boxes, compactness (lg_tile_prep_kernel), once per begin()
    DevBuf part_d2, part_idx;
    // FeatureAware (feature_aware.cpp): radii, features (row r of point i at [r * count + i]) and the
    // normalised feature-distance matrix, stored target-major ([j * n + i]) for coalesced reads
    bool feature = false;
    DevBuf f_rad, f_feat_a, f_feat_b, f_nfd;
    // kd-tree tie rule (icpx_kd.h): flagged queries, and — once a correspondence step has flagged one — the target's tree
    bool kd_rule = false;    // the method takes its neighbours from KdTree::search (vanilla 2D, both 3D)
    bool kd_ready = false;   // kd_tree holds the tree of this begin()'s target cloud
    DevBuf kd_list, kd_tree; // n query indices; [node m][pos m]
    DevBuf kd_scratch;       // kd_build_device's working arrays
    // clouds up to kKdEagerMax points: the tree is built at begin() on a stream of its own, beside the first
    // correspondence step; the first resolve launch waits for it (larger clouds: on demand, after a stall)
    hipStream_t kd_stream = nullptr;
    hipEvent_t kd_ev_cloud = nullptr, kd_ev_tree = nullptr;
    bool kd_wait = false;    // the main stream has not yet waited for kd_ev_tree
    bool kd_tied = false;    // a converge on this handle stalled at flagged queries before (large_begin: build the tree eagerly)
    int* h_done = nullptr;  // pinned: [0] done flag, [1] iterations finished, [2] stalled at a tie (written by the device, large_converge polls)
    int* h_done_dev = nullptr;   // the same memory as the device sees it
    int* poll_dev = nullptr;     // h_done_dev while large_converge runs, else null: kernels then publish nothing
    // host time at which lg_init_kernel (which samples the device's start_clock) was enqueued: the
    // time-limit test adds ONE host offset (converge's t0 -> this moment) to the device clock since then
    std::chrono::steady_clock::time_point init_at{};
};

// ------------------------------------------------------------------------------------------
// device kernels
// ------------------------------------------------------------------------------------------

namespace {

template <int D>
__global__ void lg_init_kernel(LargeDev* dev, const double* guess, int with_cost, int update_cost, int kd_ready) {
    constexpr int XN = (D + 1) * (D + 1);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int e = 0; e < 16; e++) {
        double g = e < XN ? guess[e] : 0.0;
        dev->st.transform[e] = g;
        dev->st.last_transform[e] = g;
        dev->step[e] = (e < XN && (e % (D + 2)) == 0) ? 1.0 : 0.0;
    }
    dev->st.iteration_count = 0;
    dev->st.have_last = 0;
    dev->st.done = 0;
    dev->done_m = 0;
    dev->st.cost = 0.0;
    dev->st.last_cost = 0.0;
    dev->st.start_clock = wall_clock64();
    dev->ksel = 0.0;
    dev->cost_sum = 0.0;
    dev->nn_valid = 0;
    dev->seed_bad = 0;
    dev->with_cost = with_cost;
    dev->update_cost = update_cost;
    dev->fmax_key = 0;
    dev->dmax_key = 0;
    dev->fcent[0] = dev->fcent[1] = 0.0;
    dev->kd_count = 0;
    dev->kd_ready = kd_ready;
    dev->kd_stall = 0;
    dev->tick1 = 0;
    dev->tick2 = 0;
}

// after the host has put the target's tree on the device (lg_kd_service): the stall ends
__global__ void lg_kd_resume_kernel(LargeDev* dev) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    dev->kd_ready = 1;
    dev->kd_stall = 0;
}

// sorted domain: the running cloud in the query order of the grid search
__global__ void lg_sd_gather_kernel(const double* __restrict__ cur, const uint32_t* __restrict__ perm, int64_t n,
                                    double* __restrict__ cur_s) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t i = perm[k];
#pragma unroll
    for (int d = 0; d < 3; d++) cur_s[k * 3 + d] = cur[i * 3 + d];
}

// the 3D methods' match cost and selection from the two points (what the correspondence stage publishes otherwise):
// sqrt of kdtree.h:126's Euclidean distance; Trimmed3d keeps a match when it is <= max_distance (trimmed_3d.cpp:42-49)
template <int METHOD>
ICPX_DEV bool sd_select(const double* c, const double* b, const double max_distance, double* cost_out = nullptr) {
    double df = b[0] - c[0];
    double d2 = df * df;
    df = b[1] - c[1];
    d2 = d2 + df * df;
    df = b[2] - c[2];
    d2 = d2 + df * df;
    const double cost = __builtin_sqrt(__builtin_sqrt(d2));
    if (cost_out) *cost_out = cost;
    return METHOD == ICPX_TRIMMED ? cost <= max_distance : true;
}

// sorted domain -> nb_idx / nb_cost / sel in point order (for the getters)
template <int METHOD>
__global__ void lg_sd_unpack_kernel(const double* __restrict__ cur_s, const int64_t* __restrict__ idx_s,
                                    const uint32_t* __restrict__ perm, const double* __restrict__ cloud, int64_t n,
                                    double max_distance, int64_t* __restrict__ nb_idx, double* __restrict__ nb_cost,
                                    uint8_t* __restrict__ sel) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t i = perm[k], j = idx_s[k];
    const double c[3] = {cur_s[k * 3], cur_s[k * 3 + 1], cur_s[k * 3 + 2]};
    const double b[3] = {cloud[j * 3], cloud[j * 3 + 1], cloud[j * 3 + 2]};
    double cost;
    const bool s = sd_select<METHOD>(c, b, max_distance, &cost);
    nb_idx[i] = j;
    nb_cost[i] = cost;
    sel[i] = s ? 1 : 0;
}

// NN stage.  MODE 0: the 3D per-iteration stage.  MODE 1: Vanilla::setup's compute_matches on the
// untransformed source (vanilla.cpp:68-88).  MODE 2: Trimmed::compute_matches on transform*a with
// the squared distance as cost (trimmed.cpp:95-109).  MODE 3: Vanilla's compute_matches on
// transform*a (opt-in ICPX_FLAG_MATCH_TRANSFORMED, not reference behaviour).
// qstart != null: block-cooperative (icpx_grid.h) — one workgroup per block of tiles, its queries the range
// [qstart[b], qstart[b + 1]) of the order `perm`; otherwise one query per thread.
template <int D, int METHOD, int MODE, typename Rec = GridRec>
// (six waves per SIMD: the 80-register budget costs the per-lane walk nothing and hides more of its memory latency —
// clustered config-4 converge 6.83 -> 6.55 ms; eight waves spill and lose)
#ifndef ICPX_LG_NN_WAVES
#define ICPX_LG_NN_WAVES 6
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ICPX_LG_NN_WAVES, 8)))
lg_nn_kernel(LargeDev* dev, int respect_done, const double* __restrict__ src, double* __restrict__ cur,
             int64_t n, GridParams gp, const Rec* __restrict__ recs, const int* __restrict__ start,
             int64_t* __restrict__ nb_idx, double* __restrict__ nb_cost, uint8_t* __restrict__ sel,
             double max_distance, const uint32_t* __restrict__ perm, const int* __restrict__ qstart, const BvhView bvh,
             const double* __restrict__ seed_tgt, const double* __restrict__ cloud, uint32_t* __restrict__ kd_list,
             double* __restrict__ nb_pt, double* __restrict__ cur_s = nullptr, int64_t* __restrict__ idx_s = nullptr,
             double* __restrict__ pt_s = nullptr, const GridRec32* __restrict__ filt = nullptr) {
    // filt: f32 filter records of the same grid (seeded queries scan those, exact results: grid_scan_rows_filtered)
    extern __shared__ __attribute__((aligned(16))) char grid_lds[];
    // the kd-tree methods follow the reference tree's tie rule (icpx_kd.h): flagged queries go on kd_list
    // (not in the opt-in f32 mode, GridRec32: it gives up index parity altogether)
    constexpr bool F32 = RecTraits<Rec>::f32;
    constexpr bool TIES = (D == 3 || METHOD == ICPX_VANILLA) && MODE != 2 && !F32;
    if (lg_skip(lg_head(dev), respect_done)) return;
    // the query of point i as the method defines it
    auto query_of = [&](const int64_t i, double (&q)[D]) {
        if (MODE == 1) {
#pragma unroll
            for (int d = 0; d < D; d++) q[d] = src[i * D + d];
        } else if (MODE == 2 || MODE == 3) {
            Xf<D> X;
#pragma unroll
            for (int e = 0; e < (D + 1) * (D + 1); e++) X.m[e] = dev->st.transform[e];
            XfRows<D> xr = xf_rows<D>(X);
            double p[D];
#pragma unroll
            for (int d = 0; d < D; d++) p[d] = src[i * D + d];
            xf_apply<D>(xr, p, q);  // a_current = transform * a, trimmed.cpp:40
        } else {
            // c = T * c with the step the previous iteration produced (identity after begin)
            Xf<D> S;
#pragma unroll
            for (int e = 0; e < (D + 1) * (D + 1); e++) S.m[e] = dev->step[e];
            XfRows<D> sr = xf_rows<D>(S);
            double c[D];
#pragma unroll
            for (int d = 0; d < D; d++) c[d] = cur[i * D + d];
            xf_apply<D>(sr, c, q);
#pragma unroll
            for (int d = 0; d < D; d++) cur[i * D + d] = q[d];
        }
    };
    auto publish = [&](const int64_t i, double best, const int bi, const int tie, const double* q = nullptr) {
        nb_idx[i] = bi;
        double b[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int d = 0; d < D; d++) nb_pt[i * D + d] = b[d] = cloud[static_cast<int64_t>(bi) * D + d];
        // f32 mode: the search chose the target in f32; its cost is the f64 distance to that target, as everywhere else
        if (F32 && q != nullptr && gp.m > 0) best = d2_eval<D>(b[0], b[1], b[2], q);
        if (TIES && tie) kd_list[atomicAdd(&dev->kd_count, 1)] = static_cast<uint32_t>(i);
        if (MODE == 2) {
            nb_cost[i] = best;  // trimmed.cpp:101 squaredNorm; selection comes from lg_select_kernel
        } else {
            double c = __builtin_sqrt(__builtin_sqrt(best));  // sqrt of kdtree.h:126's Euclidean distance
            nb_cost[i] = c;
            sel[i] = (D == 3 && METHOD == ICPX_TRIMMED) ? (c <= max_distance ? 1 : 0) : 1;
        }
    };
    int visited = 0;
    if (bvh.P > 0) {
        // the BVH instead of the grid (icpx_bvh.h): the launch reserved the LDS copy of the tree's top levels
        float* top = reinterpret_cast<float*>(grid_lds);
        bvh_stage_top(bvh, top);
        int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
        if (i >= n) return;
        if (perm) i = perm[i];  // Hilbert order of the initial cloud
        double q[D], best;
        int bi;
        query_of(i, q);
        int tie = 0;
        bvh_query<D, TIES>(bvh, top, q, best, bi, visited, tie, cloud);
        publish(i, best, bi, tie);
        return;
    }
    if (qstart) {
        const int qs = qstart[blockIdx.x], qe = qstart[blockIdx.x + 1];
        if (qs >= qe) return;  // (uniform)
        int b0[3], b1[3];
        const bool staged = grid_block_stage<D, Rec>(gp, recs, start, blockIdx.x, grid_lds, b0, b1);
        for (int k = qs + threadIdx.x; k < qe; k += blockDim.x) {
            const int64_t i = perm[k];
            double q[D], best;
            int bi;
            query_of(i, q);
            int tie = 0;
            float q32[3];
            if (F32) query_f32<D>(gp.lo, q, q32);
            grid_block_query<D, TIES, Rec>(gp, recs, start, q, staged, b0, b1, grid_lds, best, bi, visited, tie, cloud, q32);
            publish(i, best, bi, tie, q);
        }
        return;
    }
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (D == 3 && MODE == 0 && cur_s != nullptr) {
        // sorted domain (LargeIcp::sd): position k of the query order; only nb_pt goes to the point's own place
        const int64_t k = i;
        i = perm[k];
        double q[D], best;
        int bi, tie = 0;
        {
            Xf<D> S;
#pragma unroll
            for (int e = 0; e < (D + 1) * (D + 1); e++) S.m[e] = dev->step[e];
            XfRows<D> sr = xf_rows<D>(S);
            double c[D];
#pragma unroll
            for (int d = 0; d < D; d++) c[d] = cur_s[k * D + d];
            xf_apply<D>(sr, c, q);  // c = T * c, vanilla_3d.cpp:91 (lg_sum1 applies the same step to the cloud in point order)
#pragma unroll
            for (int d = 0; d < D; d++) cur_s[k * D + d] = q[d];
        }
        float q32[3];
        if (F32) query_f32<D>(gp.lo, q, q32);
        bool searched = false;
        int64_t sidx = -1;
        if (seed_tgt) {
            sidx = idx_s[k];
            bool ok = gp.m > 0;
            // (pt_s[k]: the coordinates of target idx_s[k], kept in query order — streamed, where seed_tgt[sidx] is a gather)
            double df = pt_s[k * D] - q[0];
            double d2s = df * df;
#pragma unroll
            for (int d = 1; d < D; d++) {
                df = pt_s[k * D + d] - q[d];
                d2s = d2s + df * df;
            }
#pragma unroll
            for (int d = 0; d < D; d++) ok = ok && finite_d(q[d]);
            ok = ok && d2s < __builtin_inf();
            if (ok) {
                best = d2s;
                bi = static_cast<int>(sidx);
                grid_query_seeded<D, TIES, Rec>(gp, recs, start, q, best, bi, visited, tie, cloud, q32, filt);
                searched = true;
            }
        }
        if (!searched) grid_query<D, TIES, Rec>(gp, recs, start, q, best, bi, visited, tie, cloud, 0, __builtin_inf(), 0x7fffffff, q32);
        // nb_pt[i] always holds the coordinates of target idx_s[k] (the resolve kernel keeps that too): a query whose
        // neighbour is the one of the previous step — most of them, late in a converge — has nothing to gather or scatter
        if (static_cast<int64_t>(bi) != sidx) {
            idx_s[k] = bi;
#pragma unroll
            for (int d = 0; d < D; d++) {
                const double v = cloud[static_cast<int64_t>(bi) * D + d];
                nb_pt[i * D + d] = v;
                pt_s[k * D + d] = v;
            }
        }
        if (TIES && tie) kd_list[atomicAdd(&dev->kd_count, 1)] = static_cast<uint32_t>(k);
        return;
    }
    if (perm) i = perm[i];  // tile order of the initial cloud: neighbouring lanes walk the same cells
    double q[D], best;
    int bi;
    query_of(i, q);
    int tie = 0;
    float q32[3];
    if (F32) query_f32<D>(gp.lo, q, q32);
    if (seed_tgt) {
        // nb_idx holds the neighbour of the previous correspondence step (the target cloud is seed_tgt): a candidate
        // whose ball bounds the search (grid_query_seeded).  Same (d2, index) minimum as the unseeded query.
        const int64_t s = nb_idx[i];
        bool ok = gp.m > 0;
        double df = seed_tgt[s * D] - q[0];
        double d2s = df * df;
#pragma unroll
        for (int d = 1; d < D; d++) {
            df = seed_tgt[s * D + d] - q[d];
            d2s = d2s + df * df;
        }
#pragma unroll
        for (int d = 0; d < D; d++) ok = ok && finite_d(q[d]);
        ok = ok && d2s < __builtin_inf();  // (false for NaN)
        if (ok) {
            best = d2s;
            bi = static_cast<int>(s);
            grid_query_seeded<D, TIES, Rec>(gp, recs, start, q, best, bi, visited, tie, cloud, q32, filt);
            publish(i, best, bi, tie, q);
            return;
        }
    }
    grid_query<D, TIES, Rec>(gp, recs, start, q, best, bi, visited, tie, cloud, 0, __builtin_inf(), 0x7fffffff, q32);
    publish(i, best, bi, tie, q);
}

// The seeded sorted-domain step of lg_nn_kernel (3D methods, per-iteration stage, filter records) as a kernel of its
// own.  Inside lg_nn_kernel that path shares its register budget (80, for six waves per SIMD) with the f64-record
// searches of the other paths — four 32-byte records in flight — and spills (132 registers); here the hot path is the
// filtered scan (grid_scan_rows_filtered: 16-byte records, exact re-evaluation of the survivors from the cloud) and
// the rare fallbacks — balls beyond 3 x 3 rows, points without a usable seed — walk the f64 records ONE at a time.
// Same operations on the same values as lg_nn_kernel's branch: bit-identical (ICPX_GRID_FILTER=0 keeps that branch).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ICPX_LG_NN_WAVES, 8)))
lg_nn_sd_kernel(LargeDev* dev, int respect_done, int64_t n, GridParams gp, const GridRec* __restrict__ recs,
                const GridRec32* __restrict__ filt, const int* __restrict__ start, const uint32_t* __restrict__ perm,
                const double* __restrict__ cloud, uint32_t* __restrict__ kd_list, double* __restrict__ nb_pt,
                double* __restrict__ cur_s, int64_t* __restrict__ idx_s, double* __restrict__ pt_s) {
    constexpr int D = 3;
    const LgHead head = lg_head(dev);
    Xf<D> S;
#pragma unroll
    for (int e = 0; e < (D + 1) * (D + 1); e++) S.m[e] = dev->step[e];
    if (lg_skip(head, respect_done)) return;
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t i = perm[k];
    double q[D], best;
    int bi, tie = 0, visited = 0;
    {
        XfRows<D> sr = xf_rows<D>(S);
        double c[D];
#pragma unroll
        for (int d = 0; d < D; d++) c[d] = cur_s[k * D + d];
        xf_apply<D>(sr, c, q);  // c = T * c, vanilla_3d.cpp:91
#pragma unroll
        for (int d = 0; d < D; d++) cur_s[k * D + d] = q[d];
    }
    const int64_t sidx = idx_s[k];
    bool ok = gp.m > 0;
    double df = pt_s[k * D] - q[0];
    double d2s = df * df;
#pragma unroll
    for (int d = 1; d < D; d++) {
        df = pt_s[k * D + d] - q[d];
        d2s = d2s + df * df;
    }
#pragma unroll
    for (int d = 0; d < D; d++) ok = ok && finite_d(q[d]);
    ok = ok && d2s < __builtin_inf();
    if (ok) {
        best = d2s;
        bi = static_cast<int>(sidx);
        grid_query_seeded<D, true, GridRec, 1>(gp, recs, start, q, best, bi, visited, tie, cloud, nullptr, filt);
    } else {
        grid_query<D, true, GridRec, 1>(gp, recs, start, q, best, bi, visited, tie, cloud);
    }
    if (static_cast<int64_t>(bi) != sidx) {
        idx_s[k] = bi;
#pragma unroll
        for (int d = 0; d < D; d++) {
            const double v = cloud[static_cast<int64_t>(bi) * D + d];
            nb_pt[i * D + d] = v;
            pt_s[k * D + d] = v;
        }
    }
    if (tie) kd_list[atomicAdd(&dev->kd_count, 1)] = static_cast<uint32_t>(k);
}

// ---- tiled brute-force correspondence for mid-size clouds --------------------------------------
// grid = (query tiles of 1024, target splits).  Each workgroup stages its target split in LDS
// (f64 SoA + f32 filter copy, icpx_filter.h) and answers its 1024 queries against it: seeded by the
// previous step's neighbour when there is one (whose exact distance bounds the answer for EVERY
// split), seedless otherwise.  Per-split results are merged by lg_tile_merge_kernel in the (d2, index)
// order.  On 256 CUs this beats the grid walk up to about 3e4 x 3e4 points.
constexpr int64_t kGridSortMinQueries = 32768;
#ifndef ICPX_TILE_THREADS
#define ICPX_TILE_THREADS 256
#endif
constexpr int kTileThreads = ICPX_TILE_THREADS;
#ifdef ICPX_TILE_QPT
constexpr int kTileQpt = ICPX_TILE_QPT;   // development: fixed queries per thread (1, 2 or 4)
constexpr bool kTileQptAuto = false;
#else
constexpr int kTileQpt = 4;
constexpr bool kTileQptAuto = true;
#endif
constexpr int kTileScratch = 96;  // doubles of reduction scratch
constexpr int kTileMaxSplit = 3072;  // targets per split at most (3D: 110 KB of LDS)
constexpr int kTileMinSplit = 512;
constexpr int64_t kFeatureMaxWork = int64_t(1) << 28;  // FeatureAware: n*m doubles of feature distances (2 GiB)
constexpr int64_t kTileMaxWork = int64_t(1) << 30;  // n*m up to which ICPX_NN_AUTO prefers the tiled search

// (f64 SoA + reduction scratch + f32 filter copy + the chunks' bounding boxes, icpx_filter.h)
inline size_t tile_lds_bytes(int dim, int scap) {
    return sizeof(double) * (static_cast<size_t>(dim) * scap + kTileScratch) + sizeof(float) * static_cast<size_t>(dim) * scap +
           sizeof(double) * 2 * static_cast<size_t>(dim) * ((scap + kChunk - 1) / kChunk + 1);
}

__global__ void lg_apply_guess_kernel(const double* __restrict__ guess, const double* __restrict__ src,
                                      double* __restrict__ cur, int64_t n) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xf<3> G;
#pragma unroll
    for (int e = 0; e < 16; e++) G.m[e] = guess[e];
    XfRows<3> gr = xf_rows<3>(G);
    double p[3] = {src[i * 3], src[i * 3 + 1], src[i * 3 + 2]}, q[3];
    xf_apply<3>(gr, p, q);
    cur[i * 3] = q[0];
    cur[i * 3 + 1] = q[1];
    cur[i * 3 + 2] = q[2];
}

// A workgroup's target split and its LDS layout (f64 SoA, scratch, f32 filter copy, chunk boxes).
// interleaved: split s of S holds the chunks s, s + S, s + 2 S ... of 32 consecutive targets (LdsView::cstride).
template <int D>
struct TileView {
    LdsView<D> L;
    double* cbox;
    int64_t t0;
    int mt, m4;
};
template <int D>
ICPX_DEV TileView<D> tile_view(double* tile_lds, const double* __restrict__ dst, const int64_t m, const int split_len,
                               const int scap, const int interleaved) {
    TileView<D> V;
    V.t0 = static_cast<int64_t>(blockIdx.y) * split_len;
    V.mt = static_cast<int>(m - V.t0 < split_len ? m - V.t0 : split_len);
    if (interleaved) {
        const int S = static_cast<int>(gridDim.y), s = static_cast<int>(blockIdx.y);
        const int nc = static_cast<int>((m + kChunk - 1) / kChunk);  // chunks of the cloud
        const int mine = s < nc ? (nc - s + S - 1) / S : 0;            // chunks s, s + S, ...
        const int tail = static_cast<int>(m - static_cast<int64_t>(nc - 1) * kChunk);  // targets of the cloud's last chunk
        V.t0 = static_cast<int64_t>(s) * kChunk;
        V.mt = mine * kChunk - ((mine > 0 && (nc - 1 - s) % S == 0) ? kChunk - tail : 0);
    }
    V.m4 = (V.mt + 3) & ~3;
    LdsView<D>& L = V.L;
    double* p = tile_lds;
    for (int d = 0; d < 3; d++) {
        L.t[d] = d < D ? p : nullptr;
        if (d < D) p += scap;
    }
    L.gs = p;
    p += kTileScratch;
    L.bc = nullptr;
    L.skey = nullptr;
    L.sval = nullptr;
    float* f = reinterpret_cast<float*>(p);
    for (int d = 0; d < 3; d++) L.tf[d] = d < D ? f + static_cast<size_t>(d) * scap : nullptr;
    V.cbox = reinterpret_cast<double*>(f + static_cast<size_t>(D) * scap);  // (scap is a multiple of four: 8-byte aligned)
    L.j0 = static_cast<int>(V.t0);
    L.cstride = interleaved ? static_cast<int>(gridDim.y) * kChunk : 0;
    L.cloud = dst;
    return V;
}

// What a split's workgroups would compute in every launch of lg_tile_nn_kernel, once per begin() (the targets do not
// change): the filter frame (stage_targets), the chunk boxes and whether they are compact (chunk_boxes_wide).
// prep + split * kTilePrepStride: [0..2] frame centre, [3] max |t - centre|, [4] compact, [8...] the boxes.
// grid = (1, splits); the dynamic LDS of lg_tile_nn_kernel.
constexpr int kTilePrepStride = 8 + 2 * 3 * (kTileMaxSplit / kChunk + 2);
template <int D>
__global__ void __launch_bounds__(kTileThreads)
lg_tile_prep_kernel(const double* __restrict__ dst, int64_t m, int split_len, int scap, int interleaved, double* __restrict__ prep) {
    extern __shared__ __attribute__((aligned(16))) double tile_lds[];
    TileView<D> V = tile_view<D>(tile_lds, dst, m, split_len, scap, interleaved);
    const FilterFrame F = stage_targets<D>(dst + V.t0 * D, V.mt, V.m4, kTileThreads, V.L, true);
    bool compact = interleaved != 0 && F.t_absmax < 1e18 && V.mt > 0;
    if (compact) compact = chunk_boxes_wide<D>(V.L, V.mt, V.cbox);
    double* out = prep + static_cast<size_t>(blockIdx.y) * kTilePrepStride;
    if (threadIdx.x == 0) {
        for (int d = 0; d < 3; d++) out[d] = F.c0[d];
        out[3] = F.t_absmax;
        out[4] = compact ? 1.0 : 0.0;
    }
    if (compact) {
        const int nw = 2 * D * ((V.mt + kChunk - 1) / kChunk);
        for (int i = threadIdx.x; i < nw; i += kTileThreads) out[8 + i] = V.cbox[i];
    }
}

// MODE as in lg_nn_kernel (0: queries = step * running cloud, computed here; 1: untransformed source;
// 2: transform * source)
template <int D, int MODE, int QPT>
__global__ void __launch_bounds__(kTileThreads)
lg_tile_nn_kernel(const LargeDev* dev, int respect_done, const double* __restrict__ src, const double* __restrict__ cur,
                  int64_t n, const double* __restrict__ dst, int64_t m, int split_len, int scap,
                  const int64_t* __restrict__ nb_idx, double* __restrict__ part_d2, int32_t* __restrict__ part_idx,
                  const int interleaved, const double* __restrict__ prep) {
    // prep (lg_tile_prep_kernel): the split's filter frame and chunk boxes, computed once per begin()
    // interleaved: split s of S holds the chunks s, s + S, s + 2 S ... of 32 consecutive targets (LdsView::cstride), and
    // the search skips the chunks no query of the wave can need (nn_filtered<..., CHUNKED>).  Clouds in a coherent
    // order — mesh vertices, scans — leave every workgroup its share of the needed chunks (with contiguous splits the
    // launch is as long as the few workgroups whose split lies in their queries' neighbourhood).  Same results;
    // ICPX_TILE_CHUNK=0: contiguous splits, every block scanned.
    extern __shared__ __attribute__((aligned(16))) double tile_lds[];
    const LgHead head = lg_head(dev);
    const int head_nn_valid = dev->nn_valid, head_seed_bad = dev->seed_bad;
    Xf<D> head_xf;
#pragma unroll
    for (int e = 0; e < (D + 1) * (D + 1); e++) head_xf.m[e] = MODE == 0 ? dev->step[e] : dev->st.transform[e];
    if (lg_skip(head, respect_done)) return;
    const int tid = threadIdx.x;
    TileView<D> V = tile_view<D>(tile_lds, dst, m, split_len, scap, interleaved);
    LdsView<D>& L = V.L;
    double* const cbox = V.cbox;
    const int64_t t0 = V.t0;
    const int mt = V.mt, m4 = V.m4;
    constexpr bool TIES = MODE != 2;  // the kd-tree methods (MODE 2 is trimmed ICP2: brute-force rule); icpx_kd.h
    // queries and seed indices first: their loads are in flight while the split is staged
    double q[QPT][D], best[QPT];
    int bidx[QPT];
    int64_t seed[QPT];
    const int64_t qbase = static_cast<int64_t>(blockIdx.x) * (kTileThreads * QPT);
    const bool seeded = head_nn_valid == 1 || (head_nn_valid == 2 && head_seed_bad == 0);
    XfRows<D> xr;
    if (MODE == 2 || MODE == 3 || MODE == 0) {
        // MODE 0: c = T * c with the step the previous iteration produced (identity after begin), vanilla_3d.cpp:91 — every
        // split's workgroup computes it from the unstepped cloud (lg_sum1 writes the stepped one behind this kernel)
        xr = xf_rows<D>(head_xf);
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        const int64_t i = qbase + tid + k * kTileThreads;
        const bool valid = i < n;
        double pnt[D];
#pragma unroll
        for (int d = 0; d < D; d++) pnt[d] = valid ? (MODE == 0 ? cur[i * D + d] : src[i * D + d]) : 0.0;
        if (MODE == 2 || MODE == 3 || MODE == 0) {
            xf_apply<D>(xr, pnt, q[k]);  // a_current = transform * a, trimmed.cpp:40
        } else {
#pragma unroll
            for (int d = 0; d < D; d++) q[k][d] = pnt[d];
        }
        int64_t j = (seeded && valid) ? nb_idx[i] : 0;
        seed[k] = j < 0 ? 0 : (j >= m ? m - 1 : j);
    }
    // frame and boxes from the begin-time pass: the split is staged in ONE pass (f64 SoA and its f32 copy about the
    // frame's centre, exactly stage_targets' values) behind one barrier — no bounding-box and |t - c| reductions
    FilterFrame F;
    const double* pr = prep + static_cast<size_t>(blockIdx.y) * kTilePrepStride;
#pragma unroll
    for (int d = 0; d < 3; d++) F.c0[d] = pr[d];
    F.t_absmax = pr[3];
    const bool compact = pr[4] != 0.0;
    {
        const double* tb = dst + t0 * D;
        for (int j = tid; j < m4; j += kTileThreads) {
            const int srcj = L.cstride ? (j >> 5) * L.cstride + (j & 31) : j;
#pragma unroll
            for (int d = 0; d < D; d++) {
                const double v = j < mt ? tb[static_cast<int64_t>(srcj) * D + d] : __builtin_inf();
                L.t[d][j] = v;
                L.tf[d][j] = j < mt ? static_cast<float>(v - F.c0[d]) : 1e18f;
            }
        }
        if (compact) {
            const int nw = 2 * D * ((mt + kChunk - 1) / kChunk);
            for (int i = tid; i < nw; i += kTileThreads) cbox[i] = pr[8 + i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        best[k] = __builtin_inf();
        bidx[k] = 0x7fffffff;
        if (seeded) {
            // exact distance to the previous neighbour: an upper bound for every split
            const int64_t j = seed[k];
            double df = dst[j * D + 0] - q[k][0];
            double acc = df * df;
#pragma unroll
            for (int d = 1; d < D; d++) {
                df = dst[j * D + d] - q[k][d];
                acc = acc + df * df;
            }
            best[k] = acc;
            bidx[k] = static_cast<int>(j);
        }
    }
    int tie[QPT];
#pragma unroll
    for (int k = 0; k < QPT; k++) tie[k] = 0;
    // (clouds with non-finite or huge targets keep the plain scans and their conventions for such values)
    // ... and so do clouds whose chunks are not compact (random order: the boxes exclude nothing)
    const bool chunk_ok = compact;  // (interleaved, usable frame, non-empty split, compact chunks: lg_tile_prep_kernel)
    if (chunk_ok && !seeded) {
        // no seed (the first search of a begin): every lane takes the chunk whose box is nearest to its query and
        // evaluates it exactly — a bound from the query's neighbourhood — then searches like a seeded query
        const int nch = (mt + kChunk - 1) / kChunk;
#pragma unroll
        for (int k = 0; k < QPT; k++) {
            double lbmin = __builtin_inf();
            int cmin = 0;
            for (int c = 0; c < nch; c++) {
                double lb = 0.0;
#pragma unroll
                for (int d = 0; d < D; d++) {
                    const double lo = cbox[(c * 2 + 0) * D + d], hi = cbox[(c * 2 + 1) * D + d];
                    const double g = __builtin_fmax(__builtin_fmax(lo - q[k][d], q[k][d] - hi), 0.0);
                    lb = d == 0 ? g * g : lb + g * g;
                }
                const bool lt = lb < lbmin;
                lbmin = lt ? lb : lbmin;
                cmin = lt ? c : cmin;
            }
            const int j1 = min((cmin + 1) * kChunk, m4);
            for (int jb = cmin * kChunk; jb < j1; jb += 4) exact_block<D, TIES>(q[k], L, jb, best[k], bidx[k], TIES ? &tie[k] : nullptr);
        }
    }
    if (chunk_ok) nn_filtered<D, QPT, TIES, true>(q, L, m4, F, best, bidx, tie, cbox);
    else if (seeded) nn_filtered<D, QPT, TIES>(q, L, m4, F, best, bidx, tie);
    else nn_twomin<D, QPT, TIES>(q, L, m4, F, best, bidx, tie);
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        const int64_t i = qbase + tid + k * kTileThreads;
        if (i < n) {
            part_d2[static_cast<int64_t>(blockIdx.y) * n + i] = best[k];
            // (the sign bit carries the split's tie flag to the merge; indices are below 2^29)
            part_idx[static_cast<int64_t>(blockIdx.y) * n + i] = bidx[k] | ((TIES && tie[k] != 0) ? static_cast<int>(0x80000000u) : 0);
        }
    }
}

// Seeds for the FIRST tiled search of a begin() (nb_idx holds nothing yet): every query takes the nearest of a strided
// sample of the targets, found in f32 — any target at a finite distance is a valid seed, the search's result does not
// depend on it — so that all splits start from ONE bound from the query's neighbourhood instead of each from the best
// of its own targets (queen.ply: first search 78 us against 25 us for a seeded one).  A query without a sample at a
// finite f32 distance (non-finite or huge values) sets seed_bad and the search runs unseeded, with its conventions.
constexpr int kTileSeedSamples = 512;
template <int D, int MODE>
__global__ void __launch_bounds__(256)
lg_tile_seed_kernel(LargeDev* dev, int respect_done, int64_t n, const double* __restrict__ src, const double* __restrict__ cur,
                    const double* __restrict__ dst, int64_t m, int stride, int64_t* __restrict__ nb_idx) {
    __shared__ __attribute__((aligned(16))) float st[D][kTileSeedSamples];
    const LgHead head = lg_head(dev);
    Xf<D> xf;
#pragma unroll
    for (int e = 0; e < (D + 1) * (D + 1); e++) xf.m[e] = MODE == 0 ? dev->step[e] : dev->st.transform[e];
    if (lg_skip(head, respect_done)) return;
    const int tid = threadIdx.x;
    const int ns = static_cast<int>((m + stride - 1) / stride);  // (<= kTileSeedSamples: the host's stride)
    const int ns4 = (ns + 3) & ~3;
    double o[D];  // f32 about the first target: clouds far from the origin keep their resolution
#pragma unroll
    for (int d = 0; d < D; d++) o[d] = dst[d];
    for (int k = tid; k < ns4; k += 256)
#pragma unroll
        for (int d = 0; d < D; d++)
            st[d][k] = k < ns ? static_cast<float>(dst[static_cast<int64_t>(k) * stride * D + d] - o[d]) : __builtin_inff();
    __syncthreads();
    // four lanes per query, each with every fourth group of four samples (a wave reads 64 contiguous bytes per row)
    const int part = tid & 3;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 64 + (tid >> 2);
    if (blockIdx.x == 0 && tid == 0) dev->nn_valid = 2;
    const bool valid = i < n;
    const int64_t ic = valid ? i : 0;
    // the query of lg_tile_nn_kernel
    double pnt[D], q[D];
#pragma unroll
    for (int d = 0; d < D; d++) pnt[d] = MODE == 0 ? cur[ic * D + d] : src[ic * D + d];
    if (MODE == 2 || MODE == 3 || MODE == 0) {
        const XfRows<D> xr = xf_rows<D>(xf);
        xf_apply<D>(xr, pnt, q);
    } else {
#pragma unroll
        for (int d = 0; d < D; d++) q[d] = pnt[d];
    }
    float qf[D];
#pragma unroll
    for (int d = 0; d < D; d++) qf[d] = static_cast<float>(q[d] - o[d]);
    float best = __builtin_inff();
    int bs = 0;
    for (int g = part * 4; g < ns4; g += 16) {
        float4 c[D];
#pragma unroll
        for (int d = 0; d < D; d++) c[d] = *reinterpret_cast<const float4*>(&st[d][g]);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < D; d++) {
                const float t = u == 0 ? c[d].x : (u == 1 ? c[d].y : (u == 2 ? c[d].z : c[d].w));
                const float df = t - qf[d];
                acc = fmaf(df, df, acc);
            }
            const bool take = acc < best;  // (NaN and inf never)
            best = take ? acc : best;
            bs = take ? g + u : bs;
        }
    }
#pragma unroll
    for (int w = 1; w <= 2; w <<= 1) {
        const float ob = __shfl_xor(best, w);
        const int os = __shfl_xor(bs, w);
        const bool take = ob < best || (ob == best && os < bs);
        best = take ? ob : best;
        bs = take ? os : bs;
    }
    if (valid && part == 0) {
        if (!(best < __builtin_inff())) dev->seed_bad = 1;
        nb_idx[i] = static_cast<int64_t>(bs) * stride;
    }
}

// (d2, index) minimum over the splits, then the epilogue of lg_nn_kernel
template <int D, int METHOD, int MODE>
__global__ void __launch_bounds__(256)
lg_tile_merge_kernel(LargeDev* dev, int respect_done, int64_t n, int splits, const double* __restrict__ part_d2,
                     const int32_t* __restrict__ part_idx, int64_t* __restrict__ nb_idx, double* __restrict__ nb_cost,
                     uint8_t* __restrict__ sel, double max_distance, const double* __restrict__ cloud,
                     uint32_t* __restrict__ kd_list, double* __restrict__ nb_pt) {
    const LgHead head = lg_head(dev);
    constexpr bool TIES = (D == 3 || METHOD == ICPX_VANILLA) && MODE != 2;
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    // (the first split's result is fetched beside the head)
    double best = i < n ? part_d2[i] : 0.0;
    int32_t bi = i < n ? part_idx[i] : 0;
    if (lg_skip(head, respect_done)) return;
    if (i == 0) dev->nn_valid = 1;
    if (i >= n) return;
    bool tie = bi < 0;
    bi &= 0x7fffffff;
    // (the partial results of eight splits are fetched together, then merged in split order: one split per trip of
    // the loop was one memory round trip per split — 12 us of the kernel's 13 for queen.ply's 19 splits)
    // (24: queen's 19 splits in one round trip after the first split's; with 8 it took three: 10.2 -> 9.2 us per iteration)
    constexpr int kFetch = 24;
    for (int s0 = 1; s0 < splits; s0 += kFetch) {
        double dd[kFetch];
        int32_t jj[kFetch];
#pragma unroll
        for (int u = 0; u < kFetch; u++) {
            const int s = s0 + u < splits ? s0 + u : splits - 1;
            dd[u] = part_d2[static_cast<int64_t>(s) * n + i];
            jj[u] = part_idx[static_cast<int64_t>(s) * n + i];
        }
#pragma unroll
        for (int u = 0; u < kFetch; u++) {
            if (s0 + u < splits) {  // (uniform)
                const double d = dd[u];
                int32_t j = jj[u];
                tie = tie || j < 0;
                j &= 0x7fffffff;
                if (TIES && j != 0x7fffffff && bi != 0x7fffffff && j != bi && d < __builtin_inf() && kd_near(d, best)) {
                    // two splits' minima within the tie range: a tie unless the two targets coincide
                    bool same = d == best;
                    for (int e = 0; e < D && same; e++) same = cloud[static_cast<int64_t>(j) * D + e] == cloud[static_cast<int64_t>(bi) * D + e];
                    tie = tie || !same;
                }
                const bool take = d < best || (d == best && j < bi);
                best = take ? d : best;
                bi = take ? j : bi;
            }
        }
    }
    if (TIES && tie && best < __builtin_inf()) kd_list[atomicAdd(&dev->kd_count, 1)] = static_cast<uint32_t>(i);
    if (bi == 0x7fffffff) bi = 0;  // nothing finite anywhere (non-finite input): the reference's untouched index
    nb_idx[i] = bi;
#pragma unroll
    for (int e = 0; e < D; e++) nb_pt[i * D + e] = cloud[static_cast<int64_t>(bi) * D + e];
    if (MODE == 2) {
        nb_cost[i] = best;
    } else {
        double c = __builtin_sqrt(__builtin_sqrt(best));
        nb_cost[i] = c;
        sel[i] = (D == 3 && METHOD == ICPX_TRIMMED) ? (c <= max_distance ? 1 : 0) : 1;
    }
}

__global__ void lg_iota_kernel(int64_t n, uint32_t* __restrict__ v) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) v[i] = static_cast<uint32_t>(i);
}

// trimmed.cpp:57-62: the first new_n matches of the sorted order are kept
__global__ void lg_select_kernel(const LargeDev* dev, int respect_done, int64_t n, int64_t new_n,
                                 const uint32_t* __restrict__ order, uint8_t* __restrict__ sel) {
    if (lg_skip(lg_head(dev), respect_done)) return;
    int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r < n) sel[order[r]] = r < new_n ? 1 : 0;
}

// Partial sums cross workgroups (and XCDs, each with an L2 of its own) inside one launch: they are written and read
// with agent-scope accesses, which go to the coherence point instead of asking for the whole L2 to be written back /
// invalidated by a fence (measured: two __threadfence per workgroup cost 7 us per sum kernel).
ICPX_DEV void st_agent(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(__double_as_longlong(v)),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
ICPX_DEV double ld_agent(const double* p) {
    return __longlong_as_double(static_cast<long long>(__hip_atomic_load(
        reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
}
ICPX_DEV void st_agent_int(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
ICPX_DEV int ld_agent_int(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// TREE64 levels 1 and 2 over this block's 4096 points.  f(i, vals) gives point i's NV terms.
template <int NV, typename F>
ICPX_DEV void block_tree(F&& f, int64_t n, double* __restrict__ part, int nblk, double* gs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kBlkPts;
    const int64_t g_total = (n + 63) >> 6;
    // gather the terms of all four slots first: their (dependent) global loads overlap instead of
    // paying one round trip per slot
    double vals[kSlots][NV];
#pragma unroll
    for (int k = 0; k < kSlots; k++) {
        int64_t i = base + k * kBlk + tid;
        if (i < n) {
            f(i, vals[k]);
        } else {
#pragma unroll
            for (int v = 0; v < NV; v++) vals[k][v] = 0.0;
        }
    }
    // level 1: the four slots of a value go through the butterfly together (xchg_sum, icpx_math.h: at xor 32 and 16 two
    // slots share one exchange; same partners, same additions, same bits as a butterfly per slot): a lane ends with
    // the group sum of slot kk = 2 b4 + b5 — 7 additions per value instead of 24 (the sum kernels of a 10^6-point
    // cloud were bound by these wave-level instructions, not by memory: 38.7 -> ... us)
    static_assert(kSlots == 4, "block_tree: four slots per thread");
    {
        const int kk = ((lane >> 4) & 1) * 2 + ((lane >> 5) & 1);
        const int64_t g = (base >> 6) + kk * (kBlk >> 6) + wave;
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const double y0 = xchg_sum<32>(vals[0][v], vals[1][v]), y1 = xchg_sum<32>(vals[2][v], vals[3][v]);
            double s = xchg_sum<16>(y0, y1);
            s = s + dpp_xor<8>(s);
            s = s + dpp_xor<4>(s);
            s = s + dpp_xor<2>(s);
            s = s + dpp_xor<1>(s);
            if ((lane & 15) == 0 && g < g_total) gs[v * 64 + kk * (kBlk >> 6) + wave] = s;
        }
    }
    __syncthreads();
    if (g_total == 1) {
        if (tid < NV) st_agent(&part[tid * nblk], gs[tid * 64]);
    } else {
        // level 2: value v by wave v (NV <= 16 waves)
        static_assert(NV <= kBlk / 64, "block_tree: one wave per value");
        const int64_t gl = g_total - (base >> 6);  // groups in this block
        if (wave < NV) {
            const double x = lane < gl ? gs[wave * 64 + lane] : 0.0;
            const double s = wave_sum(x);
            if (lane == 0) st_agent(&part[wave * nblk + blockIdx.x], s);
        }
    }
}

// TREE64 level 1 only: one 64-point group per wave, part[v * G1 + group]
template <int NV, typename F>
ICPX_DEV void group_tree(F&& f, int64_t n, double* __restrict__ part, int G1) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= G1) return;
    const int64_t i = static_cast<int64_t>(g) * 64 + lane;
    double vals[NV];
    if (i < n) {
        f(i, vals);
    } else {
#pragma unroll
        for (int v = 0; v < NV; v++) vals[v] = 0.0;
    }
#pragma unroll
    for (int v = 0; v < NV; v++) {
        double s = wave_sum(vals[v]);
        if (lane == 0) st_agent(&part[static_cast<int64_t>(v) * G1 + g], s);
    }
}

// remaining TREE64 levels over the per-block (or per-group) partials; every thread gets the total.
// NV values at once (value v's partials at part + v * nblk): the (value, group of 64 partials) pairs are spread over the
// block's waves and every wave finishes all NV second levels — the same additions as NV calls of fin_tree, with one
// pair of barriers and the loads of all values in flight together instead of a round trip per value.
// lds: NV * 64 doubles.
template <int NV>
ICPX_DEV void fin_tree_many(const double* __restrict__ part, int nblk, double* lds, double (&out)[NV]) {
    if (nblk == 1) {
#pragma unroll
        for (int v = 0; v < NV; v++) out[v] = ld_agent(&part[v]);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int G = (nblk + 63) >> 6;
    for (int t = wave; t < NV * G; t += W) {
        const int v = t / G, g = t - v * G;
        const int j = g * 64 + lane;
        const double x = j < nblk ? ld_agent(&part[static_cast<int64_t>(v) * nblk + j]) : 0.0;
        const double sum = wave_sum(x);
        if (lane == 0) lds[v * 64 + g] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; v++) {
        if (G == 1) {
            out[v] = lds[v * 64];
        } else {
            const double x = lane < G ? lds[v * 64 + lane] : 0.0;
            out[v] = wave_sum(x);
        }
    }
    __syncthreads();
}

ICPX_DEV double fin_tree(const double* __restrict__ part, int nblk, double* lds) {
    if (nblk == 1) return ld_agent(&part[0]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int G = (nblk + 63) >> 6;
    for (int g = wave; g < G; g += W) {
        int j = g * 64 + lane;
        double x = j < nblk ? ld_agent(&part[j]) : 0.0;
        double s = wave_sum(x);
        if (lane == 0) lds[g] = s;
    }
    __syncthreads();
    double r;
    if (G == 1) {
        r = lds[0];
    } else {
        double x = lane < G ? lds[lane] : 0.0;
        r = wave_sum(x);
    }
    __syncthreads();
    return r;
}

// ---- FeatureAware (lib/icp/impl/feature_aware.cpp), ICP2 --------------------------------------
ICPX_DEV double key_to_f64(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
    return __longlong_as_double(static_cast<long long>(b));
}

// a_current = transform * a (feature_aware.cpp:46,64); also clears the distance maximum of the
// matching step that follows
__global__ void fa_apply_kernel(LargeDev* dev, int respect_done, const double* __restrict__ src,
                                double* __restrict__ cur, int64_t n) {
    if (lg_skip(lg_head(dev), respect_done)) return;
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i == 0) dev->dmax_key = 0;
    if (i >= n) return;
    Xf<2> X;
#pragma unroll
    for (int e = 0; e < 9; e++) X.m[e] = dev->st.transform[e];
    XfRows<2> xr = xf_rows<2>(X);
    double p[2] = {src[i * 2], src[i * 2 + 1]}, q[2];
    xf_apply<2>(xr, p, q);
    cur[i * 2] = q[0];
    cur[i * 2 + 1] = q[1];
}

// get_centroid (geo.h:33-37) in the TREE64 order: block partials, then the remaining levels
__global__ void __launch_bounds__(kBlk)
fa_centroid_part_kernel(const double* __restrict__ pts, int64_t cnt, double* __restrict__ part, int nblk) {
    __shared__ double gs[2 * 64];
    block_tree<2>([&](int64_t i, double (&v)[2]) {
        v[0] = pts[i * 2];
        v[1] = pts[i * 2 + 1];
    }, cnt, part, nblk, gs);
}
__global__ void __launch_bounds__(kBlk)
fa_centroid_fin_kernel(LargeDev* dev, const double* __restrict__ part, int nblk, int64_t cnt) {
    __shared__ double lds[64];
    double sx = fin_tree(part, nblk, lds);
    double sy = fin_tree(part + nblk, nblk, lds);
    if (threadIdx.x == 0) {
        dev->fcent[0] = sx / static_cast<double>(cnt);
        dev->fcent[1] = sy / static_cast<double>(cnt);
    }
}
// |p_i - centroid| (feature_aware.cpp:159)
__global__ void fa_radius_kernel(const LargeDev* dev, const double* __restrict__ pts, int64_t cnt,
                                 double* __restrict__ rad) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    double dx = pts[i * 2] - dev->fcent[0], dy = pts[i * 2 + 1] - dev->fcent[1];
    rad[i] = __builtin_sqrt(dx * dx + dy * dy);
}
// feature_aware.cpp:157-174: feature r of point i, stored at feat[r * cnt + i]
__global__ void fa_features_kernel(const double* __restrict__ rad, int64_t cnt, int sn, double* __restrict__ feat) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const double ri = rad[i];
    const int64_t lower = i - sn > 0 ? i - sn : 0;
    const int64_t upper = i + sn < cnt - 1 ? i + sn : cnt - 1;
    for (int r = 0; r < sn; r++) {
        int64_t j = lower + r;
        feat[static_cast<int64_t>(r) * cnt + i] = j < i ? rad[j] - ri : 0.0;
        int64_t j2 = i + 1 + r;
        feat[static_cast<int64_t>(sn + r) * cnt + i] = j2 <= upper ? rad[j2] - ri : 0.0;
    }
}
// compute_norm_dists<Dynamic>(a_features, b_features) (feature_aware.h:33-44): |fa_i - fb_j|, terms
// added in row order; stored at nfd[j * n + i]; running maximum (feature_aware.cpp:55)
constexpr int kFaJ = 8;  // targets per thread and launch row
__global__ void __launch_bounds__(256)
fa_fdist_kernel(LargeDev* dev, const double* __restrict__ fa, const double* __restrict__ fb, int64_t n, int64_t m, int w,
                double* __restrict__ nfd) {
    __shared__ unsigned long long smax[4];
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t j0 = static_cast<int64_t>(blockIdx.y) * kFaJ;
    unsigned long long mk = 0;
    if (i < n) {
        for (int jj = 0; jj < kFaJ; jj++) {
            int64_t j = j0 + jj;
            if (j >= m) break;
            double acc = 0.0;
            for (int r = 0; r < w; r++) {
                double df = fa[static_cast<int64_t>(r) * n + i] - fb[static_cast<int64_t>(r) * m + j];
                acc = acc + df * df;
            }
            double v = __builtin_sqrt(acc);
            nfd[j * n + i] = v;
            if (v == v) {
                unsigned long long k = f64_key(v);
                mk = k > mk ? k : mk;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        unsigned long long other = __shfl_xor(mk, o);
        mk = other > mk ? other : mk;
    }
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mk;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < 4; wv++) mk = smax[wv] > mk ? smax[wv] : mk;
        if (mk) atomicMax(&dev->fmax_key, mk);
    }
}
// normalized_feature_dists /= max when max > MIN_NORM_THRESHOLD (feature_aware.cpp:56-58)
__global__ void fa_normalize_kernel(const LargeDev* dev, double* __restrict__ nfd, int64_t total) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total || dev->fmax_key == 0) return;
    const double mx = key_to_f64(dev->fmax_key);
    if (mx > 1e-6) nfd[i] = nfd[i] / mx;
}
// maxCoeff of compute_norm_dists<2>(a_current, b) (feature_aware.cpp:130-131): sqrt is monotone, so
// the maximum squared distance is tracked
__global__ void __launch_bounds__(256)
fa_maxdist_kernel(LargeDev* dev, int respect_done, const double* __restrict__ cur, int64_t n,
                  const double* __restrict__ dst, int64_t m, int split_len) {
    __shared__ unsigned long long smax[4];
    if (lg_skip(lg_head(dev), respect_done)) return;
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t t0 = static_cast<int64_t>(blockIdx.y) * split_len;
    const int64_t t1 = t0 + split_len < m ? t0 + split_len : m;
    double mx = -1.0;
    if (i < n) {
        const double qx = cur[i * 2], qy = cur[i * 2 + 1];
        for (int64_t j = t0; j < t1; j++) {
            double dx = qx - dst[j * 2], dy = qy - dst[j * 2 + 1];
            double d2 = dx * dx + dy * dy;
            mx = d2 > mx ? d2 : mx;
        }
    }
    unsigned long long mk = mx >= 0.0 ? f64_key(mx) : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        unsigned long long other = __shfl_xor(mk, o);
        mk = other > mk ? other : mk;
    }
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mk;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < 4; wv++) mk = smax[wv] > mk ? smax[wv] : mk;
        if (mk) atomicMax(&dev->dmax_key, mk);
    }
}
// FeatureAware::compute_matches (feature_aware.cpp:129-149) over one target split: cost =
// neighbor_weight * dist / max + feature_weight * normalized_feature_dist, strict <, ascending j
__global__ void __launch_bounds__(256)
fa_match_kernel(const LargeDev* dev, int respect_done, const double* __restrict__ cur, int64_t n,
                const double* __restrict__ dst, int64_t m, int split_len, const double* __restrict__ nfd,
                double feature_weight, double* __restrict__ part_d2, int32_t* __restrict__ part_idx) {
    if (lg_skip(lg_head(dev), respect_done)) return;
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t t0 = static_cast<int64_t>(blockIdx.y) * split_len;
    const int64_t t1 = t0 + split_len < m ? t0 + split_len : m;
    const double nw = 1.0 - feature_weight;  // feature_aware.cpp:31
    const double mx = dev->dmax_key ? __builtin_sqrt(key_to_f64(dev->dmax_key)) : 0.0;
    const bool norm = mx > 1e-6;
    const double qx = cur[i * 2], qy = cur[i * 2 + 1];
    double best = __builtin_inf();
    int32_t bi = 0x7fffffff;
    for (int64_t j = t0; j < t1; j++) {
        double dx = qx - dst[j * 2], dy = qy - dst[j * 2 + 1];
        double dist = __builtin_sqrt(dx * dx + dy * dy);
        if (norm) dist = dist / mx;
        double cost = nw * dist + feature_weight * nfd[j * n + i];
        if (cost < best) {
            best = cost;
            bi = static_cast<int32_t>(j);
        }
    }
    part_d2[static_cast<int64_t>(blockIdx.y) * n + i] = best;
    part_idx[static_cast<int64_t>(blockIdx.y) * n + i] = bi;
}

// The last workgroup of a sum kernel to have written its partials finishes the sum (what used to be a launch of its
// own): every thread waits for its (agent-scope) partial stores, one atomic ticket per workgroup tells which one is
// last, and that one reads the other workgroups' partials with agent-scope loads.
ICPX_DEV bool last_block_done(int* tick, int* s_flag) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (this thread's agent-scope stores have completed)
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(tick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == static_cast<int>(gridDim.x) - 1 ? 1 : 0;
        if (last) __hip_atomic_store(tick, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0;
}

// centroids, selected count and cost sum from the partials (vanilla.cpp:35-36 / trimmed.cpp:64-65, icp.h:149-155)
template <int D>
ICPX_DEV void fin1_body(LargeDev* dev, const double* __restrict__ part, const int* __restrict__ cntp, int nblk, double* lds, int* wcnt) {
    double s[2 * D + 1];
    fin_tree_many<2 * D + 1>(part, nblk, lds, s);
    int c = 0;  // <= 2^24 selected points in all
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) c += ld_agent_int(&cntp[b]);
    c = wave_sum_int(c);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long k = 0;
        for (int w = 0; w < static_cast<int>(blockDim.x >> 6); w++) k += wcnt[w];
        double kd = static_cast<double>(k);
        for (int d = 0; d < D; d++) {
            dev->ca[d] = s[d] / kd;
            dev->cb[d] = s[D + d] / kd;
        }
        dev->ksel = kd;
        dev->cost_sum = s[2 * D];
    }
}

template <int D, int METHOD>
__global__ void __launch_bounds__(kBlk)
lg_sum1_kernel(LargeDev* dev, int respect_done, const double* __restrict__ src, double* __restrict__ cur,
               const double* __restrict__ nb_pt, int64_t n, const double* __restrict__ nb_cost,
               const uint8_t* __restrict__ sel, double* __restrict__ part, int* __restrict__ cntp, int nblk, int groups,
               int* hflags, int sd_flags, double max_distance) {
    // sd_flags (3D): bit 0 — the cloud in point order takes the step here (the correspondence kernel worked on a
    // stepped copy of its own, or stepped the points on the fly); bit 1 — the selection comes from the two points
    const bool sd_step = (sd_flags & 1) != 0, sd = (sd_flags & 2) != 0;
    __shared__ double gs[(2 * D + 1) * 64];
    __shared__ int wcnt[kBlk / 64];
    __shared__ int s_last;
    const LgHead head = lg_head(dev);
    const int head_with_cost = dev->with_cost;
    Xf<D> X, head_step;
#pragma unroll
    for (int e = 0; e < (D + 1) * (D + 1); e++) {
        X.m[e] = dev->st.transform[e];
        head_step.m[e] = D == 3 ? dev->step[e] : 0.0;
    }
    if (lg_skip(head, respect_done)) return;
    if (lg_kd_gate(head, dev, hflags)) return;  // flagged queries of the kd rule and no tree on the device yet: stall
    XfRows<D> xr = xf_rows<D>(X);
    const bool with_cost = head_with_cost != 0;
    // sorted domain (LargeIcp::sd, 3D): the correspondence kernel stepped its own copy of the cloud; the cloud in point
    // order takes the same step here (c = T * c, vanilla_3d.cpp:91), and the selection comes from the two points
    XfRows<D> sr;
    if (D == 3 && sd_step) sr = xf_rows<D>(head_step);
    // the point (as the sums take it) and whether its match is selected
    auto point_of = [&](int64_t i, double (&c)[D], double (&b)[D]) -> bool {
            double p[D];
            if (D == 2) {
#pragma unroll
                for (int d = 0; d < D; d++) p[d] = src[i * D + d];
                xf_apply<D>(xr, p, c);  // a_current = transform * a, vanilla.cpp:33
            } else {
#pragma unroll
                for (int d = 0; d < D; d++) c[d] = cur[i * D + d];
                if (sd_step) {
#pragma unroll
                    for (int d = 0; d < D; d++) p[d] = c[d];
                    xf_apply<D>(sr, p, c);
#pragma unroll
                    for (int d = 0; d < D; d++) cur[i * D + d] = c[d];
                }
            }
#pragma unroll
            for (int d = 0; d < D; d++) b[d] = nb_pt[i * D + d];
            if constexpr (D == 3) {
                if (sd) return sd_select<METHOD>(c, b, max_distance);
            }
            return sel[i] != 0;
        };
    int my_sel = 0;  // selected among the points this thread took
    auto terms = [&](int64_t i, double (&v)[2 * D + 1]) {
            double c[D], b[D];
            const bool s = point_of(i, c, b);
            my_sel += s ? 1 : 0;
#pragma unroll
            for (int d = 0; d < D; d++) {
                v[d] = s ? c[d] : 0.0;
                v[D + d] = s ? b[d] : 0.0;
            }
            v[2 * D] = with_cost ? nb_cost[i] : 0.0;
        };
    if (groups) {
        const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        group_tree<2 * D + 1>(terms, n, part, nblk);
        if (g < nblk) {
            const int c = wave_sum_int(my_sel);
            if ((threadIdx.x & 63) == 0) st_agent_int(&cntp[g], c);
        }
    } else {
        block_tree<2 * D + 1>(terms, n, part, nblk, gs);
        int cnt = wave_sum_int(my_sel);
        if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < kBlk / 64; w++) t += wcnt[w];
            st_agent_int(&cntp[blockIdx.x], t);
        }
    }
    if (!last_block_done(&dev->tick1, &s_last)) return;
    fin1_body<D>(dev, part, cntp, nblk, gs, wcnt);
}

// cross-covariance partials; the last workgroup then takes the Kabsch step, updates the transform and evaluates
// ICPDriver::should_terminate (fin2_body)
template <int D, int METHOD>
ICPX_DEV void fin2_body(LargeDev* dev, const double* __restrict__ part, int nblk, double n_points, int check,
                        const icpx_criteria& crit, double clock_to_s, double host_elapsed_s, int* hflags, double* lds);

template <int D, int METHOD>
__global__ void __launch_bounds__(kBlk)
lg_sum2_kernel(LargeDev* dev, int respect_done, const double* __restrict__ src, const double* __restrict__ cur,
               const double* __restrict__ nb_pt, int64_t n, const uint8_t* __restrict__ sel, double* __restrict__ part,
               int nblk, int groups, double n_points, int check, icpx_criteria crit, double clock_to_s, double host_elapsed_s,
               int* hflags, int sd_flags, double max_distance) {
    const bool sd = (sd_flags & 2) != 0;
    __shared__ double gs[D * D * 64];
    __shared__ int s_last;
    const LgHead head = lg_head(dev);
    Xf<D> X;
#pragma unroll
    for (int e = 0; e < (D + 1) * (D + 1); e++) X.m[e] = dev->st.transform[e];
    double ca[D], cb[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        ca[d] = dev->ca[d];
        cb[d] = dev->cb[d];
    }
    if (lg_skip(head, respect_done)) return;
    XfRows<D> xr = xf_rows<D>(X);
    auto terms = [&](int64_t i, double (&v)[D * D]) {
            double p[D], c[D], b[D];
            if (D == 2) {
#pragma unroll
                for (int d = 0; d < D; d++) p[d] = src[i * D + d];
                xf_apply<D>(xr, p, c);
            } else {
#pragma unroll
                for (int d = 0; d < D; d++) c[d] = cur[i * D + d];
            }
#pragma unroll
            for (int d = 0; d < D; d++) b[d] = nb_pt[i * D + d];
            // (sorted domain: lg_sum1 has stepped the cloud; the selection comes from the two points)
            bool s = false;
            if constexpr (D == 3) {
                if (sd) s = sd_select<METHOD>(c, b, max_distance);
            }
            if (!(D == 3 && sd)) s = sel[i] != 0;
#pragma unroll
            for (int r = 0; r < D; r++)
#pragma unroll
                for (int cc = 0; cc < D; cc++) v[cc * D + r] = s ? (c[r] - ca[r]) * (b[cc] - cb[cc]) : 0.0;
        };
    if (groups)
        group_tree<D * D>(terms, n, part, nblk);
    else
        block_tree<D * D>(terms, n, part, nblk, gs);
    if (!last_block_done(&dev->tick2, &s_last)) return;
    fin2_body<D, METHOD>(dev, part, nblk, n_points, check, crit, clock_to_s, host_elapsed_s, hflags, gs);
}

// done flag and iteration count into host memory (pinned, coherent) for large_converge, which polls them instead of
// copying the flag and synchronising the stream after every chunk of iterations
ICPX_DEV void publish_progress(int* hflags, int done, int iterations) {
    if (!hflags) return;
    if (done) __hip_atomic_store(&hflags[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&hflags[1], iterations, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// setup tail: ICP::calculate_cost of the initial matches, then the loop's first should_terminate
template <int D>
__global__ void lg_setup_finish_kernel(LargeDev* dev, int64_t n, int check, icpx_criteria crit,
                                       double clock_to_s, double host_elapsed_s, int* hflags) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (dev->kd_stall) return;
    dev->kd_count = 0;
    double sum = dev->with_cost ? dev->cost_sum : 0.0;
    dev->st.cost = __builtin_sqrt(sum / static_cast<double>(n));  // icp.h:149-155
    if (check) {
        double el = host_elapsed_s + static_cast<double>(wall_clock64() - dev->st.start_clock) * clock_to_s;
        if (should_terminate<D>(crit, dev->st, el)) {
            dev->st.done = 1;
            dev->done_m = 1;
        } else {
            for (int e = 0; e < 16; e++) dev->st.last_transform[e] = dev->st.transform[e];
            dev->st.last_cost = dev->st.cost;
            dev->st.have_last = 1;
        }
    }
    publish_progress(hflags, dev->st.done, static_cast<int>(dev->st.iteration_count));
}

template <int D, int METHOD>
ICPX_DEV void fin2_body(LargeDev* dev, const double* __restrict__ part, int nblk, double n_points, int check,
                        const icpx_criteria& crit, double clock_to_s, double host_elapsed_s, int* hflags, double* lds) {
    constexpr int XN = (D + 1) * (D + 1);
    double s[D * D];
    fin_tree_many<D * D>(part, nblk, lds, s);
    if (threadIdx.x == 0) {
        Mat<D> N;
#pragma unroll
        for (int e = 0; e < D * D; e++) N.a[e] = s[e];
        double ca[D], cb[D];
        for (int d = 0; d < D; d++) {
            ca[d] = dev->ca[d];
            cb[d] = dev->cb[d];
        }
        Xf<D> step = kabsch<D>(N, ca, cb);
        Xf<D> X;
        for (int e = 0; e < XN; e++) X.m[e] = dev->st.transform[e];
        Xf<D> nx = xf_mul<D>(step, X);
        for (int e = 0; e < XN; e++) {
            dev->st.transform[e] = nx.m[e];
            dev->step[e] = step.m[e];
        }
        dev->st.iteration_count += 1;
        dev->kd_count = 0;
        // cost: 2D vanilla's matches never change (vanilla.cpp:80); ICP3 never fills them (Q4);
        // trimmed ICP2 rates the matches this iteration computed (icp.h:149-155)
        if (dev->update_cost) dev->st.cost = __builtin_sqrt(dev->cost_sum / n_points);
        if (check) {
            double el = host_elapsed_s + static_cast<double>(wall_clock64() - dev->st.start_clock) * clock_to_s;
            if (should_terminate<D>(crit, dev->st, el)) {
                dev->st.done = 1;
                dev->done_m = 1;
            } else {
                for (int e = 0; e < 16; e++) dev->st.last_transform[e] = dev->st.transform[e];
                dev->st.last_cost = dev->st.cost;
                dev->st.have_last = 1;
            }
        }
        publish_progress(hflags, dev->st.done, static_cast<int>(dev->st.iteration_count));
    }
}

// matches in `order` (sorted, trimmed ICP2 after an iterate) or in point order (order == null)
__global__ void lg_matches_kernel(int64_t n, const uint32_t* __restrict__ order, const int64_t* __restrict__ nb_idx,
                                  const double* __restrict__ nb_cost, int64_t* __restrict__ point,
                                  int64_t* __restrict__ pair, double* __restrict__ cost) {
    int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int64_t i = order ? static_cast<int64_t>(order[r]) : r;
    point[r] = i;
    pair[r] = nb_idx[i];
    cost[r] = nb_cost[i];
}

}  // namespace

// ------------------------------------------------------------------------------------------
// grid host side
// ------------------------------------------------------------------------------------------

int grid_create(GridIndex** out) {
    *out = new (std::nothrow) GridIndex();
    return *out ? ICPX_OK : fail(ICPX_ERR_NOMEM, "out of host memory");
}

void grid_destroy(GridIndex* g) {
    if (!g) return;
    for (DevBuf* b : {&g->filt, &g->recs, &g->start, &g->cell_id, &g->counts, &g->keys, &g->totals, &g->q_keys, &g->q_keys_out,
                      &g->q_iota, &g->q_perm, &g->q_tmp, &g->q_start})
        b->release();
    if (g->occ_host) (void)hipHostFree(g->occ_host);
    if (g->keys_host) (void)hipHostFree(g->keys_host);
    if (g->occ_ev) (void)hipEventDestroy(g->occ_ev);
    delete g;
}

// The cell budget could not bring the occupied cells below this many targets on average: the cloud sits in a
// few spots of a large box (far-apart clusters), every query would scan thousands of records — the BVH's case.
constexpr double kGridDegenerate = 48.0;
bool grid_degenerate(const GridIndex* g) { return g->per_occupied > kGridDegenerate; }

void grid_set_f32(GridIndex* g, bool on) { g->f32 = on; }
void grid_set_filter(GridIndex* g, bool on) { g->want_filt = on; }

int grid_count_rounds(GridIndex* g, hipStream_t stream, int first, bool defer);
int grid_scan_scatter(GridIndex* g, hipStream_t stream);
bool grid_wants_finer(GridIndex* g, int occ);

int grid_build(GridIndex* g, int dim, const double* d_t, int64_t m, hipStream_t stream, bool defer) {
    g->dim = dim;
    g->occ_pending = false;
    GridParams& gp = g->gp;
    std::memset(&gp, 0, sizeof(gp));
    gp.m = m;
    for (int d = 0; d < 3; d++) {
        gp.g[d] = 1;
        gp.h[d] = 1.0;
        gp.inv_h[d] = 1.0;
    }
    if (m <= 0) {
        g->ncell = 1;
        ICPX_TRY(g->start.reserve(sizeof(int) * 2));
        ICPX_HIP(hipMemsetAsync(g->start.p, 0, sizeof(int) * 2, stream));
        ICPX_TRY(g->recs.reserve(sizeof(GridRec)));
        return ICPX_OK;
    }
    // 1. bounding box
    ICPX_TRY(g->keys.reserve(sizeof(unsigned long long) * 6));
    unsigned long long init[6] = {~0ULL, ~0ULL, ~0ULL, 0, 0, 0};
    ICPX_HIP(hipMemcpyAsync(g->keys.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
    // (few, large workgroups: the kernel ends in six 64-bit atomics per workgroup on the same six words, and those
    // serialise — 512 workgroups: 20 us for a million points, 2048: 52 us)
    int blocks = static_cast<int>(std::min<int64_t>((m + 1023) / 1024, 128));
    if (dim == 2)
        hipLaunchKernelGGL(grid_bbox_kernel<2>, dim3(blocks), dim3(1024), 0, stream, d_t, m, g->keys.as<unsigned long long>());
    else
        hipLaunchKernelGGL(grid_bbox_kernel<3>, dim3(blocks), dim3(1024), 0, stream, d_t, m, g->keys.as<unsigned long long>());
    ICPX_HIP(hipGetLastError());
    if (!g->keys_host) ICPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&g->keys_host), sizeof(unsigned long long) * 6, hipHostMallocDefault));
    unsigned long long* keys = g->keys_host;
    ICPX_HIP(hipMemcpyAsync(keys, g->keys.p, sizeof(unsigned long long) * 6, hipMemcpyDeviceToHost, stream));
    ICPX_HIP(hipStreamSynchronize(stream));
    // 2. cell size: about 2 targets per cell over the non-degenerate axes
    GridIndex::Build& bs = g->b;
    bs.d_t = d_t;
    bs.m = m;
    bs.nz = 0;
    double vol = 1.0;
    bool any = keys[0] != ~0ULL;
    for (int d = 0; d < 3; d++) bs.ext[d] = 0.0;
    for (int d = 0; d < dim; d++) {
        gp.lo[d] = any ? f64_unkey(keys[d]) : 0.0;
        gp.hi[d] = any ? f64_unkey(keys[3 + d]) : 0.0;
        bs.ext[d] = gp.hi[d] - gp.lo[d];
        if (bs.ext[d] > 0.0 && std::isfinite(bs.ext[d])) {
            bs.nz++;
            vol *= bs.ext[d];
        } else {
            bs.ext[d] = 0.0;
        }
    }
    const double per_cell = knobs().grid_density.load() > 0 ? knobs().grid_density.load() * 0.1 : 2.0;  // development knob ICPX_GRID_DENSITY
    const double target_cells = std::max<double>(1.0, static_cast<double>(m) / per_cell);
    bs.h = 1.0;
    if (bs.nz > 0) {
        bs.h = std::pow(vol / target_cells, 1.0 / bs.nz);
        if (!(bs.h > 0.0) || !std::isfinite(bs.h)) bs.h = 1.0;
    }
    bs.max_cells = std::min<int64_t>(16 * m + 64, int64_t(1) << 26);
    ICPX_TRY(g->cell_id.reserve(sizeof(int) * m));
    ICPX_TRY(g->recs.reserve(sizeof(GridRec) * m));
    g->occ_pending = false;
    if (defer) {
        if (!g->occ_host) ICPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&g->occ_host), sizeof(int), hipHostMallocDefault));
        if (!g->occ_ev) ICPX_HIP(hipEventCreateWithFlags(&g->occ_ev, hipEventDisableTiming));
    }
    defer = defer && !g->refined_last;  // clouds that needed a finer round last time are counted the patient way
    g->refined_last = false;
    ICPX_TRY(grid_count_rounds(g, stream, 0, defer));
    return grid_scan_scatter(g, stream);
}

// 3. count per cell; surfaces and clusters leave most cells empty, so the edge is halved while the OCCUPIED cells
//    hold more than ~6 targets on average and the cell budget allows.  Rounds first .. 4; with `defer` round 0 does
//    not wait for its occupied-cell count (grid_settle does).
int grid_count_rounds(GridIndex* g, hipStream_t stream, int first, bool defer) {
    GridIndex::Build& bs = g->b;
    GridParams& gp = g->gp;
    const int dim = g->dim;
    const int64_t m = bs.m;
    const unsigned mb = static_cast<unsigned>((m + 255) / 256);
    for (int refine = first; refine < 5; refine++) {
        int64_t total = 1;
        for (int attempt = 0; attempt < 64; attempt++) {
            total = 1;
            for (int d = 0; d < dim; d++) {
                double cnt = bs.ext[d] > 0.0 ? std::floor(bs.ext[d] / bs.h) + 1.0 : 1.0;
                if (cnt > 2.0e6) cnt = 2.0e6;
                gp.g[d] = static_cast<int>(cnt);
                total *= gp.g[d];
            }
            if (total <= bs.max_cells) break;
            bs.h *= 1.26;  // ~ 2^(1/3): halve the cell count per retry in 3D
        }
        if (total > bs.max_cells) return fail(ICPX_ERR_UNSUPPORTED, "grid: degenerate cloud extents");
        for (int d = 0; d < dim; d++) {
            gp.h[d] = bs.h;
            gp.inv_h[d] = 1.0 / bs.h;
        }
        gp.rmax = std::max(gp.g[0], std::max(gp.g[1], gp.g[2])) - 1;
        g->ncell = static_cast<int>(total);
        ICPX_TRY(g->counts.reserve(sizeof(int) * (total + 2)));
        ICPX_HIP(hipMemsetAsync(g->counts.p, 0, sizeof(int) * (total + 2), stream));
        int* d_occ = g->counts.as<int>() + total + 1;
        if (dim == 2)
            hipLaunchKernelGGL(grid_count_kernel<2>, dim3(mb), dim3(256), 0, stream, bs.d_t, gp, g->cell_id.as<int>(), g->counts.as<int>());
        else
            hipLaunchKernelGGL(grid_count_kernel<3>, dim3(mb), dim3(256), 0, stream, bs.d_t, gp, g->cell_id.as<int>(), g->counts.as<int>());
        ICPX_HIP(hipGetLastError());
        g->occupied_frac = 0.0;
        g->per_occupied = static_cast<double>(m);  // (all targets in one point: one cell)
        if (bs.nz == 0) break;
        // (one atomic per workgroup on one word: few workgroups)
        hipLaunchKernelGGL(grid_occupied_kernel, dim3(static_cast<unsigned>(std::min<int64_t>((total + 4095) / 4096, 128))), dim3(256), 0,
                           stream, g->counts.as<int>(), static_cast<int>(total), d_occ);
        ICPX_HIP(hipGetLastError());
        if (defer && refine == 0) {
            ICPX_HIP(hipMemcpyAsync(g->occ_host, d_occ, sizeof(int), hipMemcpyDeviceToHost, stream));
            ICPX_HIP(hipEventRecord(g->occ_ev, stream));
            g->occ_pending = true;
            return ICPX_OK;
        }
        int occ = 0;
        ICPX_HIP(hipMemcpyAsync(&occ, d_occ, sizeof(int), hipMemcpyDeviceToHost, stream));
        ICPX_HIP(hipStreamSynchronize(stream));
        if (!grid_wants_finer(g, occ)) break;
        bs.h *= 0.5;
        g->refined_last = true;
    }
    return ICPX_OK;
}

// occupancy of the round just counted; true: the grid fits and should be counted again with half the edge
bool grid_wants_finer(GridIndex* g, int occ) {
    const GridIndex::Build& bs = g->b;
    const int64_t total = g->ncell;
    g->occupied_frac = static_cast<double>(occ) / static_cast<double>(total);
    g->per_occupied = occ > 0 ? static_cast<double>(bs.m) / occ : 0.0;
    if (occ <= 0 || static_cast<double>(bs.m) / occ <= 6.0) return false;
    double next_total = 1.0;
    for (int d = 0; d < g->dim; d++) next_total *= bs.ext[d] > 0.0 ? std::floor(bs.ext[d] / (0.5 * bs.h)) + 1.0 : 1.0;
    return next_total <= static_cast<double>(bs.max_cells);
}

// 4. exclusive scan -> cell starts, then scatter into cell order
int grid_scan_scatter(GridIndex* g, hipStream_t stream) {
    const int64_t total = g->ncell, m = g->b.m;
    const unsigned mb = static_cast<unsigned>((m + 255) / 256);
    ICPX_TRY(g->start.reserve(sizeof(int) * (total + 1)));
    const int nb = static_cast<int>((total + kScanBlk - 1) / kScanBlk);
    ICPX_TRY(g->totals.reserve(sizeof(int) * (nb + 1)));
    hipLaunchKernelGGL(grid_scan_totals_kernel, dim3(nb), dim3(256), 0, stream, g->counts.as<int>(), g->ncell, g->totals.as<int>());
    ICPX_HIP(hipGetLastError());
    hipLaunchKernelGGL(grid_scan_top_kernel, dim3(1), dim3(1024), 0, stream, g->totals.as<int>(), nb);
    ICPX_HIP(hipGetLastError());
    hipLaunchKernelGGL(grid_scan_local_kernel, dim3(nb), dim3(256), 0, stream, g->counts.as<int>(), g->ncell, g->totals.as<int>(), g->start.as<int>());
    ICPX_HIP(hipGetLastError());
    ICPX_HIP(hipMemsetAsync(g->counts.p, 0, sizeof(int) * (total + 2), stream));  // reuse as cursors
    if (g->f32) {
        if (g->dim == 2)
            hipLaunchKernelGGL((grid_scatter_kernel<2, GridRec32>), dim3(mb), dim3(256), 0, stream, g->b.d_t, m, g->cell_id.as<int>(),
                               g->start.as<int>(), g->counts.as<int>(), g->recs.as<GridRec32>(), g->gp);
        else
            hipLaunchKernelGGL((grid_scatter_kernel<3, GridRec32>), dim3(mb), dim3(256), 0, stream, g->b.d_t, m, g->cell_id.as<int>(),
                               g->start.as<int>(), g->counts.as<int>(), g->recs.as<GridRec32>(), g->gp);
    } else {
        GridRec32* filt = nullptr;
        if (g->want_filt) {
            ICPX_TRY(g->filt.reserve(sizeof(GridRec32) * m));
            filt = g->filt.as<GridRec32>();
        }
        if (g->dim == 2)
            hipLaunchKernelGGL((grid_scatter_kernel<2, GridRec>), dim3(mb), dim3(256), 0, stream, g->b.d_t, m, g->cell_id.as<int>(),
                               g->start.as<int>(), g->counts.as<int>(), g->recs.as<GridRec>(), g->gp, filt);
        else
            hipLaunchKernelGGL((grid_scatter_kernel<3, GridRec>), dim3(mb), dim3(256), 0, stream, g->b.d_t, m, g->cell_id.as<int>(),
                               g->start.as<int>(), g->counts.as<int>(), g->recs.as<GridRec>(), g->gp, filt);
    }
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

// After grid_build(defer = true) and whatever the caller enqueued behind it: reads the first round's occupied-cell
// count.  *rebuilt = the grid was counted again with a finer cell and rebuilt (the work enqueued behind the first
// build used the coarser grid: the caller repeats it).
int grid_settle(GridIndex* g, hipStream_t stream, bool* rebuilt) {
    *rebuilt = false;
    if (!g->occ_pending) return ICPX_OK;
    g->occ_pending = false;
    ICPX_HIP(hipEventSynchronize(g->occ_ev));
    if (!grid_wants_finer(g, *g->occ_host)) return ICPX_OK;
    g->b.h *= 0.5;
    g->refined_last = true;
    ICPX_TRY(grid_count_rounds(g, stream, 1, false));
    *rebuilt = true;
    return grid_scan_scatter(g, stream);
}

// rocPRIM sorts up to 2^20 items by block sort + ~log2(n / 1024) merge passes; for keys of few bits the one-sweep
// radix sort (one pass per 8 bits) is the shorter chain of kernels, so it is asked for whatever the size.
using OneSweepAlways = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                  rocprim::default_config, 0>;

// Orders the queries by the grid cell they fall in (stable radix sort of the cell ids): the query
// kernels then visit them in that order.  Worth it from a few 10^4 queries on.
int grid_sort_queries(GridIndex* g, const double* d_q, int64_t n, hipStream_t stream) {
    g->q_perm_n = 0;
    g->q_blocks = 0;
    if (n < kGridSortMinQueries || g->gp.m <= 0) return ICPX_OK;
    ICPX_TRY(g->q_keys.reserve(sizeof(uint32_t) * n));
    ICPX_TRY(g->q_keys_out.reserve(sizeof(uint32_t) * n));
    ICPX_TRY(g->q_iota.reserve(sizeof(uint32_t) * n));
    ICPX_TRY(g->q_perm.reserve(sizeof(uint32_t) * n));
    const unsigned nb = static_cast<unsigned>((n + 255) / 256);
    if (g->dim == 2)
        hipLaunchKernelGGL(grid_query_keys_kernel<2>, dim3(nb), dim3(256), 0, stream, d_q, n, g->gp,
                           g->q_keys.as<uint32_t>(), g->q_iota.as<uint32_t>());
    else
        hipLaunchKernelGGL(grid_query_keys_kernel<3>, dim3(nb), dim3(256), 0, stream, d_q, n, g->gp,
                           g->q_keys.as<uint32_t>(), g->q_iota.as<uint32_t>());
    ICPX_HIP(hipGetLastError());
    int nt[3];
    grid_tiles(g->gp, g->dim, nt);
    const int64_t max_key = (static_cast<int64_t>(nt[0]) * nt[1] * nt[2]) << 5;  // tile order, grid_query_keys_kernel
    int bits = 1;
    while (bits < 32 && (int64_t(1) << bits) < max_key) bits++;
    size_t bytes = 0;
    ICPX_HIP(rocprim::radix_sort_pairs<OneSweepAlways>(nullptr, bytes, g->q_keys.as<uint32_t>(), g->q_keys_out.as<uint32_t>(),
                                                       g->q_iota.as<uint32_t>(), g->q_perm.as<uint32_t>(),
                                                       static_cast<unsigned>(n), 0u, static_cast<unsigned>(bits), stream));
    ICPX_TRY(g->q_tmp.reserve(bytes ? bytes : 1));
    ICPX_HIP(rocprim::radix_sort_pairs<OneSweepAlways>(g->q_tmp.p, bytes, g->q_keys.as<uint32_t>(), g->q_keys_out.as<uint32_t>(),
                                                       g->q_iota.as<uint32_t>(), g->q_perm.as<uint32_t>(),
                                                       static_cast<unsigned>(n), 0u, static_cast<unsigned>(bits), stream));
    g->q_perm_n = n;
    // ranges of the sorted order that fall into each block of four tiles (block-cooperative query)
    g->q_blocks = static_cast<int64_t>(nt[0] >> 2) * nt[1] * nt[2];
    ICPX_TRY(g->q_start.reserve(sizeof(int) * (g->q_blocks + 1)));
    hipLaunchKernelGGL(grid_block_starts_kernel, dim3(nb), dim3(256), 0, stream, g->q_keys_out.as<uint32_t>(), n,
                       static_cast<int>(g->q_blocks), g->q_start.as<int>());
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

int grid_query_launch(const GridIndex* g, const double* d_q, int64_t n, int64_t* d_idx, double* d_d2,
                      unsigned long long* d_visited, hipStream_t stream, const KdTies* ties) {
    if (n <= 0) return ICPX_OK;
    const uint32_t* perm = g->q_perm_n == n ? g->q_perm.as<uint32_t>() : nullptr;
    const double* cloud = ties ? ties->cloud : nullptr;
    uint32_t* kl = ties ? ties->list : nullptr;
    int* kc = ties ? ties->count : nullptr;
    if (g->f32) {
        // opt-in f32 mode (GridRec32): same kernels over the 16-byte records; no kd rule (index parity is given up anyway)
        if (ties) return fail(ICPX_ERR_UNSUPPORTED, "the f32 grid mode has no kd-tree tie rule");
        if (perm != nullptr && g->q_blocks > 0 && knobs().grid_coop != 0 && (g->occupied_frac >= 0.5 || knobs().grid_coop == 2)) {
            auto kern = g->dim == 2 ? grid_nn_block_kernel<2, false, GridRec32> : grid_nn_block_kernel<3, false, GridRec32>;
            hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(g->q_blocks)), dim3(256), kGridBlkBytes, stream, d_q,
                               g->gp, g->recs.as<GridRec32>(), g->start.as<int>(), d_idx, d_d2, d_visited, perm, g->q_start.as<int>(),
                               nullptr, nullptr, nullptr);
        } else {
            auto kern = g->dim == 2 ? grid_nn_kernel<2, false, GridRec32> : grid_nn_kernel<3, false, GridRec32>;
            hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, d_q, n, g->gp,
                               g->recs.as<GridRec32>(), g->start.as<int>(), d_idx, d_d2, d_visited, perm, nullptr, nullptr, nullptr);
        }
        ICPX_HIP(hipGetLastError());
        return ICPX_OK;
    }
    // queries in tile order: one workgroup per block of tiles, its cells staged in LDS (icpx_grid.h)
    // — for clouds that fill their grid: with most cells empty (surfaces, clusters) the occupied blocks overflow the
    // staging area and the per-lane walk is faster (clustered config 4: 1.09 vs 1.49 ms)
    if (perm != nullptr && g->q_blocks > 0 && knobs().grid_coop != 0 && (g->occupied_frac >= 0.5 || knobs().grid_coop == 2)) {
        auto kern = g->dim == 2 ? (ties ? grid_nn_block_kernel<2, true> : grid_nn_block_kernel<2, false>)
                                : (ties ? grid_nn_block_kernel<3, true> : grid_nn_block_kernel<3, false>);
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(g->q_blocks)), dim3(256), kGridBlkBytes, stream, d_q,
                           g->gp, g->recs.as<GridRec>(), g->start.as<int>(), d_idx, d_d2, d_visited, perm, g->q_start.as<int>(),
                           cloud, kl, kc);
        ICPX_HIP(hipGetLastError());
        return ICPX_OK;
    }
    const unsigned nb = static_cast<unsigned>((n + 255) / 256);
    auto kern = g->dim == 2 ? (ties ? grid_nn_kernel<2, true> : grid_nn_kernel<2, false>)
                            : (ties ? grid_nn_kernel<3, true> : grid_nn_kernel<3, false>);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, stream, d_q, n, g->gp, g->recs.as<GridRec>(),
                       g->start.as<int>(), d_idx, d_d2, d_visited, perm, cloud, kl, kc);
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

int grid_profile(GridIndex* g, int dim, const double* d_q, int64_t n, const double* d_t, int64_t m, hipStream_t stream,
                 double* out) {
    DevBuf visited;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    int rc = visited.reserve(sizeof(unsigned long long));
    for (int i = 0; i < 3 && rc == ICPX_OK; i++)
        if (hipEventCreate(&ev[i]) != hipSuccess) rc = fail(ICPX_ERR_HIP, "hipEventCreate failed");
    auto hip = [&](hipError_t e, const char* what) {
        if (rc == ICPX_OK && e != hipSuccess) rc = fail(ICPX_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    if (rc == ICPX_OK) {
        hip(hipMemsetAsync(visited.p, 0, sizeof(unsigned long long), stream), "hipMemsetAsync");
        hip(hipEventRecord(ev[0], stream), "hipEventRecord");
        if (rc == ICPX_OK) rc = grid_build(g, dim, d_t, m, stream);
        hip(hipEventRecord(ev[1], stream), "hipEventRecord");
        if (rc == ICPX_OK) rc = grid_sort_queries(g, d_q, n, stream);
        if (rc == ICPX_OK) rc = grid_query_launch(g, d_q, n, nullptr, nullptr, visited.as<unsigned long long>(), stream);
        hip(hipEventRecord(ev[2], stream), "hipEventRecord");
        hip(hipEventSynchronize(ev[2]), "hipEventSynchronize");
    }
    if (rc == ICPX_OK) {
        float b = 0.f, q = 0.f;
        unsigned long long v = 0;
        hip(hipEventElapsedTime(&b, ev[0], ev[1]), "hipEventElapsedTime");
        hip(hipEventElapsedTime(&q, ev[1], ev[2]), "hipEventElapsedTime");
        hip(hipMemcpy(&v, visited.p, sizeof(v), hipMemcpyDeviceToHost), "hipMemcpy");
        out[0] = b;
        out[1] = q;
        out[2] = n > 0 ? static_cast<double>(v) / static_cast<double>(n) : 0.0;
        out[3] = static_cast<double>(g->ncell);
    }
    for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
    visited.release();
    return rc;
}

int grid_nn_oneshot(GridIndex* g, int dim, const double* d_q, int64_t n, const double* d_t, int64_t m, int64_t* d_idx,
                    double* d_d2, hipStream_t stream, BvhIndex* degenerate_to, const KdTies* ties) {
    // the occupied-cell count of the build is read after the query order has been enqueued (grid_settle)
    int rc = grid_build(g, dim, d_t, m, stream, true);
    if (rc == ICPX_OK) rc = grid_sort_queries(g, d_q, n, stream);
    bool rebuilt = false;
    if (rc == ICPX_OK) rc = grid_settle(g, stream, &rebuilt);
    if (rc == ICPX_OK && degenerate_to != nullptr && grid_degenerate(g))
        return bvh_nn_oneshot(degenerate_to, dim, d_q, n, d_t, m, d_idx, d_d2, stream, ties);  // (declared in icpx_large.h)
    if (rc == ICPX_OK && rebuilt) rc = grid_sort_queries(g, d_q, n, stream);  // (the order was that of the coarser grid)
    if (rc == ICPX_OK) rc = grid_query_launch(g, d_q, n, d_idx, d_d2, nullptr, stream, ties);
    if (rc == ICPX_OK) {
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = fail(ICPX_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    }
    return rc;
}

// ------------------------------------------------------------------------------------------
// BVH host side (icpx_bvh.h)
// ------------------------------------------------------------------------------------------

int bvh_create(BvhIndex** out) {
    *out = new (std::nothrow) BvhIndex();
    return *out ? ICPX_OK : fail(ICPX_ERR_NOMEM, "out of host memory");
}

void bvh_destroy(BvhIndex* b) {
    if (!b) return;
    for (DevBuf* d : {&b->recs, &b->box, &b->keys, &b->keys_out, &b->iota, &b->perm, &b->tmp, &b->bbox, &b->q_perm}) d->release();
    delete b;
}

namespace {
// Morton order of `n` points (device, AoS) in the index's frame -> out (u32 permutation)
int bvh_morton_order(BvhIndex* b, const double* d_p, int64_t n, DevBuf& out, hipStream_t stream) {
    ICPX_TRY(b->keys.reserve(sizeof(uint64_t) * n));
    ICPX_TRY(b->keys_out.reserve(sizeof(uint64_t) * n));
    ICPX_TRY(b->iota.reserve(sizeof(uint32_t) * n));
    ICPX_TRY(out.reserve(sizeof(uint32_t) * n));
    const unsigned nb = static_cast<unsigned>((n + 255) / 256);
    if (b->dim == 2)
        hipLaunchKernelGGL(bvh_keys_kernel<2>, dim3(nb), dim3(256), 0, stream, d_p, n, b->frame, b->keys.as<uint64_t>(), b->iota.as<uint32_t>());
    else
        hipLaunchKernelGGL(bvh_keys_kernel<3>, dim3(nb), dim3(256), 0, stream, d_p, n, b->frame, b->keys.as<uint64_t>(), b->iota.as<uint32_t>());
    ICPX_HIP(hipGetLastError());
    size_t bytes = 0;
    const int bits = b->dim == 3 ? 63 : 62;
    ICPX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, b->keys.as<uint64_t>(), b->keys_out.as<uint64_t>(), b->iota.as<uint32_t>(),
                                                out.as<uint32_t>(), static_cast<int>(n), 0, bits, stream));
    ICPX_TRY(b->tmp.reserve(bytes ? bytes : 1));
    ICPX_HIP(hipcub::DeviceRadixSort::SortPairs(b->tmp.p, bytes, b->keys.as<uint64_t>(), b->keys_out.as<uint64_t>(), b->iota.as<uint32_t>(),
                                                out.as<uint32_t>(), static_cast<int>(n), 0, bits, stream));
    return ICPX_OK;
}
}  // namespace

int bvh_build(BvhIndex* b, int dim, const double* d_t, int64_t m, hipStream_t stream) {
    b->dim = dim;
    b->m = m;
    b->P = 0;
    b->q_perm_n = 0;
    if (m <= 0) return ICPX_OK;
    if (m > (int64_t(1) << 30)) return fail(ICPX_ERR_UNSUPPORTED, "bvh: target cloud too large");
    // 1. bounding box of the finite targets (the grid's kernel) -> frame of the Morton codes
    ICPX_TRY(b->bbox.reserve(sizeof(unsigned long long) * 6));
    unsigned long long init[6] = {~0ULL, ~0ULL, ~0ULL, 0, 0, 0};
    ICPX_HIP(hipMemcpyAsync(b->bbox.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
    const int blocks = static_cast<int>(std::min<int64_t>((m + 1023) / 1024, 128));  // (as grid_build)
    if (dim == 2)
        hipLaunchKernelGGL(grid_bbox_kernel<2>, dim3(blocks), dim3(1024), 0, stream, d_t, m, b->bbox.as<unsigned long long>());
    else
        hipLaunchKernelGGL(grid_bbox_kernel<3>, dim3(blocks), dim3(1024), 0, stream, d_t, m, b->bbox.as<unsigned long long>());
    ICPX_HIP(hipGetLastError());
    unsigned long long keys[6];
    ICPX_HIP(hipMemcpyAsync(keys, b->bbox.p, sizeof(keys), hipMemcpyDeviceToHost, stream));
    ICPX_HIP(hipStreamSynchronize(stream));
    const bool any = keys[0] != ~0ULL;
    const double cells = dim == 3 ? 2097152.0 : 2147483648.0;  // 2^21, 2^31: the Hilbert grid (bvh_morton)
    for (int d = 0; d < 3; d++) {
        b->frame.lo[d] = 0.0;
        b->frame.scale[d] = 0.0;
        if (d < dim && any) {
            const double lo = f64_unkey(keys[d]), hi = f64_unkey(keys[3 + d]);
            b->frame.lo[d] = lo;
            const double sc = hi > lo ? cells / (hi - lo) : 0.0;
            b->frame.scale[d] = std::isfinite(sc) ? sc : 0.0;
        }
    }
    // 2. Morton order, records in that order
    ICPX_TRY(bvh_morton_order(b, d_t, m, b->perm, stream));
    ICPX_TRY(b->recs.reserve(sizeof(GridRec) * m));
    const unsigned mb = static_cast<unsigned>((m + 255) / 256);
    if (dim == 2)
        hipLaunchKernelGGL(bvh_gather_kernel<2>, dim3(mb), dim3(256), 0, stream, d_t, m, b->perm.as<uint32_t>(), b->recs.as<GridRec>());
    else
        hipLaunchKernelGGL(bvh_gather_kernel<3>, dim3(mb), dim3(256), 0, stream, d_t, m, b->perm.as<uint32_t>(), b->recs.as<GridRec>());
    ICPX_HIP(hipGetLastError());
    // 3. leaf boxes, then the inner levels bottom-up
    const int64_t leaves = (m + kBvhLeaf - 1) / kBvhLeaf;
    int P = 1;
    while (P < leaves) P <<= 1;
    ICPX_TRY(b->box.reserve(sizeof(float) * 6 * 2 * static_cast<size_t>(P)));
    const unsigned lb = static_cast<unsigned>((P + 255) / 256);
    if (dim == 2)
        hipLaunchKernelGGL(bvh_leaf_kernel<2>, dim3(lb), dim3(256), 0, stream, b->recs.as<GridRec>(), m, P, b->box.as<float>());
    else
        hipLaunchKernelGGL(bvh_leaf_kernel<3>, dim3(lb), dim3(256), 0, stream, b->recs.as<GridRec>(), m, P, b->box.as<float>());
    ICPX_HIP(hipGetLastError());
    int lvl = P >> 1;
    for (; lvl > 256; lvl >>= 1) {
        hipLaunchKernelGGL(bvh_level_kernel, dim3(static_cast<unsigned>((lvl + 255) / 256)), dim3(256), 0, stream, b->box.as<float>(), lvl);
        ICPX_HIP(hipGetLastError());
    }
    if (lvl >= 1) {
        hipLaunchKernelGGL(bvh_top_kernel, dim3(1), dim3(256), 0, stream, b->box.as<float>(), lvl);
        ICPX_HIP(hipGetLastError());
    }
    b->P = P;
    return ICPX_OK;
}

// Morton order of the queries (in the targets' frame): neighbouring lanes then take the same path through the tree
int bvh_sort_queries(BvhIndex* b, const double* d_q, int64_t n, hipStream_t stream) {
    b->q_perm_n = 0;
    if (n < kGridSortMinQueries || b->P <= 0) return ICPX_OK;
    ICPX_TRY(bvh_morton_order(b, d_q, n, b->q_perm, stream));
    b->q_perm_n = n;
    return ICPX_OK;
}

int bvh_query_launch(const BvhIndex* b, const double* d_q, int64_t n, int64_t* d_idx, double* d_d2,
                     unsigned long long* d_visited, hipStream_t stream, const KdTies* ties = nullptr) {
    if (n <= 0) return ICPX_OK;
    const uint32_t* perm = b->q_perm_n == n ? b->q_perm.as<uint32_t>() : nullptr;
    const unsigned nb = static_cast<unsigned>((n + 255) / 256);
    auto kern = b->dim == 2 ? (ties ? bvh_nn_kernel<2, true> : bvh_nn_kernel<2, false>)
                            : (ties ? bvh_nn_kernel<3, true> : bvh_nn_kernel<3, false>);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, stream, d_q, n, b->view(), d_idx, d_d2, d_visited, perm,
                       ties ? ties->cloud : nullptr, ties ? ties->list : nullptr, ties ? ties->count : nullptr);
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

int bvh_nn_oneshot(BvhIndex* b, int dim, const double* d_q, int64_t n, const double* d_t, int64_t m, int64_t* d_idx,
                   double* d_d2, hipStream_t stream, const KdTies* ties) {
    int rc = bvh_build(b, dim, d_t, m, stream);
    if (rc == ICPX_OK) rc = bvh_sort_queries(b, d_q, n, stream);
    if (rc == ICPX_OK) rc = bvh_query_launch(b, d_q, n, d_idx, d_d2, nullptr, stream, ties);
    if (rc == ICPX_OK) {
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = fail(ICPX_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    }
    return rc;
}

// ------------------------------------------------------------------------------------------
// kd-tree tie rule, host side (icpx_kd.h)
// ------------------------------------------------------------------------------------------

// The tree of the device-resident cloud d_t (m points) into d_node / d_pos (m entries each), built on the device
// (icpx_kd_build.h); everything is enqueued on the stream, nothing synchronises.  scratch: grown as needed.
constexpr int64_t kKdEagerMax = 65536;

int kd_build_device(int dim, const double* d_t, int64_t m, uint32_t* d_node, uint32_t* d_pos, DevBuf* scratch, hipStream_t stream) {
    if (m <= 0) return ICPX_OK;
    if (m > 0x7fffffff) return fail(ICPX_ERR_UNSUPPORTED, "kd rule: target cloud above 2^31 points");
    const int M = static_cast<int>(m);
    static std::mutex mu;
    static std::unordered_map<int, bool> attr_set;
    int devid = 0;
    ICPX_HIP(hipGetDevice(&devid));
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!attr_set[devid]) {
            for (const void* k : {reinterpret_cast<const void*>(kd_small_kernel<2>), reinterpret_cast<const void*>(kd_small_kernel<3>),
                                  reinterpret_cast<const void*>(kd_segment_kernel<2>), reinterpret_cast<const void*>(kd_segment_kernel<3>)})
                ICPX_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[devid] = true;
        }
    }
    if (M <= kKdOneBlock) {
        const size_t lds = ((kd_build_bytes(dim, M) + 15) & ~size_t(15)) + 4 * static_cast<size_t>(M);
        if (dim == 2) hipLaunchKernelGGL(kd_small_kernel<2>, dim3(1), dim3(kKdBuildThreads), lds, stream, d_t, M, d_node, d_pos);
        else hipLaunchKernelGGL(kd_small_kernel<3>, dim3(1), dim3(kKdBuildThreads), lds, stream, d_t, M, d_node, d_pos);
        ICPX_HIP(hipGetLastError());
        return ICPX_OK;
    }
    // scratch: keys | keys_out | iota | D orders | tmp | posA | library temp
    size_t sort_bytes = 0, scan_bytes = 0;
    ICPX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, static_cast<unsigned long long*>(nullptr),
                                                static_cast<unsigned long long*>(nullptr), static_cast<uint32_t*>(nullptr),
                                                static_cast<uint32_t*>(nullptr), M, 0, 64, stream));
    {
        KdFlag f{nullptr, nullptr, M, 0};
        auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), f);
        ICPX_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, in, static_cast<unsigned long long*>(nullptr), 0ull, static_cast<size_t>(M),
                                         rocprim::plus<unsigned long long>(), stream));
    }
    const size_t lib_bytes = (std::max(sort_bytes, scan_bytes) + 255) & ~size_t(255);
    const size_t w8 = (sizeof(uint64_t) * m + 255) & ~size_t(255), w4 = (sizeof(uint32_t) * m + 255) & ~size_t(255);
    ICPX_TRY(scratch->reserve(2 * w8 + (3 + dim) * w4 + lib_bytes));
    char* base = static_cast<char*>(scratch->p);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(base);
    unsigned long long* keys_out = reinterpret_cast<unsigned long long*>(base + w8);
    uint32_t* iota = reinterpret_cast<uint32_t*>(base + 2 * w8);
    uint32_t* bufs[4];  // D orders + one spare, rotating
    for (int b = 0; b <= dim; b++) bufs[b] = reinterpret_cast<uint32_t*>(base + 2 * w8 + (1 + b) * w4);
    uint32_t* posA = reinterpret_cast<uint32_t*>(base + 2 * w8 + (2 + dim) * w4);
    void* lib = base + 2 * w8 + (3 + dim) * w4;
    const unsigned nb = static_cast<unsigned>((M + 255) / 256);
    uint32_t* ord[3] = {bufs[0], bufs[1], bufs[2]};
    uint32_t* spare = bufs[dim];
    for (int a = 0; a < dim; a++) {
        if (dim == 2) hipLaunchKernelGGL(kd_keys_kernel<2>, dim3(nb), dim3(256), 0, stream, d_t, M, a, keys, iota);
        else hipLaunchKernelGGL(kd_keys_kernel<3>, dim3(nb), dim3(256), 0, stream, d_t, M, a, keys, iota);
        ICPX_HIP(hipGetLastError());
        size_t bytes = lib_bytes;
        ICPX_HIP(hipcub::DeviceRadixSort::SortPairs(lib, bytes, keys, keys_out, iota, ord[a], M, 0, 64, stream));
    }
    int K = 0;
    while ((M >> K) > kKdSegMax) K++;
    unsigned long long* E = keys;  // (the sort keys are no longer needed)
    for (int k = 0; k < K; k++) {
        const int a = k % dim;
        hipLaunchKernelGGL(kd_pos_kernel, dim3(nb), dim3(256), 0, stream, ord[a], M, posA);
        ICPX_HIP(hipGetLastError());
        for (int b = 0; b < dim; b++) {
            if (b == a) continue;
            KdFlag f{ord[b], posA, M, k};
            auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), f);
            size_t bytes = lib_bytes;
            ICPX_HIP(rocprim::exclusive_scan(lib, bytes, in, E, 0ull, static_cast<size_t>(M), rocprim::plus<unsigned long long>(), stream));
            hipLaunchKernelGGL(kd_scatter_kernel, dim3(nb), dim3(256), 0, stream, ord[b], posA, E, M, k, spare);
            ICPX_HIP(hipGetLastError());
            std::swap(ord[b], spare);
        }
    }
    hipLaunchKernelGGL(kd_fill_kernel, dim3(nb), dim3(256), 0, stream, ord[0], M, d_node, d_pos);
    ICPX_HIP(hipGetLastError());
    KdOrders ords{{ord[0], ord[1], dim == 3 ? ord[2] : nullptr}};
    const int seg_cap = std::min(kKdSegMax, (M >> K) + 1);
    const size_t lds = kd_segment_lds_bytes(dim, seg_cap);
    if (dim == 2) hipLaunchKernelGGL(kd_segment_kernel<2>, dim3(1u << K), dim3(kKdBuildThreads), lds, stream, ords, M, K, posA, d_node, d_pos);
    else hipLaunchKernelGGL(kd_segment_kernel<3>, dim3(1u << K), dim3(kKdBuildThreads), lds, stream, ords, M, K, posA, d_node, d_pos);
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

// test hook (icpx_debug_kd_tree): the in-order node array of a cloud's tree, as the device builds it
int kd_debug_tree(int dim, const double* d_t, int64_t m, uint32_t* h_node, hipStream_t stream) {
    DevBuf tree, scratch;
    int rc = tree.reserve(sizeof(uint32_t) * 2 * static_cast<size_t>(m));
    if (rc == ICPX_OK) rc = kd_build_device(dim, d_t, m, tree.as<uint32_t>(), tree.as<uint32_t>() + m, &scratch, stream);
    if (rc == ICPX_OK) {
        hipError_t e = hipMemcpyAsync(h_node, tree.p, sizeof(uint32_t) * m, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = fail(ICPX_ERR_HIP, "kd_debug_tree: %s", hipGetErrorString(e));
    }
    tree.release();
    scratch.release();
    return rc;
}

int kd_launch_resolve(int dim, const int* d_count, const uint32_t* d_list, int mode, const double* src, const double* cur,
                      const double* xf, const uint32_t* d_node, const uint32_t* d_pos, const double* d_t, int64_t m,
                      int64_t* d_idx, int blocks, hipStream_t stream, double* d_pt = nullptr, const uint32_t* d_perm = nullptr,
                      double* d_pt_sorted = nullptr) {
    KdView<uint32_t> kd;
    kd.node = d_node;
    kd.pos = d_pos;
    kd.pts = d_t;
    kd.m = static_cast<int>(m);
    if (dim == 2)
        hipLaunchKernelGGL(kd_resolve_kernel<2>, dim3(blocks), dim3(kKdResolveThreads), 0, stream, d_count, d_list, mode, src, cur, xf, kd, d_idx, d_pt, d_perm, d_pt_sorted);
    else
        hipLaunchKernelGGL(kd_resolve_kernel<3>, dim3(blocks), dim3(kKdResolveThreads), 0, stream, d_count, d_list, mode, src, cur, xf, kd, d_idx, d_pt, d_perm, d_pt_sorted);
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

int kd_resolve_queries(int dim, const double* d_q, int64_t n, const double* d_t, int64_t m, const KdTies& ties,
                       int64_t* d_idx, hipStream_t stream) {
    int count = 0;
    ICPX_HIP(hipMemcpyAsync(&count, ties.count, sizeof(int), hipMemcpyDeviceToHost, stream));
    ICPX_HIP(hipStreamSynchronize(stream));
    if (count <= 0 || m <= 0 || d_idx == nullptr) return ICPX_OK;
    DevBuf tree, scratch;
    int rc = tree.reserve(sizeof(uint32_t) * 2 * static_cast<size_t>(m));
    if (rc == ICPX_OK) rc = kd_build_device(dim, d_t, m, tree.as<uint32_t>(), tree.as<uint32_t>() + m, &scratch, stream);
    if (rc == ICPX_OK)
        rc = kd_launch_resolve(dim, ties.count, ties.list, 1, d_q, nullptr, nullptr, tree.as<uint32_t>(), tree.as<uint32_t>() + m, d_t, m,
                               d_idx, count < 4096 ? count : 4096, stream);
    if (rc == ICPX_OK) {
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = fail(ICPX_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    }
    tree.release();
    scratch.release();
    return rc;
}

// ------------------------------------------------------------------------------------------
// large-path host side
// ------------------------------------------------------------------------------------------

int large_create(int dim, int method, const icpx_params& prm, double clock_to_s, hipStream_t stream,
                 LargeIcp** out) {
    LargeIcp* lg = new (std::nothrow) LargeIcp();
    if (!lg) return fail(ICPX_ERR_NOMEM, "out of host memory");
    lg->dim = dim;
    lg->method = method;
    lg->prm = prm;
    lg->clock_to_s = clock_to_s;
    lg->stream = stream;
    int rc = grid_create(&lg->grid);
    if (rc != ICPX_OK) {
        delete lg;
        return rc;
    }
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&lg->h_done), sizeof(int) * 4, hipHostMallocCoherent | hipHostMallocMapped);
    if (e != hipSuccess) {
        grid_destroy(lg->grid);
        delete lg;
        return fail(ICPX_ERR_NOMEM, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&lg->h_done_dev), lg->h_done, 0) != hipSuccess) {
        (void)hipGetLastError();
        lg->h_done_dev = nullptr;  // (large_converge then reads the flag with a copy, as before)
    }
    *out = lg;
    return ICPX_OK;
}

void large_destroy(LargeIcp* lg) {
    if (!lg) return;
    (void)hipStreamSynchronize(lg->stream);
    if (lg->kd_stream) {
        (void)hipStreamSynchronize(lg->kd_stream);
        (void)hipStreamDestroy(lg->kd_stream);
    }
    if (lg->kd_ev_cloud) (void)hipEventDestroy(lg->kd_ev_cloud);
    if (lg->kd_ev_tree) (void)hipEventDestroy(lg->kd_ev_tree);
    for (DevBuf* b : {&lg->src, &lg->dst, &lg->cur, &lg->guess, &lg->nb_idx, &lg->nb_cost, &lg->sel, &lg->part,
                      &lg->cntp, &lg->dev, &lg->nb_pt, &lg->cur_s, &lg->idx_s, &lg->pt_s, &lg->sort_keys, &lg->sort_iota, &lg->sort_order, &lg->sort_tmp,
                      &lg->part_d2, &lg->part_idx, &lg->tile_prep, &lg->f_rad, &lg->f_feat_a, &lg->f_feat_b, &lg->f_nfd, &lg->kd_list, &lg->kd_tree, &lg->kd_scratch})
        b->release();
    grid_destroy(lg->grid);
    bvh_destroy(lg->bvh);
    if (lg->h_done) (void)hipHostFree(lg->h_done);
    delete lg;
}

namespace {

inline icpx_criteria zero_criteria() {
    icpx_criteria c;
    std::memset(&c, 0, sizeof(c));
    return c;
}

// flags of the sum kernels (lg_sum1_kernel): sorted domain = both; tiled 3D search = the step only
inline int lg_sd_flags(const LargeIcp* lg) { return lg->sd ? 3 : ((lg->dim == 3 && lg->tile) ? 1 : 0); }

template <int D, int METHOD>
int enqueue_sums(LargeIcp* lg, int respect_done, bool second, int check = 0, const icpx_criteria* crit = nullptr,
                 double host_elapsed_s = 0.0) {
    LargeDev* dev = lg->dev.as<LargeDev>();
    const int nblk = lg->nblk, groups = lg->groups ? 1 : 0;
    const dim3 grid(groups ? (nblk + kGroupBlk / 64 - 1) / (kGroupBlk / 64) : nblk), block(groups ? kGroupBlk : kBlk);
    if (!second) {
        // centroid / count / cost partials; the last workgroup finishes them (fin1_body)
        hipLaunchKernelGGL((lg_sum1_kernel<D, METHOD>), grid, block, 0, lg->stream, dev, respect_done,
                           lg->src.as<double>(), lg->cur.as<double>(), lg->nb_pt.as<double>(), lg->n,
                           lg->nb_cost.as<double>(), lg->sel.as<uint8_t>(),
                           lg->part.as<double>(), lg->cntp.as<int>(), nblk, groups, lg->poll_dev, lg_sd_flags(lg),
                           lg->prm.max_distance);
        ICPX_HIP(hipGetLastError());
    } else {
        // cross-covariance partials; the last workgroup takes the Kabsch step and evaluates the criteria (fin2_body)
        hipLaunchKernelGGL((lg_sum2_kernel<D, METHOD>), grid, block, 0, lg->stream, dev, respect_done,
                           lg->src.as<double>(), lg->cur.as<double>(), lg->nb_pt.as<double>(), lg->n,
                           lg->sel.as<uint8_t>(), lg->part.as<double>(), nblk, groups, static_cast<double>(lg->n), check,
                           crit ? *crit : zero_criteria(), lg->clock_to_s, host_elapsed_s, lg->poll_dev, lg_sd_flags(lg),
                           lg->prm.max_distance);
        ICPX_HIP(hipGetLastError());
    }
    return ICPX_OK;
}

// FeatureAware::compute_matches (feature_aware.cpp:129-149): a_current, the maximum pairwise
// distance, the weighted-cost argmin per target split, the merge
int enqueue_feature_match(LargeIcp* lg, int respect) {
    LargeDev* dev = lg->dev.as<LargeDev>();
    const unsigned nb = static_cast<unsigned>((lg->n + 255) / 256);
    dim3 grid(nb, static_cast<unsigned>(lg->splits));
    hipLaunchKernelGGL(fa_apply_kernel, dim3(nb), dim3(256), 0, lg->stream, dev, respect, lg->src.as<double>(),
                       lg->cur.as<double>(), lg->n);
    ICPX_HIP(hipGetLastError());
    hipLaunchKernelGGL(fa_maxdist_kernel, grid, dim3(256), 0, lg->stream, dev, respect, lg->cur.as<double>(), lg->n,
                       lg->dst.as<double>(), lg->m, lg->split_len);
    ICPX_HIP(hipGetLastError());
    hipLaunchKernelGGL(fa_match_kernel, grid, dim3(256), 0, lg->stream, dev, respect, lg->cur.as<double>(), lg->n,
                       lg->dst.as<double>(), lg->m, lg->split_len, lg->f_nfd.as<double>(), lg->prm.feature_weight,
                       lg->part_d2.as<double>(), lg->part_idx.as<int32_t>());
    ICPX_HIP(hipGetLastError());
    hipLaunchKernelGGL((lg_tile_merge_kernel<2, ICPX_TRIMMED, 2>), dim3(nb), dim3(256), 0, lg->stream, dev, respect, lg->n,
                       lg->splits, lg->part_d2.as<double>(), lg->part_idx.as<int32_t>(), lg->nb_idx.as<int64_t>(),
                       lg->nb_cost.as<double>(), lg->sel.as<uint8_t>(), lg->prm.max_distance, lg->dst.as<double>(), nullptr,
                       lg->nb_pt.as<double>());
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

// compute_features of one cloud (feature_aware.cpp:151-177) into feat[r * cnt + i]
int enqueue_features(LargeIcp* lg, const double* pts, int64_t cnt, double* feat) {
    LargeDev* dev = lg->dev.as<LargeDev>();
    const int nblk = static_cast<int>((cnt + kBlkPts - 1) / kBlkPts);
    const unsigned nb = static_cast<unsigned>((cnt + 255) / 256);
    hipLaunchKernelGGL(fa_centroid_part_kernel, dim3(nblk), dim3(kBlk), 0, lg->stream, pts, cnt, lg->part.as<double>(), nblk);
    ICPX_HIP(hipGetLastError());
    hipLaunchKernelGGL(fa_centroid_fin_kernel, dim3(1), dim3(kBlk), 0, lg->stream, dev, lg->part.as<double>(), nblk, cnt);
    ICPX_HIP(hipGetLastError());
    hipLaunchKernelGGL(fa_radius_kernel, dim3(nb), dim3(256), 0, lg->stream, dev, pts, cnt, lg->f_rad.as<double>());
    ICPX_HIP(hipGetLastError());
    if (lg->prm.symmetric_neighbors > 0) {
        hipLaunchKernelGGL(fa_features_kernel, dim3(nb), dim3(256), 0, lg->stream, lg->f_rad.as<double>(), cnt,
                           lg->prm.symmetric_neighbors, feat);
        ICPX_HIP(hipGetLastError());
    }
    return ICPX_OK;
}

// FeatureAware::setup up to (not including) compute_matches (feature_aware.cpp:45-59)
int enqueue_feature_setup(LargeIcp* lg) {
    LargeDev* dev = lg->dev.as<LargeDev>();
    const unsigned nb = static_cast<unsigned>((lg->n + 255) / 256);
    hipLaunchKernelGGL(fa_apply_kernel, dim3(nb), dim3(256), 0, lg->stream, dev, 0, lg->src.as<double>(),
                       lg->cur.as<double>(), lg->n);
    ICPX_HIP(hipGetLastError());
    ICPX_TRY(enqueue_features(lg, lg->cur.as<double>(), lg->n, lg->f_feat_a.as<double>()));
    ICPX_TRY(enqueue_features(lg, lg->dst.as<double>(), lg->m, lg->f_feat_b.as<double>()));
    const int w = 2 * lg->prm.symmetric_neighbors;
    dim3 grid(nb, static_cast<unsigned>((lg->m + kFaJ - 1) / kFaJ));
    hipLaunchKernelGGL(fa_fdist_kernel, grid, dim3(256), 0, lg->stream, dev, lg->f_feat_a.as<double>(),
                       lg->f_feat_b.as<double>(), lg->n, lg->m, w, lg->f_nfd.as<double>());
    ICPX_HIP(hipGetLastError());
    const int64_t total = lg->n * lg->m;
    hipLaunchKernelGGL(fa_normalize_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, lg->stream,
                       dev, lg->f_nfd.as<double>(), total);
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

// the correspondence stage of one step: the tiled brute force (mid-size clouds) or the grid walk
// kd rule: once the target's tree is on the device, the flagged queries of a correspondence step are resolved right
// behind it (a launch that finds an empty list otherwise)
template <int D, int MODE>
int enqueue_kd_resolve(LargeIcp* lg) {
    if (!lg->kd_rule || !lg->kd_ready || MODE == 2) return ICPX_OK;
    LargeDev* dev = lg->dev.as<LargeDev>();
    if (lg->kd_wait) {
        ICPX_HIP(hipStreamWaitEvent(lg->stream, lg->kd_ev_tree, 0));
        lg->kd_wait = false;
    }
    // (sorted domain: the list holds positions of the query order; queries and neighbours live in cur_s / idx_s)
    const bool sd = lg->sd && MODE == 0;
    // (tiled 3D search: its queries are step * cur, the cloud itself is stepped later, by lg_sum1)
    const bool on_the_fly = D == 3 && MODE == 0 && lg->tile;
    return kd_launch_resolve(D, &dev->kd_count, lg->kd_list.as<uint32_t>(), on_the_fly ? 4 : MODE, lg->src.as<double>(),
                             sd ? lg->cur_s.as<double>() : lg->cur.as<double>(),
                             on_the_fly ? dev->step : dev->st.transform, lg->kd_tree.as<uint32_t>(), lg->kd_tree.as<uint32_t>() + lg->m,
                             lg->dst.as<double>(), lg->m, sd ? lg->idx_s.as<int64_t>() : lg->nb_idx.as<int64_t>(), 1024, lg->stream,
                             lg->nb_pt.as<double>(), sd ? lg->grid->q_perm.as<uint32_t>() : nullptr, sd ? lg->pt_s.as<double>() : nullptr);
}

template <int D, int METHOD, int MODE>
int enqueue_nn_search(LargeIcp* lg, int respect);

template <int D, int METHOD, int MODE>
int enqueue_nn(LargeIcp* lg, int respect) {
    ICPX_TRY((enqueue_nn_search<D, METHOD, MODE>(lg, respect)));
    return enqueue_kd_resolve<D, MODE>(lg);
}

template <int D, int METHOD, int MODE>
int enqueue_nn_search(LargeIcp* lg, int respect) {
    LargeDev* dev = lg->dev.as<LargeDev>();
    const unsigned nb = static_cast<unsigned>((lg->n + 255) / 256);
    if (lg->feature) return enqueue_feature_match(lg, respect);
    if (!lg->tile) {
        if (lg->use_bvh) {
            const BvhIndex* bv = lg->bvh;
            hipLaunchKernelGGL((lg_nn_kernel<D, METHOD, MODE, GridRec>), dim3(nb), dim3(256), sizeof(float) * 6 * kBvhTopNodes, lg->stream, dev,
                               respect, lg->src.as<double>(), lg->cur.as<double>(), lg->n, GridParams{}, static_cast<const GridRec*>(nullptr), nullptr,
                               lg->nb_idx.as<int64_t>(), lg->nb_cost.as<double>(), lg->sel.as<uint8_t>(), lg->prm.max_distance,
                               bv->q_perm_n == lg->n ? bv->q_perm.as<uint32_t>() : nullptr, nullptr, bv->view(), nullptr,
                               lg->dst.as<double>(), lg->kd_list.as<uint32_t>(), lg->nb_pt.as<double>());
            ICPX_HIP(hipGetLastError());
            return ICPX_OK;
        }
        const GridIndex* g = lg->grid;
        const uint32_t* perm = g->q_perm_n == lg->n ? g->q_perm.as<uint32_t>() : nullptr;
        // block-cooperative only on request (ICPX_GRID_COOP=2): the order is that of begin(), and once the running cloud has
        // moved by more than a cell most queries have left their block's staged cells (config 4 converge: 26.6 vs 19.4 ms)
        const bool blk = perm != nullptr && g->q_blocks > 0 && knobs().grid_coop == 2;
        const bool sd_step = lg->sd && D == 3 && MODE == 0 && !blk && perm != nullptr;
        if (g->f32) {
            // opt-in f32 mode (3D methods' per-iteration stage only: large_begin refuses the rest)
            if constexpr (D == 3 && MODE == 0) {
                hipLaunchKernelGGL((lg_nn_kernel<D, METHOD, MODE, GridRec32>), dim3(blk ? static_cast<unsigned>(g->q_blocks) : nb), dim3(256),
                                   blk ? kGridBlkBytes : 0, lg->stream, dev, respect,
                                   lg->src.as<double>(), lg->cur.as<double>(), lg->n, g->gp,
                                   g->recs.as<GridRec32>(), g->start.as<int>(), lg->nb_idx.as<int64_t>(),
                                   lg->nb_cost.as<double>(), lg->sel.as<uint8_t>(), lg->prm.max_distance, perm,
                                   blk ? g->q_start.as<int>() : nullptr, BvhView{},
                                   (!blk && lg->nn_seeded && knobs().grid_seed != 0) ? lg->dst.as<double>() : nullptr,
                                   lg->dst.as<double>(), lg->kd_list.as<uint32_t>(), lg->nb_pt.as<double>(),
                                   (sd_step ? lg->cur_s.as<double>() : nullptr), (sd_step ? lg->idx_s.as<int64_t>() : nullptr),
                                   (sd_step ? lg->pt_s.as<double>() : nullptr));
                ICPX_HIP(hipGetLastError());
                if (sd_step) lg->sd_dirty = true;
                lg->nn_seeded = true;
                return ICPX_OK;
            } else {
                return fail(ICPX_ERR_UNSUPPORTED, "nn = grid_f32 serves the 3D methods without flags");
            }
        }
        if constexpr (D == 3 && MODE == 0) {
            // seeded sorted-domain steps through the f32 filter records: the lean kernel (same results)
            if (sd_step && lg->nn_seeded && knobs().grid_seed != 0 && g->want_filt && g->filt.p != nullptr) {
                hipLaunchKernelGGL(lg_nn_sd_kernel, dim3(nb), dim3(256), 0, lg->stream, dev, respect, lg->n, g->gp, g->recs.as<GridRec>(),
                                   g->filt.as<GridRec32>(), g->start.as<int>(), perm, lg->dst.as<double>(), lg->kd_list.as<uint32_t>(),
                                   lg->nb_pt.as<double>(), lg->cur_s.as<double>(), lg->idx_s.as<int64_t>(), lg->pt_s.as<double>());
                ICPX_HIP(hipGetLastError());
                lg->sd_dirty = true;
                return ICPX_OK;
            }
        }
        hipLaunchKernelGGL((lg_nn_kernel<D, METHOD, MODE, GridRec>), dim3(blk ? static_cast<unsigned>(g->q_blocks) : nb), dim3(256),
                           blk ? kGridBlkBytes : 0, lg->stream, dev, respect,
                           lg->src.as<double>(), lg->cur.as<double>(), lg->n, g->gp,
                           g->recs.as<GridRec>(), g->start.as<int>(), lg->nb_idx.as<int64_t>(),
                           lg->nb_cost.as<double>(), lg->sel.as<uint8_t>(), lg->prm.max_distance, perm,
                           blk ? g->q_start.as<int>() : nullptr, BvhView{},
                           (!blk && lg->nn_seeded && knobs().grid_seed != 0) ? lg->dst.as<double>() : nullptr,
                           lg->dst.as<double>(), lg->kd_list.as<uint32_t>(), lg->nb_pt.as<double>(),
                           (sd_step ? lg->cur_s.as<double>() : nullptr), (sd_step ? lg->idx_s.as<int64_t>() : nullptr),
                           (sd_step ? lg->pt_s.as<double>() : nullptr),
                           (g->want_filt && g->filt.p != nullptr && D == 3 && MODE == 0) ? g->filt.as<GridRec32>() : nullptr);
        ICPX_HIP(hipGetLastError());
        if (sd_step) lg->sd_dirty = true;
        lg->nn_seeded = true;  // nb_idx (idx_s) now holds this step's neighbours
        return ICPX_OK;
    }
    // (MODE 0: the queries are step * cur, computed by every workgroup on the fly; lg_sum1 writes the stepped cloud)
    // queries per thread (large_begin): 4, or fewer for clouds whose tiles would not fill the chip
    if (lg->tile_first && lg->splits >= 4) {
        // the first search of a begin: seeds from a sample of the targets, so that every split starts from one bound
        const int stride = static_cast<int>((lg->m + kTileSeedSamples - 1) / kTileSeedSamples);
        hipLaunchKernelGGL((lg_tile_seed_kernel<D, MODE>), dim3(static_cast<unsigned>((lg->n + 63) / 64)), dim3(256), 0, lg->stream,
                           dev, respect, lg->n, lg->src.as<double>(), lg->cur.as<double>(), lg->dst.as<double>(), lg->m, stride,
                           lg->nb_idx.as<int64_t>());
        ICPX_HIP(hipGetLastError());
    }
    lg->tile_first = false;
    auto kern = lg->tile_qpt == 1 ? lg_tile_nn_kernel<D, MODE, 1> : (lg->tile_qpt == 2 ? lg_tile_nn_kernel<D, MODE, 2> : lg_tile_nn_kernel<D, MODE, 4>);
    const size_t lds = tile_lds_bytes(D, lg->split_len);
    static std::mutex mu;
    static std::unordered_map<int, bool> attr_set;
    {
        int devid = 0;
        ICPX_HIP(hipGetDevice(&devid));
        std::lock_guard<std::mutex> lk(mu);
        if (!attr_set[devid * 8 + lg->tile_qpt]) {
            ICPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(tile_lds_bytes(3, kTileMaxSplit))));
            attr_set[devid * 8 + lg->tile_qpt] = true;
        }
    }
    const int64_t tile_queries = static_cast<int64_t>(kTileThreads) * lg->tile_qpt;
    dim3 grid(static_cast<unsigned>((lg->n + tile_queries - 1) / tile_queries), static_cast<unsigned>(lg->splits));
    hipLaunchKernelGGL(kern, grid, dim3(kTileThreads), lds, lg->stream, dev, respect, lg->src.as<double>(),
                       lg->cur.as<double>(), lg->n, lg->dst.as<double>(), lg->m, lg->split_len, lg->split_len,
                       lg->nb_idx.as<int64_t>(), lg->part_d2.as<double>(), lg->part_idx.as<int32_t>(), lg->tile_interleaved ? 1 : 0,
                       lg->tile_prep.as<double>());
    ICPX_HIP(hipGetLastError());
    hipLaunchKernelGGL((lg_tile_merge_kernel<D, METHOD, MODE>), dim3(nb), dim3(256), 0, lg->stream, dev, respect, lg->n,
                       lg->splits, lg->part_d2.as<double>(), lg->part_idx.as<int32_t>(), lg->nb_idx.as<int64_t>(),
                       lg->nb_cost.as<double>(), lg->sel.as<uint8_t>(), lg->prm.max_distance, lg->dst.as<double>(),
                       lg->kd_list.as<uint32_t>(), lg->nb_pt.as<double>());
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

// what follows the correspondence step of setup (also: what remains of it after a stall at a tie, lg_kd_service)
template <int D, int METHOD>
int enqueue_setup_tail(LargeIcp* lg, int check, const icpx_criteria& crit, double host_elapsed_s) {
    LargeDev* dev = lg->dev.as<LargeDev>();
    const bool fill3 = D == 3 && (lg->prm.flags & ICPX_FLAG_FILL_MATCHES);
    if (D == 2 || fill3) ICPX_TRY((enqueue_sums<D, METHOD>(lg, 0, false)));
    hipLaunchKernelGGL(lg_setup_finish_kernel<D>, dim3(1), dim3(64), 0, lg->stream, dev, lg->n, check, crit,
                       lg->clock_to_s, host_elapsed_s, lg->poll_dev);
    ICPX_HIP(hipGetLastError());
    return ICPX_OK;
}

template <int D, int METHOD>
int enqueue_setup(LargeIcp* lg, int check, const icpx_criteria& crit, double host_elapsed_s) {
    LargeDev* dev = lg->dev.as<LargeDev>();
    const bool match_tf = D == 2 && METHOD == ICPX_VANILLA && (lg->prm.flags & ICPX_FLAG_MATCH_TRANSFORMED);
    const bool fill3 = D == 3 && (lg->prm.flags & ICPX_FLAG_FILL_MATCHES);
    const int with_cost = (D == 2 || fill3) ? 1 : 0;
    const int update_cost = ((D == 2 && METHOD != ICPX_VANILLA) || match_tf || fill3) ? 1 : 0;
    lg->init_at = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(lg_init_kernel<D>, dim3(1), dim3(64), 0, lg->stream, dev, lg->guess.as<double>(), with_cost,
                       update_cost, lg->kd_ready ? 1 : 0);
    ICPX_HIP(hipGetLastError());
    if (D == 2) {
        if (lg->feature) ICPX_TRY(enqueue_feature_setup(lg));
        if (match_tf) ICPX_TRY((enqueue_nn<D, METHOD, 3>(lg, 0)));
        else ICPX_TRY((enqueue_nn<D, METHOD, METHOD == ICPX_TRIMMED ? 2 : 1>(lg, 0)));
    } else if (fill3) {
        // opt-in: the 3D methods fill `matches` in setup too (step is the identity here)
        ICPX_TRY((enqueue_nn<D, METHOD, 0>(lg, 0)));
    }
    return enqueue_setup_tail<D, METHOD>(lg, check, crit, host_elapsed_s);
}

template <int D, int METHOD>
int enqueue_iteration_tail(LargeIcp* lg, int check, const icpx_criteria& crit, double host_elapsed_s);

template <int D, int METHOD>
int enqueue_iteration(LargeIcp* lg, int check, const icpx_criteria& crit, double host_elapsed_s) {
    const int respect = check;
    if (D == 3 || METHOD == ICPX_TRIMMED) ICPX_TRY((enqueue_nn<D, METHOD, D == 3 ? 0 : 2>(lg, respect)));
    else if (lg->prm.flags & ICPX_FLAG_MATCH_TRANSFORMED) ICPX_TRY((enqueue_nn<D, METHOD, 3>(lg, respect)));
    return enqueue_iteration_tail<D, METHOD>(lg, check, crit, host_elapsed_s);
}

// what follows the correspondence step of an iteration (also: what remains of it after a stall at a tie)
template <int D, int METHOD>
int enqueue_iteration_tail(LargeIcp* lg, int check, const icpx_criteria& crit, double host_elapsed_s) {
    LargeDev* dev = lg->dev.as<LargeDev>();
    const int respect = check;
    if (D == 2 && METHOD == ICPX_TRIMMED) {
        // std::sort by cost (trimmed.cpp:55-56) made total: a stable LSD radix sort over the
        // cost's bit pattern (costs are >= 0, so bit order is numeric order) with the point index
        // as payload yields the (cost, point) order.  After `done` the inputs no longer change,
        // so the redundant sorts of a chunk's tail reproduce the same arrays.
        ICPX_HIP(hipcub::DeviceRadixSort::SortPairs(lg->sort_tmp.p, lg->sort_tmp_bytes,
                                                    lg->nb_cost.as<unsigned long long>(),
                                                    lg->sort_keys.as<unsigned long long>(),
                                                    lg->sort_iota.as<uint32_t>(), lg->sort_order.as<uint32_t>(),
                                                    static_cast<int>(lg->n), 0, 64, lg->stream));
        unsigned nb = static_cast<unsigned>((lg->n + 255) / 256);
        hipLaunchKernelGGL(lg_select_kernel, dim3(nb), dim3(256), 0, lg->stream, dev, respect, lg->n, lg->new_n,
                           lg->sort_order.as<uint32_t>(), lg->sel.as<uint8_t>());
        ICPX_HIP(hipGetLastError());
    }
    ICPX_TRY((enqueue_sums<D, METHOD>(lg, respect, false)));
    return enqueue_sums<D, METHOD>(lg, respect, true, check, &crit, host_elapsed_s);
}

#define LG_DISPATCH(fn, lg, ...)                                                             \
    ((lg)->dim == 2 ? ((lg)->method == ICPX_VANILLA ? fn<2, ICPX_VANILLA>(lg, __VA_ARGS__)   \
                                                    : fn<2, ICPX_TRIMMED>(lg, __VA_ARGS__))  \
                    : ((lg)->method == ICPX_VANILLA ? fn<3, ICPX_VANILLA>(lg, __VA_ARGS__)   \
                                                    : fn<3, ICPX_TRIMMED>(lg, __VA_ARGS__)))

}  // namespace

namespace {
// kd rule, host side of a stall.  Precondition: the stream is idle.  *stalled: the device had stopped at flagged
// queries without a tree; now the tree is on the device, the flagged queries are being resolved and the stall is
// over — the caller enqueues what remained of the stalled step.  mode: MODE of the stalled correspondence step.
int lg_kd_service(LargeIcp* lg, int mode, bool* stalled) {
    *stalled = false;
    if (!lg->kd_rule || lg->kd_ready) return ICPX_OK;
    LargeDev* dev = lg->dev.as<LargeDev>();
    int stall = 0;
    ICPX_HIP(hipMemcpyAsync(&stall, &dev->kd_stall, sizeof(int), hipMemcpyDeviceToHost, lg->stream));
    ICPX_HIP(hipStreamSynchronize(lg->stream));
    if (!stall) return ICPX_OK;
    ICPX_TRY(lg->kd_tree.reserve(sizeof(uint32_t) * 2 * static_cast<size_t>(lg->m)));
    ICPX_TRY(kd_build_device(lg->dim, lg->dst.as<double>(), lg->m, lg->kd_tree.as<uint32_t>(), lg->kd_tree.as<uint32_t>() + lg->m,
                             &lg->kd_scratch, lg->stream));
    lg->kd_ready = true;
    hipLaunchKernelGGL(lg_kd_resume_kernel, dim3(1), dim3(64), 0, lg->stream, dev);
    ICPX_HIP(hipGetLastError());
    const bool sd = lg->sd && mode == 0;
    const bool on_the_fly = lg->dim == 3 && mode == 0 && lg->tile;
    ICPX_TRY(kd_launch_resolve(lg->dim, &dev->kd_count, lg->kd_list.as<uint32_t>(), on_the_fly ? 4 : mode, lg->src.as<double>(),
                               sd ? lg->cur_s.as<double>() : lg->cur.as<double>(),
                               on_the_fly ? dev->step : dev->st.transform, lg->kd_tree.as<uint32_t>(), lg->kd_tree.as<uint32_t>() + lg->m,
                               lg->dst.as<double>(), lg->m, sd ? lg->idx_s.as<int64_t>() : lg->nb_idx.as<int64_t>(), 1024, lg->stream,
                               lg->nb_pt.as<double>(), sd ? lg->grid->q_perm.as<uint32_t>() : nullptr, sd ? lg->pt_s.as<double>() : nullptr));
    *stalled = true;
    lg->kd_tied = true;
    return ICPX_OK;
}
int lg_iteration_mode(const LargeIcp* lg) { return lg->dim == 3 ? 0 : 3; }
int lg_setup_mode(const LargeIcp* lg) {
    if (lg->dim == 3) return 0;
    return (lg->prm.flags & ICPX_FLAG_MATCH_TRANSFORMED) ? 3 : 1;
}
}  // namespace

int large_begin(LargeIcp* lg, const double* source, int64_t n, const double* target, int64_t m,
                const double* guess) {
    const int D = lg->dim;
    if (n <= 0 || m <= 0) return fail(ICPX_ERR_INVALID, "large path needs non-empty clouds");
    if (n > kLargeMaxN) return fail(ICPX_ERR_UNSUPPORTED, "source cloud above %lld points", (long long)kLargeMaxN);
    lg->n = n;
    lg->m = m;
    lg->groups = n <= kGroupSumMaxN && knobs().no_group_sums == 0;
    lg->nblk = static_cast<int>(lg->groups ? (n + 63) / 64 : (n + kBlkPts - 1) / kBlkPts);
    ICPX_TRY(lg->src.reserve(sizeof(double) * D * n));
    ICPX_TRY(lg->dst.reserve(sizeof(double) * D * m));
    ICPX_TRY(lg->cur.reserve(sizeof(double) * D * n));
    ICPX_TRY(lg->guess.reserve(sizeof(double) * 16));
    ICPX_TRY(lg->nb_idx.reserve(sizeof(int64_t) * n));
    ICPX_TRY(lg->nb_pt.reserve(sizeof(double) * D * n));
    ICPX_TRY(lg->nb_cost.reserve(sizeof(double) * n));
    ICPX_TRY(lg->sel.reserve(n));
    ICPX_TRY(lg->part.reserve(sizeof(double) * 16 * lg->nblk));
    ICPX_TRY(lg->cntp.reserve(sizeof(int) * lg->nblk));
    ICPX_TRY(lg->dev.reserve(sizeof(LargeDev)));
    ICPX_HIP(hipMemcpyAsync(lg->src.p, source, sizeof(double) * D * n, hipMemcpyDefault, lg->stream));
    ICPX_HIP(hipMemcpyAsync(lg->dst.p, target, sizeof(double) * D * m, hipMemcpyDefault, lg->stream));
    ICPX_HIP(hipMemcpyAsync(lg->guess.p, guess, sizeof(double) * (D + 1) * (D + 1), hipMemcpyDefault, lg->stream));
    // c = a (vanilla_3d.cpp:72): the guess is NOT applied to the running cloud (Q5)
    ICPX_HIP(hipMemcpyAsync(lg->cur.p, lg->src.p, sizeof(double) * D * n, hipMemcpyDeviceToDevice, lg->stream));
    if (D == 3 && (lg->prm.flags & ICPX_FLAG_APPLY_GUESS)) {
        // opt-in correction of Q5: c = guess * a
        hipLaunchKernelGGL(lg_apply_guess_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, lg->stream,
                           lg->guess.as<double>(), lg->src.as<double>(), lg->cur.as<double>(), n);
        ICPX_HIP(hipGetLastError());
    }
    ICPX_HIP(hipMemsetAsync(lg->sel.p, 1, n, lg->stream));
    lg->sorted = false;
    lg->feature = D == 2 && lg->method == ICPX_FEATURE_AWARE;
    const bool f32 = lg->prm.nn_mode == ICPX_NN_GRID_F32;
    if (f32 && (D != 3 || lg->feature || lg->prm.flags != 0))
        return fail(ICPX_ERR_UNSUPPORTED, "nn = grid_f32 serves the 3D methods (vanilla, trimmed) without flags");
    // (the f32 mode gives up index parity: no kd-tree tie rule there)
    lg->kd_rule = !lg->feature && (D == 3 || lg->method == ICPX_VANILLA) && !f32;
    lg->kd_ready = false;
    lg->kd_wait = false;
    if (lg->kd_rule) ICPX_TRY(lg->kd_list.reserve(sizeof(uint32_t) * n));
    // The tree of a small target at begin(), on a stream of its own — only for a handle that met tie partners before
    // (kd_tied: lg_kd_service): clouds of measured coordinates hardly ever hold two targets within 4 ulps of a query's
    // minimum, and then neither the build nor the resolve launch behind every search is needed (queen.ply: 1.13 ->
    // 0.93 ms).  A handle's first cloud with ties pays one stall (lg_kd_service) and builds eagerly from then on.
    // ICPX_KD_EAGER (development): 0 never, 1 always.
    const int eager = knobs().kd_eager.load();
    if (lg->kd_rule && m <= kKdEagerMax && (eager == 1 || (eager < 0 && lg->kd_tied))) {
        if (!lg->kd_stream) {
            ICPX_HIP(hipStreamCreateWithFlags(&lg->kd_stream, hipStreamNonBlocking));
            ICPX_HIP(hipEventCreateWithFlags(&lg->kd_ev_cloud, hipEventDisableTiming));
            ICPX_HIP(hipEventCreateWithFlags(&lg->kd_ev_tree, hipEventDisableTiming));
        }
        ICPX_TRY(lg->kd_tree.reserve(sizeof(uint32_t) * 2 * static_cast<size_t>(m)));
        ICPX_HIP(hipEventRecord(lg->kd_ev_cloud, lg->stream));  // (the target cloud is on the device; earlier converges are over)
        ICPX_HIP(hipStreamWaitEvent(lg->kd_stream, lg->kd_ev_cloud, 0));
        ICPX_TRY(kd_build_device(D, lg->dst.as<double>(), m, lg->kd_tree.as<uint32_t>(), lg->kd_tree.as<uint32_t>() + m, &lg->kd_scratch,
                                 lg->kd_stream));
        ICPX_HIP(hipEventRecord(lg->kd_ev_tree, lg->kd_stream));
        lg->kd_ready = true;
        lg->kd_wait = true;
    }
    if (D == 2 && lg->method != ICPX_VANILLA) {
        // trimmed.cpp:57-58, feature_aware.cpp:92-93 (overlap_rate > 1 would read out of range there; clamped here)
        long long k = static_cast<long long>(lg->prm.overlap_rate * static_cast<double>(n));
        lg->new_n = std::min<long long>(std::max<long long>(k, 1), n);
        ICPX_TRY(lg->sort_keys.reserve(sizeof(unsigned long long) * n));
        ICPX_TRY(lg->sort_iota.reserve(sizeof(uint32_t) * n));
        ICPX_TRY(lg->sort_order.reserve(sizeof(uint32_t) * n));
        size_t bytes = 0;
        ICPX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, lg->nb_cost.as<unsigned long long>(),
                                                    lg->sort_keys.as<unsigned long long>(),
                                                    lg->sort_iota.as<uint32_t>(), lg->sort_order.as<uint32_t>(),
                                                    static_cast<int>(n), 0, 64, lg->stream));
        lg->sort_tmp_bytes = bytes ? bytes : 1;
        ICPX_TRY(lg->sort_tmp.reserve(lg->sort_tmp_bytes));
        hipLaunchKernelGGL(lg_iota_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, lg->stream, n,
                           lg->sort_iota.as<uint32_t>());
        ICPX_HIP(hipGetLastError());
    }
    // correspondence engine: tiled brute force for mid-size clouds, the uniform grid beyond
    // (ICPX_NN_GRID / ICPX_NN_BRUTE force one or the other)
    lg->use_bvh = false;
    lg->nn_seeded = false;
    lg->sd = false;
    lg->sd_dirty = false;
    lg->tile = lg->feature || lg->prm.nn_mode == ICPX_NN_BRUTE ||
               (lg->prm.nn_mode != ICPX_NN_GRID && lg->prm.nn_mode != ICPX_NN_GRID_F32 && lg->prm.nn_mode != ICPX_NN_BVH &&
                static_cast<double>(n) * static_cast<double>(m) <= static_cast<double>(kTileMaxWork));
    if (lg->tile) {
        const int64_t min_split = knobs().tile_min.load() > 0 ? knobs().tile_min.load() : kTileMinSplit;  // development knob ICPX_TILE_MIN
        const int64_t most = (m + min_split - 1) / min_split;  // splits of at least that many targets
        // queries per thread: four, unless the cloud then gives fewer than ~3 workgroups per CU even with the most
        // splits — one wave per SIMD leaves every load of the search exposed (queen, 10 720 points: 231 workgroups,
        // 1.43 ms; 882 workgroups with one query per thread, 1.21 ms); ICPX_TILE_QPT fixes it (development)
        const int64_t want_wgs = knobs().tile_wgs.load() > 0 ? knobs().tile_wgs.load() : 768;  // development knob ICPX_TILE_WGS
        lg->tile_qpt = kTileQpt;
        if (kTileQptAuto)
            for (int qpt = 4; qpt >= 1; qpt >>= 1) {
                lg->tile_qpt = qpt;
                if (((n + kTileThreads * qpt - 1) / (kTileThreads * qpt)) * most >= want_wgs) break;
            }
        const int64_t tile_queries = static_cast<int64_t>(kTileThreads) * lg->tile_qpt;
        const int64_t ntq = (n + tile_queries - 1) / tile_queries;
        int64_t splits = (want_wgs + ntq - 1) / ntq;                  // aim at ~3 workgroups per CU
        if (splits > most) splits = most;
        if (splits < 1) splits = 1;
        int64_t len = ((m + splits - 1) / splits + 3) & ~int64_t(3);
        if (len > kTileMaxSplit) len = kTileMaxSplit;
        splits = (m + len - 1) / len;
        if (splits > 65535) return fail(ICPX_ERR_UNSUPPORTED, "target cloud too large for the tiled search; use nn = grid");
        // interleaved splits with chunk skipping (lg_tile_nn_kernel): a split's capacity is a whole number of chunks
        lg->tile_interleaved = !lg->feature && knobs().tile_chunk != 0;
        lg->tile_first = true;
        if (lg->tile_interleaved) {
            const int64_t nc = (m + kChunk - 1) / kChunk;
            if (splits > nc) splits = nc;
            while (((nc + splits - 1) / splits) * kChunk > kTileMaxSplit) splits++;
            len = ((nc + splits - 1) / splits) * kChunk;
        }
        lg->splits = static_cast<int>(splits);
        lg->split_len = static_cast<int>(len);
        if (!lg->feature) {
            // the splits' filter frames and chunk boxes, once per begin (the targets are on the stream's way already)
            ICPX_TRY(lg->tile_prep.reserve(sizeof(double) * kTilePrepStride * static_cast<size_t>(splits)));
            auto pk = D == 2 ? lg_tile_prep_kernel<2> : lg_tile_prep_kernel<3>;
            ICPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pk), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(tile_lds_bytes(3, kTileMaxSplit))));
            hipLaunchKernelGGL(pk, dim3(1, static_cast<unsigned>(splits)), dim3(kTileThreads), tile_lds_bytes(D, static_cast<int>(len)),
                               lg->stream, lg->dst.as<double>(), m, static_cast<int>(len), static_cast<int>(len),
                               lg->tile_interleaved ? 1 : 0, lg->tile_prep.as<double>());
            ICPX_HIP(hipGetLastError());
        }
        ICPX_TRY(lg->part_d2.reserve(sizeof(double) * static_cast<size_t>(splits) * n));
        ICPX_TRY(lg->part_idx.reserve(sizeof(int32_t) * static_cast<size_t>(splits) * n));
    } else {
        // grid, or the BVH: on request (nn = bvh), or when the grid's cell budget leaves the cloud in a few cells
        // (far-apart clusters) and nothing was forced
        lg->use_bvh = lg->prm.nn_mode == ICPX_NN_BVH;
        if (!lg->use_bvh) {
            grid_set_f32(lg->grid, f32);
            // f32 filter records beside the f64 ones for the 3D methods' seeded per-iteration queries (ICPX_GRID_FILTER=0: none)
            grid_set_filter(lg->grid, D == 3 && !f32 && !(lg->prm.flags & ICPX_FLAG_FILL_MATCHES) && knobs().grid_filter != 0);
            ICPX_TRY(grid_build(lg->grid, D, lg->dst.as<double>(), m, lg->stream));
            lg->use_bvh = lg->prm.nn_mode == ICPX_NN_AUTO && grid_degenerate(lg->grid);
        }
        if (lg->use_bvh) {
            if (!lg->bvh) ICPX_TRY(bvh_create(&lg->bvh));
            ICPX_TRY(bvh_build(lg->bvh, D, lg->dst.as<double>(), m, lg->stream));
            ICPX_TRY(bvh_sort_queries(lg->bvh, lg->cur.as<double>(), n, lg->stream));
        } else {
            // the running cloud moves little between iterations: its initial cell order stays coherent
            ICPX_TRY(grid_sort_queries(lg->grid, lg->cur.as<double>(), n, lg->stream));
            // sorted domain (see LargeIcp::sd); development knob ICPX_GRID_SD=0
            if (D == 3 && lg->grid->q_perm_n == n && !(lg->prm.flags & ICPX_FLAG_FILL_MATCHES) && knobs().grid_coop != 2 &&
                knobs().grid_sd != 0) {
                ICPX_TRY(lg->cur_s.reserve(sizeof(double) * D * n));
                ICPX_TRY(lg->idx_s.reserve(sizeof(int64_t) * n));
                ICPX_TRY(lg->pt_s.reserve(sizeof(double) * D * n));
                hipLaunchKernelGGL(lg_sd_gather_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, lg->stream,
                                   lg->cur.as<double>(), lg->grid->q_perm.as<uint32_t>(), n, lg->cur_s.as<double>());
                ICPX_HIP(hipGetLastError());
                lg->sd = true;
            }
        }
    }
    if (lg->feature) {
        // the reference keeps the dense n x m feature-distance matrix too (feature_aware.h:60)
        if (static_cast<double>(n) * static_cast<double>(m) > static_cast<double>(kFeatureMaxWork))
            return fail(ICPX_ERR_UNSUPPORTED, "feature_aware keeps an n x m matrix: n*m is limited to %lld",
                        (long long)kFeatureMaxWork);
        if (lg->prm.symmetric_neighbors < 0 || lg->prm.symmetric_neighbors > 4096)
            return fail(ICPX_ERR_INVALID, "symmetric_neighbors out of range");
        const size_t w = 2 * static_cast<size_t>(lg->prm.symmetric_neighbors);
        const int64_t big = n > m ? n : m;
        ICPX_TRY(lg->part.reserve(sizeof(double) * 16 * static_cast<size_t>((big + kBlkPts - 1) / kBlkPts)));
        ICPX_TRY(lg->f_rad.reserve(sizeof(double) * big));
        ICPX_TRY(lg->f_feat_a.reserve(sizeof(double) * (w * n + 1)));
        ICPX_TRY(lg->f_feat_b.reserve(sizeof(double) * (w * m + 1)));
        ICPX_TRY(lg->f_nfd.reserve(sizeof(double) * static_cast<size_t>(n) * m));
    }
    ICPX_TRY(LG_DISPATCH(enqueue_setup, lg, 0, zero_criteria(), 0.0));
    if (lg->kd_rule && (D == 2 || (lg->prm.flags & ICPX_FLAG_FILL_MATCHES))) {
        // setup ran a correspondence step: it may have stopped at a tie of the kd rule
        bool stalled = false;
        ICPX_TRY(lg_kd_service(lg, lg_setup_mode(lg), &stalled));
        if (stalled) ICPX_TRY(LG_DISPATCH(enqueue_setup_tail, lg, 0, zero_criteria(), 0.0));
    }
    return ICPX_OK;
}

int large_iterate(LargeIcp* lg) {
    lg->sorted = true;
    ICPX_TRY(LG_DISPATCH(enqueue_iteration, lg, 0, zero_criteria(), 0.0));
    bool stalled = false;
    ICPX_TRY(lg_kd_service(lg, lg_iteration_mode(lg), &stalled));  // (no-op once the tree is on the device)
    if (stalled) ICPX_TRY(LG_DISPATCH(enqueue_iteration_tail, lg, 0, zero_criteria(), 0.0));
    return ICPX_OK;
}

int large_converge(LargeIcp* lg, const icpx_criteria* crit, std::chrono::steady_clock::time_point t0,
                   icpx_state* out) {
    // elapsed = (t0 -> enqueue of lg_init_kernel, host clock, sampled once) + (device clock since lg_init_kernel
    // ran).  Passing a fresh host time with every enqueue would count the device time twice from the second
    // chunk on (the host synchronises between chunks), and the limit would fire at half its value.
    const double host_offset = std::chrono::duration<double>(lg->init_at - t0).count();
    auto elapsed = [&]() { return host_offset > 0.0 ? host_offset : 0.0; };
    LargeDev* dev = lg->dev.as<LargeDev>();
    // large_begin already ran setup without a check: evaluate the loop's first should_terminate.
    // progress flags in pinned memory, written by the two kernels that can end the loop; the host polls them
    // (a copy of the flag + a stream synchronisation per chunk left the GPU idle for ~25 us each time)
    struct PollGuard {
        LargeIcp* l;
        ~PollGuard() { l->poll_dev = nullptr; }
    } guard{lg};
    lg->poll_dev = knobs().poll_done != 0 ? lg->h_done_dev : nullptr;
    lg->h_done[0] = 0;
    lg->h_done[1] = -1;
    lg->h_done[2] = 0;
    if (lg->dim == 2)
        hipLaunchKernelGGL(lg_setup_finish_kernel<2>, dim3(1), dim3(64), 0, lg->stream, dev, lg->n, 1, *crit,
                           lg->clock_to_s, elapsed(), lg->poll_dev);
    else
        hipLaunchKernelGGL(lg_setup_finish_kernel<3>, dim3(1), dim3(64), 0, lg->stream, dev, lg->n, 1, *crit,
                           lg->clock_to_s, elapsed(), lg->poll_dev);
    ICPX_HIP(hipGetLastError());
    uint64_t enq = 0;
    int chunk = 4;
    while (true) {
        // wait until the device reports `done`, a stall at a tie of the kd rule, or all `enq` iterations enqueued so far
        bool have = false;
        if (lg->poll_dev) {
            for (uint64_t spin = 1;; spin++) {
                const int ticks = __atomic_load_n(&lg->h_done[1], __ATOMIC_ACQUIRE);
                if (__atomic_load_n(&lg->h_done[0], __ATOMIC_ACQUIRE) != 0 || __atomic_load_n(&lg->h_done[2], __ATOMIC_ACQUIRE) != 0 ||
                    (ticks >= 0 && static_cast<uint64_t>(ticks) >= enq)) {
                    have = true;
                    break;
                }
                if ((spin & 0x3fff) == 0) {
                    const hipError_t q = hipStreamQuery(lg->stream);
                    if (q == hipSuccess) {
                        // the stream has drained: its kernels' stores are visible now (or never came: read the flag the patient way)
                        const int t2 = __atomic_load_n(&lg->h_done[1], __ATOMIC_ACQUIRE);
                        have = __atomic_load_n(&lg->h_done[0], __ATOMIC_ACQUIRE) != 0 || __atomic_load_n(&lg->h_done[2], __ATOMIC_ACQUIRE) != 0 ||
                               (t2 >= 0 && static_cast<uint64_t>(t2) >= enq);
                        break;
                    }
                    if (q != hipErrorNotReady) return fail(ICPX_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
                }
            }
        }
        if (!have) {
            ICPX_HIP(hipMemcpyAsync(lg->h_done, &dev->st.done, sizeof(int), hipMemcpyDeviceToHost, lg->stream));
            ICPX_HIP(hipStreamSynchronize(lg->stream));
        }
        if (!lg->h_done[0] && lg->kd_rule && !lg->kd_ready) {
            // (without the polled flags the stall shows only in device memory: lg_kd_service reads it)
            bool stalled = false;
            if (lg->h_done[2] || !have || !lg->poll_dev) {
                ICPX_HIP(hipStreamSynchronize(lg->stream));  // the kernels enqueued behind the stall return at once
                ICPX_TRY(lg_kd_service(lg, lg_iteration_mode(lg), &stalled));
            }
            if (stalled) {
                uint64_t it = 0;
                ICPX_HIP(hipMemcpyAsync(&it, &dev->st.iteration_count, sizeof(it), hipMemcpyDeviceToHost, lg->stream));
                ICPX_HIP(hipStreamSynchronize(lg->stream));
                lg->h_done[2] = 0;
                ICPX_TRY(LG_DISPATCH(enqueue_iteration_tail, lg, 1, *crit, elapsed()));  // the rest of the stalled iteration
                enq = it + 1;
                continue;
            }
        }
        if (lg->h_done[0]) break;
        uint64_t k = static_cast<uint64_t>(chunk);
        if ((crit->has & ICPX_HAS_MAX_ITER) && enq + k > crit->max_iterations)
            k = crit->max_iterations > enq ? crit->max_iterations - enq : 1;
        for (uint64_t j = 0; j < k; j++) ICPX_TRY(LG_DISPATCH(enqueue_iteration, lg, 1, *crit, elapsed()));
        enq += k;
        if (chunk < 16) chunk *= 2;
    }
    PairState st;
    ICPX_HIP(hipMemcpyAsync(&st, &dev->st, sizeof(PairState), hipMemcpyDeviceToHost, lg->stream));
    ICPX_HIP(hipStreamSynchronize(lg->stream));
    icpx_state res;
    std::memset(&res, 0, sizeof(res));
    res.cost = st.cost;
    res.iteration_count = st.iteration_count;
    if (st.iteration_count > 0) lg->sorted = true;
    const int XN = (lg->dim + 1) * (lg->dim + 1);
    for (int e = 0; e < XN; e++) res.transform[e] = st.transform[e];
    ICPX_HIP(hipMemcpy(out, &res, sizeof(res), hipMemcpyDefault));
    return ICPX_OK;
}

int large_cost(LargeIcp* lg, double* out) {
    ICPX_HIP(hipMemcpyAsync(out, &lg->dev.as<LargeDev>()->st.cost, sizeof(double), hipMemcpyDeviceToHost, lg->stream));
    ICPX_HIP(hipStreamSynchronize(lg->stream));
    return ICPX_OK;
}

int large_transform(LargeIcp* lg, double* out) {
    const int XN = (lg->dim + 1) * (lg->dim + 1);
    ICPX_HIP(hipMemcpyAsync(out, lg->dev.as<LargeDev>()->st.transform, sizeof(double) * XN, hipMemcpyDeviceToHost, lg->stream));
    ICPX_HIP(hipStreamSynchronize(lg->stream));
    return ICPX_OK;
}

// sorted domain: nb_idx / nb_cost / sel in point order, when a getter wants them
static int sd_unpack(LargeIcp* lg) {
    if (!lg->sd || !lg->sd_dirty) return ICPX_OK;
    const unsigned nb = static_cast<unsigned>((lg->n + 255) / 256);
    auto kern = lg->method == ICPX_TRIMMED ? lg_sd_unpack_kernel<ICPX_TRIMMED> : lg_sd_unpack_kernel<ICPX_VANILLA>;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, lg->stream, lg->cur_s.as<double>(), lg->idx_s.as<int64_t>(),
                       lg->grid->q_perm.as<uint32_t>(), lg->dst.as<double>(), lg->n, lg->prm.max_distance,
                       lg->nb_idx.as<int64_t>(), lg->nb_cost.as<double>(), lg->sel.as<uint8_t>());
    ICPX_HIP(hipGetLastError());
    lg->sd_dirty = false;
    return ICPX_OK;
}

int large_matches(LargeIcp* lg, int64_t* point, int64_t* pair, double* cost) {
    // 2D only (3D is answered by the caller).  Vanilla: point order.  Trimmed: point order after
    // begin, the sorted order Trimmed::iterate leaves behind (trimmed.cpp:55-56) afterwards.
    const int64_t n = lg->n;
    ICPX_TRY(sd_unpack(lg));
    const bool by_order = lg->dim == 2 && lg->method != ICPX_VANILLA && lg->sorted;
    DevBuf tp, tq, tc;
    int rc = tp.reserve(sizeof(int64_t) * n);
    if (rc == ICPX_OK) rc = tq.reserve(sizeof(int64_t) * n);
    if (rc == ICPX_OK) rc = tc.reserve(sizeof(double) * n);
    if (rc == ICPX_OK) {
        hipLaunchKernelGGL(lg_matches_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, lg->stream, n,
                           by_order ? lg->sort_order.as<uint32_t>() : nullptr, lg->nb_idx.as<int64_t>(),
                           lg->nb_cost.as<double>(), tp.as<int64_t>(), tq.as<int64_t>(), tc.as<double>());
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && point) e = hipMemcpyAsync(point, tp.p, sizeof(int64_t) * n, hipMemcpyDefault, lg->stream);
        if (e == hipSuccess && pair) e = hipMemcpyAsync(pair, tq.p, sizeof(int64_t) * n, hipMemcpyDefault, lg->stream);
        if (e == hipSuccess && cost) e = hipMemcpyAsync(cost, tc.p, sizeof(double) * n, hipMemcpyDefault, lg->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(lg->stream);
        if (e != hipSuccess) rc = fail(ICPX_ERR_HIP, "matches copy failed: %s", hipGetErrorString(e));
    }
    tp.release();
    tq.release();
    tc.release();
    return rc;
}

int large_neighbors(LargeIcp* lg, int64_t* idx, double* dist) {
    const int64_t n = lg->n;
    ICPX_TRY(sd_unpack(lg));
    if (idx) ICPX_HIP(hipMemcpyAsync(idx, lg->nb_idx.p, sizeof(int64_t) * n, hipMemcpyDefault, lg->stream));
    if (dist) ICPX_HIP(hipMemcpyAsync(dist, lg->nb_cost.p, sizeof(double) * n, hipMemcpyDefault, lg->stream));
    ICPX_HIP(hipStreamSynchronize(lg->stream));
    return ICPX_OK;
}

}  // namespace icpx
