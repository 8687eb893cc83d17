// icpx_large.h — host interface of the multi-kernel ("large cloud") ICP path and of the
// uniform-grid nearest-neighbour search.  Implemented in icpx_large.hip.
#pragma once

#include <chrono>

#include "icpx_host.h"

namespace icpx {

struct LargeIcp;

int large_create(int dim, int method, const icpx_params& prm, double clock_to_s, hipStream_t stream,
                 LargeIcp** out);
void large_destroy(LargeIcp* lg);
// ICP::begin + setup() (icp.h:120-133; vanilla.cpp:21-26, vanilla_3d.cpp:71-81, trimmed_3d.cpp:77-87)
int large_begin(LargeIcp* lg, const double* source, int64_t n, const double* target, int64_t m,
                const double* guess);
int large_iterate(LargeIcp* lg);
// ICPDriver::converge loop (driver.h:63-69): iterations are chained on the stream, termination is
// decided on the device; the host only polls a done flag every few iterations.
int large_converge(LargeIcp* lg, const icpx_criteria* crit,
                   std::chrono::steady_clock::time_point t0, icpx_state* out);
int large_cost(LargeIcp* lg, double* out);
int large_transform(LargeIcp* lg, double* out);
int large_matches(LargeIcp* lg, int64_t* point, int64_t* pair, double* cost);
int large_neighbors(LargeIcp* lg, int64_t* idx, double* dist);

// kd-tree tie rule of a stand-alone query (icpx_kd.h): the searches append the queries that have a tie partner
struct KdTies {
    const double* cloud = nullptr;  // the targets (device, AoS, cloud order)
    uint32_t* list = nullptr;       // n entries (device)
    int* count = nullptr;           // one word (device), cleared by the caller
};
// The reference's median-split tree (kdtree.h:96-116; equal coordinates ordered by index) of m host-resident points,
// in in-order layout: node[p] = point at in-order position p, pos = its inverse.
int kd_debug_tree(int dim, const double* d_t, int64_t m, uint32_t* h_node, hipStream_t stream);
// After a query with KdTies on `stream`: resolves the flagged queries (tree built on the host when there are any).
// d_q: the queries, d_idx: their (d2, index) answers, corrected in place.  Synchronises the stream.
int kd_resolve_queries(int dim, const double* d_q, int64_t n, const double* d_t, int64_t m, const KdTies& ties,
                       int64_t* d_idx, hipStream_t stream);

// Reusable search structure over one target cloud (device memory).
struct GridIndex;
int grid_create(GridIndex** out);
void grid_destroy(GridIndex* g);
// opt-in f32 correspondence mode (ICPX_NN_GRID_F32): the next grid_build writes 16-byte f32 records and the queries
// evaluate candidates in f32 (icpx_grid.h, GridRec32).  Off by default; set before grid_build.
void grid_set_f32(GridIndex* g, bool on);
// the next grid_build also writes f32 filter records (exact seeded queries through them: grid_scan_rows_filtered)
void grid_set_filter(GridIndex* g, bool on);
// d_t: DEVICE pointer, AoS dim x m.  Synchronises the stream once (bounding box read-back).
// defer: the first round's occupied-cell count is not waited for; call grid_settle before using occupancy or querying
int grid_build(GridIndex* g, int dim, const double* d_t, int64_t m, hipStream_t stream, bool defer = false);
int grid_settle(GridIndex* g, hipStream_t stream, bool* rebuilt);
// d_q / d_idx / d_d2: DEVICE pointers (idx, d2 may be null).  visited (device u64, may be null)
// accumulates the number of candidate targets examined.
int grid_query_launch(const GridIndex* g, const double* d_q, int64_t n, int64_t* d_idx, double* d_d2,
                      unsigned long long* d_visited, hipStream_t stream, const KdTies* ties = nullptr);
int grid_profile(GridIndex* g, int dim, const double* d_q, int64_t n, const double* d_t, int64_t m, hipStream_t stream, double* out);
struct BvhIndex;
// degenerate_to (may be null): the index to use instead when the grid turns out degenerate (grid_degenerate)
int grid_nn_oneshot(GridIndex* g, int dim, const double* d_q, int64_t n, const double* d_t, int64_t m,
                    int64_t* d_idx, double* d_d2, hipStream_t stream, BvhIndex* degenerate_to, const KdTies* ties = nullptr);

// Linear BVH over one target cloud (icpx_bvh.h): the alternative index for clouds that fill their bounding box unevenly.
struct BvhIndex;
int bvh_create(BvhIndex** out);
void bvh_destroy(BvhIndex* b);
// d_t: DEVICE pointer, AoS dim x m.  Synchronises the stream once (bounding box read-back).
int bvh_build(BvhIndex* b, int dim, const double* d_t, int64_t m, hipStream_t stream);
int bvh_nn_oneshot(BvhIndex* b, int dim, const double* d_q, int64_t n, const double* d_t, int64_t m,
                   int64_t* d_idx, double* d_d2, hipStream_t stream, const KdTies* ties = nullptr);

}  // namespace icpx
