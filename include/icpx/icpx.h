/*
 * icpx.h — the Eigen-free C ABI of the MI355X (gfx950) ICP engine.
 *
 * This is the drop-in boundary for cornellev/icp's hot path
 *   NN correspondence -> (trimmed) selection -> centroid + cross-covariance ->
 *   2x2 / 3x3 Kabsch SVD -> transform update -> ICPDriver convergence loop.
 * The reference has no process/device boundary there (everything is one C++ thread), so the
 * entry points below are what a binding of the reference's public classes calls; each one names
 * the reference interface it stands in for (paths relative to the reference checkout).
 * include/icp/ holds the C++ classes (icp::ICP2/ICP3, icp::ICPDriver, ...) written on top of
 * it; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - clouds: `const double*` to Dim x n COLUMN-MAJOR data, i.e. x0 y0 (z0) x1 ... — exactly
 *     icp::PointCloud<Dim>::data() (include/icp/geo.h:26-29).
 *   - transforms: (Dim+1) x (Dim+1) column-major — icp::RBTransform<Dim>::matrix().data().
 *   - every cloud/transform/output pointer may be a HOST pointer or a DEVICE (HIP) pointer; the
 *     library looks at the pointer's attributes and copies accordingly.  Inputs are always
 *     copied (ICP::begin deep-copies, include/icp/icp.h:125-126).
 *   - all functions return ICPX_OK (0) or an error code; icpx_last_error() gives the text.
 *     Nothing throws, nothing aborts.  A handle is not thread-safe (the reference's instances
 *     are not either); different handles may be used from different threads.
 *   - all arithmetic is IEEE f64, unfused (no FMA contraction), in the reference's operand order.
 *
 * Deterministic tie-break (documented rules), d2 = sum over axes, in axis order, of (target - query)^2:
 *   brute-force methods (Trimmed 2D, FeatureAware; icpx_nn): the nearest neighbour is the lexicographic
 *     minimum of (d2, target index) — bit for bit Trimmed::compute_matches (lib/icp/impl/trimmed.cpp:95-109)
 *     and the brute force the reference's kd-tree test treats as ground truth (tests/test.cpp:53-64);
 *   kd-tree methods (Vanilla 2D, Vanilla3d, Trimmed3d; icpx_nn_kdtree): KdTree::search's answer
 *     (include/algo/kdtree.h:132-154) — it compares sqrt(d2) with strict < and keeps the first node it
 *     visits, so among the targets that share the minimal sqrt(d2) (coincident targets, symmetric
 *     configurations, two d2 that round to one square root) the one that comes first in the
 *     near-child-first pre-order of the median-split tree (kdtree.h:96-116) wins.  Where a split
 *     subset holds equal coordinates the reference's tree is std::nth_element's choice
 *     (implementation-defined); here such coordinates are ordered by index, and among coincident
 *     targets the lowest index is returned.
 */
#ifndef ICPX_H
#define ICPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICPX_VERSION 10000 /* 1.0.0 */

enum icpx_status {
    ICPX_OK = 0,
    ICPX_ERR_INVALID = 1,     /* bad argument (null pointer, dim not 2/3, negative size ...) */
    ICPX_ERR_HIP = 2,         /* a HIP runtime call failed                                     */
    ICPX_ERR_NOMEM = 3,       /* device or host allocation failed                              */
    ICPX_ERR_STATE = 4,       /* call order violated (iterate before begin, icp.h:42-43)       */
    ICPX_ERR_UNSUPPORTED = 5, /* valid request this build does not implement                   */
    ICPX_ERR_NO_DEVICE = 6    /* no gfx950 device visible                                      */
};

/* ICP<Dim>::methods registry names (lib/icp/icp.cpp:17-27) */
enum icpx_method {
    ICPX_VANILLA = 0, /* "vanilla": Vanilla (2D, vanilla.cpp) / Vanilla3d (vanilla_3d.cpp) */
    ICPX_TRIMMED = 1, /* "trimmed": Trimmed (2D, trimmed.cpp) / Trimmed3d (trimmed_3d.cpp) */
    ICPX_FEATURE_AWARE = 2 /* "feature_aware": FeatureAware (2D only, feature_aware.cpp)      */
};

enum icpx_nn_mode {
    ICPX_NN_AUTO = 0,  /* brute force (LDS tiles, f32 filter + exact f64) up to ~3e4 x 3e4, uniform grid beyond */
    ICPX_NN_BRUTE = 1, /* LDS-tiled brute force                                     */
    ICPX_NN_GRID = 2,  /* exact uniform-grid search                                 */
    ICPX_NN_BVH = 3,   /* exact search through a linear bounding-volume hierarchy (Morton order, implicit
                          binary tree): the stand-in for the reference's kd-tree (kdtree.h:96-154) on clouds
                          that fill their bounding box very unevenly.  Same results as the other modes */
    ICPX_NN_GRID_F32 = 4 /* OPT-IN, NOT reference arithmetic: the uniform grid with 16-byte f32 records
                          (coordinates relative to the grid's lower corner) and candidates evaluated in f32.  The
                          returned target is the nearest up to that rounding — two targets whose distances from the
                          query differ by less than about 2^-21 of the cloud's extent may swap — so neither index
                          parity nor the kd-tree tie rule holds.  Everything downstream of the choice (match cost,
                          Trimmed3d's selection, sums, Kabsch) stays f64 on the original coordinates.  icpx_nn: d2 is
                          the f32-evaluated squared distance, widened; handles: Vanilla3d / Trimmed3d without flags
                          (Config nn = "grid_f32"); icpx_nn_kdtree and the 2D methods: ICPX_ERR_UNSUPPORTED.
                          Clouds whose extent is beyond ~1e18 overflow f32 squares: use the f64 modes */
};

/* Opt-in corrections of reference quirks (§8 f4).  NOT reference behaviour:
 * no counterpart in cornellev/icp; defaults (0) keep parity.  Handles only; such handles run on the
 * multi-kernel path. */
enum icpx_flags {
    ICPX_FLAG_MATCH_TRANSFORMED = 1, /* Vanilla (2D): match transform*a, every iteration, instead of a once (vanilla.cpp:80) */
    ICPX_FLAG_FILL_MATCHES = 2,      /* 3D methods: fill `matches` (point, neighbour, distance) in setup and iterate, so calculate_cost works */
    ICPX_FLAG_APPLY_GUESS = 4        /* 3D methods: the running cloud starts as guess*a (vanilla_3d.cpp:72 starts from a) */
};

/* What icp::Config carries for these methods (include/icp/config.h) plus placement. */
typedef struct icpx_params {
    double overlap_rate; /* Config "overlap_rate" — Trimmed(2D), trimmed.cpp:28, default 0.9   */
    double max_distance; /* Config "max_distance" — Trimmed3d, trimmed_3d.cpp:27, default 1.0  */
    int32_t nn_mode;     /* enum icpx_nn_mode                                                   */
    int32_t device;      /* HIP device ordinal; -1 = the calling thread's current device        */
    void* stream;        /* hipStream_t to run on; NULL = the null stream                       */
    double feature_weight;       /* Config "feature_weight" — FeatureAware, feature_aware.cpp:23, default 0.7      */
    int32_t symmetric_neighbors; /* Config "symmetric_neighbors" — FeatureAware, feature_aware.cpp:24, default 10 */
    int32_t flags;               /* enum icpx_flags; 0 = the reference's behaviour */
} icpx_params;

/* ICPDriver's optional termination criteria (include/icp/driver.h:80-151, 226-233). */
enum icpx_has {
    ICPX_HAS_MIN_ITER = 1,      /* set_min_iterations          driver.h:80  */
    ICPX_HAS_MAX_ITER = 2,      /* set_max_iterations          driver.h:90  */
    ICPX_HAS_STOP_COST = 4,     /* set_stop_cost               driver.h:101 */
    ICPX_HAS_REL_TOL = 8,       /* set_relative_cost_tolerance driver.h:113 */
    ICPX_HAS_ABS_TOL = 16,      /* set_absolute_cost_tolerance driver.h:124 */
    ICPX_HAS_TRANSFORM_TOL = 32,/* set_transform_tolerance     driver.h:138 */
    ICPX_HAS_TIME_LIMIT = 64    /* set_time_limit              driver.h:149 */
};

typedef struct icpx_criteria {
    uint32_t has; /* bit set of enum icpx_has */
    uint32_t _pad;
    uint64_t min_iterations;
    uint64_t max_iterations;
    double stop_cost;
    double relative_cost_tolerance;
    double absolute_cost_tolerance;
    double angle_tolerance_rad;
    double translation_tolerance;
    double time_limit_s;
} icpx_criteria;

/* ICPDriver::ConvergenceState (driver.h:24-33) */
typedef struct icpx_state {
    double cost;
    uint64_t iteration_count;
    double transform[16]; /* first (dim+1)^2 entries, column-major */
} icpx_state;

typedef struct icpx_ctx* icpx_handle;

/* Library-level ------------------------------------------------------------------------- */

/* No reference counterpart (the reference has no device).  Checks that `device` (-1 = current)
 * is a gfx950 GPU and warms the HIP runtime.  Optional: every entry point initialises lazily. */
int icpx_init(int device);

int icpx_version(void);

/* Text of the calling thread's most recent error ("" when none). */
const char* icpx_last_error(void);

/* Defaults as the reference's constructors read them from an empty Config. */
void icpx_default_params(icpx_params* p);

/* Instance life cycle ------------------------------------------------------------------- */

/* ICP<Dim>::from_method(name, config) (include/icp/icp.h:95-102) and the concrete
 * constructors Vanilla/Trimmed/Vanilla3d/Trimmed3d(const Config&).  dim in {2,3}. */
int icpx_create(int dim, int method, const icpx_params* params, icpx_handle* out);

/* ~ICP() (icp.h:116) */
int icpx_destroy(icpx_handle h);

/* ICP<Dim>::begin(source, target, guess) (icp.h:120-133) including the method's setup():
 * vanilla.cpp:21-26, trimmed.cpp:31-36 (initial matches), vanilla_3d.cpp:71-81,
 * trimmed_3d.cpp:77-87 (running cloud c = a, search structure on b). */
int icpx_begin(icpx_handle h, const double* source, int64_t n, const double* target, int64_t m,
               const double* guess);

/* ICP<Dim>::iterate() (icp.h:141): vanilla.cpp:28-66, trimmed.cpp:38-93,
 * vanilla_3d.cpp:83-96, trimmed_3d.cpp:89-116.  One device-side iteration, no host round trip
 * of any per-point data. */
int icpx_iterate(icpx_handle h);

/* ICP<Dim>::calculate_cost() (icp.h:149-155) */
int icpx_cost(icpx_handle h, double* out);

/* ICP<Dim>::current_transform() (icp.h:158-160); out: (dim+1)^2 doubles, column-major */
int icpx_transform(icpx_handle h, double* out);

/* ICP<Dim>::get_matches() (icp.h:167-169); each array has n entries (any may be NULL).
 * As in the reference: Trimmed (2D) leaves them sorted by cost (trimmed.cpp:55-56; order here
 * is (cost, point)); the 3D methods never fill them (all zero). */
int icpx_matches(icpx_handle h, int64_t* point, int64_t* pair, double* cost);

/* Source / target sizes given to the last begin/converge. */
int icpx_sizes(icpx_handle h, int64_t* n, int64_t* m);

/* No public reference counterpart: the Neighbors{distances, indices} that
 * Vanilla3d::nearest_neighbor (vanilla_3d.cpp:30-45) produced in the last iterate(), for
 * per-iteration index parity checks.  3D only; arrays of n entries (any may be NULL). */
int icpx_debug_neighbors(icpx_handle h, int64_t* indices, double* distances);

/* ICPDriver<Dim>::converge(source, target, guess) (driver.h:51-72) with should_terminate
 * (driver.h:154-222) evaluated on the device: begin + the whole iterate loop are ONE call, the
 * host reads back only the final ConvergenceState.  After it returns, icpx_cost/transform/
 * matches refer to the final iteration, as they would on the reference's instance. */
int icpx_converge(icpx_handle h, const double* source, int64_t n, const double* target, int64_t m,
                  const double* guess, const icpx_criteria* criteria, icpx_state* out);

/* Batched ICPDriver::converge over independent scan pairs (trajectory / loop-closure
 * workloads; no reference counterpart beyond a caller's for-loop, e.g. bench/bench.cpp:56-60, 123).
 * Pair p uses source[source_offsets[p] .. source_offsets[p+1]) (offsets in POINTS, n_pairs+1
 * entries, HOST memory, free to reuse when the call returns) and likewise for target.  guesses:
 * n_pairs x (dim+1)^2 column-major, or NULL for identity.  out_states: n_pairs entries (host or
 * device memory).  All pairs run on params->device (icpx_converge_batch_multi shards over devices).
 * Every registered method is accepted: pairs that fit the fused kernel (vanilla / trimmed, <= 4096
 * source points, target in LDS) run in fused launches, one workgroup per pair; the others
 * (feature_aware, flags, larger clouds) run one after the other on the multi-kernel path.
 * Ordering: when source, target, guesses and out_states are all DEVICE memory and every pair runs
 * fused, the call is STREAM-ORDERED on params->stream — it returns once the work is enqueued, and
 * out_states is valid for work enqueued on that stream afterwards (or after synchronising it).
 * Calls on different streams of one device run concurrently.  With any host array the call
 * returns after the results have arrived.
 * criteria must hold max_iterations or a time limit (ICPX_ERR_INVALID otherwise): the loop runs on
 * the device and cannot be interrupted like the reference's host loop (driver.h:43-44, 63). */
int icpx_converge_batch(int dim, int method, const icpx_params* params, int64_t n_pairs,
                        const double* source, const int64_t* source_offsets,
                        const double* target, const int64_t* target_offsets,
                        const double* guesses, const icpx_criteria* criteria,
                        icpx_state* out_states);

/* icpx_converge_batch sharded over the devices of one node: shard r of `device_count`
 * takes the contiguous block of pairs [ceil(n_pairs*r/device_count), ceil(n_pairs*(r+1)/device_count))
 * and runs it on devices[r] (NULL: device r) with one host thread and one stream per shard; there is no
 * exchange between devices, every shard writes its records straight into out_states (the "gather" of the
 * 144-byte results).  A device may be listed more than once (its shards then share it on separate
 * streams).  source / target / guesses / out_states: host memory, or memory every listed device can
 * reach with hipMemcpyDefault.  Returns after all shards have finished.  No reference counterpart. */
int icpx_converge_batch_multi(int dim, int method, const icpx_params* params, int device_count,
                              const int* devices, int64_t n_pairs,
                              const double* source, const int64_t* source_offsets,
                              const double* target, const int64_t* target_offsets,
                              const double* guesses, const icpx_criteria* criteria,
                              icpx_state* out_states);

/* icpx_converge_batch for callers that cannot allocate (tight serving loops, harnesses that forbid
 * hipMalloc; NOT usable under HIP graph capture as it stands: it copies the host offset arrays and
 * synchronises the stream before it returns).  NO counterpart in the reference.  Same arguments
 * and results as icpx_converge_batch, but the library calls no hipMalloc / hipFree: all device memory
 * beyond the caller's inputs and outputs (offsets, per-pair states, scheduler words, running clouds,
 * seeds, per-pair grids) lives in one static device array of about 20 MB that every GPU the library
 * runs on holds until the process ends.  Limits (ICPX_ERR_UNSUPPORTED beyond them): n_pairs <=
 * ICPX_NOALLOC_MAX_PAIRS, total source points and total target points <= ICPX_NOALLOC_MAX_POINTS
 * each, methods vanilla / trimmed without flags, every pair within the fused kernel's limits.
 * source, target, guesses and out_states must be DEVICE memory (nothing is staged); the offsets are
 * host memory as in icpx_converge_batch and the call returns after the stream has drained.  All
 * calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two
 * threads) race. */
#define ICPX_NOALLOC_MAX_PAIRS 256
#define ICPX_NOALLOC_MAX_POINTS 524288
int icpx_converge_batch_noalloc(int dim, int method, const icpx_params* params, int64_t n_pairs,
                                const double* source, const int64_t* source_offsets,
                                const double* target, const int64_t* target_offsets,
                                const double* guesses, const icpx_criteria* criteria,
                                icpx_state* out_states);

/* Development knobs (no reference counterpart).  The ICPX_* environment variables (ICPX_PRUNE,
 * ICPX_FILTER, ICPX_SCHED, ICPX_SCHED_STEPS, ICPX_PRUNE_HANDLES, ICPX_NO_GROUP_SUMS, ICPX_COOP,
 * ICPX_GRID_COOP, ICPX_GRID_SEED, ICPX_GRID_SD, ICPX_GRID_FILTER, ICPX_TILE_CHUNK, ICPX_KD_EAGER, ICPX_DEFER_SPARSE, ICPX_CHUNK_SKIP, ICPX_ZERO_COPY, ICPX_GRID_DENSITY, ICPX_TILE_WGS, ICPX_TILE_MIN, ICPX_POLL_DONE, ICPX_FUSE_PREPASS,
 * ICPX_FUSED_SINGLE_MAX, ICPX_GEOM) are read once, when the library is first used; this call changes one afterwards
 * (value NULL = built-in default); an unknown name is ICPX_ERR_INVALID.  Every setting computes the same
 * results; they select between kernels.  Not thread-safe against running calls. */
int icpx_debug_knob(const char* name, const char* value);

/* Test hook (no reference counterpart as a call; the structure is KdTree::build_tree, kdtree.h:96-116): the
 * median-split tree the kd-tree tie rule uses, as the DEVICE builds it, in in-order layout — node_out[p] = the
 * target at in-order position p (m entries, host memory); equal split coordinates are ordered by index.
 * targets: host or device memory. */
int icpx_debug_kd_tree(int dim, const double* targets, int64_t m, uint32_t* node_out, int device);

/* Measurement helper (no counterpart in the reference): one build + one query of the uniform-grid
 * search with device-memory inputs (queries, targets must be device pointers), timed with HIP
 * events on `stream`.  out[0] = build ms, out[1] = query ms, out[2] = mean number of target records
 * a query examined (the c-bar of SURVEY 8(d)'s bytes-per-query figure), out[3] = grid cells. */
int icpx_nn_grid_profile(int dim, const double* queries, int64_t n, const double* targets, int64_t m,
                         int device, void* stream, double* out);

/* Stand-alone correspondence query: the arithmetic of Trimmed::compute_matches
 * (trimmed.cpp:95-109) / KdTree::search (kdtree.h:79-89) under the documented tie-break.
 * idx: n x int64 target indices; d2: n x f64 squared distances (either may be NULL).
 * mode: enum icpx_nn_mode.  device/stream as in icpx_params. */
int icpx_nn(int dim, const double* queries, int64_t n, const double* targets, int64_t m,
            int64_t* idx, double* d2, int mode, int device, void* stream);

/* Stand-alone KdTree::search (include/algo/kdtree.h:79-89, 132-154) for n queries against the tree of
 * `targets` (kdtree.h:42-52, 96-116): idx = the target the reference's tree walk returns, dist = its
 * EUCLIDEAN distance (kdtree.h:126 takes the square root).  Same searches as icpx_nn; the queries whose
 * minimum has a tie partner at the same sqrt(d2) are then resolved by the tree's visiting order (see the
 * tie-break note at the top of this file).  mode / device / stream as icpx_nn. */
int icpx_nn_kdtree(int dim, const double* queries, int64_t n, const double* targets, int64_t m,
                   int64_t* idx, double* dist, int mode, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICPX_H */
