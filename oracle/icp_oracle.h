/*
 * icp_oracle.h — C ABI of the CPU ORACLE for the ICP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the reported CPU baseline.  The product path
 * (icp_amd/, include/icpx/icpx.h) never links, imports or calls it.
 *
 * What it is: a single-threaded, Eigen-free restatement of cornellev/icp's
 * algorithm for the path  NN correspondence -> (trimmed) selection -> centroid +
 * cross-covariance -> Kabsch SVD -> transform update -> convergence test.
 * Every function cites the reference file:line it follows (paths relative to the
 * reference checkout).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - kd-tree / NN arithmetic and tie behaviour: PINNED against the reference's
 *     real include/algo/kdtree.h compiled into oracle/_ref/libref_kdtree.so
 *     (tests/test_oracle.py) and against the reference's own kd-tree KAT setup
 *     (tests/test.cpp:39-81).
 *   - everything that goes through Eigen in the reference (JacobiSVD, transform
 *     products, rowwise().mean()): Eigen is an un-vendored, un-pinned dependency
 *     (CMakeLists.txt:39) that is absent from this image, so the reference's
 *     method classes cannot be built here.  Those parts restate Eigen 3.4's
 *     published algorithms and are anchored on the reference's own tests
 *     (tests/test.cpp:83-216, tests/test3d.cpp:36-202) whose tolerances are
 *     0.5 units / 0.01 rad.  Beyond those tolerances: PARITY UNPINNED.
 */
#ifndef ICP_ORACLE_H
#define ICP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_VANILLA = 0, ORC_TRIMMED = 1, ORC_FEATURE_AWARE = 2 /* ICP2 only */ };

/* how correspondences are found for the kd-tree methods (vanilla 2D, both 3D) */
enum {
    ORC_NN_KDTREE = 0, /* restated reference kd-tree: first-visited wins ties (kdtree.h:132-154) */
    ORC_NN_BRUTE = 1,  /* brute-force rule: lexicographic min of (d2, target index)             */
    ORC_NN_KDCANON = 2 /* the kd-tree's rule in canonical form (icp_oracle.cpp, KdCanon): first-visited among the
                          minimal sqrt(d2), tree ordered by (coordinate, index), lowest index among coincident
                          targets.  Equals ORC_NN_KDTREE wherever split coordinates are distinct; the rule the
                          HIP kernels implement for vanilla 2D and both 3D methods */
};

/* floating-point summation order of the centroid / covariance / cost reductions */
enum {
    ORC_SUM_SEQUENTIAL = 0, /* index order, as the reference's explicit loops (vanilla.cpp:38-48) */
    ORC_SUM_TREE64 = 1      /* radix-64 butterfly tree: the order the HIP kernels use            */
};

typedef struct {
    double overlap_rate; /* Trimmed (2D): trimmed.cpp:28, default 0.9  */
    double max_distance; /* Trimmed3d:    trimmed_3d.cpp:27, default 1 */
    int32_t nn_mode;     /* ORC_NN_*  */
    int32_t sum_mode;    /* ORC_SUM_* */
    double feature_weight;       /* FeatureAware: feature_aware.cpp:23, default 0.7 */
    int32_t symmetric_neighbors; /* FeatureAware: feature_aware.cpp:24, default 10  */
    int32_t flags;               /* ORC_FLAG_*: opt-in corrections of reference quirks; 0 = reference behaviour */
} orc_params;

/* "corrected" modes — NOT reference behaviour, defined by this restatement */
enum {
    ORC_FLAG_MATCH_TRANSFORMED = 1, /* Vanilla (2D): match transform*a instead of a, every iteration (fixes Q1) */
    ORC_FLAG_FILL_MATCHES = 2,      /* 3D methods: fill `matches` (point, neighbour, distance) in setup and in every iterate, so calculate_cost is meaningful (fixes Q4) */
    ORC_FLAG_APPLY_GUESS = 4        /* 3D methods: the running cloud starts as guess*a (fixes Q5) */
};

/* driver.h:226-233 — one optional per criterion */
enum {
    ORC_HAS_MIN_ITER = 1,
    ORC_HAS_MAX_ITER = 2,
    ORC_HAS_STOP_COST = 4,
    ORC_HAS_REL_TOL = 8,
    ORC_HAS_ABS_TOL = 16,
    ORC_HAS_TRANSFORM_TOL = 32,
    ORC_HAS_TIME_LIMIT = 64
};

typedef struct {
    uint32_t has;
    uint32_t _pad;
    uint64_t min_iterations;
    uint64_t max_iterations;
    double stop_cost;
    double relative_cost_tolerance;
    double absolute_cost_tolerance;
    double angle_tolerance_rad;
    double translation_tolerance;
    double time_limit_s;
} orc_criteria;

typedef struct {
    double cost;
    uint64_t iteration_count;
    double transform[16]; /* (dim+1)x(dim+1), column-major, first (dim+1)^2 entries */
} orc_state;

typedef struct orc_icp orc_icp;

/* ---- stand-alone pieces ---- */

/* trimmed.cpp:95-109 arithmetic: d2 = sum_axis (t - q)^2 in axis order, strict <, ascending j */
void orc_nn_brute(int dim, const double* q, int64_t n, const double* t, int64_t m, int64_t* idx,
                  double* d2);

/* kdtree.h:42-52,79-89,96-154 restated; dist is Euclidean (sqrt), as kdtree.h:126 */
void orc_nn_kdtree(int dim, const double* q, int64_t n, const double* t, int64_t m, int64_t* idx,
                   double* dist);

/* ORC_NN_KDCANON stand-alone.  by_key != 0: the brute-force formulation (minimum, tied set, smallest pre-order
 * key) instead of the tree walk — same answers, cross-checked in tests.  nodes (may be NULL): the canonical
 * tree in in-order layout, m entries (node of the range [lo, hi) sits at lo + (hi - lo - 1) / 2) */
void orc_nn_kdcanon(int dim, const double* q, int64_t n, const double* t, int64_t m, int64_t* idx,
                    double* dist, int by_key, int32_t* nodes);

/* Eigen 3.4 JacobiSVD (FullU|FullV) for dim x dim, column-major in/out */
void orc_jacobi_svd(int dim, const double* a, double* u, double* s, double* v);

/* best rigid step from a cross-covariance: R = V U^T with last-column reflection fix,
 * t = cb - R ca  (vanilla.cpp:50-63, vanilla_3d.cpp:53-66).  step is (dim+1)^2 column-major */
void orc_kabsch(int dim, const double* N, const double* ca, const double* cb, double* step);

double orc_sum(const double* v, int64_t n, int sum_mode);

/* ---- the ICP state machine (icp.h and the impl sources) ---- */
orc_icp* orc_create(int dim, int method, const orc_params* p);
void orc_destroy(orc_icp*);
void orc_begin(orc_icp*, const double* src, int64_t n, const double* dst, int64_t m,
               const double* guess);
void orc_iterate(orc_icp*);
double orc_cost(const orc_icp*);
void orc_transform(const orc_icp*, double* out);
/* icp.h:167-169. 3D methods never fill matches (all zero) */
void orc_matches(const orc_icp*, int64_t* point, int64_t* pair, double* cost);
/* 3D only: Neighbors of the last iterate() (vanilla_3d.cpp:30-45) */
void orc_neighbors(const orc_icp*, int64_t* idx, double* dist);
/* current (transformed) source cloud: a_current (2D) or c (3D) */
void orc_current_cloud(const orc_icp*, double* out);

/* driver.h:51-72 + 154-222.  trace_* may be NULL; when given they receive, per executed
 * iteration k (k < trace_cap): cost, transform ((dim+1)^2 each) */
void orc_converge(orc_icp*, const double* src, int64_t n, const double* dst, int64_t m,
                  const double* guess, const orc_criteria* crit, orc_state* out, int64_t trace_cap,
                  double* trace_cost, double* trace_transform);

#ifdef __cplusplus
}
#endif
#endif
