// ref_kdtree_shim.cpp — builds oracle/_ref/libref_kdtree.so from the REFERENCE's own
// include/algo/kdtree.h, compiled from where it lies under /root/reference (never copied).
// This file is only the C-ABI driver around that header: it instantiates icp::KdTree with a
// plain POD point type (the header is templated on any PointT with operator[] and size()).
// TEST INFRASTRUCTURE: used to pin oracle/icp_oracle.cpp's kd-tree restatement (tests/test_oracle.py,
// tests/test_gpu_parity.py::test_nn_against_reference_kdtree_semantics).  bench.py's cpu_baseline times the
// oracle's own restatement ("port"), not this library.
#include <array>    // kdtree.h:67 uses std::array without including it
#include <cstdint>
#include <vector>

#include "algo/kdtree.h"

namespace {
template <int D>
struct Pt {
    double v[D];
    double operator[](int i) const { return v[i]; }
    int size() const { return D; }
};

template <int D>
struct Tree {
    icp::KdTree<Pt<D>> kt;
    explicit Tree(const double* t, int64_t m) {
        std::vector<Pt<D>> pts(static_cast<size_t>(m));
        for (int64_t j = 0; j < m; j++)
            for (int k = 0; k < D; k++) pts[j].v[k] = t[j * D + k];
        kt.build(pts, D);
    }
    void query(const double* q, int64_t n, int64_t* idx, double* dist) const {
        for (int64_t i = 0; i < n; i++) {
            Pt<D> p;
            for (int k = 0; k < D; k++) p.v[k] = q[i * D + k];
            double md = 0.0;
            idx[i] = kt.search(p, &md);
            dist[i] = md;
        }
    }
};
}  // namespace

extern "C" {
// one-shot: build on t (m points), answer n queries.  dist = what KdTree::search reports
// (Euclidean distance, kdtree.h:126).
void ref_kdtree_nn(int dim, const double* q, int64_t n, const double* t, int64_t m, int64_t* idx,
                   double* dist) {
    if (dim == 2) Tree<2>(t, m).query(q, n, idx, dist);
    else Tree<3>(t, m).query(q, n, idx, dist);
}

// split build / query so the two can be timed separately
void* ref_kdtree_build(int dim, const double* t, int64_t m) {
    if (dim == 2) return new Tree<2>(t, m);
    return new Tree<3>(t, m);
}
void ref_kdtree_query(int dim, void* tree, const double* q, int64_t n, int64_t* idx, double* dist) {
    if (dim == 2) static_cast<Tree<2>*>(tree)->query(q, n, idx, dist);
    else static_cast<Tree<3>*>(tree)->query(q, n, idx, dist);
}
void ref_kdtree_free(int dim, void* tree) {
    if (dim == 2) delete static_cast<Tree<2>*>(tree);
    else delete static_cast<Tree<3>*>(tree);
}
}
