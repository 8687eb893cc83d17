// icp_oracle.cpp — CPU ORACLE (test infrastructure, see icp_oracle.h).
//
// Single-threaded, Eigen-free restatement of cornellev/icp's hot path.  Compile with
//   g++ -std=c++17 -O2 -ffp-contract=off
// (no FMA contraction: the reference's CI builds are plain x86-64 / aarch64 Release builds;
// unfused arithmetic is also what the HIP kernels are held to).
//
// All clouds are column-major Dim x n, i.e. interleaved x0 y0 (z0) x1 ... exactly like
// Eigen::Matrix<double, Dim, Dynamic> (geo.h:27).  Transforms are (Dim+1)x(Dim+1) column-major
// like Eigen::Transform::matrix().

#include "icp_oracle.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// summation orders
// ---------------------------------------------------------------------------------------------

// 64-lane xor butterfly: the order a wavefront's shuffle reduction uses.
double butterfly64(double* v) {
    for (int s = 32; s >= 1; s >>= 1) {
        double nv[64];
        for (int l = 0; l < 64; l++) nv[l] = v[l] + v[l ^ s];
        std::memcpy(v, nv, sizeof(nv));
    }
    return v[0];
}

// radix-64 tree: groups of 64 consecutive values are butterfly-summed (zero padded), then the
// group sums are reduced the same way, until one value is left.
double sum_tree64(const double* v, int64_t n) {
    if (n <= 0) return 0.0;
    std::vector<double> level(v, v + n);
    while (true) {
        int64_t groups = (static_cast<int64_t>(level.size()) + 63) / 64;
        std::vector<double> next(groups);
        for (int64_t g = 0; g < groups; g++) {
            double lane[64];
            for (int l = 0; l < 64; l++) {
                int64_t i = g * 64 + l;
                lane[l] = i < static_cast<int64_t>(level.size()) ? level[i] : 0.0;
            }
            next[g] = butterfly64(lane);
        }
        if (groups == 1) return next[0];
        level.swap(next);
    }
}

double sum_seq(const double* v, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += v[i];
    return s;
}

double sum_mode(const std::vector<double>& v, int mode) {
    return mode == ORC_SUM_TREE64 ? sum_tree64(v.data(), static_cast<int64_t>(v.size()))
                                  : sum_seq(v.data(), static_cast<int64_t>(v.size()));
}

// ---------------------------------------------------------------------------------------------
// nearest neighbour
// ---------------------------------------------------------------------------------------------

// squared distance in axis order, unfused (trimmed.cpp:101 `(b.col(j) - a_current.col(i)).squaredNorm()`,
// kdtree.h:121-125 `dist += (p[i]-q[i])*(p[i]-q[i])`; the two differ only by the sign of the
// difference, which squaring removes exactly).
template <int D>
inline double sqdist(const double* t, const double* q) {
    double d = 0.0;
    for (int k = 0; k < D; k++) {
        double df = t[k] - q[k];
        d += df * df;
    }
    return d;
}

template <int D>
void nn_brute(const double* q, int64_t n, const double* t, int64_t m, int64_t* idx, double* d2) {
    // trimmed.cpp:95-109
    for (int64_t i = 0; i < n; i++) {
        double best = std::numeric_limits<double>::infinity();
        int64_t bj = 0;  // the reference leaves `pair` untouched when m == 0; we define 0
        for (int64_t j = 0; j < m; j++) {
            double d = sqdist<D>(t + j * D, q + i * D);
            if (d < best) {
                best = d;
                bj = j;
            }
        }
        idx[i] = bj;
        d2[i] = best;
    }
}

#ifdef ORC_REFERENCE_KDTREE
}  // namespace
// oracle/_ref/liboracle_ref.so (oracle/Makefile): the SAME oracle with ORC_NN_KDTREE served by the reference's own
// include/algo/kdtree.h, compiled from where it lies (-I$(REF)/include) — bench.py's cpu_baseline ("kind":
// "reference" for the kd-tree legs).  This wrapper only adapts the point type.
#include "algo/kdtree.h"
namespace {
template <int D>
class KdTree {
public:
    void build(const double* pts, int64_t m) {
        std::vector<Pt> p(static_cast<size_t>(m));
        for (int64_t j = 0; j < m; j++)
            for (int k = 0; k < D; k++) p[j].v[k] = pts[j * D + k];
        kt_.build(p, D);
    }
    int search(const double* query, double* min_dist) const {
        Pt q;
        for (int k = 0; k < D; k++) q.v[k] = query[k];
        return kt_.search(q, min_dist);
    }

private:
    struct Pt {
        double v[D];
        double operator[](int i) const { return v[i]; }
        int size() const { return D; }
    };
    icp::KdTree<Pt> kt_;
};
#else
// kdtree.h restated.  Nodes are kept in a flat vector instead of unique_ptrs; the build order,
// split rule and traversal are the reference's.
template <int D>
class KdTree {
public:
    void build(const double* pts, int64_t m) {
        // kdtree.h:42-52
        pts_ = pts;
        nodes_.clear();
        nodes_.reserve(static_cast<size_t>(m));
        std::vector<int> indices(static_cast<size_t>(m));
        std::iota(indices.begin(), indices.end(), 0);
        root_ = build_recursive(indices.data(), static_cast<int>(m), 0);
    }

    int search(const double* query, double* min_dist) const {
        // kdtree.h:79-89
        int guess = -1;
        double md = std::numeric_limits<double>::max();
        search_recursive(query, root_, &guess, &md);
        if (min_dist != nullptr) *min_dist = md;
        return guess;
    }

private:
    struct Node {
        int idx;
        int axis;
        int next[2];
    };

    int build_recursive(int* indices, int npoints, int depth) {
        // kdtree.h:96-116
        if (npoints <= 0) return -1;
        const int axis = depth % D;
        const int mid = (npoints - 1) / 2;
        const double* p = pts_;
        std::nth_element(indices, indices + mid, indices + npoints,
                         [p, axis](int lhs, int rhs) { return p[lhs * D + axis] < p[rhs * D + axis]; });
        int me = static_cast<int>(nodes_.size());
        nodes_.push_back(Node{indices[mid], axis, {-1, -1}});
        int l = build_recursive(indices, mid, depth + 1);
        int r = build_recursive(indices + mid + 1, npoints - mid - 1, depth + 1);
        nodes_[me].next[0] = l;
        nodes_[me].next[1] = r;
        return me;
    }

    static double distance(const double* p, const double* q) {
        // kdtree.h:121-127 — returns the EUCLIDEAN distance (sqrt), despite the comment there
        double dist = 0;
        for (int i = 0; i < D; i++) dist += (p[i] - q[i]) * (p[i] - q[i]);
        return std::sqrt(dist);
    }

    void search_recursive(const double* query, int node, int* guess, double* min_dist) const {
        // kdtree.h:132-154
        if (node < 0) return;
        const Node& nd = nodes_[node];
        const double* train = pts_ + static_cast<int64_t>(nd.idx) * D;
        const double dist = distance(query, train);
        if (dist < *min_dist) {
            *min_dist = dist;
            *guess = nd.idx;
        }
        const int axis = nd.axis;
        const int dir = query[axis] < train[axis] ? 0 : 1;
        search_recursive(query, nd.next[dir], guess, min_dist);
        const double diff = std::fabs(query[axis] - train[axis]);
        if (diff < *min_dist) search_recursive(query, nd.next[dir == 0], guess, min_dist);
    }

    const double* pts_ = nullptr;
    std::vector<Node> nodes_;
    int root_ = -1;
};

#endif  // ORC_REFERENCE_KDTREE

// ---------------------------------------------------------------------------------------------
// The kd-tree RULE, canonical form (ORC_NN_KDCANON) — what the HIP kernels are held to for the
// kd-tree-backed methods (vanilla 2D, both 3D).
//
// kdtree.h:132-154 compares sqrt(d2) with strict < and keeps the first node it visits, children
// near side first.  Pruning never hides that node, so the reference's answer is
//   among the targets whose sqrt(d2) equals the minimal sqrt(d2), bit for bit, the one that comes
//   FIRST in the near-child-first pre-order of the tree for this query.
// The tree (kdtree.h:96-116) is the median-split tree, median = element (n - 1) / 2 of the subset
// ordered by the split coordinate.  With distinct coordinates that tree is unique; where a subset
// holds equal coordinates std::nth_element's pick is implementation-defined, and this canonical
// form orders (coordinate, index) — still a tree the reference could have built.  Coincident targets
// (all coordinates equal) are indistinguishable to the reference's callers except by index, whose
// choice among them is again nth_element's: the canonical answer is the LOWEST index among the
// targets coincident with the first-visited one.
// Two formulations, cross-checked by tests/test_oracle.py: search() walks the tree like the
// reference; search_by_key() takes the brute-force minimum, collects the tied set and compares
// pre-order keys (the formulation of the kernels' tie resolution, icp_amd/csrc/icpx_kd.h).
// ---------------------------------------------------------------------------------------------
inline uint64_t ord_key(double v) {  // order-preserving f64 -> u64 (a total order, NaN included)
    uint64_t b;
    std::memcpy(&b, &v, sizeof(b));
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}

template <int D>
class KdCanon {
public:
    void build(const double* pts, int64_t m) {
        pts_ = pts;
        m_ = static_cast<int>(m);
        node_.resize(static_cast<size_t>(m));
        std::iota(node_.begin(), node_.end(), 0);
        build_range(0, m_, 0);
        pos_.assign(static_cast<size_t>(m), 0);
        for (int p = 0; p < m_; p++) pos_[node_[p]] = p;
        // lowest index among coincident targets (-0.0 == +0.0; NaN coincides with nothing)
        std::vector<int> order(static_cast<size_t>(m));
        std::iota(order.begin(), order.end(), 0);
        auto nk = [this](int i, int d) { double v = pts_[static_cast<int64_t>(i) * D + d]; return ord_key(v == 0.0 ? 0.0 : v); };
        std::sort(order.begin(), order.end(), [&](int x, int y) {
            for (int d = 0; d < D; d++) {
                uint64_t kx = nk(x, d), ky = nk(y, d);
                if (kx != ky) return kx < ky;
            }
            return x < y;
        });
        dupmin_.assign(static_cast<size_t>(m), 0);
        for (int r = 0; r < m_; r++) {
            int i = order[r];
            bool same = r > 0;
            if (same) {
                int j = order[r - 1];
                for (int d = 0; d < D; d++) same = same && pts_[static_cast<int64_t>(i) * D + d] == pts_[static_cast<int64_t>(j) * D + d];
            }
            dupmin_[i] = same ? dupmin_[order[r - 1]] : i;
        }
    }

    // tree walk of the reference (kdtree.h:132-154) on the canonical tree; dist is Euclidean
    int search(const double* q, double* dist) const {
        int guess = -1;
        double md = std::numeric_limits<double>::max();
        walk(q, 0, m_, 0, &guess, &md);
        if (guess < 0) {  // nothing at a finite distance: the (d2, index) rule's convention
            *dist = std::numeric_limits<double>::infinity();
            return 0;
        }
        *dist = md;
        return dupmin_[guess];
    }

    // pre-order key of target t for query q: 2 bits per level from the root, 0 = this node, 1 = near child, 2 = far
    uint64_t key(const double* q, int t) const {
        const int P = pos_[t];
        int lo = 0, hi = m_;
        uint64_t k = 0;
        for (int depth = 0; depth < 31; depth++) {
            const int mid = lo + (hi - lo - 1) / 2;
            if (P == mid) break;
            const int axis = depth % D;
            const bool near_left = q[axis] < pts_[static_cast<int64_t>(node_[mid]) * D + axis];  // kdtree.h:146
            const bool go_left = P < mid;
            k |= static_cast<uint64_t>(go_left == near_left ? 1 : 2) << (60 - 2 * depth);
            if (go_left) hi = mid;
            else lo = mid + 1;
        }
        return k;
    }

    // the same answer without walking: brute-force minimum, tied set, smallest key
    int search_by_key(const double* q, double* dist) const {
        double best = std::numeric_limits<double>::infinity();
        for (int j = 0; j < m_; j++) {
            double d = sqdist<D>(pts_ + static_cast<int64_t>(j) * D, q);
            if (d < best) best = d;
        }
        if (!(best < std::numeric_limits<double>::infinity())) {
            *dist = std::numeric_limits<double>::infinity();
            return 0;
        }
        const double s = std::sqrt(best);
        int win = -1;
        uint64_t wk = 0;
        for (int j = 0; j < m_; j++) {
            double d = sqdist<D>(pts_ + static_cast<int64_t>(j) * D, q);
            if (std::sqrt(d) == s) {
                uint64_t k = key(q, j);
                if (win < 0 || k < wk) {
                    win = j;
                    wk = k;
                }
            }
        }
        *dist = s;
        return dupmin_[win];
    }

    const std::vector<int>& nodes() const { return node_; }

private:
    void build_range(int lo, int hi, int depth) {
        if (hi - lo <= 0) return;
        const int axis = depth % D;
        const int mid = lo + (hi - lo - 1) / 2;
        const double* p = pts_;
        std::nth_element(node_.begin() + lo, node_.begin() + mid, node_.begin() + hi, [p, axis](int x, int y) {
            uint64_t kx = ord_key(p[static_cast<int64_t>(x) * D + axis]), ky = ord_key(p[static_cast<int64_t>(y) * D + axis]);
            return kx < ky || (kx == ky && x < y);
        });
        build_range(lo, mid, depth + 1);
        build_range(mid + 1, hi, depth + 1);
    }

    void walk(const double* q, int lo, int hi, int depth, int* guess, double* md) const {
        if (hi - lo <= 0) return;
        const int mid = lo + (hi - lo - 1) / 2;
        const double* train = pts_ + static_cast<int64_t>(node_[mid]) * D;
        double dist = 0;
        for (int i = 0; i < D; i++) dist += (q[i] - train[i]) * (q[i] - train[i]);
        dist = std::sqrt(dist);
        if (dist < *md) {
            *md = dist;
            *guess = node_[mid];
        }
        const int axis = depth % D;
        const bool left_first = q[axis] < train[axis];
        if (left_first) walk(q, lo, mid, depth + 1, guess, md);
        else walk(q, mid + 1, hi, depth + 1, guess, md);
        const double diff = std::fabs(q[axis] - train[axis]);
        if (diff < *md) {
            if (left_first) walk(q, mid + 1, hi, depth + 1, guess, md);
            else walk(q, lo, mid, depth + 1, guess, md);
        }
    }

    const double* pts_ = nullptr;
    int m_ = 0;
    std::vector<int> node_, pos_, dupmin_;
};

// ---------------------------------------------------------------------------------------------
// small dense algebra (column-major, D x D) restating the Eigen 3.4 routines the reference calls
// ---------------------------------------------------------------------------------------------

struct Rot {  // Eigen::JacobiRotation<double>
    double c, s;
};

// JacobiRotation::makeJacobi(x, y, z)  (Eigen/src/Jacobi/Jacobi.h)
Rot make_jacobi(double x, double y, double z) {
    Rot r;
    double deno = 2.0 * std::fabs(y);
    if (deno < std::numeric_limits<double>::min()) {
        r.c = 1.0;
        r.s = 0.0;
    } else {
        double tau = (x - z) / deno;
        double w = std::sqrt(tau * tau + 1.0);
        double t;
        if (tau > 0.0) {
            t = 1.0 / (tau + w);
        } else {
            t = 1.0 / (tau - w);
        }
        double sign_t = t > 0.0 ? 1.0 : -1.0;
        double n = 1.0 / std::sqrt(t * t + 1.0);
        r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
        r.c = n;
    }
    return r;
}

// JacobiRotation::operator* and transpose (real case)
inline Rot rot_mul(Rot a, Rot b) { return Rot{a.c * b.c - a.s * b.s, a.c * b.s + a.s * b.c}; }
inline Rot rot_t(Rot a) { return Rot{a.c, -a.s}; }

// internal::apply_rotation_in_the_plane(x, y, j): x_i = c x_i + s y_i ; y_i = -s x_i + c y_i
inline void rot_apply(double& x, double& y, Rot j) {
    double xi = x, yi = y;
    x = j.c * xi + j.s * yi;
    y = -j.s * xi + j.c * yi;
}

template <int D>
struct Mat {
    double a[D * D];  // column-major
    double& operator()(int r, int c) { return a[c * D + r]; }
    double operator()(int r, int c) const { return a[c * D + r]; }
    static Mat identity() {
        Mat m;
        for (int i = 0; i < D * D; i++) m.a[i] = 0.0;
        for (int i = 0; i < D; i++) m(i, i) = 1.0;
        return m;
    }
    // MatrixBase::applyOnTheLeft(p, q, j): rows p and q
    void apply_left(int p, int q, Rot j) {
        if (j.c == 1.0 && j.s == 0.0) return;
        for (int c = 0; c < D; c++) rot_apply((*this)(p, c), (*this)(q, c), j);
    }
    // MatrixBase::applyOnTheRight(p, q, j): columns p and q with j.transpose()
    void apply_right(int p, int q, Rot j) {
        Rot jt = rot_t(j);
        if (jt.c == 1.0 && jt.s == 0.0) return;
        for (int r = 0; r < D; r++) rot_apply((*this)(r, p), (*this)(r, q), jt);
    }
};

// internal::real_2x2_jacobi_svd (Eigen/src/misc/RealSvd2x2.h)
template <int D>
void real_2x2_jacobi_svd(const Mat<D>& w, int p, int q, Rot* j_left, Rot* j_right) {
    Mat<2> m;
    m(0, 0) = w(p, p);
    m(0, 1) = w(p, q);
    m(1, 0) = w(q, p);
    m(1, 1) = w(q, q);
    Rot rot1;
    double t = m(0, 0) + m(1, 1);
    double d = m(1, 0) - m(0, 1);
    if (std::fabs(d) < std::numeric_limits<double>::min()) {
        rot1.s = 0.0;
        rot1.c = 1.0;
    } else {
        double u = t / d;
        double tmp = std::sqrt(1.0 + u * u);
        rot1.s = 1.0 / tmp;
        rot1.c = u / tmp;
    }
    m.apply_left(0, 1, rot1);
    *j_right = make_jacobi(m(0, 0), m(0, 1), m(1, 1));
    *j_left = rot_mul(rot1, rot_t(*j_right));
}

// Eigen::JacobiSVD<...>::compute for a square real matrix with ComputeFullU | ComputeFullV
// (Eigen/src/SVD/JacobiSVD.h): scale, two-sided Jacobi sweeps over (p, q<p), sign fix, sort.
// Call sites: vanilla.cpp:50, trimmed.cpp:75, vanilla_3d.cpp:53, trimmed_3d.cpp:59.
template <int D>
void jacobi_svd(const Mat<D>& in, Mat<D>& U, double* S, Mat<D>& V) {
    const double precision = 2.0 * std::numeric_limits<double>::epsilon();
    const double consider_as_zero = std::numeric_limits<double>::min();

    double scale = 0.0;
    bool finite = true;
    for (int i = 0; i < D * D; i++) {
        double v = std::fabs(in.a[i]);
        if (!std::isfinite(v)) finite = false;
        if (v > scale) scale = v;
    }
    U = Mat<D>::identity();
    V = Mat<D>::identity();
    if (!finite) {
        // InvalidInput: Eigen leaves U, V, S unset; define NaN so the caller propagates NaN
        for (int i = 0; i < D; i++) S[i] = std::numeric_limits<double>::quiet_NaN();
        for (int i = 0; i < D * D; i++) U.a[i] = V.a[i] = std::numeric_limits<double>::quiet_NaN();
        return;
    }
    if (scale == 0.0) scale = 1.0;

    Mat<D> w;
    for (int i = 0; i < D * D; i++) w.a[i] = in.a[i] / scale;

    double max_diag = 0.0;
    for (int i = 0; i < D; i++) max_diag = std::max(max_diag, std::fabs(w(i, i)));

    bool finished = false;
    while (!finished) {
        finished = true;
        for (int p = 1; p < D; ++p) {
            for (int q = 0; q < p; ++q) {
                double threshold = std::max(consider_as_zero, precision * max_diag);
                if (std::fabs(w(p, q)) > threshold || std::fabs(w(q, p)) > threshold) {
                    finished = false;
                    Rot j_left, j_right;
                    real_2x2_jacobi_svd(w, p, q, &j_left, &j_right);
                    w.apply_left(p, q, j_left);
                    U.apply_right(p, q, rot_t(j_left));
                    w.apply_right(p, q, j_right);
                    V.apply_right(p, q, j_right);
                    max_diag = std::max(max_diag, std::max(std::fabs(w(p, p)), std::fabs(w(q, q))));
                }
            }
        }
    }

    for (int i = 0; i < D; i++) {
        double a = w(i, i);
        S[i] = std::fabs(a);
        if (a < 0.0)
            for (int r = 0; r < D; r++) U(r, i) = -U(r, i);
    }
    for (int i = 0; i < D; i++) S[i] *= scale;

    for (int i = 0; i < D; i++) {
        int pos = 0;
        double best = S[i];
        for (int k = i + 1; k < D; k++)
            if (S[k] > best) {
                best = S[k];
                pos = k - i;
            }
        if (best == 0.0) break;
        if (pos) {
            pos += i;
            std::swap(S[i], S[pos]);
            for (int r = 0; r < D; r++) std::swap(U(r, i), U(r, pos));
            for (int r = 0; r < D; r++) std::swap(V(r, i), V(r, pos));
        }
    }
}

template <int D>
double det(const Mat<D>& m);
template <>
double det<2>(const Mat<2>& m) {
    return m(0, 0) * m(1, 1) - m(1, 0) * m(0, 1);
}
template <>
double det<3>(const Mat<3>& m) {
    // Eigen bruteforce_det3_helper order
    auto h = [&](int a, int b, int c) {
        return m(0, a) * (m(1, b) * m(2, c) - m(1, c) * m(2, b));
    };
    return h(0, 1, 2) - h(1, 0, 2) + h(2, 0, 1);
}

// R = V * U^T
template <int D>
Mat<D> v_ut(const Mat<D>& V, const Mat<D>& U) {
    Mat<D> R;
    for (int i = 0; i < D; i++)
        for (int j = 0; j < D; j++) {
            double s = V(i, 0) * U(j, 0);
            for (int k = 1; k < D; k++) s += V(i, k) * U(j, k);
            R(i, j) = s;
        }
    return R;
}

// affine (D+1)x(D+1) column-major helpers
template <int D>
struct Xf {
    double m[(D + 1) * (D + 1)];
    double& operator()(int r, int c) { return m[c * (D + 1) + r]; }
    double operator()(int r, int c) const { return m[c * (D + 1) + r]; }
    static Xf identity() {
        Xf x;
        for (int i = 0; i < (D + 1) * (D + 1); i++) x.m[i] = 0.0;
        for (int i = 0; i <= D; i++) x(i, i) = 1.0;
        return x;
    }
};

// Transform * Transform (vanilla.cpp:65 `transform = step * transform`):
// linear = L1*L2, translation = L1*t2 + t1, sums left to right.
template <int D>
Xf<D> xf_mul(const Xf<D>& a, const Xf<D>& b) {
    Xf<D> r = Xf<D>::identity();
    for (int i = 0; i < D; i++) {
        for (int j = 0; j < D; j++) {
            double s = a(i, 0) * b(0, j);
            for (int k = 1; k < D; k++) s += a(i, k) * b(k, j);
            r(i, j) = s;
        }
        double s = a(i, 0) * b(0, D);
        for (int k = 1; k < D; k++) s += a(i, k) * b(k, D);
        r(i, D) = s + a(i, D);
    }
    return r;
}

// Transform * point (vanilla.cpp:33 `a_current = transform * a`): (L*p) + t, sums left to right.
template <int D>
inline void xf_apply(const Xf<D>& x, const double* p, double* out) {
    double tmp[D];
    for (int i = 0; i < D; i++) {
        double s = x(i, 0) * p[0];
        for (int k = 1; k < D; k++) s += x(i, k) * p[k];
        tmp[i] = s + x(i, D);
    }
    for (int i = 0; i < D; i++) out[i] = tmp[i];
}

// Kabsch step from cross-covariance N (vanilla.cpp:50-63, trimmed.cpp:75-90,
// vanilla_3d.cpp:53-66, trimmed_3d.cpp:59-72)
template <int D>
Xf<D> kabsch(const Mat<D>& N, const double* ca, const double* cb) {
    Mat<D> U, V;
    double S[D];
    jacobi_svd<D>(N, U, S, V);
    Mat<D> R = v_ut<D>(V, U);
    if (det<D>(R) < 0) {
        for (int r = 0; r < D; r++) V(r, D - 1) = -V(r, D - 1);  // V.col(Dim-1) *= -1
        R = v_ut<D>(V, U);
    }
    Xf<D> step = Xf<D>::identity();
    for (int i = 0; i < D; i++) {
        for (int j = 0; j < D; j++) step(i, j) = R(i, j);
        double s = R(i, 0) * ca[0];
        for (int k = 1; k < D; k++) s += R(i, k) * ca[k];
        step(i, D) = cb[i] - s;
    }
    return step;
}

// ---------------------------------------------------------------------------------------------
// the ICP instances
// ---------------------------------------------------------------------------------------------

struct Match {  // icp.h:61-70
    int64_t point;
    int64_t pair;
    double cost;
};

struct IcpBase {
    virtual ~IcpBase() = default;
    virtual int dim() const = 0;
    virtual void begin(const double* src, int64_t n, const double* dst, int64_t m,
                       const double* guess) = 0;
    virtual void iterate() = 0;
    virtual double cost() const = 0;
    virtual void get_transform(double* out) const = 0;
    virtual void get_matches(int64_t* point, int64_t* pair, double* cost) const = 0;
    virtual void get_neighbors(int64_t* idx, double* dist) const = 0;
    virtual void get_current(double* out) const = 0;
};

template <int D>
struct Icp : IcpBase {
    int method;
    orc_params prm;
    Xf<D> transform = Xf<D>::identity();
    std::vector<double> a, b;  // D x n, D x m
    int64_t n = 0, m = 0;
    std::vector<Match> matches;
    std::vector<double> cur;  // a_current (2D) / c (3D)
    KdTree<D> kdtree;         // 3D: built once in setup (vanilla_3d.cpp:75-80)
    KdCanon<D> kdcanon;       // ORC_NN_KDCANON: the same rule in canonical form (device order)
    bool canon_by_key = false;  // non-finite input: the brute-force formulation (no pruning to reason about)
    std::vector<int64_t> nb_idx;
    std::vector<double> nb_dist;
    std::vector<double> nfd;  // FeatureAware: normalized_feature_dists, n x m row-major (feature_aware.h:60)

    Icp(int method_, const orc_params& p) : method(method_), prm(p) {}
    int dim() const override { return D; }

    // ICP<Dim>::begin  icp.h:120-133
    void begin(const double* src, int64_t n_, const double* dst, int64_t m_,
               const double* guess) override {
        for (int i = 0; i < (D + 1) * (D + 1); i++) transform.m[i] = guess[i];
        n = n_;
        m = m_;
        a.assign(src, src + n * D);
        b.assign(dst, dst + m * D);
        matches.assign(static_cast<size_t>(n), Match{0, 0, 0.0});
        nb_idx.assign(static_cast<size_t>(n), 0);
        nb_dist.assign(static_cast<size_t>(n), 0.0);
        setup();
    }

    void apply_transform_to(const std::vector<double>& in, std::vector<double>& out,
                            const Xf<D>& x) {
        out.resize(in.size());
        for (int64_t i = 0; i < n; i++) xf_apply<D>(x, &in[i * D], &out[i * D]);
    }

    // nearest neighbours of `q` (n points) in b for the kd-tree methods; dist is Euclidean
    void nn_tree_or_brute(const std::vector<double>& q, bool rebuild, std::vector<int64_t>& idx,
                          std::vector<double>& dist) {
        idx.resize(static_cast<size_t>(n));
        dist.resize(static_cast<size_t>(n));
        if (prm.nn_mode == ORC_NN_BRUTE) {
            std::vector<double> d2(static_cast<size_t>(n));
            nn_brute<D>(q.data(), n, b.data(), m, idx.data(), d2.data());
            for (int64_t i = 0; i < n; i++) dist[i] = std::sqrt(d2[i]);
        } else if (prm.nn_mode == ORC_NN_KDCANON) {
            if (rebuild) build_canon();
            bool by_key = canon_by_key;
            for (size_t e = 0; e < q.size() && !by_key; e++) by_key = !std::isfinite(q[e]);
            for (int64_t i = 0; i < n; i++) {
                double md = 0.0;
                idx[i] = by_key ? kdcanon.search_by_key(&q[i * D], &md) : kdcanon.search(&q[i * D], &md);
                dist[i] = md;
            }
        } else {
            if (rebuild) kdtree.build(b.data(), m);
            for (int64_t i = 0; i < n; i++) {
                double md = 0.0;
                idx[i] = kdtree.search(&q[i * D], &md);
                dist[i] = md;
            }
        }
    }
    void build_canon() {
        kdcanon.build(b.data(), m);
        canon_by_key = false;
        for (double v : b) canon_by_key = canon_by_key || !std::isfinite(v);
    }

    // ---- FeatureAware (lib/icp/impl/feature_aware.cpp), ICP2 only ----
    // feature_aware.cpp:151-177.  Feature r of point i: |q_j| - |q_i| for the up to `sn` points before
    // (packed from row 0) and after (from row sn) it in scan order, q = p - centroid; 0 elsewhere.
    // The centroid is get_centroid (geo.h:33-37): summation order per prm.sum_mode.
    std::vector<double> compute_features(const std::vector<double>& pts, int64_t cnt) const {
        const int64_t sn = prm.symmetric_neighbors;
        std::vector<double> f(static_cast<size_t>(2 * sn * cnt), 0.0);
        if (cnt == 0) return f;
        double c[2];
        std::vector<double> tmp(static_cast<size_t>(cnt));
        for (int d = 0; d < 2; d++) {
            for (int64_t i = 0; i < cnt; i++) tmp[i] = pts[i * 2 + d];
            c[d] = sum_mode(tmp, prm.sum_mode) / static_cast<double>(cnt);
        }
        std::vector<double> rad(static_cast<size_t>(cnt));
        for (int64_t i = 0; i < cnt; i++) {
            double dx = pts[i * 2] - c[0], dy = pts[i * 2 + 1] - c[1];
            rad[i] = std::sqrt(dx * dx + dy * dy);  // (point - centroid).norm()
        }
        for (int64_t i = 0; i < cnt; i++) {
            int64_t lower = std::max<int64_t>(0, i - sn);
            for (int64_t j = lower; j < i; j++) f[(j - lower) + 2 * sn * i] = rad[j] - rad[i];
            int64_t upper = std::min<int64_t>(cnt - 1, i + sn);
            for (int64_t j = i + 1; j <= upper; j++) f[(j - i - 1 + sn) + 2 * sn * i] = rad[j] - rad[i];
        }
        return f;
    }

    // feature_aware.cpp:45-63
    void setup_feature_aware() {
        apply_transform_to(a, cur, transform);
        const int64_t w = 2 * static_cast<int64_t>(prm.symmetric_neighbors);
        std::vector<double> fa = compute_features(cur, n), fb = compute_features(b, m);
        nfd.assign(static_cast<size_t>(n * m), 0.0);
        double mx = -std::numeric_limits<double>::infinity();
        for (int64_t i = 0; i < n; i++)
            for (int64_t j = 0; j < m; j++) {
                // (first.col(i) - second.col(j)).norm(), feature_aware.h:39-41; terms added in row
                // order (Eigen's packet order for a 2*sn vector is build dependent: parity unpinned)
                double acc = 0.0;
                for (int64_t r = 0; r < w; r++) {
                    double df = fa[r + w * i] - fb[r + w * j];
                    acc = acc + df * df;
                }
                double v = std::sqrt(acc);
                nfd[i * m + j] = v;
                if (v > mx) mx = v;
            }
        if (mx > 1e-6)  // MIN_NORM_THRESHOLD, feature_aware.cpp:25,56-58
            for (double& v : nfd) v = v / mx;
        compute_matches_feature_aware();
    }

    // feature_aware.cpp:129-149
    void compute_matches_feature_aware() {
        if (n == 0) return;
        const double fw = prm.feature_weight, nw = 1.0 - prm.feature_weight;  // feature_aware.cpp:31
        std::vector<double> nd(static_cast<size_t>(n * m));
        double mx = -std::numeric_limits<double>::infinity();
        for (int64_t i = 0; i < n; i++)
            for (int64_t j = 0; j < m; j++) {
                double dx = cur[i * 2] - b[j * 2], dy = cur[i * 2 + 1] - b[j * 2 + 1];
                double v = std::sqrt(dx * dx + dy * dy);
                nd[i * m + j] = v;
                if (v > mx) mx = v;
            }
        const bool norm = mx > 1e-6;
        for (int64_t i = 0; i < n; i++) {
            matches[i].point = i;
            matches[i].cost = std::numeric_limits<double>::infinity();
            // (a match that never finds a cost < inf keeps the pair stored at this position; only
            // reachable with non-finite input)
            for (int64_t j = 0; j < m; j++) {
                double dist = norm ? nd[i * m + j] / mx : nd[i * m + j];
                double cost = nw * dist + fw * nfd[i * m + j];
                if (cost < matches[i].cost) {
                    matches[i].cost = cost;
                    matches[i].pair = j;
                }
            }
        }
    }
    void setup() {
        if (D == 2 && method == ORC_FEATURE_AWARE) {
            setup_feature_aware();
        } else if (D == 2) {
            // vanilla.cpp:21-26, trimmed.cpp:31-36
            apply_transform_to(a, cur, transform);
            compute_matches_2d();
        } else {
            // vanilla_3d.cpp:71-81, trimmed_3d.cpp:77-87: c = a (guess NOT applied), build tree once
            cur = a;
            if (prm.flags & ORC_FLAG_APPLY_GUESS) apply_transform_to(a, cur, transform);  // opt-in, fixes Q5
            if (prm.nn_mode == ORC_NN_KDTREE) kdtree.build(b.data(), m);
            if (prm.nn_mode == ORC_NN_KDCANON) build_canon();
            if ((prm.flags & ORC_FLAG_FILL_MATCHES) && n > 0 && m > 0) {  // opt-in, fixes Q4
                nn_tree_or_brute(cur, false, nb_idx, nb_dist);
                for (int64_t i = 0; i < n; i++) {
                    nb_dist[i] = std::sqrt(nb_dist[i]);
                    matches[i] = Match{i, nb_idx[i], nb_dist[i]};
                }
            }
        }
    }

    void compute_matches_2d() {
        if (method == ORC_VANILLA) {
            // vanilla.cpp:68-88: tree rebuilt per call, query is the UNTRANSFORMED a.col(i)
            if (n == 0 || m == 0) return;
            std::vector<int64_t> idx;
            std::vector<double> dist;
            // Q1: the reference queries the UNTRANSFORMED cloud; the opt-in correction queries a_current
            nn_tree_or_brute((prm.flags & ORC_FLAG_MATCH_TRANSFORMED) ? cur : a, true, idx, dist);
            for (int64_t i = 0; i < n; i++) {
                matches[i].cost = std::sqrt(dist[i]);  // vanilla.cpp:84 sqrt of an Euclidean distance
                matches[i].point = i;
                matches[i].pair = idx[i];
            }
        } else {
            // trimmed.cpp:95-109
            std::vector<int64_t> idx(static_cast<size_t>(n));
            std::vector<double> d2(static_cast<size_t>(n));
            nn_brute<D>(cur.data(), n, b.data(), m, idx.data(), d2.data());
            for (int64_t i = 0; i < n; i++) {
                matches[i].point = i;
                matches[i].cost = d2[i];
                if (m > 0) matches[i].pair = idx[i];
            }
        }
    }

    // centroid + cross-covariance + Kabsch over the pairs (cur[src_i], b[dst_i]) with weight
    // mask[i] in {0,1}; k = number of selected pairs.
    Xf<D> fit(const std::vector<int64_t>& src_i, const std::vector<int64_t>& dst_i) {
        const int64_t k = static_cast<int64_t>(src_i.size());
        double ca[D], cb[D];
        std::vector<double> tmp(static_cast<size_t>(k));
        for (int d = 0; d < D; d++) {
            for (int64_t i = 0; i < k; i++) tmp[i] = cur[src_i[i] * D + d];
            ca[d] = sum_mode(tmp, prm.sum_mode) / static_cast<double>(k);
            for (int64_t i = 0; i < k; i++) tmp[i] = b[dst_i[i] * D + d];
            cb[d] = sum_mode(tmp, prm.sum_mode) / static_cast<double>(k);
        }
        Mat<D> N;
        for (int r = 0; r < D; r++)
            for (int c = 0; c < D; c++) {
                for (int64_t i = 0; i < k; i++)
                    tmp[i] = (cur[src_i[i] * D + r] - ca[r]) * (b[dst_i[i] * D + c] - cb[c]);
                N(r, c) = sum_mode(tmp, prm.sum_mode);
            }
        return kabsch<D>(N, ca, cb);
    }

    // same, but over ALL n source points with a 0/1 mask (the TREE64 device order keeps every
    // point at its own tree position and adds +0.0 for unselected ones)
    Xf<D> fit_masked(const std::vector<uint8_t>& sel, const std::vector<int64_t>& pair,
                     int64_t k) {
        double ca[D], cb[D];
        std::vector<double> tmp(static_cast<size_t>(n));
        for (int d = 0; d < D; d++) {
            for (int64_t i = 0; i < n; i++) tmp[i] = sel[i] ? cur[i * D + d] : 0.0;
            ca[d] = sum_mode(tmp, prm.sum_mode) / static_cast<double>(k);
            for (int64_t i = 0; i < n; i++) tmp[i] = sel[i] ? b[pair[i] * D + d] : 0.0;
            cb[d] = sum_mode(tmp, prm.sum_mode) / static_cast<double>(k);
        }
        Mat<D> N;
        for (int r = 0; r < D; r++)
            for (int c = 0; c < D; c++) {
                for (int64_t i = 0; i < n; i++)
                    tmp[i] = sel[i] ? (cur[i * D + r] - ca[r]) * (b[pair[i] * D + c] - cb[c]) : 0.0;
                N(r, c) = sum_mode(tmp, prm.sum_mode);
            }
        return kabsch<D>(N, ca, cb);
    }

    void iterate() override {
        if (D == 2 && method == ORC_VANILLA) iterate_vanilla2();
        else if (D == 2) iterate_trimmed2();
        else iterate_3d();
    }

    // vanilla.cpp:28-66
    void iterate_vanilla2() {
        if (n == 0 || m == 0) return;
        apply_transform_to(a, cur, transform);
        compute_matches_2d();
        std::vector<int64_t> si(static_cast<size_t>(n)), di(static_cast<size_t>(n));
        for (int64_t i = 0; i < n; i++) {
            si[i] = i;
            di[i] = matches[i].pair;
        }
        Xf<D> step = fit(si, di);
        transform = xf_mul<D>(step, transform);
    }

    // trimmed.cpp:38-93; feature_aware.cpp:65-127 is the same step around its own compute_matches
    void iterate_trimmed2() {
        if (n == 0 || m == 0) return;  // Q10: the reference is UB here; defined as a no-op
        apply_transform_to(a, cur, transform);
        if (method == ORC_FEATURE_AWARE) compute_matches_feature_aware();  // feature_aware.cpp:66-86
        else compute_matches_2d();
        if (prm.sum_mode == ORC_SUM_SEQUENTIAL) {
            // trimmed.cpp:55-56: unstable std::sort on cost only — same libstdc++ here
            std::sort(matches.begin(), matches.end(),
                      [](const Match& x, const Match& y) { return x.cost < y.cost; });
        } else {
            // device rule: total order (cost, point)
            std::sort(matches.begin(), matches.end(), [](const Match& x, const Match& y) {
                return x.cost < y.cost || (x.cost == y.cost && x.point < y.point);
            });
        }
        int64_t new_n = static_cast<int64_t>(prm.overlap_rate * static_cast<double>(n));
        new_n = std::max<int64_t>(new_n, 1);
        new_n = std::min<int64_t>(new_n, n);  // overlap_rate > 1 would index out of range (UB) in the reference
        Xf<D> step;
        if (prm.sum_mode == ORC_SUM_SEQUENTIAL) {
            std::vector<int64_t> si(static_cast<size_t>(new_n)), di(static_cast<size_t>(new_n));
            for (int64_t i = 0; i < new_n; i++) {
                si[i] = matches[i].point;
                di[i] = matches[i].pair;
            }
            step = fit(si, di);
        } else {
            std::vector<uint8_t> sel(static_cast<size_t>(n), 0);
            std::vector<int64_t> pair(static_cast<size_t>(n), 0);
            for (int64_t i = 0; i < n; i++) pair[matches[i].point] = matches[i].pair;
            for (int64_t i = 0; i < new_n; i++) sel[matches[i].point] = 1;
            step = fit_masked(sel, pair, new_n);
        }
        transform = xf_mul<D>(step, transform);
    }

    // vanilla_3d.cpp:83-96, trimmed_3d.cpp:89-116
    void iterate_3d() {
        nn_tree_or_brute(cur, false, nb_idx, nb_dist);
        for (int64_t i = 0; i < n; i++) nb_dist[i] = std::sqrt(nb_dist[i]);  // vanilla_3d.cpp:41
        if (prm.flags & ORC_FLAG_FILL_MATCHES)
            for (int64_t i = 0; i < n; i++) matches[i] = Match{i, nb_idx[i], nb_dist[i]};
        Xf<D> T;
        if (prm.sum_mode == ORC_SUM_SEQUENTIAL) {
            std::vector<int64_t> si, di;
            for (int64_t i = 0; i < n; i++) {
                if (method == ORC_TRIMMED && !(nb_dist[i] <= prm.max_distance)) continue;  // trimmed_3d.cpp:98
                si.push_back(i);
                di.push_back(nb_idx[i]);
            }
            T = fit(si, di);
        } else {
            std::vector<uint8_t> sel(static_cast<size_t>(n), 1);
            int64_t k = n;
            if (method == ORC_TRIMMED) {
                k = 0;
                for (int64_t i = 0; i < n; i++) {
                    sel[i] = nb_dist[i] <= prm.max_distance ? 1 : 0;
                    k += sel[i];
                }
            }
            T = fit_masked(sel, nb_idx, k);
        }
        std::vector<double> next;
        apply_transform_to(cur, next, T);  // c = T * c  (vanilla_3d.cpp:91)
        cur.swap(next);
        transform = xf_mul<D>(T, transform);
    }

    // icp.h:149-155
    double cost() const override {
        std::vector<double> c(static_cast<size_t>(n));
        if (prm.sum_mode == ORC_SUM_SEQUENTIAL) {
            // the reference walks `matches` in its current order (sorted by cost after
            // Trimmed::iterate, trimmed.cpp:55-56)
            for (int64_t i = 0; i < n; i++) c[i] = matches[i].cost;
        } else {
            // device order: every point keeps its own position in the tree
            for (int64_t i = 0; i < n; i++) c[matches[i].point] = matches[i].cost;
        }
        double sum = sum_mode(c, prm.sum_mode);
        return std::sqrt(sum / static_cast<double>(n));
    }

    void get_transform(double* out) const override {
        std::memcpy(out, transform.m, sizeof(transform.m));
    }
    void get_matches(int64_t* point, int64_t* pair, double* cost) const override {
        for (int64_t i = 0; i < n; i++) {
            point[i] = matches[i].point;
            pair[i] = matches[i].pair;
            cost[i] = matches[i].cost;
        }
    }
    void get_neighbors(int64_t* idx, double* dist) const override {
        for (int64_t i = 0; i < n; i++) {
            idx[i] = nb_idx[i];
            dist[i] = nb_dist[i];
        }
    }
    void get_current(double* out) const override {
        std::memcpy(out, cur.data(), cur.size() * sizeof(double));
    }
};

// ---------------------------------------------------------------------------------------------
// driver.h:154-222
// ---------------------------------------------------------------------------------------------

template <int D>
double rotation_angle(const double* cur, const double* last);

// Rotation2Dd(rotation_step).smallestAngle()  (driver.h:202-203)
template <>
double rotation_angle<2>(const double* cur, const double* last) {
    // rotation_step = R_cur * R_last^T ; column-major 3x3 inputs
    auto C = [&](int r, int c) { return cur[c * 3 + r]; };
    auto L = [&](int r, int c) { return last[c * 3 + r]; };
    double m00 = C(0, 0) * L(0, 0) + C(0, 1) * L(0, 1);
    double m10 = C(1, 0) * L(0, 0) + C(1, 1) * L(0, 1);
    double angle = std::atan2(m10, m00);  // Rotation2D::fromRotationMatrix
    double tmp = std::fmod(angle, 2.0 * M_PI);  // smallestAngle
    if (tmp > M_PI) tmp -= 2.0 * M_PI;
    else if (tmp < -M_PI) tmp += 2.0 * M_PI;
    return tmp;
}

// AngleAxisd(rotation_step).angle()  (driver.h:208-209): matrix -> quaternion -> angle-axis
template <>
double rotation_angle<3>(const double* cur, const double* last) {
    auto C = [&](int r, int c) { return cur[c * 4 + r]; };
    auto L = [&](int r, int c) { return last[c * 4 + r]; };
    double mat[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = C(i, 0) * L(j, 0);
            for (int k = 1; k < 3; k++) s += C(i, k) * L(j, k);
            mat[i][j] = s;
        }
    double q[4];  // x y z w
    double t = mat[0][0] + mat[1][1] + mat[2][2];
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (mat[2][1] - mat[1][2]) * t;
        q[1] = (mat[0][2] - mat[2][0]) * t;
        q[2] = (mat[1][0] - mat[0][1]) * t;
    } else {
        int i = 0;
        if (mat[1][1] > mat[0][0]) i = 1;
        if (mat[2][2] > mat[i][i]) i = 2;
        int j = (i + 1) % 3;
        int k = (j + 1) % 3;
        t = std::sqrt(mat[i][i] - mat[j][j] - mat[k][k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (mat[k][j] - mat[j][k]) * t;
        q[j] = (mat[j][i] + mat[i][j]) * t;
        q[k] = (mat[k][i] + mat[i][k]) * t;
    }
    double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (nrm != 0.0) return 2.0 * std::atan2(nrm, std::fabs(q[3]));
    return 0.0;
}

template <int D>
bool should_terminate(const orc_criteria& c, const orc_state& cur, const orc_state* last,
                      std::chrono::steady_clock::time_point start) {
    if ((c.has & ORC_HAS_MIN_ITER) && cur.iteration_count < c.min_iterations) return false;
    if ((c.has & ORC_HAS_MAX_ITER) && cur.iteration_count >= c.max_iterations) return true;
    if ((c.has & ORC_HAS_STOP_COST) && cur.cost < c.stop_cost) return true;
    if (c.has & ORC_HAS_TIME_LIMIT) {
        std::chrono::duration<double> el = std::chrono::steady_clock::now() - start;
        if (el.count() > c.time_limit_s) return true;
    }
    if (last == nullptr) return false;
    double delta_cost = cur.cost - last->cost;
    if ((c.has & ORC_HAS_ABS_TOL) && std::fabs(delta_cost) < c.absolute_cost_tolerance) return true;
    double rel = std::fabs(delta_cost) / cur.cost;
    if ((c.has & ORC_HAS_REL_TOL) && rel < c.relative_cost_tolerance) return true;
    if (c.has & ORC_HAS_TRANSFORM_TOL) {
        double angle = rotation_angle<D>(cur.transform, last->transform);
        double tn = 0.0;
        for (int i = 0; i < D; i++) {
            double d = cur.transform[D * (D + 1) + i] - last->transform[D * (D + 1) + i];
            tn += d * d;
        }
        tn = std::sqrt(tn);
        if (std::fabs(angle) < c.angle_tolerance_rad && tn < c.translation_tolerance) return true;
    }
    return false;
}

template <int D>
void converge(IcpBase* icp, const double* src, int64_t n, const double* dst, int64_t m,
              const double* guess, const orc_criteria* crit, orc_state* out, int64_t trace_cap,
              double* trace_cost, double* trace_transform) {
    // driver.h:51-72
    auto start = std::chrono::steady_clock::now();
    icp->begin(src, n, dst, m, guess);
    orc_state state;
    std::memset(&state, 0, sizeof(state));
    state.iteration_count = 0;
    state.cost = icp->cost();
    icp->get_transform(state.transform);
    orc_state last;
    bool have_last = false;
    while (!should_terminate<D>(*crit, state, have_last ? &last : nullptr, start)) {
        last = state;
        have_last = true;
        icp->iterate();
        state.iteration_count++;
        state.cost = icp->cost();
        icp->get_transform(state.transform);
        int64_t k = static_cast<int64_t>(state.iteration_count) - 1;
        if (k < trace_cap) {
            if (trace_cost) trace_cost[k] = state.cost;
            if (trace_transform)
                std::memcpy(trace_transform + k * (D + 1) * (D + 1), state.transform,
                            sizeof(double) * (D + 1) * (D + 1));
        }
    }
    *out = state;
}

template <int D>
void nn_kdcanon(const double* q, int64_t n, const double* t, int64_t m, int64_t* idx, double* dist, int by_key, int32_t* nodes) {
    KdCanon<D> kt;
    kt.build(t, m);
    for (int64_t i = 0; i < n; i++) idx[i] = by_key ? kt.search_by_key(q + i * D, dist + i) : kt.search(q + i * D, dist + i);
    if (nodes) std::memcpy(nodes, kt.nodes().data(), sizeof(int32_t) * static_cast<size_t>(m));
}
}  // namespace

struct orc_icp {
    std::unique_ptr<IcpBase> impl;
};

extern "C" {

void orc_nn_brute(int dim, const double* q, int64_t n, const double* t, int64_t m, int64_t* idx,
                  double* d2) {
    if (dim == 2) nn_brute<2>(q, n, t, m, idx, d2);
    else nn_brute<3>(q, n, t, m, idx, d2);
}

void orc_nn_kdtree(int dim, const double* q, int64_t n, const double* t, int64_t m, int64_t* idx,
                   double* dist) {
    if (dim == 2) {
        KdTree<2> kt;
        kt.build(t, m);
        for (int64_t i = 0; i < n; i++) idx[i] = kt.search(q + i * 2, dist + i);
    } else {
        KdTree<3> kt;
        kt.build(t, m);
        for (int64_t i = 0; i < n; i++) idx[i] = kt.search(q + i * 3, dist + i);
    }
}

void orc_nn_kdcanon(int dim, const double* q, int64_t n, const double* t, int64_t m, int64_t* idx,
                    double* dist, int by_key, int32_t* nodes) {
    if (m <= 0) {
        for (int64_t i = 0; i < n; i++) {
            idx[i] = 0;
            dist[i] = std::numeric_limits<double>::infinity();
        }
        return;
    }
    if (dim == 2) nn_kdcanon<2>(q, n, t, m, idx, dist, by_key, nodes);
    else nn_kdcanon<3>(q, n, t, m, idx, dist, by_key, nodes);
}

void orc_jacobi_svd(int dim, const double* a, double* u, double* s, double* v) {
    if (dim == 2) {
        Mat<2> A, U, V;
        std::memcpy(A.a, a, sizeof(A.a));
        jacobi_svd<2>(A, U, s, V);
        std::memcpy(u, U.a, sizeof(U.a));
        std::memcpy(v, V.a, sizeof(V.a));
    } else {
        Mat<3> A, U, V;
        std::memcpy(A.a, a, sizeof(A.a));
        jacobi_svd<3>(A, U, s, V);
        std::memcpy(u, U.a, sizeof(U.a));
        std::memcpy(v, V.a, sizeof(V.a));
    }
}

void orc_kabsch(int dim, const double* N, const double* ca, const double* cb, double* step) {
    if (dim == 2) {
        Mat<2> n;
        std::memcpy(n.a, N, sizeof(n.a));
        Xf<2> s = kabsch<2>(n, ca, cb);
        std::memcpy(step, s.m, sizeof(s.m));
    } else {
        Mat<3> n;
        std::memcpy(n.a, N, sizeof(n.a));
        Xf<3> s = kabsch<3>(n, ca, cb);
        std::memcpy(step, s.m, sizeof(s.m));
    }
}

double orc_sum(const double* v, int64_t n, int mode) {
    return mode == ORC_SUM_TREE64 ? sum_tree64(v, n) : sum_seq(v, n);
}

orc_icp* orc_create(int dim, int method, const orc_params* p) {
    if ((dim != 2 && dim != 3) || method < ORC_VANILLA || method > ORC_FEATURE_AWARE) return nullptr;
    if (method == ORC_FEATURE_AWARE && dim != 2) return nullptr;  // registered for ICP2 only
    orc_params prm;
    if (p) prm = *p;
    else prm = orc_params{0.9, 1.0, ORC_NN_KDTREE, ORC_SUM_SEQUENTIAL, 0.7, 10, 0};
    if (prm.symmetric_neighbors < 0) return nullptr;
    auto* h = new orc_icp;
    if (dim == 2) h->impl = std::make_unique<Icp<2>>(method, prm);
    else h->impl = std::make_unique<Icp<3>>(method, prm);
    return h;
}

void orc_destroy(orc_icp* h) { delete h; }

void orc_begin(orc_icp* h, const double* src, int64_t n, const double* dst, int64_t m,
               const double* guess) {
    h->impl->begin(src, n, dst, m, guess);
}
void orc_iterate(orc_icp* h) { h->impl->iterate(); }
double orc_cost(const orc_icp* h) { return h->impl->cost(); }
void orc_transform(const orc_icp* h, double* out) { h->impl->get_transform(out); }
void orc_matches(const orc_icp* h, int64_t* point, int64_t* pair, double* cost) {
    h->impl->get_matches(point, pair, cost);
}
void orc_neighbors(const orc_icp* h, int64_t* idx, double* dist) {
    h->impl->get_neighbors(idx, dist);
}
void orc_current_cloud(const orc_icp* h, double* out) { h->impl->get_current(out); }

void orc_converge(orc_icp* h, const double* src, int64_t n, const double* dst, int64_t m,
                  const double* guess, const orc_criteria* crit, orc_state* out, int64_t trace_cap,
                  double* trace_cost, double* trace_transform) {
    if (h->impl->dim() == 2)
        converge<2>(h->impl.get(), src, n, dst, m, guess, crit, out, trace_cap, trace_cost,
                    trace_transform);
    else
        converge<3>(h->impl.get(), src, n, dst, m, guess, crit, out, trace_cap, trace_cost,
                    trace_transform);
}

}  // extern "C"
