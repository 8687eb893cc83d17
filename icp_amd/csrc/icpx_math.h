// icpx_math.h — small f64 dense algebra used by every ICP kernel (device side).
//
// All of it runs on ONE lane: the 2x2 / 3x3 two-sided Jacobi SVD, the Kabsch step, the transform
// composition and the driver's termination test.  Arithmetic is plain IEEE f64 with contraction
// off (the translation unit is built with -ffp-contract=off), in the operand order documented in
// DESIGN.md, so that results are reproducible bit for bit.
//
// Reference call sites this stands in for:
//   Eigen::JacobiSVD(N, FullU|FullV), R = V U^T, reflection fix, t = cb - R ca
//     lib/icp/impl/vanilla.cpp:50-63, trimmed.cpp:75-90, vanilla_3d.cpp:53-66, trimmed_3d.cpp:59-72
//   transform = step * transform        vanilla.cpp:65, trimmed.cpp:92, vanilla_3d.cpp:93
//   ICPDriver::should_terminate         include/icp/driver.h:154-222
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/icpx/icpx.h"

#define ICPX_DEV __device__ __forceinline__

namespace icpx {

constexpr double kDblMin = 2.2250738585072014e-308;
constexpr double kDblEps = 2.220446049250313e-16;
constexpr double kPi = 3.14159265358979323846;

ICPX_DEV double dabs(double x) { return __builtin_fabs(x); }
ICPX_DEV double dmax(double a, double b) { return a < b ? b : a; }

struct Rot {
    double c, s;
};

// Jacobi rotation that diagonalises the symmetric 2x2 [[x, y], [y, z]]
ICPX_DEV Rot make_jacobi(double x, double y, double z) {
    Rot r;
    double deno = 2.0 * dabs(y);
    if (deno < kDblMin) {
        r.c = 1.0;
        r.s = 0.0;
    } else {
        double tau = (x - z) / deno;
        double w = __builtin_sqrt(tau * tau + 1.0);
        double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
        double sign_t = t > 0.0 ? 1.0 : -1.0;
        double n = 1.0 / __builtin_sqrt(t * t + 1.0);
        r.s = -sign_t * (y / dabs(y)) * dabs(t) * n;
        r.c = n;
    }
    return r;
}

ICPX_DEV Rot rot_mul(Rot a, Rot b) { return Rot{a.c * b.c - a.s * b.s, a.c * b.s + a.s * b.c}; }
ICPX_DEV Rot rot_t(Rot a) { return Rot{a.c, -a.s}; }
ICPX_DEV void rot_apply(double& x, double& y, Rot j) {
    double xi = x, yi = y;
    x = j.c * xi + j.s * yi;
    y = -j.s * xi + j.c * yi;
}

// column-major D x D
template <int D>
struct Mat {
    double a[D * D];
    ICPX_DEV double& operator()(int r, int c) { return a[c * D + r]; }
    ICPX_DEV double operator()(int r, int c) const { return a[c * D + r]; }
    ICPX_DEV void set_identity() {
#pragma unroll
        for (int i = 0; i < D * D; i++) a[i] = 0.0;
#pragma unroll
        for (int i = 0; i < D; i++) a[i * D + i] = 1.0;
    }
};

template <int D>
ICPX_DEV void rows_rotate(Mat<D>& m, int p, int q, Rot j) {
    if (j.c == 1.0 && j.s == 0.0) return;
#pragma unroll
    for (int c = 0; c < D; c++) rot_apply(m.a[c * D + p], m.a[c * D + q], j);
}
// columns p,q rotated by j^T (Eigen's applyOnTheRight(p, q, j))
template <int D>
ICPX_DEV void cols_rotate(Mat<D>& m, int p, int q, Rot j) {
    Rot jt = rot_t(j);
    if (jt.c == 1.0 && jt.s == 0.0) return;
#pragma unroll
    for (int r = 0; r < D; r++) rot_apply(m.a[p * D + r], m.a[q * D + r], jt);
}

// two-sided Jacobi SVD, full U and V, singular values sorted descending
template <int D>
ICPX_DEV void jacobi_svd(const Mat<D>& in, Mat<D>& U, double* S, Mat<D>& V) {
    const double precision = 2.0 * kDblEps;
    double scale = 0.0;
    bool finite = true;
#pragma unroll
    for (int i = 0; i < D * D; i++) {
        double v = dabs(in.a[i]);
        if (!(v <= 1.7976931348623157e308)) finite = false;  // inf or NaN
        if (v > scale) scale = v;
    }
    U.set_identity();
    V.set_identity();
    if (!finite) {
        double nan = __builtin_nan("");
#pragma unroll
        for (int i = 0; i < D; i++) S[i] = nan;
#pragma unroll
        for (int i = 0; i < D * D; i++) U.a[i] = V.a[i] = nan;
        return;
    }
    if (scale == 0.0) scale = 1.0;
    Mat<D> w;
#pragma unroll
    for (int i = 0; i < D * D; i++) w.a[i] = in.a[i] / scale;
    double max_diag = 0.0;
#pragma unroll
    for (int i = 0; i < D; i++) max_diag = dmax(max_diag, dabs(w(i, i)));

    bool finished = false;
    while (!finished) {
        finished = true;
#pragma unroll
        for (int p = 1; p < D; ++p) {
#pragma unroll
            for (int q = 0; q < p; ++q) {
                double threshold = dmax(kDblMin, precision * max_diag);
                if (dabs(w(p, q)) > threshold || dabs(w(q, p)) > threshold) {
                    finished = false;
                    // 2x2 kernel: first a rotation that makes the block symmetric, then Jacobi
                    double m00 = w(p, p), m01 = w(p, q), m10 = w(q, p), m11 = w(q, q);
                    Rot rot1;
                    double t = m00 + m11;
                    double d = m10 - m01;
                    if (dabs(d) < kDblMin) {
                        rot1.s = 0.0;
                        rot1.c = 1.0;
                    } else {
                        double u = t / d;
                        double tmp = __builtin_sqrt(1.0 + u * u);
                        rot1.s = 1.0 / tmp;
                        rot1.c = u / tmp;
                    }
                    if (!(rot1.c == 1.0 && rot1.s == 0.0)) {
                        rot_apply(m00, m10, rot1);
                        rot_apply(m01, m11, rot1);
                    }
                    Rot j_right = make_jacobi(m00, m01, m11);
                    Rot j_left = rot_mul(rot1, rot_t(j_right));
                    rows_rotate<D>(w, p, q, j_left);
                    cols_rotate<D>(U, p, q, rot_t(j_left));
                    cols_rotate<D>(w, p, q, j_right);
                    cols_rotate<D>(V, p, q, j_right);
                    max_diag = dmax(max_diag, dmax(dabs(w(p, p)), dabs(w(q, q))));
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; i++) {
        double a = w(i, i);
        S[i] = dabs(a);
        if (a < 0.0) {
#pragma unroll
            for (int r = 0; r < D; r++) U(r, i) = -U(r, i);
        }
    }
#pragma unroll
    for (int i = 0; i < D; i++) S[i] *= scale;
#pragma unroll
    for (int i = 0; i < D; i++) {
        // largest remaining singular value moves to position i (first one wins ties)
        int pos = i;
        double best = S[i];
#pragma unroll
        for (int k = i + 1; k < D; k++)
            if (S[k] > best) {
                best = S[k];
                pos = k;
            }
        if (best == 0.0) break;
#pragma unroll
        for (int k = i + 1; k < D; k++) {
            if (pos == k) {
                double ts = S[i];
                S[i] = S[k];
                S[k] = ts;
#pragma unroll
                for (int r = 0; r < D; r++) {
                    double tu = U(r, i);
                    U(r, i) = U(r, k);
                    U(r, k) = tu;
                    double tv = V(r, i);
                    V(r, i) = V(r, k);
                    V(r, k) = tv;
                }
            }
        }
    }
}

template <int D>
ICPX_DEV double det(const Mat<D>& m) {
    if constexpr (D == 2) {
        return m(0, 0) * m(1, 1) - m(1, 0) * m(0, 1);
    } else {
        double h0 = m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1));
        double h1 = m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0));
        double h2 = m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
        return h0 - h1 + h2;
    }
}

template <int D>
ICPX_DEV Mat<D> v_ut(const Mat<D>& V, const Mat<D>& U) {
    Mat<D> R;
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int j = 0; j < D; j++) {
            double s = V(i, 0) * U(j, 0);
#pragma unroll
            for (int k = 1; k < D; k++) s += V(i, k) * U(j, k);
            R(i, j) = s;
        }
    return R;
}

// (D+1)x(D+1) column-major rigid transform
template <int D>
struct Xf {
    double m[(D + 1) * (D + 1)];
    ICPX_DEV double& operator()(int r, int c) { return m[c * (D + 1) + r]; }
    ICPX_DEV double operator()(int r, int c) const { return m[c * (D + 1) + r]; }
    ICPX_DEV void set_identity() {
#pragma unroll
        for (int i = 0; i < (D + 1) * (D + 1); i++) m[i] = 0.0;
#pragma unroll
        for (int i = 0; i <= D; i++) m[i * (D + 1) + i] = 1.0;
    }
};

// a * b  (linear = La Lb, translation = La tb + ta; sums left to right)
template <int D>
ICPX_DEV Xf<D> xf_mul(const Xf<D>& a, const Xf<D>& b) {
    Xf<D> r;
    r.set_identity();
#pragma unroll
    for (int i = 0; i < D; i++) {
#pragma unroll
        for (int j = 0; j < D; j++) {
            double s = a(i, 0) * b(0, j);
#pragma unroll
            for (int k = 1; k < D; k++) s += a(i, k) * b(k, j);
            r(i, j) = s;
        }
        double s = a(i, 0) * b(0, D);
#pragma unroll
        for (int k = 1; k < D; k++) s += a(i, k) * b(k, D);
        r(i, D) = s + a(i, D);
    }
    return r;
}

// x * p = (L p) + t, sums left to right.  `x` is given as its first D rows flattened
// row-major into registers by the caller (see XfRows) to keep the hot loops free of indexing.
template <int D>
struct XfRows {
    double l[D][D];
    double t[D];
};
template <int D>
ICPX_DEV XfRows<D> xf_rows(const Xf<D>& x) {
    XfRows<D> r;
#pragma unroll
    for (int i = 0; i < D; i++) {
#pragma unroll
        for (int j = 0; j < D; j++) r.l[i][j] = x(i, j);
        r.t[i] = x(i, D);
    }
    return r;
}
template <int D>
ICPX_DEV void xf_apply(const XfRows<D>& x, const double* p, double* out) {
    double tmp[D];
#pragma unroll
    for (int i = 0; i < D; i++) {
        double s = x.l[i][0] * p[0];
#pragma unroll
        for (int k = 1; k < D; k++) s += x.l[i][k] * p[k];
        tmp[i] = s + x.t[i];
    }
#pragma unroll
    for (int i = 0; i < D; i++) out[i] = tmp[i];
}

// Kabsch step from the cross-covariance N and the two centroids
template <int D>
ICPX_DEV Xf<D> kabsch(const Mat<D>& N, const double* ca, const double* cb) {
    Mat<D> U, V;
    double S[D];
    jacobi_svd<D>(N, U, S, V);
    Mat<D> R = v_ut<D>(V, U);
    if (det<D>(R) < 0) {
#pragma unroll
        for (int r = 0; r < D; r++) V(r, D - 1) = -V(r, D - 1);
        R = v_ut<D>(V, U);
    }
    Xf<D> step;
    step.set_identity();
#pragma unroll
    for (int i = 0; i < D; i++) {
#pragma unroll
        for (int j = 0; j < D; j++) step(i, j) = R(i, j);
        double s = R(i, 0) * ca[0];
#pragma unroll
        for (int k = 1; k < D; k++) s += R(i, k) * ca[k];
        step(i, D) = cb[i] - s;
    }
    return step;
}

// |rotation angle| between two transforms: Rotation2Dd(Rc Rl^T).smallestAngle() in 2D,
// AngleAxisd(Rc Rl^T).angle() in 3D (driver.h:196-210)
template <int D>
ICPX_DEV double rotation_angle(const double* cur, const double* last) {
    constexpr int L = D + 1;
    if constexpr (D == 2) {
        double m00 = cur[0 * L + 0] * last[0 * L + 0] + cur[1 * L + 0] * last[1 * L + 0];
        double m10 = cur[0 * L + 1] * last[0 * L + 0] + cur[1 * L + 1] * last[1 * L + 0];
        double angle = atan2(m10, m00);
        double tmp = fmod(angle, 2.0 * kPi);
        if (tmp > kPi) tmp -= 2.0 * kPi;
        else if (tmp < -kPi) tmp += 2.0 * kPi;
        return tmp;
    } else {
        double mat[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                double s = cur[0 * L + i] * last[0 * L + j];
#pragma unroll
                for (int k = 1; k < 3; k++) s += cur[k * L + i] * last[k * L + j];
                mat[i][j] = s;
            }
        // rotation matrix -> quaternion (x, y, z, w); explicit cases keep everything in registers
        double qx, qy, qz, qw;
        double t = mat[0][0] + mat[1][1] + mat[2][2];
        if (t > 0.0) {
            t = __builtin_sqrt(t + 1.0);
            qw = 0.5 * t;
            t = 0.5 / t;
            qx = (mat[2][1] - mat[1][2]) * t;
            qy = (mat[0][2] - mat[2][0]) * t;
            qz = (mat[1][0] - mat[0][1]) * t;
        } else {
            int i = 0;
            if (mat[1][1] > mat[0][0]) i = 1;
            double mii = i == 0 ? mat[0][0] : mat[1][1];
            if (mat[2][2] > mii) i = 2;
            if (i == 0) {  // (i, j, k) = (0, 1, 2)
                t = __builtin_sqrt(mat[0][0] - mat[1][1] - mat[2][2] + 1.0);
                qx = 0.5 * t;
                t = 0.5 / t;
                qw = (mat[2][1] - mat[1][2]) * t;
                qy = (mat[1][0] + mat[0][1]) * t;
                qz = (mat[2][0] + mat[0][2]) * t;
            } else if (i == 1) {  // (1, 2, 0)
                t = __builtin_sqrt(mat[1][1] - mat[2][2] - mat[0][0] + 1.0);
                qy = 0.5 * t;
                t = 0.5 / t;
                qw = (mat[0][2] - mat[2][0]) * t;
                qz = (mat[2][1] + mat[1][2]) * t;
                qx = (mat[0][1] + mat[1][0]) * t;
            } else {  // (2, 0, 1)
                t = __builtin_sqrt(mat[2][2] - mat[0][0] - mat[1][1] + 1.0);
                qz = 0.5 * t;
                t = 0.5 / t;
                qw = (mat[1][0] - mat[0][1]) * t;
                qx = (mat[0][2] + mat[2][0]) * t;
                qy = (mat[1][2] + mat[2][1]) * t;
            }
        }
        double nrm = __builtin_sqrt(qx * qx + qy * qy + qz * qz);
        if (nrm != 0.0) return 2.0 * atan2(nrm, dabs(qw));
        return 0.0;
    }
}

// Device image of one pair's ICP state; lives in global memory between launches.
struct PairState {
    double transform[16];
    double last_transform[16];
    double cost;
    double last_cost;
    uint64_t iteration_count;
    int32_t have_last;
    int32_t done;
    uint64_t start_clock;  // wall_clock64() when converge started
    uint64_t kd_flags;     // kd-tree tie rule (icpx_kd.h): bit 0 the pair's tree is built, bit 1 stopped at a tie (take-over pending)
};

// ICPDriver::should_terminate (driver.h:154-222).  elapsed_s: seconds since converge() began.
template <int D>
ICPX_DEV bool should_terminate(const icpx_criteria& c, const PairState& s, double elapsed_s) {
    if ((c.has & ICPX_HAS_MIN_ITER) && s.iteration_count < c.min_iterations) return false;
    if ((c.has & ICPX_HAS_MAX_ITER) && s.iteration_count >= c.max_iterations) return true;
    if ((c.has & ICPX_HAS_STOP_COST) && s.cost < c.stop_cost) return true;
    if ((c.has & ICPX_HAS_TIME_LIMIT) && elapsed_s > c.time_limit_s) return true;
    if (!s.have_last) return false;
    double delta_cost = s.cost - s.last_cost;
    if ((c.has & ICPX_HAS_ABS_TOL) && dabs(delta_cost) < c.absolute_cost_tolerance) return true;
    double rel = dabs(delta_cost) / s.cost;
    if ((c.has & ICPX_HAS_REL_TOL) && rel < c.relative_cost_tolerance) return true;
    if (c.has & ICPX_HAS_TRANSFORM_TOL) {
        double angle = rotation_angle<D>(s.transform, s.last_transform);
        double tn = 0.0;
#pragma unroll
        for (int i = 0; i < D; i++) {
            double d = s.transform[D * (D + 1) + i] - s.last_transform[D * (D + 1) + i];
            tn += d * d;
        }
        tn = __builtin_sqrt(tn);
        if (dabs(angle) < c.angle_tolerance_rad && tn < c.translation_tolerance) return true;
    }
    return false;
}

// A value that is the same in every lane, moved to scalar registers (the compiler cannot prove uniformity
// of what comes out of LDS or of a butterfly): frees the vector registers it would occupy.
ICPX_DEV int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
ICPX_DEV double uniform_double(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// ---- wave-level helpers -------------------------------------------------------------------

// 64-lane xor butterfly sum: every lane ends with the same value; the association order is the
// one DESIGN.md calls TREE64 (v[l] + v[l^32], then ^16, ... ^1).
// value of lane (l ^ DIST) for DIST in {8, 4, 2, 1}: DPP moves inside a row of 16 lanes (VALU, no LDS
// round trip): row_ror:8; row_shl:4 into banks 0, 2 + row_shr:4 into banks 1, 3; quad_perm
template <int DIST>
ICPX_DEV double dpp_xor(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (DIST == 8) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false);
    } else if (DIST == 4) {
        int l2 = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xf, 0x5, false);   // lanes 0-3, 8-11 <- lane + 4
        int h2 = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xf, 0x5, false);
        lo = __builtin_amdgcn_update_dpp(l2, lo, 0x114, 0xf, 0xa, false);      // lanes 4-7, 12-15 <- lane - 4
        hi = __builtin_amdgcn_update_dpp(h2, hi, 0x114, 0xf, 0xa, false);
    } else if (DIST == 2) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false);        // quad_perm [2,3,0,1]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false);
    }
    return __hiloint2double(hi, lo);
}

// value of lane (l ^ 32) / (l ^ 16): gfx950's v_permlane32_swap / v_permlane16_swap of the value with
// itself leave {lower half | even rows} in one result and {upper half | odd rows} in the other
// (tools/micro/permlane_probe.hip); each lane picks the one that holds its partner
template <int DIST>
ICPX_DEV double swap_xor(double v) {
    const unsigned lo = static_cast<unsigned>(__double2loint(v)), hi = static_cast<unsigned>(__double2hiint(v));
    const bool upper = (threadIdx.x & DIST) != 0;
    unsigned plo, phi;
    if (DIST == 32) {
        auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        plo = upper ? a[0] : a[1];
        phi = upper ? b[0] : b[1];
    } else {
        auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        plo = upper ? a[0] : a[1];
        phi = upper ? b[0] : b[1];
    }
    return __hiloint2double(static_cast<int>(phi), static_cast<int>(plo));
}

ICPX_DEV double wave_sum(double v) {
    v = v + swap_xor<32>(v);
    v = v + swap_xor<16>(v);
    v = v + dpp_xor<8>(v);
    v = v + dpp_xor<4>(v);
    v = v + dpp_xor<2>(v);
    v = v + dpp_xor<1>(v);
    return v;
}
// Transposed butterfly levels (same additions as wave_sum's, half the adds per level): given TWO values per
// lane, lanes whose bit DIST is clear end with x[l] + x[l ^ DIST], lanes whose bit is set with y[l] + y[l ^ DIST].
// DIST 32 / 16: ONE v_permlane*_swap per 32-bit half does the whole exchange (x's upper half / odd rows trade
// places with y's lower half / even rows); DIST 8: each lane sends the value it does not keep through DPP.
template <int DIST>
ICPX_DEV double xchg_sum(double x, double y) {
    if constexpr (DIST == 32 || DIST == 16) {
        const unsigned xl = static_cast<unsigned>(__double2loint(x)), xh = static_cast<unsigned>(__double2hiint(x));
        const unsigned yl = static_cast<unsigned>(__double2loint(y)), yh = static_cast<unsigned>(__double2hiint(y));
        double a, b;
        if constexpr (DIST == 32) {
            auto lo = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
            auto hi = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
            a = __hiloint2double(static_cast<int>(hi[0]), static_cast<int>(lo[0]));
            b = __hiloint2double(static_cast<int>(hi[1]), static_cast<int>(lo[1]));
        } else {
            auto lo = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
            auto hi = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
            a = __hiloint2double(static_cast<int>(hi[0]), static_cast<int>(lo[0]));
            b = __hiloint2double(static_cast<int>(hi[1]), static_cast<int>(lo[1]));
        }
        return a + b;
    } else {
        static_assert(DIST == 8, "xchg_sum: DIST must be 32, 16 or 8");
        const bool up = (threadIdx.x & 8) != 0;
        const double keep = up ? y : x, send = up ? x : y;
        return keep + dpp_xor<8>(send);
    }
}

ICPX_DEV int wave_sum_int(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
    return v;
}

}  // namespace icpx
