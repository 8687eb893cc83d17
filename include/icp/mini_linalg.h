// mini_linalg.h — the few Eigen types the ICP interface exposes (geo.h:16-29), in a minimal
// built-in form, used ONLY when <Eigen/Core> is not available to the consumer (this image has no
// Eigen).  Memory layout is Eigen's: column-major, so .data() feeds the C ABI unchanged.
// With Eigen present, include/icp/geo.h uses the real Eigen typedefs and this file is not seen.
#pragma once

#include <array>
#include <cmath>
#include <cstddef>
#include <vector>

namespace icp {
namespace mini {

    using Index = std::ptrdiff_t;

    template<int N>
    struct Vec {
        std::array<double, N> v{};
        Vec() = default;
        Vec(double x, double y) { static_assert(N == 2); v = {x, y}; }
        Vec(double x, double y, double z) { static_assert(N == 3); v = {x, y, z}; }
        static Vec Zero() { return Vec(); }
        double& operator()(int i) { return v[i]; }
        double operator()(int i) const { return v[i]; }
        double& operator[](int i) { return v[i]; }
        double operator[](int i) const { return v[i]; }
        double x() const { return v[0]; }
        double y() const { return v[1]; }
        double z() const { static_assert(N >= 3); return v[2]; }
        static constexpr int size() { return N; }
        double norm() const { double s = 0; for (double e : v) s += e * e; return std::sqrt(s); }
        Vec operator-(const Vec& o) const { Vec r; for (int i = 0; i < N; i++) r.v[i] = v[i] - o.v[i]; return r; }
        Vec operator+(const Vec& o) const { Vec r; for (int i = 0; i < N; i++) r.v[i] = v[i] + o.v[i]; return r; }
        const double* data() const { return v.data(); }
        double* data() { return v.data(); }
    };

    template<int R, int C>
    struct Mat {  // fixed size, column-major
        std::array<double, R * C> m{};
        static Mat Identity() { Mat r; for (int i = 0; i < (R < C ? R : C); i++) r(i, i) = 1.0; return r; }
        static Mat Zero() { return Mat(); }
        double& operator()(int r, int c) { return m[c * R + r]; }
        double operator()(int r, int c) const { return m[c * R + r]; }
        const double* data() const { return m.data(); }
        double* data() { return m.data(); }
        Mat<C, R> transpose() const { Mat<C, R> t; for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) t(c, r) = (*this)(r, c); return t; }
        template<int K>
        Mat<R, K> operator*(const Mat<C, K>& o) const {
            Mat<R, K> out;
            for (int r = 0; r < R; r++) for (int k = 0; k < K; k++) { double s = 0; for (int c = 0; c < C; c++) s += (*this)(r, c) * o(c, k); out(r, k) = s; }
            return out;
        }
        Vec<R> operator*(const Vec<C>& o) const {
            Vec<R> out;
            for (int r = 0; r < R; r++) { double s = 0; for (int c = 0; c < C; c++) s += (*this)(r, c) * o(c); out(r) = s; }
            return out;
        }
    };

    // Dim x n, column-major, dynamic column count
    template<int R>
    class Cloud {
        std::vector<double> d_;
        Index cols_ = 0;

    public:
        class ColRef {
            double* p_;
            int fill_ = 0;
        public:
            explicit ColRef(double* p): p_(p) {}
            ColRef& operator<<(double v) { p_[fill_++] = v; return *this; }   // a.col(0) << x, y;
            ColRef& operator,(double v) { p_[fill_++] = v; return *this; }
            ColRef& operator=(const Vec<R>& v) { for (int i = 0; i < R; i++) p_[i] = v(i); return *this; }
            ColRef& operator+=(const Vec<R>& v) { for (int i = 0; i < R; i++) p_[i] += v(i); return *this; }
            operator Vec<R>() const { Vec<R> v; for (int i = 0; i < R; i++) v(i) = p_[i]; return v; }
            double& operator()(int i) { return p_[i]; }
        };
        Cloud() = default;
        Cloud(Index rows, Index cols): d_(static_cast<size_t>(R * cols), 0.0), cols_(cols) { (void)rows; }
        Index rows() const { return R; }
        Index cols() const { return cols_; }
        Index size() const { return R * cols_; }
        void resize(Index rows, Index cols) { (void)rows; d_.assign(static_cast<size_t>(R * cols), 0.0); cols_ = cols; }
        double& operator()(Index r, Index c) { return d_[static_cast<size_t>(c * R + r)]; }
        double operator()(Index r, Index c) const { return d_[static_cast<size_t>(c * R + r)]; }
        ColRef col(Index c) { return ColRef(d_.data() + c * R); }
        Vec<R> col(Index c) const { Vec<R> v; for (int i = 0; i < R; i++) v(i) = d_[static_cast<size_t>(c * R + i)]; return v; }
        const double* data() const { return d_.data(); }
        double* data() { return d_.data(); }
    };

    // Eigen::Transform<double, Dim, Isometry>: (Dim+1)x(Dim+1) column-major matrix
    template<int D>
    class Transform {
        Mat<D + 1, D + 1> m_ = Mat<D + 1, D + 1>::Identity();

    public:
        class LinearRef {
            Transform& t_;
        public:
            explicit LinearRef(Transform& t): t_(t) {}
            LinearRef& operator=(const Mat<D, D>& r) { for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) t_.m_(i, j) = r(i, j); return *this; }
            operator Mat<D, D>() const { return static_cast<const Transform&>(t_).linear(); }
        };
        class TranslationRef {
            Transform& t_;
        public:
            explicit TranslationRef(Transform& t): t_(t) {}
            TranslationRef& operator=(const Vec<D>& v) { for (int i = 0; i < D; i++) t_.m_(i, D) = v(i); return *this; }
            operator Vec<D>() const { return static_cast<const Transform&>(t_).translation(); }
            double x() const { return t_.m_(0, D); }
            double y() const { return t_.m_(1, D); }
            double z() const { static_assert(D >= 3); return t_.m_(2, D); }
            double operator()(int i) const { return t_.m_(i, D); }
        };
        static Transform Identity() { return Transform(); }
        Mat<D + 1, D + 1>& matrix() { return m_; }
        const Mat<D + 1, D + 1>& matrix() const { return m_; }
        LinearRef linear() { return LinearRef(*this); }
        TranslationRef translation() { return TranslationRef(*this); }
        Mat<D, D> linear() const { Mat<D, D> r; for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) r(i, j) = m_(i, j); return r; }
        Mat<D, D> rotation() const { return linear(); }  // Isometry mode: rotation() == linear()
        Vec<D> translation() const { Vec<D> v; for (int i = 0; i < D; i++) v(i) = m_(i, D); return v; }
        Transform operator*(const Transform& o) const { Transform r; r.m_ = m_ * o.m_; return r; }
        Vec<D> operator*(const Vec<D>& p) const { Vec<D> r = linear() * p; for (int i = 0; i < D; i++) r(i) += m_(i, D); return r; }
        Cloud<D> operator*(const Cloud<D>& c) const {
            Cloud<D> out(D, c.cols());
            for (Index k = 0; k < c.cols(); k++) out.col(k) = (*this) * c.col(k);
            return out;
        }
    };

    inline Mat<2, 2> rotation2d(double angle) {
        Mat<2, 2> r;
        r(0, 0) = std::cos(angle); r(0, 1) = -std::sin(angle);
        r(1, 0) = std::sin(angle); r(1, 1) = std::cos(angle);
        return r;
    }
    // axis 0/1/2 = X/Y/Z, as Eigen::AngleAxisd(angle, UnitX/Y/Z).toRotationMatrix()
    inline Mat<3, 3> rotation3d(double angle, int axis) {
        Mat<3, 3> r = Mat<3, 3>::Identity();
        int a = (axis + 1) % 3, b = (axis + 2) % 3;
        r(a, a) = std::cos(angle); r(a, b) = -std::sin(angle);
        r(b, a) = std::sin(angle); r(b, b) = std::cos(angle);
        return r;
    }
}  // namespace mini
}  // namespace icp
