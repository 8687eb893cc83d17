// Drop-in for cornellev/icp include/icp/geo.h:16-37.
// With Eigen available the types ARE the reference's Eigen typedefs; without it (ICP_NO_EIGEN or
// no <Eigen/Core> on the include path) a minimal built-in stand-in with the same memory layout is
// used so that the interface still compiles.
#pragma once

#include "icp/dim.h"

#if !defined(ICP_NO_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Core>)
#define ICP_HAVE_EIGEN 1
#endif
#endif

#ifdef ICP_HAVE_EIGEN
#include <Eigen/Core>
#include <Eigen/Geometry>
#include "icp/neighbor.h"

namespace icp {
    using Index = Eigen::Index;

    template<const Dimension Dim>
    using Vector = Eigen::Vector<double, Dim>;
    template<const Dimension Dim>
    using RBTransform = Eigen::Transform<double, Dim, Eigen::Isometry>;
    template<const Dimension Dim>
    using PointCloud = Eigen::Matrix<double, Dim, Eigen::Dynamic>;

    template<const auto Dim>
    Eigen::Vector<double, Dim> get_centroid(const Eigen::Matrix<double, Dim, Eigen::Dynamic>& points) {
        return points.rowwise().mean();
    }
}
#else
#include "icp/mini_linalg.h"
#include "icp/neighbor.h"

namespace icp {
    using Index = mini::Index;

    template<const Dimension Dim>
    using Vector = mini::Vec<Dim>;
    template<const Dimension Dim>
    using RBTransform = mini::Transform<Dim>;
    template<const Dimension Dim>
    using PointCloud = mini::Cloud<Dim>;

    template<int Dim>
    mini::Vec<Dim> get_centroid(const mini::Cloud<Dim>& points) {
        mini::Vec<Dim> c;
        for (Index k = 0; k < points.cols(); k++)
            for (int i = 0; i < Dim; i++) c(i) += points(i, k);
        for (int i = 0; i < Dim; i++) c(i) /= static_cast<double>(points.cols());
        return c;
    }
}
#endif

namespace icp {
    using Vector2 = Vector<Dimension::TwoD>;
    using Vector3 = Vector<Dimension::ThreeD>;
    using RBTransform2 = RBTransform<Dimension::TwoD>;
    using RBTransform3 = RBTransform<Dimension::ThreeD>;
    using PointCloud2 = PointCloud<Dimension::TwoD>;
    using PointCloud3 = PointCloud<Dimension::ThreeD>;
}
