// test_dropin.cpp — the reference's own integration tests (tests/test.cpp:83-233 and
// tests/test3d.cpp:36-206), re-typed against this repository's drop-in headers (include/icp/*.h)
// with the reference's tolerances, plus checks of the interface semantics a consumer relies on.
// Needs a GPU: the built-in methods run in libicpx.so.  Eigen-free (the image has none): the
// headers fall back to icp/mini_linalg.h; with Eigen on the include path the same source builds
// against the Eigen types.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <variant>

#include "icp/driver.h"
#include "icp/geo.h"
#include "icp/icp.h"
#include "icp/impl/vanilla_3d.h"

static int g_failed = 0, g_checked = 0;
#define assert_true(cond)                                                        \
    do {                                                                         \
        g_checked++;                                                             \
        if (!(cond)) {                                                           \
            g_failed++;                                                          \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
        }                                                                        \
    } while (0)

constexpr double TRANS_EPS = 0.5;  // tests/test.cpp:21
constexpr double RAD_EPS = 0.01;   // tests/test.cpp:22

#ifndef ICP_HAVE_EIGEN
using Mat2 = icp::mini::Mat<2, 2>;
using Mat3 = icp::mini::Mat<3, 3>;
static Mat2 rot2(double a) { return icp::mini::rotation2d(a); }
static Mat3 rotx(double a) { return icp::mini::rotation3d(a, 0); }
static Mat3 roty(double a) { return icp::mini::rotation3d(a, 1); }
static Mat3 rotz(double a) { return icp::mini::rotation3d(a, 2); }
static double angle_of(const Mat2& r) { return std::atan2(r(1, 0), r(0, 0)); }  // Rotation2Dd(r).smallestAngle()
static bool is_approx(const Mat3& a, const Mat3& b, double eps) {
    double d = 0, na = 0, nb = 0;
    for (int i = 0; i < 9; i++) { d += (a.m[i] - b.m[i]) * (a.m[i] - b.m[i]); na += a.m[i] * a.m[i]; nb += b.m[i] * b.m[i]; }
    return d <= eps * eps * std::min(na, nb);
}
#else
using Mat2 = Eigen::Matrix2d;
using Mat3 = Eigen::Matrix3d;
static Mat2 rot2(double a) { return Eigen::Rotation2Dd(a).toRotationMatrix(); }
static Mat3 rotx(double a) { return Eigen::AngleAxisd(a, Eigen::Vector3d::UnitX()).toRotationMatrix(); }
static Mat3 roty(double a) { return Eigen::AngleAxisd(a, Eigen::Vector3d::UnitY()).toRotationMatrix(); }
static Mat3 rotz(double a) { return Eigen::AngleAxisd(a, Eigen::Vector3d::UnitZ()).toRotationMatrix(); }
static double angle_of(const Mat2& r) { return Eigen::Rotation2Dd(r).smallestAngle(); }
static bool is_approx(const Mat3& a, const Mat3& b, double eps) { return a.isApprox(b, eps); }
#endif

#define assert_translation2_eps(ex, ey, real, eps)       \
    do {                                                 \
        assert_true(std::abs((real).x() - (ex)) < (eps)); \
        assert_true(std::abs((real).y() - (ey)) < (eps)); \
    } while (0)
#define assert_rotation2_eps(expected_angle, real, eps) assert_true(std::abs(angle_of(real) - (expected_angle)) < (eps))

static icp::RBTransform2 make_tf2(double angle, double tx, double ty) {
    icp::RBTransform2 t = icp::RBTransform2::Identity();
    t.linear() = rot2(angle);
    t.translation() = icp::Vector2(tx, ty);
    return t;
}

// tests/test.cpp:83-216
void test_icp_generic(const std::string& method, const icp::Config& config) {
    std::unique_ptr<icp::ICP2> icp = icp::ICP2::from_method(method, config).value();
    icp::ICPDriver driver(std::move(icp));
    driver.set_max_iterations(100);
    driver.set_transform_tolerance(0.1 * M_PI / 180, 0.1);

    {  // case 1: single point translation
        icp::PointCloud2 a(2, 1);
        a.col(0) << 0, 0;
        icp::PointCloud2 b(2, 1);
        b.col(0) << 100, 0;
        auto result = driver.converge(a, b, icp::RBTransform2::Identity());
        assert_true(result.iteration_count <= 10);
        assert_translation2_eps(100, 0, result.transform.translation(), TRANS_EPS);
        assert_rotation2_eps(0.0, Mat2(result.transform.rotation()), RAD_EPS);
    }
    {  // case 2: identity
        icp::PointCloud2 a(2, 2);
        a.col(0) << 0, 0;
        a.col(1) << 100, 100;
        icp::PointCloud2 b = a;
        auto result = driver.converge(a, b, icp::RBTransform2::Identity());
        assert_translation2_eps(0, 0, result.transform.translation(), TRANS_EPS);
        assert_rotation2_eps(0.0, Mat2(result.transform.rotation()), RAD_EPS);
    }
    for (int deg = 0; deg < 10; deg++) {  // case 3a
        double angle = (double)deg * M_PI / 180.0;
        icp::PointCloud2 a(2, 2);
        a.col(0) << -100, -100;
        a.col(1) << 100, 100;
        icp::PointCloud2 b = make_tf2(angle, 0, 0) * a;
        auto result = driver.converge(a, b, icp::RBTransform2::Identity());
        assert_translation2_eps(0, 0, result.transform.translation(), TRANS_EPS);
        assert_rotation2_eps(angle, Mat2(result.transform.rotation()), RAD_EPS);
    }
    for (int deg = 0; deg < 10; deg++) {  // case 3b
        double angle = deg * M_PI / 180.0;
        icp::PointCloud2 a(2, 2);
        a.col(0) << 0, 0;
        a.col(1) << 1, 0;
        icp::PointCloud2 b = make_tf2(angle, 0, 0) * a;
        auto result = driver.converge(a, b, icp::RBTransform2::Identity());
        assert_translation2_eps(0, 0, result.transform.translation(), TRANS_EPS);
        assert_rotation2_eps(angle, Mat2(result.transform.rotation()), RAD_EPS);
    }
    {  // case 4: pure translation
        icp::PointCloud2 a(2, 2);
        a.col(0) << 0, 0;
        a.col(1) << 0, 100;
        icp::PointCloud2 b(2, 2);
        b.col(0) << 100, 0;
        b.col(1) << 100, 100;
        auto result = driver.converge(a, b, icp::RBTransform2::Identity());
        assert_translation2_eps(100, 0, result.transform.translation(), TRANS_EPS);
        assert_rotation2_eps(0.0, Mat2(result.transform.rotation()), RAD_EPS);
    }
    {  // case 5: translation + rotation
        double angle = 45 * M_PI / 180.0;
        icp::PointCloud2 a(2, 2);
        a.col(0) << 0, 0;
        a.col(1) << 100, 0;
        icp::PointCloud2 b = make_tf2(angle, 5, 5) * a;
        auto result = driver.converge(a, b, icp::RBTransform2::Identity());
        assert_translation2_eps(5, 5, result.transform.translation(), TRANS_EPS);
        assert_rotation2_eps(angle, Mat2(result.transform.rotation()), RAD_EPS);
    }
    {  // case 6: noise
        double angle = 30 * M_PI / 180.0;
        std::default_random_engine generator;
        std::normal_distribution<double> noise_dist(0.0, 1.0);
        icp::PointCloud2 a(2, 4);
        a.col(0) << 0, 0;
        a.col(1) << 100, 0;
        a.col(2) << 50, 50;
        a.col(3) << 0, 50;
        icp::PointCloud2 b = make_tf2(angle, 20, 10) * a;
        for (icp::Index i = 0; i < b.cols(); i++) {
            double nx = noise_dist(generator), ny = noise_dist(generator);
            b(0, i) += nx;
            b(1, i) += ny;
        }
        auto result = driver.converge(a, b, icp::RBTransform2::Identity());
        assert_translation2_eps(20, 10, result.transform.translation(), TRANS_EPS * 3);
        assert_rotation2_eps(angle, Mat2(result.transform.rotation()), RAD_EPS * 5);
    }
}

#define assert_translation3_eps(e, real, eps)                \
    do {                                                     \
        assert_true(std::abs((real).x() - (e).x()) < (eps)); \
        assert_true(std::abs((real).y() - (e).y()) < (eps)); \
        assert_true(std::abs((real).z() - (e).z()) < (eps)); \
    } while (0)

static icp::RBTransform3 make_tf3(const Mat3& r, const icp::Vector3& t) {
    icp::RBTransform3 x = icp::RBTransform3::Identity();
    x.linear() = r;
    x.translation() = t;
    return x;
}

// tests/test3d.cpp:36-202
void test_icp_3d(const icp::Config& config) {
    std::unique_ptr<icp::ICP3> icp = std::make_unique<icp::Vanilla3d>(config);
    icp::ICPDriver driver(std::move(icp));
    driver.set_max_iterations(100);
    driver.set_transform_tolerance(0.1 * M_PI / 180, 0.1);

    {  // case 1
        icp::PointCloud3 a(3, 1);
        a.col(0) << 0, 0, 0;
        icp::PointCloud3 b(3, 1);
        b.col(0) << 100, 0, 0;
        auto result = driver.converge(a, b, icp::RBTransform3::Identity());
        assert_true(result.iteration_count <= 10);
        assert_translation3_eps(icp::Vector3(100, 0, 0), result.transform.translation(), TRANS_EPS);
        assert_true(is_approx(Mat3(result.transform.rotation()), Mat3::Identity(), RAD_EPS));
    }
    auto unit = [](double s) {
        icp::PointCloud3 a(3, 3);
        a.col(0) << s, 0, 0;
        a.col(1) << 0, s, 0;
        a.col(2) << 0, 0, s;
        return a;
    };
    auto about_centroid = [&](const Mat3& R, const icp::PointCloud3& a, icp::Vector3& expected_t) {
        icp::Vector3 center = icp::get_centroid(a);
        icp::Vector3 rc = R * center;
        expected_t = center - rc;
        return make_tf3(R, expected_t);
    };
    for (int deg = 0; deg < 10; deg++) {  // case 2
        icp::PointCloud3 a = unit(1);
        Mat3 R = rotx((double)deg * M_PI / 180.0);
        icp::Vector3 expected_t;
        icp::PointCloud3 b = about_centroid(R, a, expected_t) * a;
        auto result = driver.converge(a, b, icp::RBTransform3::Identity());
        assert_translation3_eps(expected_t, result.transform.translation(), TRANS_EPS);
        assert_true(is_approx(Mat3(result.transform.rotation()), R, RAD_EPS));
    }
    {  // case 3
        icp::PointCloud3 a = unit(1);
        double r30 = 30.0 * M_PI / 180.0;
        Mat3 R = rotz(r30) * roty(r30) * rotx(r30);
        icp::Vector3 expected_t;
        icp::PointCloud3 b = about_centroid(R, a, expected_t) * a;
        auto result = driver.converge(a, b, icp::RBTransform3::Identity());
        assert_translation3_eps(expected_t, result.transform.translation(), TRANS_EPS);
        assert_true(is_approx(Mat3(result.transform.rotation()), R, RAD_EPS));
    }
    double r10 = 10.0 * M_PI / 180.0;
    Mat3 R10 = rotz(r10) * roty(r10) * rotx(r10);
    {  // case 4
        icp::PointCloud3 a = unit(100);
        icp::Vector3 t(50, 50, 50);
        icp::PointCloud3 b = make_tf3(R10, t) * a;
        auto result = driver.converge(a, b, icp::RBTransform3::Identity());
        assert_translation3_eps(t, result.transform.translation(), TRANS_EPS);
        assert_true(is_approx(Mat3(result.transform.rotation()), R10, RAD_EPS));
    }
    {  // case 5
        icp::PointCloud3 a = unit(100);
        icp::Vector3 t(20, 10, 30);
        std::default_random_engine generator;
        std::normal_distribution<double> noise_dist(0.0, 1.0);
        icp::PointCloud3 b = make_tf3(R10, t) * a;
        for (icp::Index i = 0; i < a.cols(); ++i) {
            b(0, i) += noise_dist(generator);
            b(1, i) += noise_dist(generator);
            b(2, i) += noise_dist(generator);
        }
        auto result = driver.converge(a, b, icp::RBTransform3::Identity());
        assert_translation3_eps(t, result.transform.translation(), TRANS_EPS + 1.0);
        assert_true(is_approx(Mat3(result.transform.rotation()), R10, 0.1));
    }
}

// A consumer-written ICP instance (book/writing_icp.md:20-39): must still compile against the
// protected surface and run through the driver's host loop.
class ShiftToCentroid final : public icp::ICP2 {
public:
    int setups = 0;

protected:
    void setup() override {
        setups++;
        for (icp::Index i = 0; i < a.cols(); i++) matches[static_cast<size_t>(i)] = Match{i, i, 1.0};
    }

public:
    void iterate() override {
        Vector ca = icp::get_centroid(a), cb = icp::get_centroid(b);
        RBTransform step = RBTransform::Identity();
        step.translation() = Vector(cb.x() - ca.x(), cb.y() - ca.y());
        transform = step;
        for (auto& m : matches) m.cost = 0.25;
    }
};

// A 3D consumer-written instance whose k-th iteration turns the transform to a given angle about z: exercises the host
// loop's transform tolerance (driver.h:196-210), i.e. the quaternion angle — both branches of the matrix -> quaternion
// conversion (trace > 0 and trace <= 0).
class SpinAboutZ final : public icp::ICP3 {
public:
    std::vector<double> angles;
    size_t k = 0;

protected:
    void setup() override {
        k = 0;
        for (icp::Index i = 0; i < a.cols(); i++) matches[static_cast<size_t>(i)] = Match{i, i, 1.0};
    }

public:
    void iterate() override {
        const double th = angles[k < angles.size() ? k : angles.size() - 1];
        k++;
        RBTransform t = RBTransform::Identity();
        t.matrix()(0, 0) = std::cos(th);
        t.matrix()(0, 1) = -std::sin(th);
        t.matrix()(1, 0) = std::sin(th);
        t.matrix()(1, 1) = std::cos(th);
        transform = t;
        for (auto& m : matches) m.cost = 1.0 + static_cast<double>(k);  // (costs keep changing: only the transform can stop it)
    }
};

void test_host_loop_transform_tolerance_3d() {
    icp::PointCloud3 a(3, 2), b(3, 2);
    for (int i = 0; i < 2; i++) {
        a.col(i) << 1.0 * i, 0, 0;
        b.col(i) << 1.0 * i, 0, 0;
    }
    // steps of 3.0 (trace <= 0: the other quaternion branch), 0.1, 0.002, 0.0005: stops at the first step below 1e-3
    auto spin = std::make_unique<SpinAboutZ>();
    spin->angles = {3.0, 3.1, 3.102, 3.1025, 3.1026};
    icp::ICPDriver<icp::Dimension::ThreeD> driver(std::move(spin));
    driver.set_max_iterations(20);
    driver.set_transform_tolerance(1e-3, 1e-3);
    auto r = driver.converge(a, b, icp::RBTransform3::Identity());
    assert_true(r.iteration_count == 4);
    // a step of exactly pi about z (trace = -1) is not "no rotation"
    auto half = std::make_unique<SpinAboutZ>();
    half->angles = {3.141592653589793, 0.0, 0.0};
    icp::ICPDriver<icp::Dimension::ThreeD> driver2(std::move(half));
    driver2.set_max_iterations(20);
    driver2.set_transform_tolerance(1e-3, 1e-3);
    auto r2 = driver2.converge(a, b, icp::RBTransform3::Identity());
    assert_true(r2.iteration_count == 3);  // pi (no stop), back to 0 (a turn of pi again), 0 -> 0 stops
}

void test_interface_semantics() {
    // registry (icp.h:95-114)
    assert_true(icp::ICP2::is_method_registered("vanilla"));
    assert_true(icp::ICP2::is_method_registered("trimmed"));
    assert_true(icp::ICP2::is_method_registered("feature_aware"));
    assert_true(icp::ICP2::registered_methods().size() == 3);  // lib/icp/icp.cpp:17-21
    assert_true(!icp::ICP3::is_method_registered("feature_aware"));
    assert_true(!icp::ICP2::is_method_registered("nope"));
    assert_true(!icp::ICP2::from_method("nope", icp::Config()).has_value());
    assert_true(icp::ICP3::registered_methods().size() == 2);
    // Config (config.h:18-32)
    icp::Config c;
    c.set("overlap_rate", 0.7);
    assert_true(c.get<double>("overlap_rate", 0.9) == 0.7);
    assert_true(c.get<int>("missing", 3) == 3);
    bool threw = false;
    try {
        (void)c.get<int>("overlap_rate", 1);
    } catch (const std::bad_variant_access&) {
        threw = true;
    }
    assert_true(threw);
    // step-wise use (README.md:28-38): begin / iterate / cost / transform / matches
    auto icp = icp::ICP2::from_method("trimmed", icp::Config()).value();
    icp::PointCloud2 a(2, 3), b(2, 3);
    a.col(0) << 0, 0;
    a.col(1) << 1, 0;
    a.col(2) << 0, 2;
    b.col(0) << 0.25, 0;
    b.col(1) << 1.25, 0;
    b.col(2) << 0.25, 2;
    icp->begin(a, b, icp::RBTransform2::Identity());
    double c0 = icp->calculate_cost();
    assert_true(std::abs(c0 - 0.25) < 1e-12);  // RMS of d2 = 0.0625 each
    icp->iterate();
    icp->iterate();
    assert_true(icp->calculate_cost() < 1e-9);
    assert_true(std::abs(icp->current_transform().translation().x() - 0.25) < 1e-9);
    assert_true(icp->get_matches().size() == 3);
    // third-party subclass through the driver's host loop
    auto mine = std::make_unique<ShiftToCentroid>();
    ShiftToCentroid* raw = mine.get();
    icp::ICPDriver<icp::Dimension::TwoD> driver(std::move(mine));
    driver.set_max_iterations(3);
    auto r = driver.converge(a, b, icp::RBTransform2::Identity());
    assert_true(r.iteration_count == 3 && raw->setups == 1);
    assert_true(std::abs(r.transform.translation().x() - 0.25) < 1e-12);
    assert_true(std::abs(r.cost - 0.5) < 1e-12);
}

int main() {
    if (icpx_init(-1) != ICPX_OK) {
        std::fprintf(stderr, "icpx_init failed: %s\n", icpx_last_error());
        return 2;
    }
    test_icp_generic("vanilla", icp::Config());
    icp::Config trimmed_config;
    trimmed_config.set("overlap_rate", 1.0);  // tests/test.cpp:223-226
    test_icp_generic("trimmed", trimmed_config);
    // tests/test.cpp:228-232
    icp::Config feature_config;
    feature_config.set("overlap_rate", 1.0);
    feature_config.set("feature_weight", 0.7);
    feature_config.set("symmetric_neighbors", 1);
    test_icp_generic("feature_aware", feature_config);
    test_icp_3d(icp::Config());
    test_interface_semantics();
    test_host_loop_transform_tolerance_3d();
    std::printf("%d checks, %d failed\n", g_checked, g_failed);
    return g_failed ? 1 : 0;
}
