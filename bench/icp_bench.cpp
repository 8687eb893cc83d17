// icp_bench — the reference's `make bench` report (bench/bench.cpp:88-142) on the MI355X path.
//
// For every registered ICP2 method and every ex_data scan pair it runs ICPDriver::converge a fixed
// number of times with the reference's settings (50 iterations at most, 0.1 degree / 0.01 unit
// transform tolerance, bench.cpp:109-113) and prints the reference's report lines in the
// reference's order, so that existing tooling reading that report keeps working.  Lines prefixed
// "+ " are added per block (the reference never prints them): point counts, NN correspondence
// queries/s, converged pairs/s and the number of sqrt-tie queries (see sqrt_tie_queries).
//
// usage: icp_bench [data_root=. ] [invocations=10]      (reads <data_root>/ex_data/scanK/{first,second}.csv)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "common/parse_scan.h"
#include "icp/config.h"
#include "icp/driver.h"
#include "icp/geo.h"
#include "icp/icp.h"

namespace {

template <typename T>
struct Summary {
    T lo{}, hi{}, median{};
    double mean = 0.0, total = 0.0;
    explicit Summary(std::vector<T> v) {
        std::sort(v.begin(), v.end());
        lo = v.front();
        hi = v.back();
        median = v[v.size() / 2];  // upper median, as bench.cpp:71,77
        for (const T& x : v) total += static_cast<double>(x);
        mean = total / static_cast<double>(v.size());
    }
};

struct Settings {
    std::string root = ".";
    std::size_t invocations = 10;
    std::size_t max_iterations = 50;
    double angle_tol = 0.1 * 3.14159265358979323846 / 180.0;
    double trans_tol = 0.01;
};

// Source points whose nearest target is ambiguous under the reference's comparison: the kd-tree compares
// sqrt(d2) (kdtree.h:140,151), so two targets with DIFFERENT d2 but bit-identical sqrt(d2) at the minimum are a
// tie there, resolved by traversal order, while this library compares d2 (then the index).  Such a query may get
// another partner than in the reference; counted at the identity guess, the
// correspondence step setup() runs.
std::size_t sqrt_tie_queries(const icp::PointCloud2& source, const icp::PointCloud2& target) {
    std::size_t ties = 0;
    for (icp::Index i = 0; i < source.cols(); i++) {
        double best = INFINITY;
        for (icp::Index j = 0; j < target.cols(); j++) {
            const double dx = target(0, j) - source(0, i), dy = target(1, j) - source(1, i);
            best = std::min(best, dx * dx + dy * dy);
        }
        const double root = std::sqrt(best);
        bool tie = false;
        for (icp::Index j = 0; j < target.cols(); j++) {
            const double dx = target(0, j) - source(0, i), dy = target(1, j) - source(1, i);
            const double d2 = dx * dx + dy * dy;
            tie = tie || (d2 != best && std::sqrt(d2) == root);
        }
        ties += tie ? 1 : 0;
    }
    return ties;
}

void report(const std::string& method, unsigned scan, const Settings& s) {
    const std::string dir = s.root + "/ex_data/scan" + std::to_string(scan);
    const icp::PointCloud2 source = parse_lidar_scan(dir + "/first.csv");
    const icp::PointCloud2 target = parse_lidar_scan(dir + "/second.csv");

    icp::ICPDriver driver(icp::ICP2::from_method(method, icp::Config()).value());
    driver.set_max_iterations(s.max_iterations);
    driver.set_transform_tolerance(s.angle_tol, s.trans_tol);

    std::vector<double> costs;
    std::vector<std::size_t> iterations;
    const auto t0 = std::chrono::steady_clock::now();
    for (std::size_t k = 0; k < s.invocations; k++) {
        const auto state = driver.converge(source, target, icp::RBTransform2::Identity());
        costs.push_back(state.cost);
        iterations.push_back(state.iteration_count);
    }
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const Summary<double> c(costs);
    const Summary<std::size_t> it(iterations);

    std::cout << "* Method: " << method << "\n"
              << "* Number of invocations: " << s.invocations << "\n"
              << "* Angle tolerance: " << s.angle_tol << " rad\n"
              << "* Translation tolerance: " << s.trans_tol << " scan units\n"
              << "* Min cost: " << c.lo << "\n"
              << "* Max cost: " << c.hi << "\n"
              << "* Median cost: " << c.median << "\n"
              << "* Mean cost: " << c.mean << "\n"
              << "* Min iterations: " << it.lo << "\n"
              << "* Max iterations: " << it.hi << "\n"
              << "* Median iterations: " << it.median << "\n"
              << "* Mean iterations: " << it.mean << "\n"
              << "* Average time per invocation: " << seconds / static_cast<double>(s.invocations) << "s\n"
              << "* Average time per iteration: " << seconds / it.total << "s\n";
    // correspondence steps: one per iterate, plus the one setup runs (vanilla.cpp:25, trimmed.cpp:35)
    const double nn_steps = it.total + static_cast<double>(s.invocations);
    std::cout << "+ Source / target points: " << source.cols() << " / " << target.cols() << "\n"
              << "+ NN correspondence queries/s: " << nn_steps * static_cast<double>(source.cols()) / seconds << "\n"
              << "+ Converged pairs/s: " << static_cast<double>(s.invocations) / seconds << "\n"
              << "+ sqrt-tie queries at the first correspondence step: " << sqrt_tie_queries(source, target) << "\n";
}

}  // namespace

int main(int argc, char** argv) {
    Settings s;
    if (argc > 1) s.root = argv[1];
    if (argc > 2) s.invocations = static_cast<std::size_t>(std::max(1L, std::atol(argv[2])));
    const char* bar = "=======================================\n";
    std::cout << "ICP ALGORITHM BENCHMARKING\n";
    try {
        for (const std::string& method : icp::ICP2::registered_methods()) {
            std::cout << bar;
            for (unsigned scan = 1; scan <= 3; scan++) {
                if (scan > 1) std::cout << "---------------------------------------\n";
                report(method, scan, s);
            }
        }
    } catch (const std::exception& e) {
        std::cerr << "icp_bench: " << e.what() << "\n";
        return 1;
    }
    std::cout << bar;
    return 0;
}
