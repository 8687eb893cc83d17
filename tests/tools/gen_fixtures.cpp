// gen_fixtures.cpp — captures the exact pseudo-random inputs the reference's tests draw from
// libstdc++ engines, so that they can be committed as data under tests/golden/:
//   * kd-tree KAT cloud: std::mt19937(123), uniform_real_distribution(-100, 100), 1000 x 3
//     (reference tests/test.cpp:39-49)
//   * 2D noise: default-seeded std::default_random_engine, normal_distribution(0, 1), 4 x 2 draws
//     in the order tests/test.cpp:197-210 consumes them
//   * 3D noise: same engine, 3 x 3 draws in the order tests/test3d.cpp:178-195 consumes them
// Build + run: g++ -O2 -o /tmp/gen_fixtures tests/tools/gen_fixtures.cpp && /tmp/gen_fixtures tests/golden
#include <cstdio>
#include <random>
#include <string>

static void dump(const std::string& path, const double* v, size_t n) {
    FILE* f = std::fopen(path.c_str(), "wb");
    std::fwrite(v, sizeof(double), n, f);
    std::fclose(f);
}

int main(int argc, char** argv) {
    std::string dir = argc > 1 ? argv[1] : ".";
    {
        std::mt19937 rng(123);
        std::uniform_real_distribution<double> dist(-100.0, 100.0);
        static double pts[3000];
        for (int i = 0; i < 1000; ++i) {
            // emplace_back(dist(rng), dist(rng), dist(rng)): g++ evaluates the arguments of a
            // call in unspecified order; this file fixes left-to-right (any uniform cloud serves the KAT).
            pts[i * 3 + 0] = dist(rng);
            pts[i * 3 + 1] = dist(rng);
            pts[i * 3 + 2] = dist(rng);
        }
        dump(dir + "/kdtree_kat_points_f64.bin", pts, 3000);
    }
    {
        std::default_random_engine generator;
        std::normal_distribution<double> noise(0.0, 1.0);
        double v[8];
        for (int i = 0; i < 8; i++) v[i] = noise(generator);
        dump(dir + "/noise2d_f64.bin", v, 8);
    }
    {
        std::default_random_engine generator;
        std::normal_distribution<double> noise(0.0, 1.0);
        double v[9];
        for (int i = 0; i < 9; i++) v[i] = noise(generator);
        dump(dir + "/noise3d_f64.bin", v, 9);
    }
    return 0;
}
