// device_icp.h — common base of the four built-in methods: an icp::ICP<Dim> whose state lives in an
// icpx handle (include/icpx/icpx.h).  No counterpart in the reference (its methods compute on the
// host); it only moves pointers across the C ABI.
#pragma once

#include <stdexcept>
#include <string>

#include "icp/icp.h"

namespace icp {

    template<const Dimension Dim>
    class DeviceICP : public ICP<Dim> {
    protected:
        using Base = ICP<Dim>;
        icpx_handle h_ = nullptr;
        mutable bool transform_stale_ = false;
        mutable bool matches_stale_ = false;

        DeviceICP(int method, const Config& config) {
            icpx_params p;
            icpx_default_params(&p);
            p.overlap_rate = config.get<double>("overlap_rate", p.overlap_rate);  // trimmed.cpp:28
            p.max_distance = config.get<double>("max_distance", p.max_distance);  // trimmed_3d.cpp:27
            p.feature_weight = config.get<double>("feature_weight", p.feature_weight);                // feature_aware.cpp:40
            p.symmetric_neighbors = config.get<int>("symmetric_neighbors", p.symmetric_neighbors);    // feature_aware.cpp:41
            // optional, non-reference keys
            p.flags = (config.get<int>("match_transformed", 0) ? ICPX_FLAG_MATCH_TRANSFORMED : 0) |
                      (config.get<int>("fill_matches", 0) ? ICPX_FLAG_FILL_MATCHES : 0) |
                      (config.get<int>("apply_guess", 0) ? ICPX_FLAG_APPLY_GUESS : 0);
            p.device = config.get<int>("device", -1);
            const std::string nn = config.get<std::string>("nn", "auto");
            if (nn == "auto") p.nn_mode = ICPX_NN_AUTO;
            else if (nn == "brute") p.nn_mode = ICPX_NN_BRUTE;
            else if (nn == "grid") p.nn_mode = ICPX_NN_GRID;
            else if (nn == "bvh") p.nn_mode = ICPX_NN_BVH;
            else if (nn == "grid_f32") p.nn_mode = ICPX_NN_GRID_F32;  // opt-in f32 correspondence mode (not reference arithmetic)
            else throw std::invalid_argument("icp::Config nn must be auto, brute, grid, bvh or grid_f32 (got \"" + nn + "\")");
            icpx_check(icpx_create(static_cast<int>(Dim), method, &p, &h_));
        }

        void setup() override {
            icpx_check(icpx_begin(h_, this->a.data(), this->a.cols(), this->b.data(), this->b.cols(),
                this->transform.matrix().data()));
            transform_stale_ = false;
            matches_stale_ = true;
        }

        bool device_cost(double& out) const override {
            icpx_check(icpx_cost(h_, &out));
            return true;
        }

        void device_sync_transform() const override {
            if (!transform_stale_) return;
            icpx_check(icpx_transform(h_, this->transform.matrix().data()));
            transform_stale_ = false;
        }

        void device_sync_matches() const override {
            if (!matches_stale_) return;
            const size_t n = static_cast<size_t>(this->a.cols());
            std::vector<int64_t> point(n), pair(n);
            std::vector<double> cost(n);
            icpx_check(icpx_matches(h_, point.data(), pair.data(), cost.data()));
            this->matches.resize(n);
            for (size_t i = 0; i < n; i++)
                this->matches[i] = typename Base::Match{static_cast<icp::Index>(point[i]),
                    static_cast<icp::Index>(pair[i]), cost[i]};
            matches_stale_ = false;
        }

        icpx_handle device_handle() const override { return h_; }

        /** ICPDriver::converge hands the whole loop to the device: keep host mirrors coherent. */
        void device_adopt_converged(const PointCloud<Dim>& source, const PointCloud<Dim>& target,
            const double* final_transform) override {
            this->a = source;
            this->b = target;
            for (int i = 0; i < (Dim + 1) * (Dim + 1); i++) this->transform.matrix().data()[i] = final_transform[i];
            transform_stale_ = false;
            matches_stale_ = true;
        }

    public:
        ~DeviceICP() override { icpx_destroy(h_); }
        DeviceICP(const DeviceICP&) = delete;
        DeviceICP& operator=(const DeviceICP&) = delete;

        void iterate() override {
            icpx_check(icpx_iterate(h_));
            transform_stale_ = true;
            matches_stale_ = true;
        }
    };
}
