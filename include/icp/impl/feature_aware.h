// Drop-in for cornellev/icp include/icp/impl/feature_aware.h + lib/icp/impl/feature_aware.cpp (2D
// feature-aware trimmed ICP; Config "overlap_rate" 0.9, "feature_weight" 0.7,
// "symmetric_neighbors" 10 — feature_aware.cpp:22-41).
#pragma once
#include "icp/impl/device_icp.h"
namespace icp {
    class FeatureAware final : public DeviceICP<Dimension::TwoD> {
        static Config make(double overlap_rate, double feature_weight, int symmetric_neighbors) {
            Config c;
            c.set("overlap_rate", overlap_rate);
            c.set("feature_weight", feature_weight);
            c.set("symmetric_neighbors", symmetric_neighbors);
            return c;
        }

    public:
        explicit FeatureAware(const Config& config): DeviceICP(ICPX_FEATURE_AWARE, config) {}
        // feature_aware.h:14
        FeatureAware(double overlap_rate, double feature_weight, int symmetric_neighbors)
            : DeviceICP(ICPX_FEATURE_AWARE, make(overlap_rate, feature_weight, symmetric_neighbors)) {}
    };
}
