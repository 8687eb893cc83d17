// Drop-in for cornellev/icp include/icp/impl/trimmed_3d.h + lib/icp/impl/trimmed_3d.cpp
// (Config "max_distance", default 1.0 — trimmed_3d.cpp:27; the reference's default constructor
// leaves max_distance uninitialised, here it gets the same 1.0).
#pragma once
#include "icp/impl/device_icp.h"
namespace icp {
    class Trimmed3d final : public DeviceICP<Dimension::ThreeD> {
    public:
        Trimmed3d(): DeviceICP(ICPX_TRIMMED, Config()) {}
        explicit Trimmed3d(const Config& config): DeviceICP(ICPX_TRIMMED, config) {}
    };
}
