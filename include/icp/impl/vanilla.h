// Drop-in for cornellev/icp include/icp/impl/vanilla.h + lib/icp/impl/vanilla.cpp (2D vanilla ICP).
#pragma once
#include "icp/impl/device_icp.h"
namespace icp {
    class Vanilla final : public DeviceICP<Dimension::TwoD> {
    public:
        Vanilla(): DeviceICP(ICPX_VANILLA, Config()) {}
        explicit Vanilla(const Config& config): DeviceICP(ICPX_VANILLA, config) {}
    };
}
