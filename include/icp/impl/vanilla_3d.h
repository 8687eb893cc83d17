// Drop-in for cornellev/icp include/icp/impl/vanilla_3d.h + lib/icp/impl/vanilla_3d.cpp.
#pragma once
#include "icp/impl/device_icp.h"
namespace icp {
    class Vanilla3d final : public DeviceICP<Dimension::ThreeD> {
    public:
        Vanilla3d(): DeviceICP(ICPX_VANILLA, Config()) {}
        explicit Vanilla3d(const Config& config): DeviceICP(ICPX_VANILLA, config) {}
    };
}
