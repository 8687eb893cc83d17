// Drop-in for cornellev/icp include/icp/impl/trimmed.h + lib/icp/impl/trimmed.cpp (2D trimmed ICP;
// Config "overlap_rate", default 0.9 — trimmed.cpp:28).
#pragma once
#include "icp/impl/device_icp.h"
namespace icp {
    class Trimmed final : public DeviceICP<Dimension::TwoD> {
    public:
        explicit Trimmed(const Config& config): DeviceICP(ICPX_TRIMMED, config) {}
    };
}
