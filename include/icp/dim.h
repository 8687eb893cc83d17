// Drop-in for cornellev/icp include/icp/dim.h:4
#ifndef ICP_DIM_H
#define ICP_DIM_H
#include <cstdint>
namespace icp {
    enum Dimension : std::uint8_t { TwoD = 2, ThreeD = 3 };
}
#endif
