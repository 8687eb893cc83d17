// Drop-in for cornellev/icp include/icp/neighbor.h:10-13
#ifndef ICP_NEIGHBOR_H
#define ICP_NEIGHBOR_H
#include <cstddef>
#include <vector>
namespace icp {
    struct Neighbors {
        std::vector<double> distances;
        std::vector<std::ptrdiff_t> indices;  // Eigen::Index is std::ptrdiff_t
    };
}
#endif
