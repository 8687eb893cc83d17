// parse_scan.h — drop-in for the reference's LiDAR CSV reader (common/parse_scan.h:9,
// common/parse_scan.cpp:18-50), header-only.
//
// One "x,y" pair per line; a line whose x or y is infinite is skipped (the scans in ex_data mark
// missing returns with +-inf); an unreadable file throws std::runtime_error
// (parse_scan.cpp:20-22).  Returns the 2 x n cloud in file order.
#pragma once

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "icp/geo.h"

inline icp::PointCloud2 parse_lidar_scan(const std::string& path) {
    std::ifstream file(path);
    if (!file.is_open()) throw std::runtime_error("failed to read lidar scan: failed to open file");
    std::vector<double> xy;
    for (std::string row; std::getline(file, row);) {
        // strtod semantics as in the reference: leading blanks skipped, unparsable text reads as 0;
        // a row without a comma parses its whole text for both fields (npos + 1 == 0)
        const std::string::size_type comma = row.find(',');
        const double x = std::strtod(row.substr(0, comma).c_str(), nullptr);
        const double y = std::strtod(row.c_str() + (comma == std::string::npos ? 0 : comma + 1), nullptr);
        if (std::isinf(x) || std::isinf(y)) continue;
        xy.push_back(x);
        xy.push_back(y);
    }
    icp::PointCloud2 cloud(2, static_cast<long>(xy.size() / 2));
    for (std::size_t i = 0; i < xy.size(); i++) cloud.data()[i] = xy[i];
    return cloud;
}
