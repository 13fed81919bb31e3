// orb.hpp — shared between orb.hip (device pipeline) and orb_host.cpp (quadtree).
#pragma once
#include <vector>

#include "../../include/sivo_hip.h"

namespace sivo {

// ORBextractor::DistributeOctTree (reference ORBextractor.cc:544-750); host only.
int distribute_quadtree(const SivoKeyPoint *kp, int n, int min_x, int max_x, int min_y, int max_y, int target,
                        std::vector<SivoKeyPoint> &out);

}  // namespace sivo
