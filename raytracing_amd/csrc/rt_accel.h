// Search index of the extend stage (see rt_accel.cpp).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/rt_amd.h"

namespace rt {

// leaf ref: LEAF | (count-1) << 28 | first index into triIdx; with directLeaves (one triangle
// per leaf): LEAF | triangle number
constexpr uint32_t ACCEL_LEAF_BIT = 0x80000000u;

struct AccelNode {          // 64 bytes: both children's (grown) boxes and their refs
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    uint32_t left, right;
    uint32_t pad0, pad1;
};
static_assert(sizeof(AccelNode) == 64, "AccelNode layout");

struct Accel {
    std::vector<AccelNode> nodes;
    std::vector<uint32_t> triIdx;   // triangle numbers (uploaded order) in leaf order
    uint32_t rootRef = 0;
    float rootMin[3] = {0, 0, 0}, rootMax[3] = {0, 0, 0};
    int maxDepth = 0;               // interior levels; bounds the traversal stack
    bool directLeaves = false;
};

// sceneMaxAbs: largest |coordinate| of the scene's bounding box.
int build_accel(const RtTriangle *tris, int64_t n, float sceneMaxAbs, int leafMax, Accel &out);

}  // namespace rt
