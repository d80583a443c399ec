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

// 4-wide node of the index as the GPU reads it (64 bytes).  The boxes of up to four children are
// stored as 8-bit offsets from the node's own box corner, on a per-axis power-of-two grid, and
// are rounded outwards, so the quantised boxes contain the (already grown) child boxes.
//   decode: lo = origin[k] + qlo * scale_k, hi = origin[k] + qhi * scale_k (scale_k a power of two)
// qlo[k]/qhi[k] hold one byte per child (child c in bits 8c..8c+7); an empty slot has
// ref == 0xFFFFFFFF and an inverted box (qlo 255, qhi 0).
struct WideNode {
    float origin[3];
    float scaleX;           // the powers of two 2^exp_k as floats: scaleX here, scaleY and scaleZ at the end
    uint32_t ref[4];        // interior: index into the wide node array; leaf: LEAF | triangle number
    uint32_t qlo[3];
    uint32_t qhi[3];
    float scaleY, scaleZ;
};
static_assert(sizeof(WideNode) == 64, "WideNode layout");

// log2 of the growth of the index's triangle boxes relative to the scene scale (times a shape
// factor): INDEX_EPS_LOG2, or the experiment override RT_AMD_INDEX_EPS_LOG2.
constexpr int INDEX_EPS_LOG2 = -18;
int index_eps_log2();

struct Accel {
    std::vector<AccelNode> nodes;
    std::vector<uint32_t> triIdx;   // triangle numbers (uploaded order) in leaf order
    uint32_t rootRef = 0;
    float rootMin[3] = {0, 0, 0}, rootMax[3] = {0, 0, 0};
    int maxDepth = 0;               // interior levels; bounds the traversal stack
    bool directLeaves = false;
    std::vector<WideNode> wide;     // 4-wide collapse of `nodes` (only with directLeaves)
    uint32_t wideRootRef = 0;
    int wideDepth = 0;              // levels of wide interior nodes
    int wideStack = 0;              // bound on deferred entries during a traversal
};

// sceneMaxAbs: largest |coordinate| of the scene's bounding box.
int build_accel(const RtTriangle *tris, int64_t n, float sceneMaxAbs, int leafMax, Accel &out);

}  // namespace rt
