// GPU-side build of the search index (see rt_gpubuild.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "rt_accel.h"

namespace rt {

struct GpuIndex {
    WideNode *wide = nullptr;       // device memory, owned by the caller after a successful build (hipFree)
    uint32_t wideCount = 0;
    uint32_t rootRef = 0;           // 0 (wide node 0), or LEAF | 0 for a one-triangle scene
    int wideDepth = 0;              // levels of wide nodes
    float rootMin[3] = {0, 0, 0}, rootMax[3] = {0, 0, 0};
};

// dTris: the device triangle records (3 x float4 each, vertices in the first nine floats).
// sceneMin/sceneMax: box of all vertices (normalises the Morton codes); sceneMaxAbs: its largest
// |coordinate| (sizes the growth of the boxes, as in build_accel).  Synchronises `st`.
int build_index_gpu(const float4 *dTris, int64_t n, const float sceneMin[3], const float sceneMax[3], float sceneMaxAbs,
                    hipStream_t st, GpuIndex &out, std::string &err);

// Tests: the build's radix sort (stable, in place) and exclusive prefix sum on host arrays.
int selftest_device_sort(uint64_t *keys, uint32_t *vals, int64_t n, hipStream_t st, std::string &err);
int selftest_device_scan(uint64_t *values, int64_t n, hipStream_t st, std::string &err);

}  // namespace rt
