// Device-side data layout and kernels of the tracer (gfx950, wave64).
//
// What the reference does per pixel in one GLSL fragment shader
// (RayTracerFragmentShader.glsl, "frag") is split here into the stages of a wavefront path
// tracer that share the device functions below:
//   generate    frag:677-689   seed + camera ray
//   extend      frag:373-456   closest triangle over the BVH (hitBVH without the material draw)
//   shade       frag:457-488, 558-675   material draw, spheres, scatter, Russian roulette
//   accumulate  frag:683,697-698        running mean (+ a plain sum for multi-GPU reduction)
// A megakernel that runs the same functions in one thread per pixel is kept beside it.
//
// HBM layout (built once by rt_scene_upload from the reference's arrays):
//   nodes: one 64-byte record per INTERIOR node of the reference tree holding the boxes of
//          both children, so one fetch decides both (the reference fetches 9 floats per child
//          and tests a box only after descending to it).  Leaves (always one triangle,
//          BVH.h:207-211) have no record: the parent stores LEAF|triangle index.
//   tris:  one 48-byte record per triangle: v0 v1 v2 and the unit normal N that hitTriangle
//          (frag:257-259) recomputes for every ray (it depends on the triangle only, and is
//          computed with the same arithmetic).  Three 16-byte loads per test.
//   mats:  one 80-byte record per triangle, read once per final hit: ka kd ks (frag:220-230),
//          the shading normal (frag:296-300) and its tangent frame (frag:494-500), which depend
//          on the triangle only.
//   vnrm:  only with RtOptions.shading_normals: one 48-byte record per triangle, the vertex
//          normals n0 n1 n2 (frag:131), read once per final hit.
//   uvs:   only with RtOptions.albedo_texture: one 32-byte record per triangle, the texture
//          coordinates uv0 uv1 uv2 (frag:132), read once per final hit, and the texels of the image.
// The reference's 33-float triangle texel and 9-float node texel carry 15 resp. 0 floats the
// shader never uses (normals, uvs: frag:196-218).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"
#include "rt_math.h"

namespace rtd {

using rtm::vec3;

constexpr int TRI_RECORD = 3;                  // float4 per triangle record (v0 v1 v2 N)
constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;     // also "no triangle"

// Lanes of the wave for which p holds.  HIP's __ballot takes an int, which costs a
// v_cndmask + v_cmp round trip for every predicate that is not already a fresh compare.
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

struct SceneD {
    const float4 *nodes;    // 4 x float4 per interior node
    const float4 *tris;     // TRI_RECORD x float4 per triangle
    const float4 *mats;     // 5 x float4 per triangle
    const float4 *vnrm;     // RtOptions.shading_normals: 3 x float4 per triangle (n0 n1 n2 | pad), else nullptr
    float rootMin[3];
    float rootMax[3];
    uint32_t rootRef;       // LEAF_BIT|0 for a one-triangle scene, 0 otherwise, REF_NONE if empty
    float cullEps;          // 2^-10 * largest |coordinate| of the root box
    int32_t stackEntries;   // LDS stack slots per lane (tree height)
    int32_t refill;         // idle lanes of a wave that trigger a fetch of new rays
    int32_t leafBatch;      // lanes waiting at a leaf that make the wave run the triangle tests
    // search index (rt_accel.cpp); anodes == nullptr when absent
    const float4 *anodes;   // 4 x float4 per 4-wide node (rt_accel.h WideNode)
    uint32_t arootRef;
    int32_t aldsEntries;    // deferred-node slots per lane kept in LDS; deeper ones go to aovf
    uint32_t *aovf;         // overflow slots, [slot][thread of the grid]
    uint32_t aovfStride;
    uint32_t aovfRetrace;   // first overflow slot of the exact re-trace's stack: above every slot a deferred node can take
    float arootMin[3];
    float arootMax[3];
    float originLimit;      // rays starting farther than this from the origin skip the index
    unsigned long long *timeline;   // experiment builds (RT_EXP_TIMELINE): per wave start, end, rays, iterations
    // RtOptions.albedo_texture, else nullptr
    const float4 *uvs;      // 2 x float4 per triangle: uv0 uv1 | uv2 pad pad
    const uint8_t *texels;  // texW x texH x 3 bytes, row 0 at v = 0
    int32_t texW, texH;
};

struct FrameD {
    RtCamera cam;
    int32_t width, height;
    int32_t maxBounces;
    int32_t sphereNum;
    RtSphere sph[RT_MAX_SPHERES];
};

// Path state of the wavefront pipeline, structure of arrays over queue slots.
struct PathSoA {
    float *ox, *oy, *oz;
    float *dx, *dy, *dz;
    float *tr, *tg, *tb;    // throughput ("color" of frag:622)
    uint32_t *seed;         // wseed, frag:35
    uint32_t *pix;          // py << 16 | px
};

struct WorkCounters {       // device-side twin of RtStats
    unsigned long long rays, node_tests, tri_tests, tri_hits, retraced;
};

}  // namespace rtd
