// GPU-side construction of the search index of the extend stage (see rt_accel.cpp for what the
// index is and why any conservative index gives the reference's image).
//
// The host builder (rt_accel.cpp, binned SAH) takes 0.8 s for 1.3M triangles and 4.5 s for 10M.
// This one builds a linear BVH on the device in a few milliseconds: triangle boxes and 63-bit
// Morton codes of their centres, a radix sort, the binary radix tree of Karras (one thread per
// internal node), a bottom-up refit, and a level-by-level collapse into the same 4-wide nodes with
// 8-bit quantised child boxes that the traversal kernels read (rt_accel.h WideNode).  The boxes
// are grown and quantised with the host builder's formulas, so the search stays conservative.
// A Morton-order tree is not as good as a SAH tree (more boxes per ray), so this is the build
// for callers that upload often; RtOptions.index_build selects it.
//
// The reference has no counterpart: its only tree is the median split of BVH.h:188-264, built
// on the CPU (its author's alternative split, BVH.h:225-256, is rt_bvh_build_sweep).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>

#include "rt_accel.h"
#include "rt_gpubuild.h"

namespace rt {

namespace {

constexpr uint32_t LEAF = ACCEL_LEAF_BIT;
constexpr int BLOCK = 256;
constexpr int MAX_LEVELS = 96;

struct Box6 { float mn[3], mx[3]; };

__device__ __forceinline__ uint64_t spread21(uint32_t v) {     // 21 bits -> every third bit of 63
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// Grown box of every triangle (the host builder's formula: eps = 2^INDEX_EPS_LOG2 * scene scale * shape
// factor |A||B| / |A x B| clamped to 1..4096), and the Morton code of its centre.
__global__ __launch_bounds__(BLOCK) void k_prims(const float4 *__restrict__ tris, const uint32_t n, const double epsBase,
                                                 const float3 smin, const float3 sinv, Box6 *__restrict__ boxes,
                                                 uint64_t *__restrict__ keys, uint32_t *__restrict__ ids) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 q0 = tris[(size_t)i * 3], q1 = tris[(size_t)i * 3 + 1], q2 = tris[(size_t)i * 3 + 2];
    const float v[3][3] = {{q0.x, q0.y, q0.z}, {q0.w, q1.x, q1.y}, {q1.z, q1.w, q2.x}};
    double A[3], B[3];
    for (int k = 0; k < 3; k++) { A[k] = (double)v[1][k] - v[0][k]; B[k] = (double)v[2][k] - v[0][k]; }
    const double cx = A[1] * B[2] - A[2] * B[1], cy = A[2] * B[0] - A[0] * B[2], cz = A[0] * B[1] - A[1] * B[0];
    const double cr = sqrt(cx * cx + cy * cy + cz * cz);
    const double ab = sqrt((A[0] * A[0] + A[1] * A[1] + A[2] * A[2]) * (B[0] * B[0] + B[1] * B[1] + B[2] * B[2]));
    double f = (cr > 0.0 && isfinite(ab / cr)) ? ab / cr : 1.0;
    f = fmin(fmax(f, 1.0), 4096.0);
    const float eps = (float)(epsBase * f);
    Box6 b;
    float c[3];
    for (int k = 0; k < 3; k++) {
        const float lo = fminf(fminf(v[0][k], v[1][k]), v[2][k]);
        const float hi = fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]);
        b.mn[k] = lo - eps; b.mx[k] = hi + eps;
        c[k] = 0.5f * lo + 0.5f * hi;
    }
    boxes[i] = b;
    const float s[3] = {smin.x, smin.y, smin.z}, inv[3] = {sinv.x, sinv.y, sinv.z};
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        float t = (c[k] - s[k]) * inv[k];           // 0..1 over the scene box; NaN and out of range clamp
        t = (t > 0.0f) ? ((t < 1.0f) ? t : 1.0f) : 0.0f;
        const uint32_t u = (uint32_t)(t * 2097151.0f);
        q[k] = u > 2097151u ? 2097151u : u;
    }
    // Two leading bits order the triangles by size class first (a quarter, a sixteenth, a sixty-fourth
    // of the scene's extent, smaller): a few large triangles in Morton order among small ones - the
    // walls of a room around a dense mesh - would give every node above them a room-sized box; this
    // way they get subtrees of their own below the root.
    float frac = 0.0f;
    for (int k = 0; k < 3; k++) frac = fmaxf(frac, (b.mx[k] - b.mn[k]) * inv[k]);
    const uint64_t cls = (frac >= 0.25f) ? 0ull : ((frac >= 0.0625f) ? 1ull : ((frac >= 0.015625f) ? 2ull : 3ull));
    keys[i] = cls << 62 | (spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2])) >> 1;
    ids[i] = i;
}

// Common leading bits of the keys at sorted positions i and j (equal keys: told apart by position).
__device__ __forceinline__ int delta(const uint64_t *__restrict__ keys, const int n, const int i, const int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// Karras, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees" (2012):
// internal node i of the binary radix tree over the sorted keys.  Child refs: LEAF | sorted
// position, or an internal index.
__global__ __launch_bounds__(BLOCK) void k_hierarchy(const uint64_t *__restrict__ keys, const int n, uint32_t *__restrict__ left,
                                                     uint32_t *__restrict__ right, uint32_t *__restrict__ parentOfInternal,
                                                     uint32_t *__restrict__ parentOfLeaf) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t L = (lo == gamma) ? (LEAF | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t R = (hi == gamma + 1) ? (LEAF | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    left[i] = L; right[i] = R;
    if (L & LEAF) parentOfLeaf[gamma] = (uint32_t)i; else parentOfInternal[gamma] = (uint32_t)i;
    if (R & LEAF) parentOfLeaf[gamma + 1] = (uint32_t)i; else parentOfInternal[gamma + 1] = (uint32_t)i;
    if (i == 0) parentOfInternal[0] = 0xFFFFFFFFu;
}

__device__ __forceinline__ Box6 child_box(const uint32_t ref, const Box6 *__restrict__ prim, const uint32_t *__restrict__ ids,
                                          const Box6 *nodeBox) {
    return (ref & LEAF) ? prim[ids[ref & ~LEAF]] : nodeBox[ref];
}

// The same inside the refit kernel, where an internal box was written by another workgroup moments
// ago: read past this CU's vector cache (a neighbouring entry may have brought the line in earlier).
__device__ __forceinline__ Box6 child_box_fresh(const uint32_t ref, const Box6 *__restrict__ prim, const uint32_t *__restrict__ ids,
                                                const Box6 *nodeBox) {
    if (ref & LEAF) return prim[ids[ref & ~LEAF]];
    const volatile float *p = (const volatile float *)&nodeBox[ref];
    Box6 b;
    for (int k = 0; k < 3; k++) { b.mn[k] = p[k]; b.mx[k] = p[3 + k]; }
    return b;
}

// Boxes of the internal nodes, bottom up: the second thread to arrive at a node computes it.
__global__ __launch_bounds__(BLOCK) void k_refit(const int n, const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                                 const uint32_t *__restrict__ parentOfInternal, const uint32_t *__restrict__ parentOfLeaf,
                                                 const Box6 *__restrict__ prim, const uint32_t *__restrict__ ids,
                                                 Box6 *nodeBox, uint32_t *arrivals) {
    const int s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= n) return;
    uint32_t cur = parentOfLeaf[s];
    while (cur != 0xFFFFFFFFu) {
        __threadfence();
        if (atomicAdd(&arrivals[cur], 1u) == 0u) return;
        __threadfence();
        const Box6 a = child_box_fresh(left[cur], prim, ids, nodeBox), b = child_box_fresh(right[cur], prim, ids, nodeBox);
        Box6 u;
        for (int k = 0; k < 3; k++) { u.mn[k] = fminf(a.mn[k], b.mn[k]); u.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
        // written so that the other CUs see it (the reader is whoever arrives second at the parent)
        volatile float *o = (volatile float *)&nodeBox[cur];
        for (int k = 0; k < 3; k++) { o[k] = u.mn[k]; o[3 + k] = u.mx[k]; }
        cur = parentOfInternal[cur];
    }
}

__device__ __forceinline__ float half_area(const Box6 &b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return (dx < 0.0f) ? 0.0f : dx * dy + dy * dz + dz * dx;
}

// One level of the collapse: every item (binary node, wide slot) of the frontier becomes a
// 4-wide node: start from the binary node's two children and replace the largest interior one
// by its own children until there are four.  Interior children get a wide slot and join the
// next frontier.  counters: [0] wide nodes allocated, [1] deepest level reached, [2 + level] size
// of the frontier of `level`.
__global__ __launch_bounds__(BLOCK) void k_collapse(const int level, const uint2 *__restrict__ in, uint2 *__restrict__ out,
                                                    uint32_t *counters, const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                                    const Box6 *__restrict__ prim, const uint32_t *__restrict__ ids,
                                                    const Box6 *__restrict__ nodeBox, WideNode *__restrict__ wide) {
    const uint32_t count = counters[2 + level];
    if (count == 0u) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&counters[1], (uint32_t)(level + 1));
    for (uint32_t t = blockIdx.x * BLOCK + threadIdx.x; t < count; t += gridDim.x * BLOCK) {
        const uint2 item = in[t];
        uint32_t ref[4]; Box6 box[4];
        int m = 2;
        ref[0] = left[item.x]; ref[1] = right[item.x];
        box[0] = child_box(ref[0], prim, ids, nodeBox); box[1] = child_box(ref[1], prim, ids, nodeBox);
        while (m < 4) {
            int pick = -1; float area = -1.0f;
            for (int c = 0; c < m; c++)
                if (!(ref[c] & LEAF)) { const float a = half_area(box[c]); if (a > area) { area = a; pick = c; } }
            if (pick < 0) break;
            const uint32_t src = ref[pick];
            ref[pick] = left[src]; box[pick] = child_box(ref[pick], prim, ids, nodeBox);
            ref[m] = right[src]; box[m] = child_box(ref[m], prim, ids, nodeBox);
            m++;
        }
        Box6 nb = box[0];
        for (int c = 1; c < m; c++)
            for (int k = 0; k < 3; k++) { nb.mn[k] = fminf(nb.mn[k], box[c].mn[k]); nb.mx[k] = fmaxf(nb.mx[k], box[c].mx[k]); }
        WideNode w;
        for (int k = 0; k < 3; k++) { w.qlo[k] = 0u; w.qhi[k] = 0u; }
        double scale[3];
        for (int k = 0; k < 3; k++) {
            w.origin[k] = nb.mn[k];
            const double ext = (double)nb.mx[k] - (double)nb.mn[k];
            int e = (ext > 0.0) ? (int)ceil(log2(ext / 255.0)) : -100;
            e = e < -100 ? -100 : (e > 100 ? 100 : e);
            while (255.0 * ldexp(1.0, e) < ext) e++;
            scale[k] = ldexp(1.0, e);
            (k == 0 ? w.scaleX : (k == 1 ? w.scaleY : w.scaleZ)) = (float)scale[k];
        }
        for (int c = 0; c < 4; c++) {
            if (c >= m) {
                w.ref[c] = 0xFFFFFFFFu;
                for (int k = 0; k < 3; k++) w.qlo[k] |= 255u << (8 * c);
                continue;
            }
            for (int k = 0; k < 3; k++) {
                double lo = floor(((double)box[c].mn[k] - (double)nb.mn[k]) / scale[k]);
                double hi = ceil(((double)box[c].mx[k] - (double)nb.mn[k]) / scale[k]);
                lo = fmin(fmax(lo, 0.0), 255.0); hi = fmin(fmax(hi, 0.0), 255.0);
                w.qlo[k] |= (uint32_t)lo << (8 * c);
                w.qhi[k] |= (uint32_t)hi << (8 * c);
            }
            if (ref[c] & LEAF) {
                w.ref[c] = LEAF | ids[ref[c] & ~LEAF];
            } else {
                const uint32_t slot = atomicAdd(&counters[0], 1u);
                const uint32_t pos = atomicAdd(&counters[2 + level + 1], 1u);
                out[pos] = make_uint2(ref[c], slot);
                w.ref[c] = slot;
            }
        }
        wide[item.y] = w;
    }
}

__global__ void k_seed_frontier(uint2 *frontier, uint32_t *counters) {
    frontier[0] = make_uint2(0u, 0u);       // binary root -> wide slot 0
    counters[0] = 1u; counters[1] = 0u; counters[2] = 1u;
}

#define GB_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); rc = RT_E_HIP; goto done; } \
    } while (0)

}  // namespace

int build_index_gpu(const float4 *dTris, int64_t n, const float sceneMin[3], const float sceneMax[3], float sceneMaxAbs,
                    hipStream_t st, GpuIndex &out, std::string &err) {
    out = GpuIndex();
    if (n <= 0 || n >= ((int64_t)1 << 28)) { err = "GPU index build: triangle count out of range"; return RT_E_INVALID; }
    int rc = RT_OK;
    const int N = (int)n;
    Box6 *prim = nullptr, *nodeBox = nullptr;
    uint64_t *keys = nullptr, *keysSorted = nullptr;
    uint32_t *ids = nullptr, *idsSorted = nullptr, *left = nullptr, *right = nullptr, *parI = nullptr, *parL = nullptr, *arrivals = nullptr, *counters = nullptr;
    uint2 *frontier[2] = {nullptr, nullptr};
    void *tmp = nullptr; size_t tmpBytes = 0;
    WideNode *wide = nullptr;
    const size_t nInt = (size_t)std::max(N - 1, 1);
    const int blocksN = (N + BLOCK - 1) / BLOCK;
    uint32_t hc[2] = {0, 0};
    float3 smin = make_float3(sceneMin[0], sceneMin[1], sceneMin[2]);
    float3 sinv;
    {
        const float ex = sceneMax[0] - sceneMin[0], ey = sceneMax[1] - sceneMin[1], ez = sceneMax[2] - sceneMin[2];
        sinv = make_float3(ex > 0.0f ? 1.0f / ex : 0.0f, ey > 0.0f ? 1.0f / ey : 0.0f, ez > 0.0f ? 1.0f / ez : 0.0f);
    }
    GB_TRY(hipMalloc(&prim, (size_t)N * sizeof(Box6)));
    GB_TRY(hipMalloc(&keys, (size_t)N * 8));
    GB_TRY(hipMalloc(&keysSorted, (size_t)N * 8));
    GB_TRY(hipMalloc(&ids, (size_t)N * 4));
    GB_TRY(hipMalloc(&idsSorted, (size_t)N * 4));
    hipLaunchKernelGGL(k_prims, dim3(blocksN), dim3(BLOCK), 0, st, dTris, (uint32_t)N, std::ldexp((double)sceneMaxAbs, index_eps_log2()), smin, sinv, prim, keys, ids);
    GB_TRY(hipGetLastError());
    if (N == 1) {
        // a single triangle: the root is a leaf, its box the triangle's grown box
        Box6 hb;
        GB_TRY(hipMemcpyAsync(&hb, prim, sizeof hb, hipMemcpyDeviceToHost, st));
        GB_TRY(hipStreamSynchronize(st));
        for (int k = 0; k < 3; k++) { out.rootMin[k] = hb.mn[k]; out.rootMax[k] = hb.mx[k]; }
        out.rootRef = LEAF | 0u; out.wideDepth = 0; out.wideCount = 0; out.wide = nullptr;
        goto done;
    }
    GB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, keys, keysSorted, ids, idsSorted, N, 0, 64, st));
    GB_TRY(hipMalloc(&tmp, std::max<size_t>(tmpBytes, 16)));
    GB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmpBytes, keys, keysSorted, ids, idsSorted, N, 0, 64, st));
    GB_TRY(hipMalloc(&left, nInt * 4)); GB_TRY(hipMalloc(&right, nInt * 4));
    GB_TRY(hipMalloc(&parI, nInt * 4)); GB_TRY(hipMalloc(&parL, (size_t)N * 4));
    GB_TRY(hipMalloc(&arrivals, nInt * 4));
    GB_TRY(hipMalloc(&nodeBox, nInt * sizeof(Box6)));
    GB_TRY(hipMalloc(&counters, (MAX_LEVELS + 4) * 4));
    GB_TRY(hipMalloc(&frontier[0], nInt * sizeof(uint2))); GB_TRY(hipMalloc(&frontier[1], nInt * sizeof(uint2)));
    GB_TRY(hipMalloc(&wide, nInt * sizeof(WideNode)));
    GB_TRY(hipMemsetAsync(arrivals, 0, nInt * 4, st));
    GB_TRY(hipMemsetAsync(counters, 0, (MAX_LEVELS + 4) * 4, st));
    hipLaunchKernelGGL(k_hierarchy, dim3((N - 1 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, keysSorted, N, left, right, parI, parL);
    hipLaunchKernelGGL(k_refit, dim3(blocksN), dim3(BLOCK), 0, st, N, left, right, parI, parL, prim, idsSorted, nodeBox, arrivals);
    hipLaunchKernelGGL(k_seed_frontier, dim3(1), dim3(1), 0, st, frontier[0], counters);
    for (int level = 0; level < MAX_LEVELS; level++) {
        // level L holds at most 4^L items; the kernel strides over whatever it finds
        const int64_t most = std::min<int64_t>((int64_t)nInt, level < 12 ? ((int64_t)1 << (2 * level)) : (int64_t)nInt);
        const int blocks = (int)std::min<int64_t>((most + BLOCK - 1) / BLOCK, 2048);
        hipLaunchKernelGGL(k_collapse, dim3(blocks), dim3(BLOCK), 0, st, level, frontier[level & 1], frontier[(level + 1) & 1], counters,
                           left, right, prim, idsSorted, nodeBox, wide);
    }
    GB_TRY(hipGetLastError());
    {
        Box6 rb;
        uint32_t lastFrontier = 0;
        GB_TRY(hipMemcpyAsync(hc, counters, sizeof hc, hipMemcpyDeviceToHost, st));
        GB_TRY(hipMemcpyAsync(&lastFrontier, counters + 2 + MAX_LEVELS, 4, hipMemcpyDeviceToHost, st));
        GB_TRY(hipMemcpyAsync(&rb, nodeBox, sizeof rb, hipMemcpyDeviceToHost, st));
        GB_TRY(hipStreamSynchronize(st));
        if (lastFrontier != 0u) { err = "GPU index build: tree deeper than the level limit"; rc = RT_E_INVALID; goto done; }
        for (int k = 0; k < 3; k++) { out.rootMin[k] = rb.mn[k]; out.rootMax[k] = rb.mx[k]; }
    }
    out.wide = wide; wide = nullptr;
    out.wideCount = hc[0];
    out.wideDepth = (int)hc[1];
    out.rootRef = 0u;
done:
    hipFree(prim); hipFree(nodeBox); hipFree(keys); hipFree(keysSorted); hipFree(ids); hipFree(idsSorted);
    hipFree(left); hipFree(right); hipFree(parI); hipFree(parL); hipFree(arrivals); hipFree(counters);
    hipFree(frontier[0]); hipFree(frontier[1]); hipFree(tmp); hipFree(wide);
    return rc;
}

}  // namespace rt
