// GPU-side construction of the search index of the extend stage (see rt_accel.cpp for what the
// index is and why any conservative index gives the reference's image).
//
// The host builder (rt_accel.cpp, binned SAH) takes 0.2 s for 1.3M triangles and 1.5 s for 10M.
// This one builds on the device in a few milliseconds: triangle boxes and 63-bit Morton codes of
// their centres, a radix sort (rt_devprims.h), agglomerative clustering of neighbours in that order (parallel
// locally-ordered clustering) into a binary tree, and a level-by-level collapse into the same
// 4-wide nodes with 8-bit quantised child boxes that the traversal kernels read (rt_accel.h
// WideNode).  The boxes are grown and quantised with the host builder's formulas, so the search
// stays conservative.  The tree is not as good as the host's top-down SAH tree (5-15 % more boxes
// per ray; a Karras radix tree over the same order, which this build replaced, had 3 % more
// again), so this is the build for callers that upload often; RtOptions.index_build selects it.
//
// The reference has no counterpart: its only tree is the median split of BVH.h:188-264, built
// on the CPU (its author's alternative split, BVH.h:225-256, is rt_bvh_build_sweep).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>

#include "rt_accel.h"
#include "rt_devprims.h"
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

__device__ __forceinline__ Box6 child_box(const uint32_t ref, const Box6 *__restrict__ prim, const uint32_t *__restrict__ ids,
                                          const Box6 *nodeBox) {
    return (ref & LEAF) ? prim[ids[ref & ~LEAF]] : nodeBox[ref];
}

__device__ __forceinline__ float half_area(const Box6 &b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return (dx < 0.0f) ? 0.0f : dx * dy + dy * dz + dz * dx;
}

__device__ __forceinline__ Box6 box_union(const Box6 &a, const Box6 &b) {
    Box6 u;
    for (int k = 0; k < 3; k++) { u.mn[k] = fminf(a.mn[k], b.mn[k]); u.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
    return u;
}

// ---- agglomerative build: parallel locally-ordered clustering (Meister and Bittner, "Parallel
// locally-ordered clustering for bounding volume hierarchy construction", 2018).  The clusters
// (at first the triangles in Morton order) each look for the neighbour within PLOC_RADIUS positions
// whose union with them has the smallest area; pairs that chose each other merge into a new
// binary node, the others stay, the array is compacted in order, and so on until one is left.

constexpr int PLOC_RADIUS = 16;
constexpr int PLOC_RADIUS_MAX = 128;

// nearest[i]: position of the neighbour of cluster i with the smallest union.  A tile of boxes is
// staged in LDS.  Ties go to the lower position, and an area that is not finite counts as the
// largest float, so the choice is a total order and some pair always chooses each other.
__global__ __launch_bounds__(BLOCK) void k_ploc_nearest(const uint32_t count, const int radius, const Box6 *__restrict__ cbox, uint32_t *__restrict__ nearest) {
    __shared__ Box6 tile[BLOCK + 2 * PLOC_RADIUS_MAX];
    const int first = (int)(blockIdx.x * BLOCK) - radius;
    for (int t = threadIdx.x; t < BLOCK + 2 * radius; t += BLOCK) {
        const int j = first + t;
        if (j >= 0 && j < (int)count) tile[t] = cbox[j];
    }
    __syncthreads();
    const int i = (int)(blockIdx.x * BLOCK + threadIdx.x);
    if (i >= (int)count) return;
    const Box6 mine = tile[threadIdx.x + radius];
    float best = INFINITY; int bestJ = -1;
    for (int d = -radius; d <= radius; d++) {
        const int j = i + d;
        if (d == 0 || j < 0 || j >= (int)count) continue;
        float a = half_area(box_union(mine, tile[threadIdx.x + radius + d]));
        if (!(a < 3.0e38f)) a = 3.0e38f;
        if (a < best) { best = a; bestJ = j; }
    }
    nearest[i] = (uint32_t)bestJ;
}

// flags[i]: low word 1 if position i survives the round (unmerged, or the lower position of a
// merging pair), high word 1 if it is the lower position of a merging pair.
__global__ __launch_bounds__(BLOCK) void k_ploc_flags(const uint32_t count, const uint32_t *__restrict__ nearest, uint64_t *__restrict__ flags) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint32_t j = nearest[i];
    const bool mutual = nearest[j] == i;
    flags[i] = (mutual && j < i) ? 0ull : ((mutual ? 1ull << 32 : 0ull) | 1ull);
}

// Writes the next round's clusters at their scanned positions; merging pairs become binary node
// nodeBase + (rank among the merges).  state: [0] clusters left, [1] binary nodes made so far.
__global__ __launch_bounds__(BLOCK) void k_ploc_merge(const uint32_t count, const uint32_t nodeBase, const uint32_t *__restrict__ nearest,
                                                      const uint64_t *__restrict__ flags, const uint64_t *__restrict__ scan,
                                                      const uint32_t *__restrict__ cl, const Box6 *__restrict__ cbox,
                                                      uint32_t *__restrict__ clOut, Box6 *__restrict__ cboxOut,
                                                      uint32_t *__restrict__ left, uint32_t *__restrict__ right, Box6 *__restrict__ nodeBox,
                                                      uint32_t *__restrict__ state) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint64_t f = flags[i], ps = scan[i];
    const uint32_t pos = (uint32_t)ps, rank = (uint32_t)(ps >> 32);
    if (f >> 32) {
        const uint32_t j = nearest[i], node = nodeBase + rank;
        const Box6 u = box_union(cbox[i], cbox[j]);
        left[node] = cl[i]; right[node] = cl[j]; nodeBox[node] = u;
        clOut[pos] = node; cboxOut[pos] = u;
    } else if (f & 1ull) {
        clOut[pos] = cl[i]; cboxOut[pos] = cbox[i];
    }
    if (i == count - 1u) {
        const uint64_t total = ps + f;
        state[0] = (uint32_t)total;
        state[1] = nodeBase + (uint32_t)(total >> 32);
    }
}

// The clusters of the first round: the triangles in sorted order.
__global__ __launch_bounds__(BLOCK) void k_ploc_init(const uint32_t n, const Box6 *__restrict__ prim, const uint32_t *__restrict__ ids,
                                                     uint32_t *__restrict__ cl, Box6 *__restrict__ cbox) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    cl[i] = LEAF | i;
    cbox[i] = prim[ids[i]];
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

__global__ void k_seed_frontier(uint2 *frontier, uint32_t *counters, const uint32_t root) {
    frontier[0] = make_uint2(root, 0u);     // binary root -> wide slot 0
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
    uint32_t *ids = nullptr, *idsSorted = nullptr, *left = nullptr, *right = nullptr, *clB = nullptr, *dState = nullptr, *counters = nullptr;
    uint2 *frontier[2] = {nullptr, nullptr};
    Box6 *cbox[2] = {nullptr, nullptr};
    uint64_t *flags = nullptr, *scan = nullptr;
    uint32_t root = 0;
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
    // stable radix sort of (Morton key, triangle) and, below, the prefix sums of the clustering rounds
    // (rt_devprims.h); one scratch buffer serves both
    tmpBytes = std::max(rtp::sort_scratch_bytes((uint32_t)N), rtp::scan_scratch_bytes<uint64_t>((uint32_t)N));
    GB_TRY(hipMalloc(&tmp, std::max<size_t>(tmpBytes, 16)));
    GB_TRY(rtp::radix_sort_pairs(keys, keysSorted, ids, idsSorted, (uint32_t)N, tmp, st));
    // after its eight passes the sorted pairs are back in the buffers they came from: those are
    // the sorted ones from here on, the others are free
    std::swap(keys, keysSorted);
    std::swap(ids, idsSorted);
    GB_TRY(hipMalloc(&left, nInt * 4)); GB_TRY(hipMalloc(&right, nInt * 4));
    GB_TRY(hipMalloc(&nodeBox, nInt * sizeof(Box6)));
    GB_TRY(hipMalloc(&counters, (MAX_LEVELS + 4) * 4));
    GB_TRY(hipMalloc(&frontier[0], nInt * sizeof(uint2))); GB_TRY(hipMalloc(&frontier[1], nInt * sizeof(uint2)));
    GB_TRY(hipMalloc(&wide, nInt * sizeof(WideNode)));
    GB_TRY(hipMemsetAsync(counters, 0, (MAX_LEVELS + 4) * 4, st));
    {
        // clusters in Morton order merge with their nearest neighbours until one is left; its node is
        // the last one made, binary node N - 2
        GB_TRY(hipMalloc(&clB, (size_t)N * 4));
        uint32_t *cl[2] = {ids, clB};          // the sort's input ids are free once it has run
        GB_TRY(hipMalloc(&cbox[0], (size_t)N * sizeof(Box6))); GB_TRY(hipMalloc(&cbox[1], (size_t)N * sizeof(Box6)));
        GB_TRY(hipMalloc(&flags, (size_t)N * 8)); GB_TRY(hipMalloc(&scan, (size_t)N * 8));
        GB_TRY(hipMalloc(&dState, 16));         // clusters left, nodes made
        hipLaunchKernelGGL(k_ploc_init, dim3(blocksN), dim3(BLOCK), 0, st, (uint32_t)N, prim, idsSorted, cl[0], cbox[0]);
        uint32_t count = (uint32_t)N, made = 0;
        int radius = PLOC_RADIUS;
        if (const char *env = getenv("RT_AMD_PLOC_RADIUS")) { const int v = atoi(env); if (v >= 1 && v <= PLOC_RADIUS_MAX) radius = v; }
        int cur = 0;
        for (int round = 0; count > 1u; round++) {
            if (round >= 4096) { err = "GPU index build: clustering did not finish"; rc = RT_E_INVALID; goto done; }
            const int blocks = (int)((count + BLOCK - 1) / BLOCK);
            hipLaunchKernelGGL(k_ploc_nearest, dim3(blocks), dim3(BLOCK), 0, st, count, radius, cbox[cur], (uint32_t *)keys);     // keys is free after the sort: the neighbour choices
            hipLaunchKernelGGL(k_ploc_flags, dim3(blocks), dim3(BLOCK), 0, st, count, (const uint32_t *)keys, flags);
            GB_TRY(rtp::exclusive_sum<uint64_t>(flags, scan, count, tmp, st));
            hipLaunchKernelGGL(k_ploc_merge, dim3(blocks), dim3(BLOCK), 0, st, count, made, (const uint32_t *)keys, flags, scan, cl[cur], cbox[cur],
                               cl[cur ^ 1], cbox[cur ^ 1], left, right, nodeBox, dState);
            uint32_t state[2] = {0, 0};
            GB_TRY(hipMemcpyAsync(state, dState, sizeof state, hipMemcpyDeviceToHost, st));
            GB_TRY(hipStreamSynchronize(st));
            if (state[0] >= count || state[0] == 0u || state[1] - made != count - state[0]) {
                err = "GPU index build: clustering made no progress"; rc = RT_E_INVALID; goto done;
            }
            count = state[0]; made = state[1];
            cur ^= 1;
        }
        if (made != (uint32_t)(N - 1)) { err = "GPU index build: clustering left nodes over"; rc = RT_E_INVALID; goto done; }
        root = (uint32_t)(N - 2);
    }
    hipLaunchKernelGGL(k_seed_frontier, dim3(1), dim3(1), 0, st, frontier[0], counters, root);
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
        GB_TRY(hipMemcpyAsync(&rb, nodeBox + root, sizeof rb, hipMemcpyDeviceToHost, st));
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
    hipFree(left); hipFree(right); hipFree(clB); hipFree(dState); hipFree(counters);
    hipFree(frontier[0]); hipFree(frontier[1]); hipFree(tmp); hipFree(wide);
    hipFree(cbox[0]); hipFree(cbox[1]); hipFree(flags); hipFree(scan);
    return rc;
}

// The two device primitives of the build on caller data, for tests: sorts n (key, value) pairs
// in place (stable), or replaces n values by their exclusive prefix sums.  Host arrays.
int selftest_device_sort(uint64_t *keys, uint32_t *vals, int64_t n, hipStream_t st, std::string &err) {
    if (n < 0 || n >= ((int64_t)1 << 28)) { err = "device sort: n out of range"; return RT_E_INVALID; }
    if (n == 0) return RT_OK;
    int rc = RT_OK;
    uint64_t *dk = nullptr, *dk2 = nullptr; uint32_t *dv = nullptr, *dv2 = nullptr; void *tmp = nullptr;
    GB_TRY(hipMalloc(&dk, (size_t)n * 8)); GB_TRY(hipMalloc(&dk2, (size_t)n * 8));
    GB_TRY(hipMalloc(&dv, (size_t)n * 4)); GB_TRY(hipMalloc(&dv2, (size_t)n * 4));
    GB_TRY(hipMalloc(&tmp, rtp::sort_scratch_bytes((uint32_t)n)));
    GB_TRY(hipMemcpyAsync(dk, keys, (size_t)n * 8, hipMemcpyHostToDevice, st));
    GB_TRY(hipMemcpyAsync(dv, vals, (size_t)n * 4, hipMemcpyHostToDevice, st));
    GB_TRY(rtp::radix_sort_pairs(dk, dk2, dv, dv2, (uint32_t)n, tmp, st));
    GB_TRY(hipMemcpyAsync(keys, dk, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    GB_TRY(hipMemcpyAsync(vals, dv, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    GB_TRY(hipStreamSynchronize(st));
done:
    hipFree(dk); hipFree(dk2); hipFree(dv); hipFree(dv2); hipFree(tmp);
    return rc;
}

int selftest_device_scan(uint64_t *values, int64_t n, hipStream_t st, std::string &err) {
    if (n < 0 || n >= ((int64_t)1 << 30)) { err = "device scan: n out of range"; return RT_E_INVALID; }
    if (n == 0) return RT_OK;
    int rc = RT_OK;
    uint64_t *d = nullptr; void *tmp = nullptr;
    GB_TRY(hipMalloc(&d, (size_t)n * 8));
    GB_TRY(hipMalloc(&tmp, rtp::scan_scratch_bytes<uint64_t>((uint32_t)n)));
    GB_TRY(hipMemcpyAsync(d, values, (size_t)n * 8, hipMemcpyHostToDevice, st));
    GB_TRY(rtp::exclusive_sum<uint64_t>(d, d, (uint32_t)n, tmp, st));
    GB_TRY(hipMemcpyAsync(values, d, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    GB_TRY(hipStreamSynchronize(st));
done:
    hipFree(d); hipFree(tmp);
    return rc;
}

}  // namespace rt
