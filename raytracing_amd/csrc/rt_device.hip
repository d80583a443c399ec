// HIP kernels (gfx950) and the device half of the C ABI declared in include/rt_amd.h.
// Compile with -ffp-contract=off: every fused multiply-add is written out (rt_math.h).
#include <hip/hip_runtime.h>
#include <link.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <chrono>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "rt_accel.h"
#include "rt_gpubuild.h"
#include "rt_internal.h"
#include "rt_kernels.h"
#include "rt_shade.h"
#include "rt_traverse.h"

using namespace rtd;
using rtm::vec3;

namespace {

constexpr int BLOCK = 256;          // 4 waves
#ifndef RT_PATHS_BLOCK
#define RT_PATHS_BLOCK 256
#endif
constexpr int PATHS_BLOCK = RT_PATHS_BLOCK;     // threads per block of the persistent path kernel
// Ray queues are split into NSHARD independent sub-queues, each with its own fill count and
// dequeue head on its own 128-byte line: one device-scope atomic word sustains only ~90
// operations per microsecond, and a 1080p frame needs ~32k wave-granular pushes and pops per
// bounce.  Wave-sized chunk c of a producer goes to shard c % NSHARD.
constexpr int NSHARD = 16;
constexpr int CTL_STRIDE = 32;      // words between counters (128 bytes)

__device__ __forceinline__ void flush_counters(WorkCounters *wc, uint32_t rays, uint32_t nNode, uint32_t nTri, uint32_t nHit) {
    // wave-level reduction, then one atomic per wave and counter
    for (int off = 32; off > 0; off >>= 1) {
        rays += __shfl_down(rays, off);
        nNode += __shfl_down(nNode, off);
        nTri += __shfl_down(nTri, off);
        nHit += __shfl_down(nHit, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (rays) atomicAdd(&wc->rays, (unsigned long long)rays);
        if (nNode) atomicAdd(&wc->node_tests, (unsigned long long)nNode);
        if (nTri) atomicAdd(&wc->tri_tests, (unsigned long long)nTri);
        if (nHit) atomicAdd(&wc->tri_hits, (unsigned long long)nHit);
    }
}

// ------------------------------------------------------------------ megakernel

// One thread per pixel, `count` frames per launch; accumulators stay in registers.
template <bool CULL, bool COUNT, bool EXT = false>
__global__ __launch_bounds__(BLOCK) void k_megakernel(const SceneD S, const FrameD F, const int firstLoopNum,
                                                      const int count, const float *__restrict__ randOrigin,
                                                      const uint32_t *__restrict__ sobV,
                                                      float *__restrict__ sum, float *__restrict__ mean,
                                                      const int history8, WorkCounters *__restrict__ wc) {
    extern __shared__ uint32_t lds_stack[];
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    uint32_t px, py;
    const bool valid = tile_pixel(tile, lane, F.width, F.height, px, py);
    uint32_t cRays = 0, cNode = 0, cTri = 0, cHit = 0;
    if (valid) {
        const size_t o3 = ((size_t)py * (size_t)F.width + px) * 3;
        float s0 = sum[o3], s1 = sum[o3 + 1], s2 = sum[o3 + 2];
        float m0 = mean[o3], m1 = mean[o3 + 1], m2 = mean[o3 + 2];
        for (int f = 0; f < count; f++) {
            const uint32_t loopNum = (uint32_t)(firstLoopNum + f);
            Path p; float uvx, uvy;
            generate_path(F, randOrigin[f], px, py, p, uvx, uvy);
            vec3 sample = rtm::mk(0.0f, 0.0f, 0.0f);
            for (int b = 0; b < F.maxBounces; b++) {
                float t; uint32_t tri;
                closest_triangle<CULL, COUNT>(S, p.o, p.d, lds_stack + threadIdx.x, BLOCK, t, tri, cNode, cTri);
                cRays++;
                if (COUNT && tri != REF_NONE) cHit++;
                float su, sv;
                sobol_pair(sobV, loopNum, (uint32_t)b, su, sv);
                if (!shade_hit<EXT>(F, S, p, t, tri, b, su, sv, px, py, sample)) break;
            }
            // frag:697 with an fp32 history, and the plain sum
            const float n = (float)(int)loopNum;
            const float wa = 1.0f / n, wb = (float)((int)loopNum - 1) / n;
            m0 = rtm::madd(wa, sample.x, wb * m0);
            m1 = rtm::madd(wa, sample.y, wb * m1);
            m2 = rtm::madd(wa, sample.z, wb * m2);
            if (history8) { m0 = store_unorm8(m0); m1 = store_unorm8(m1); m2 = store_unorm8(m2); }
            s0 += sample.x; s1 += sample.y; s2 += sample.z;
        }
        sum[o3] = s0; sum[o3 + 1] = s1; sum[o3 + 2] = s2;
        mean[o3] = m0; mean[o3 + 1] = m1; mean[o3 + 2] = m2;
    }
    flush_counters(wc, cRays, cNode, cTri, cHit);
}

// ------------------------------------------------------------------ wavefront pipeline

// Reserves consecutive queue slots for the lanes of this wave that pass `want`.
__device__ __forceinline__ uint32_t wave_push(uint32_t *counter, bool want) {
    const unsigned long long mask = ballot64(want);
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (mask != 0ull) {
        const int leader = __ffsll((long long)mask) - 1;
        if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);      // scalar: the chunk bookkeeping stays wave uniform
    }
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ void store_path(const PathSoA &Q, uint32_t slot, const Path &p, uint32_t pix) {
    Q.ox[slot] = p.o.x; Q.oy[slot] = p.o.y; Q.oz[slot] = p.o.z;
    Q.dx[slot] = p.d.x; Q.dy[slot] = p.d.y; Q.dz[slot] = p.d.z;
    Q.tr[slot] = p.thr.x; Q.tg[slot] = p.thr.y; Q.tb[slot] = p.thr.z;
    Q.seed[slot] = p.seed; Q.pix[slot] = pix;
}

// generate: one wave per 8x8 pixel tile; writes the camera rays of one frame into queue 0
// (a tile stays together in one chunk, so a wave of the extend stage starts on a coherent
// tile) and clears the frame's sample image.
// frameSlot / bitsX / bitsY: the frame's place in a batch of frames that share the queues, and the
// layout of the pixel word px | py << bitsX | frameSlot << (bitsX + bitsY); cur is the frame's own image.
__global__ __launch_bounds__(BLOCK) void k_generate(const FrameD F, const float randOrigin, const PathSoA Q,
                                                    const uint32_t cap, uint32_t *__restrict__ qcount,
                                                    float *__restrict__ cur, const uint32_t frameSlot, const int bitsX, const int bitsY) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    uint32_t px, py;
    const bool valid = tile_pixel(tile, lane, F.width, F.height, px, py);
    Path p; float uvx, uvy;
    if (valid) {
        generate_path(F, randOrigin, px, py, p, uvx, uvy);
        const size_t o3 = ((size_t)py * (size_t)F.width + px) * 3;
        cur[o3] = 0.0f; cur[o3 + 1] = 0.0f; cur[o3 + 2] = 0.0f;
    }
    const uint32_t shard = (uint32_t)tile % NSHARD;
    const uint32_t slot = shard * cap + wave_push(qcount + shard * CTL_STRIDE, valid);
    if (valid) store_path(Q, slot, p, px | (py << bitsX) | (frameSlot << (bitsX + bitsY)));
}

// extend: persistent waves pull rays from the sharded queue until every shard is empty.  A lane
// that finishes its ray does not wait for the slowest lane of its wave: as soon as S.refill lanes
// are idle the wave reserves that many queue entries with one atomic and the idle lanes start
// on new rays while the others keep traversing (traversal lengths differ by several times
// between neighbouring rays, and a wave pays for its longest lane).

template <bool CULL, bool COUNT>
__global__ __launch_bounds__(BLOCK) void k_extend(const SceneD S, const PathSoA Q, const uint32_t cap,
                                                  const uint32_t *__restrict__ qcount, uint32_t *__restrict__ qhead,
                                                  float *__restrict__ hitT, uint32_t *__restrict__ hitTri,
                                                  WorkCounters *__restrict__ wc) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t cNode = 0, cTri = 0, cHit = 0;
    Trav<CULL, COUNT> tv;
    tv.cur = REF_NONE;
    bool have = false;
    uint32_t slot = 0;
    uint32_t k = 0, shard = wave % NSHARD;
    uint32_t n = qcount[shard * CTL_STRIDE];
    bool more = true;                                   // wave-uniform: some shard may still hold rays
    while (true) {
        const unsigned long long idle = ballot64(!have);
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        if (more && (nIdle >= (uint32_t)S.refill)) {
            uint32_t base = 0;
            const int leader = __ffsll((long long)idle) - 1;
            if ((int)lane == leader) base = atomicAdd(qhead + shard * CTL_STRIDE, nIdle);
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);      // scalar: the chunk bookkeeping stays wave uniform
            if (base >= n) {                            // this shard is drained: move on
                k++;
                if (k == NSHARD) more = false;
                else { shard = (wave + k) % NSHARD; n = qcount[shard * CTL_STRIDE]; }
                continue;
            }
            const uint32_t idx = base + (uint32_t)__popcll(idle & below);
            if (!have && idx < n) {
                slot = shard * cap + idx;
                const vec3 o = rtm::mk(Q.ox[slot], Q.oy[slot], Q.oz[slot]);
                const vec3 d = rtm::mk(Q.dx[slot], Q.dy[slot], Q.dz[slot]);
                tv.begin(S, o, d, cNode);
                have = true;
            }
        }
        if (ballot64(have) == 0ull) {
            if (!more) break;
            continue;
        }
        if (have) {
            if (tv.at_interior()) tv.node_step(S, stack, BLOCK, cNode);
            else if (tv.at_leaf()) tv.leaf_step(S, stack, BLOCK, cTri);
            if (tv.done()) {
                hitT[slot] = tv.hitMin;
                hitTri[slot] = tv.best;
                if (COUNT && tv.best != REF_NONE) cHit++;
                have = false;
            }
        }
    }
    if (COUNT) flush_counters(wc, 0, cNode, cTri, cHit);
}

// extend through the search index.  Persistent waves; each lane owns one ray at a time and is
// in one of three conditions: it can take a node step, it is waiting for a triangle pass
// (its queue is full, or its traversal is over but triangles are still queued), or it is
// finished.  Per iteration the wave runs at most one of each of three code blocks, each only
// when enough lanes need it: node steps (every iteration), a triangle pass (when S.leafBatch
// lanes wait for one or nobody can step), and the finish + refill block (when S.refill lanes
// are finished: their results are written, then they start on new rays from the queue).
// A finished ray whose result cannot be proven equal to the reference's (tie, unusable ray, or
// the reference not reaching the triangle) is traced again in the reference's order on the spot.
template <bool COUNT>
__global__ __launch_bounds__(BLOCK) void k_extend_fast(const SceneD S, const PathSoA Q, const uint32_t cap,
                                                       const uint32_t *__restrict__ qcount, uint32_t *__restrict__ qhead,
                                                       float *__restrict__ hitT, uint32_t *__restrict__ hitTri,
                                                       WorkCounters *__restrict__ wc) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
    uint32_t *ovf = S.aovf + (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t cNode = 0, cTri = 0, cHit = 0;
    FastTrav<COUNT> tv;
    tv.cur = REF_NONE; tv.npend = 0;
    bool have = false;                                  // this lane owns a ray (possibly finished, not yet written)
    uint32_t slot = 0;
    uint32_t k = 0, shard = wave % NSHARD;
    uint32_t n = qcount[shard * CTL_STRIDE];
    bool more = true;                                   // wave-uniform: some shard may still hold rays
    while (true) {
        // ---- finish + refill
        const bool fin = have && tv.finished();
        const unsigned long long freeMask = ballot64(!have || fin);
        const uint32_t nFree = (uint32_t)__popcll(freeMask);
        const unsigned long long busyMask = ballot64(have && !fin);
        if ((more && nFree >= (uint32_t)S.refill) || busyMask == 0ull) {
            bool exact = false;
            if (fin) {
                exact = tv.tie || !tv.usable(S);
                if (!exact && tv.best != REF_NONE) exact = !verify_reference_reaches(S, tv.o, tv.inv, tv.best);
            }
            if (ballot64(exact) != 0ull) {
                // Rare (a few rays per million): trace those rays again in the reference's order,
                // right here.  The wave's other lanes wait; the other waves of the CU do not.
                // The lane's global overflow column is empty at this point and serves as the stack.
                if (exact) {
                    float t; uint32_t tri;
                    closest_triangle<true, COUNT>(S, tv.o, tv.d, ovf + (size_t)S.aovfRetrace * S.aovfStride, (int)S.aovfStride, t, tri, cNode, cTri);
                    tv.hitMin = t; tv.best = tri;
                    atomicAdd(&wc->retraced, 1ull);
                }
            }
            if (fin) {
                hitT[slot] = tv.hitMin;
                hitTri[slot] = tv.best;
                if (COUNT && tv.best != REF_NONE) cHit++;
                have = false;
            }
            if (!more) {
                if (busyMask == 0ull) break;
            } else {
                uint32_t base = 0;
                const int leader = __ffsll((long long)freeMask) - 1;
                if ((int)lane == leader) base = atomicAdd(qhead + shard * CTL_STRIDE, nFree);
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);      // scalar: the chunk bookkeeping stays wave uniform
                if (base >= n) {                        // this shard is drained: move on
                    k++;
                    if (k == NSHARD) more = false;
                    else { shard = (wave + k) % NSHARD; n = qcount[shard * CTL_STRIDE]; }
                    continue;
                }
                const uint32_t idx = base + (uint32_t)__popcll(freeMask & below);
                if (!have && idx < n) {
                    slot = shard * cap + idx;
                    const vec3 o = rtm::mk(Q.ox[slot], Q.oy[slot], Q.oz[slot]);
                    const vec3 d = rtm::mk(Q.dx[slot], Q.dy[slot], Q.dz[slot]);
                    tv.begin(S, stack, BLOCK, o, d, cNode);
                    have = true;
                }
            }
        }
        // ---- node steps
        const bool step = have && tv.can_step();
        if (step) tv.node_step(S, stack, BLOCK, ovf, cNode);
        // ---- triangle pass
        const unsigned long long waiting = ballot64(have && tv.wants_test());
        const unsigned long long stepping = ballot64(have && tv.can_step());
        if (waiting != 0ull && ((int)__popcll(waiting) >= S.leafBatch || stepping == 0ull)) {
            if (have && tv.npend > 0) tv.test_step(S, stack, BLOCK, cTri);
        }
    }
    if (COUNT) flush_counters(wc, 0, cNode, cTri, cHit);
}

// shade: one thread per ray of the current queue; finished paths write their sample,
// continuing paths are compacted into the next queue.
template <bool EXT>
__global__ __launch_bounds__(BLOCK) void k_shade(const SceneD S, const FrameD F, const int bounce, const uint32_t loopNum,
                                                 const uint32_t *__restrict__ sobV, const PathSoA Qin, const uint32_t cap, const uint32_t *__restrict__ qcountIn,
                                                 const float *__restrict__ hitT, const uint32_t *__restrict__ hitTri,
                                                 const PathSoA Qout, uint32_t *__restrict__ qcountOut,
                                                 float *__restrict__ cur, const int bitsX, const int bitsY,
                                                 const float *__restrict__ sobTable, const uint32_t frameFloats) {
    const uint32_t stride = gridDim.x * BLOCK;
    const uint32_t total = NSHARD * cap;        // cap is a multiple of 64: a wave never straddles shards
    // one frame: its Sobol pair for this bounce; a batch of frames: sobTable[(frame slot * maxBounces + bounce) * 2]
    float su = 0.0f, sv = 0.0f;
    if (!sobTable) sobol_pair(sobV, loopNum, (uint32_t)bounce, su, sv);
    const uint32_t maskX = (1u << bitsX) - 1u, maskY = (1u << bitsY) - 1u;
    for (uint32_t v0 = blockIdx.x * BLOCK; v0 < total; v0 += stride) {
        const uint32_t v = v0 + threadIdx.x;
        const uint32_t shardIn = v / cap, idx = v - shardIn * cap;
        const bool live = (v < total) && (idx < qcountIn[shardIn * CTL_STRIDE]);
        bool goOn = false;
        Path p; uint32_t pix = 0;
        if (live) {
            const uint32_t slot = v;
            p.o = rtm::mk(Qin.ox[slot], Qin.oy[slot], Qin.oz[slot]);
            p.d = rtm::mk(Qin.dx[slot], Qin.dy[slot], Qin.dz[slot]);
            p.thr = rtm::mk(Qin.tr[slot], Qin.tg[slot], Qin.tb[slot]);
            p.seed = Qin.seed[slot];
            pix = Qin.pix[slot];
            const uint32_t px = pix & maskX, py = (pix >> bitsX) & maskY, fslot = pix >> (bitsX + bitsY);
            if (sobTable) { const int si = ((int)fslot * F.maxBounces + bounce) * 2; su = sobTable[si]; sv = sobTable[si + 1]; }
            vec3 sample;
            goOn = shade_hit<EXT>(F, S, p, hitT[slot], hitTri[slot], bounce, su, sv, px, py, sample);
            if (!goOn && (sample.x != 0.0f || sample.y != 0.0f || sample.z != 0.0f)) {
                const size_t o3 = (size_t)fslot * frameFloats + ((size_t)py * (size_t)F.width + px) * 3;
                cur[o3] = sample.x; cur[o3 + 1] = sample.y; cur[o3 + 2] = sample.z;
            }
        }
        const uint32_t shardOut = (v >> 6) % NSHARD;
        const uint32_t outSlot = shardOut * cap + wave_push(qcountOut + shardOut * CTL_STRIDE, goOn);
        if (goOn) store_path(Qout, outSlot, p, pix);
    }
}

// accumulate: frag:697 (fp32 history) and the plain sum.
// Thread 0 also adds the frame's ray count (the fill counts of all queues) to the counters.
__global__ __launch_bounds__(BLOCK) void k_accumulate(const float *__restrict__ cur, float *__restrict__ sum,
                                                      float *__restrict__ mean, const size_t nfloats, const int loopNum,
                                                      const uint32_t *__restrict__ qcounts, const int nqueues,
                                                      const int history8, WorkCounters *__restrict__ wc) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) {
        unsigned long long rays = 0;
        for (int q = 0; q < nqueues * NSHARD; q++) rays += qcounts[q * CTL_STRIDE];
        atomicAdd(&wc->rays, rays);
    }
    if (i >= nfloats) return;
    const float n = (float)loopNum;
    const float wa = 1.0f / n, wb = (float)(loopNum - 1) / n;
    const float c = cur[i];
    const float m = rtm::madd(wa, c, wb * mean[i]);
    mean[i] = history8 ? store_unorm8(m) : m;
    sum[i] += c;
}

// The rays of a batch of the wavefront pipeline: the fill counts of all its queues.
__global__ void k_count_queue_rays(const uint32_t *__restrict__ qcounts, const int nqueues, WorkCounters *__restrict__ wc) {
    unsigned long long rays = 0;
    for (int q = 0; q < nqueues * NSHARD; q++) rays += qcounts[q * CTL_STRIDE];
    atomicAdd(&wc->rays, rays);
}

// Every fp32 bit pattern through recip_exact / sqrt_exact / rsqrt_exact against the compiler's IEEE
// division and square root: out[0..2] = number of inputs whose results differ (NaNs compare equal); out[3]: pixel_coord.
__global__ __launch_bounds__(256) void k_selftest_exact_math(unsigned long long *__restrict__ out) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
    unsigned long long bad0 = 0, bad1 = 0, bad2 = 0;
    // all lanes of a wave run the same number of rounds (the functions vote across the wave)
    for (unsigned long long b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float r = rtm::recip_exact(x), rw = 1.0f / x;
        const float q = rtm::sqrt_exact(x), qw = __builtin_sqrtf(x);
        const float i = rtm::rsqrt_exact(x), iw = 1.0f / __builtin_sqrtf(x);
        bad0 += !((__float_as_uint(r) == __float_as_uint(rw)) || (r != r && rw != rw));
        bad1 += !((__float_as_uint(q) == __float_as_uint(qw)) || (q != q && qw != qw));
        bad2 += !((__float_as_uint(i) == __float_as_uint(iw)) || (i != i && iw != iw));
    }
    // pixel_coord for every pixel X of every width W <= 16384
    unsigned long long bad3 = 0;
    for (uint32_t W = blockIdx.x + 1; W <= 16384u; W += gridDim.x)
        for (uint32_t X = threadIdx.x; X < W; X += 256u) {
            const float g = rtm::pixel_coord(X, (float)W), w = ((float)X + 0.5f) / (float)W;
            bad3 += __float_as_uint(g) != __float_as_uint(w);
        }
    if (bad3) atomicAdd(&out[3], bad3);
    if (bad0) atomicAdd(&out[0], bad0);
    if (bad1) atomicAdd(&out[1], bad1);
    if (bad2) atomicAdd(&out[2], bad2);
}

// ------------------------------------------------------------------ persistent paths

struct BatchD {                     // the frames one launch of k_paths renders
    int32_t firstLoopNum;
    int32_t nframes;                // 1..PATHS_MAX_FRAMES, and < 2^(32 - bitsX - bitsY)
    int32_t bitsX, bitsY;           // a lane's pixel word: px | py << bitsX | frame slot << (bitsX + bitsY)
    float randOrigin[64];
};
constexpr int PATHS_MAX_FRAMES = 64;

// Sobol numbers of every (frame slot, bounce) of a batch: they do not depend on the pixel
// (frag:99-104).  table[(slot * maxBounces + bounce) * 2 + {0,1}].
__global__ void k_sobol_table(const BatchD B, const int maxBounces, const uint32_t *__restrict__ sobV, float *__restrict__ table) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B.nframes * maxBounces * 2) return;
    const int f = t / (maxBounces * 2), r = t - f * (maxBounces * 2);
    const uint32_t loopNum = (uint32_t)(B.firstLoopNum + f);
    uint32_t g = loopNum ^ (loopNum >> 1), acc = 0u;
    const uint32_t base = (uint32_t)(r >> 1) * 64u + (uint32_t)(r & 1) * 32u;
    for (uint32_t j = 0u; g != 0u; g >>= 1, j++)
        if (g & 1u) acc ^= sobV[base + j];
    table[t] = (float)acc * (1.0f / 4294967296.0f);
}

// The arguments of k_paths as the kernel-argument segment holds them (each argument at the next
// offset of its alignment, i.e. the layout of this struct).  The part of the loop that runs once in
// several iterations - shading, new camera rays, the start of the next ray - reads what it needs
// of the scene, the frame and the batch from there at the point of use (scalar loads), so that
// these values do not sit in scalar registers across the traversal and get spilled around it.
struct PathsArgs {
    SceneD S; FrameD F; BatchD B;
    const float *sob; uint32_t *workHead; float *cur; WorkCounters *wc;
};

// Whole paths in one launch: persistent waves, each lane owns one path at a time and runs it
// from the camera ray to its end (generate, extend, shade, extend, ... as frag:677-689 +
// frag:617-675 do per pixel), then takes the next (frame, pixel) item.  Several frames are in
// flight in one launch, so the only point where lanes wait for the slowest ray is the end of
// the whole batch; a launch per bounce pays that wait (a few hundred microseconds on a
// 1M-triangle scene) every time.  Samples go to cur[frame slot][pixel]; k_accumulate_batch
// folds them into the running mean in frame order.
//
// Items are numbered frame-major, pixels in 8x8 tiles; a wave reserves a chunk of consecutive
// items with one atomic and hands them to its idle lanes, the chunk size shrinking towards
// the end of the batch so that the last waves finish together.
//
// FAST: closest hit through the search index with the proof/re-trace of k_extend_fast;
// otherwise the reference's tree in the reference's order (Trav<CULL>).
template <bool FAST, bool CULL, bool COUNT, int OCC, bool EXT = false>
__global__ __launch_bounds__(PATHS_BLOCK) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_paths(const SceneD S, const FrameD F, const BatchD B,
                                                 const float *__restrict__ sob, uint32_t *__restrict__ workHead, float *__restrict__ cur,
                                                 WorkCounters *__restrict__ wc) {
    extern __shared__ uint32_t lds_dyn[];
    uint32_t *stack = lds_dyn + threadIdx.x;
    uint32_t *ovf = FAST ? S.aovf + (size_t)blockIdx.x * PATHS_BLOCK + threadIdx.x : nullptr;
    const uint32_t lane = threadIdx.x & 63;

    uint32_t cRays = 0, cNode = 0, cTri = 0, cHit = 0;
#ifdef RT_EXP_TIMELINE
    const unsigned long long tlStart = wall_clock64();
    unsigned long long tlIter = 0, tlExhausted = 0;
    uint32_t tlRetrace = 0;
    // shader-clock cycles of the wave inside the finish, node and triangle blocks (their waits for memory included)
    unsigned long long tlGo = 0, tlNode = 0, tlTest = 0, tlNodeWait = 0;
#define RT_TL_BEGIN() const unsigned long long tlT0 = (__builtin_amdgcn_sched_barrier(0), __builtin_amdgcn_s_memtime())
#define RT_TL_END(acc) do { __builtin_amdgcn_sched_barrier(0); acc += __builtin_amdgcn_s_memtime() - tlT0; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RT_TL_BEGIN() do {} while (0)
#define RT_TL_END(acc) do {} while (0)
#endif
    typename std::conditional<FAST, FastTrav<COUNT>, Trav<CULL, COUNT>>::type tv;
    tv.cur = REF_NONE;
    if constexpr (FAST) tv.npend = 0;
    bool have = false;
    unsigned long long haveMask = 0ull;                 // wave-uniform: the lanes with `have` set (changes in the finish block only)
    uint32_t pix = 0;                                   // px | py << bitsX | frame slot << (bitsX + bitsY)
    int bounce = 0;
    vec3 thr = rtm::mk(0.0f, 0.0f, 0.0f);
    uint32_t seed = 0;
    uint32_t cNext = 0, cEnd = 0;                       // wave-uniform: the wave's current chunk of items
    uint32_t cSlot = 0, cTy = 0, cTx = 0, cLane = 0;    // wave-uniform: where item cNext lies (frame slot, tile row/column, pixel of the 8x8 tile)
    // first chunk: a quarter of a wave's share of the batch, 64..256 items (a one-frame launch has
    // only ~290 items per wave; handing out 256 at once leaves nothing to balance with)
    uint32_t chunk;
    {
        const uint32_t total0 = (((uint32_t)(F.width + 7) >> 3) * ((uint32_t)(F.height + 7) >> 3) * 64u) * (uint32_t)B.nframes;
        chunk = (total0 / (4u * (gridDim.x * (PATHS_BLOCK / 64)))) & ~63u;
        chunk = chunk < 64u ? 64u : (chunk > 256u ? 256u : chunk);
    }
    bool exhausted = false;                             // wave-uniform: the item counter has passed `total`
#ifdef RT_EXP_PROFILE
    unsigned long long xIter = 0, xNode = 0, xTest = 0, xGo = 0;
#endif
    while (true) {
#ifdef RT_EXP_PROFILE
        xIter += 1ull + ((unsigned long long)__popcll(ballot64(have)) << 32);
#endif
#ifdef RT_EXP_TIMELINE
        tlIter++;
        if (exhausted && cNext >= cEnd && tlExhausted == 0) tlExhausted = wall_clock64();
#endif
        // wave-wide masks from fresh compares and scalar logic (a ballot of a derived lane mask costs two
        // vector instructions; blocks are whole waves, so every lane is here)
        bool fin;
        unsigned long long finMask;
        if constexpr (FAST) {
            fin = have && tv.finished();
            finMask = ballot64(tv.cur == REF_NONE) & ballot64(tv.npend == 0) & haveMask;
        } else {
            fin = have && tv.done();
            finMask = ballot64(tv.cur == REF_NONE) & haveMask;
        }
        const uint32_t nFin = (uint32_t)__popcll(finMask);
        const uint32_t nEmpty = 64u - (uint32_t)__popcll(haveMask);
        const unsigned long long busyMask = haveMask & ~finMask;
        const bool more = !exhausted || cNext < cEnd;
        bool go = busyMask == 0ull;
        if (more) go = go || (nFin + nEmpty >= (uint32_t)S.refill);
        else {
            // no new paths: finished rays are shaded (and their paths continued) once half of
            // the lanes that still carry a path wait for it
            const uint32_t live = 64u - nEmpty;
            const uint32_t need = (live / 2u < (uint32_t)S.refill) ? (live / 2u) : (uint32_t)S.refill;
            go = go || (nFin >= (need > 0u ? need : 1u));
        }
        if (go) {
            RT_TL_BEGIN();
#ifdef RT_EXP_PROFILE
            xGo += 1ull + ((unsigned long long)nFin << 32);
#endif
            // scene, frame and batch as the kernel-argument segment holds them, read where they are used
            // (the empty asm keeps the compiler from merging these loads with the ones at kernel entry)
            const __attribute__((address_space(4))) PathsArgs *ka = (const __attribute__((address_space(4))) PathsArgs *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            const auto &Sf = ka->S; const auto &Ff = ka->F; const auto &Bf = ka->B;
            // what only this block needs of the image and the batch is worked out here from those arguments
            // (a few scalar instructions once in several iterations) instead of living in registers across the traversal
            const uint32_t tilesX = (uint32_t)(Ff.width + 7) >> 3, tilesY = (uint32_t)(Ff.height + 7) >> 3;
            const uint32_t framePix = tilesX * tilesY * 64u;
            const uint32_t total = framePix * (uint32_t)Bf.nframes;
            const uint32_t frameFloats = (uint32_t)Ff.width * (uint32_t)Ff.height * 3u;     // a launch holds at most 2 GiB of sample images (paths_frames_cap)
            const uint32_t nWaves = gridDim.x * (PATHS_BLOCK / 64);
            const int bitsX = Bf.bitsX, bitsY = Bf.bitsY, shiftSlot = bitsX + bitsY;
            const uint32_t maskX = (1u << bitsX) - 1u, maskY = (1u << bitsY) - 1u;
            const float *sobF = ka->sob; uint32_t *workHeadF = ka->workHead; float *curF = ka->cur; WorkCounters *wcF = ka->wc;
            if constexpr (FAST) {
                bool exact = false;
                if (fin) {
                    exact = tv.tie || !tv.usable(Sf);
                    if (!exact && tv.best != REF_NONE) exact = !verify_reference_reaches(Sf, tv.o, tv.inv, tv.best);
#ifdef RT_EXP_TIMELINE
                    // experiment builds: why rays are traced again (the work counters are free when COUNT is off)
                    if (!COUNT && exact) {
                        if (tv.tie) atomicAdd(&wcF->node_tests, 1ull);
                        else if (!tv.usable(Sf)) atomicAdd(&wcF->tri_tests, 1ull);
                        else atomicAdd(&wcF->tri_hits, 1ull);
                        tlRetrace++;
                    }
#endif
                }
                if (ballot64(exact) != 0ull) {
                    if (exact) {
                        float t; uint32_t tri;
                        closest_triangle<true, COUNT>(S, tv.o, tv.d, ovf + (size_t)S.aovfRetrace * S.aovfStride, (int)S.aovfStride, t, tri, cNode, cTri);
                        tv.hitMin = t; tv.best = tri;
                        atomicAdd(&wcF->retraced, 1ull);
                    }
                }
            }
            bool startRay = false;
            vec3 nextO = rtm::mk(0.0f, 0.0f, 0.0f), nextD = rtm::mk(0.0f, 0.0f, 0.0f);
            if (fin) {
                cRays++;
                if (COUNT && tv.best != REF_NONE) cHit++;
                const uint32_t px = pix & maskX, py = (pix >> bitsX) & maskY, slot = pix >> shiftSlot;
                const int si = ((int)slot * Ff.maxBounces + bounce) * 2;
                Path p;
                p.o = tv.o; p.d = tv.d; p.thr = thr; p.seed = seed;
                vec3 sample;
                const bool goOn = shade_hit<EXT>(Ff, Sf, p, tv.hitMin, tv.best, bounce, sobF[si], sobF[si + 1], px, py, sample);
                if (goOn) {
                    bounce++;
                    thr = p.thr; seed = p.seed;
                    nextO = p.o; nextD = p.d; startRay = true;
                } else {
                    float *dst = curF + (slot * frameFloats + (py * (uint32_t)Ff.width + px) * 3u);
                    dst[0] = sample.x; dst[1] = sample.y; dst[2] = sample.z;
                    have = false;
                }
            }
            // items for the idle lanes (a second round when the wave's chunk ran out half way).  The
            // position of item cNext - frame slot, tile row, tile column, pixel within the tile - is kept
            // in scalar registers and advanced with the counter, so a lane only adds its rank to it.
            bool gotItem = false;
            uint32_t iSlot = 0, iPx = 0, iPy = 0;
            for (int round = 0; round < 2; round++) {
                const unsigned long long freeMask = ballot64(!have && !gotItem);
                if (freeMask == 0ull || (exhausted && cNext >= cEnd)) break;
                if (cNext >= cEnd) {
                    uint32_t base = 0;
                    const int leader = __ffsll((long long)freeMask) - 1;
                    if ((int)lane == leader) base = atomicAdd(workHeadF, chunk);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);      // scalar: the chunk bookkeeping stays wave uniform
                    if (base >= total) {
                        exhausted = true;
#ifdef RT_EXP_TAILPRIO
                        __builtin_amdgcn_s_setprio(RT_EXP_TAILPRIO);
#endif
                        break;
                    }
                    cNext = base;
                    cEnd = (total - base < chunk) ? total : base + chunk;
                    cSlot = base / framePix;
                    const uint32_t rem = base - cSlot * framePix, tile = rem >> 6;
                    cLane = rem & 63u;
                    cTy = tile / tilesX;
                    cTx = tile - cTy * tilesX;
                    // guided self-scheduling: about half a share of what is left, 64..256 items
                    const uint32_t share = ((total - cEnd) / (2u * nWaves)) & ~63u;
                    chunk = share < 64u ? 64u : (share > 256u ? 256u : share);
                }
                const uint32_t nFree = (uint32_t)__popcll(freeMask);
                const uint32_t left = cEnd - cNext;
                const uint32_t give = nFree < left ? nFree : left;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(freeMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)freeMask, 0u));
                if (!have && !gotItem && rank < give) {
                    gotItem = true;
                    const uint32_t l = cLane + rank;                // < 128: at most one carry into the tile column
                    uint32_t tx = cTx + (l >> 6), ty = cTy;
                    iSlot = cSlot;
                    if (tx >= tilesX) { tx = 0u; ty++; }
                    if (ty >= tilesY) { ty = 0u; iSlot++; }
                    iPx = tx * 8u + (l & 7u);
                    iPy = ty * 8u + ((l >> 3) & 7u);
                }
                cNext += give;
                cLane += give;                                      // give <= 64
                if (cLane >= 64u) {
                    cLane -= 64u;
                    if (++cTx >= tilesX) { cTx = 0u; if (++cTy >= tilesY) { cTy = 0u; cSlot++; } }
                }
                if (give == nFree) break;
            }
            if (gotItem && iPx < (uint32_t)Ff.width && iPy < (uint32_t)Ff.height) {
                Path p; float uvx, uvy;
                generate_path(Ff, Bf.randOrigin[iSlot], iPx, iPy, p, uvx, uvy);
                thr = p.thr; seed = p.seed; bounce = 0;
                pix = iPx | (iPy << bitsX) | (iSlot << shiftSlot);
                nextO = p.o; nextD = p.d; startRay = true;
                have = true;
            }
            // continued and new paths start their rays together
            if (startRay) {
                if constexpr (FAST) tv.begin(Sf, stack, PATHS_BLOCK, nextO, nextD, cNode); else tv.begin(S, nextO, nextD, cNode);
            }
            {   // the value is wave uniform; saying so keeps it in scalar registers across the loop
                const unsigned long long m = ballot64(have);
                haveMask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
            }
            RT_TL_END(tlGo);
            if (haveMask == 0ull) {
                if (exhausted && cNext >= cEnd) break;
                continue;
            }
        }
        // ---- advance the rays by one step
        if constexpr (FAST) {
            // a lane without a path keeps cur == REF_NONE and an empty queue, so none of these needs `have`
            const bool step = tv.can_step();
#ifdef RT_EXP_PROFILE
            { const unsigned long long ms = ballot64(step); if (ms) xNode += 1ull + ((unsigned long long)__popcll(ms) << 32); }
#ifdef RT_EXP_TOPK
            // node steps at the first RT_EXP_TOPK nodes of the index (reported in place of the test blocks)
            { const unsigned long long mt = ballot64(step && tv.cur < (uint32_t)(RT_EXP_TOPK)); xTest += (mt ? 1ull : 0ull) + ((unsigned long long)__popcll(mt) << 32); }
#endif
#endif
#if defined(RT_EXP_PROFILE) && defined(RT_EXP_EMPTY)
            {   // node steps in which no child was hit (reported in place of the test blocks): the node had
                // been deferred and lies beyond the closest hit by now, or the ray passes between its children
                const int npBefore = tv.npend, spBefore = tv.sp;
                if (step) tv.node_step(S, stack, PATHS_BLOCK, ovf, cNode);
                const unsigned long long me = ballot64(step && tv.npend == npBefore && (tv.sp < spBefore || (spBefore == 0 && tv.cur == REF_NONE)));
#if RT_EXP_EMPTY == 2
                // ... and of those, the steps in which a child would have been hit but for the closest hit known by now
                const unsigned long long mc = me & ballot64(tv.dbgWouldHit);
                xTest += (mc ? 1ull : 0ull) + ((unsigned long long)__popcll(mc) << 32);
#else
                xTest += (me ? 1ull : 0ull) + ((unsigned long long)__popcll(me) << 32);
#endif
            }
#elif defined(RT_EXP_TIMELINE)
            {
                // the node block in two parts: until the records have arrived, and the step proper
                RT_TL_BEGIN();
                float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
                if (step) { const float4 *rec = RT_REC(S.anodes, tv.cur, 64u); q0 = rec[0]; q1 = rec[1]; q2 = rec[2]; q3 = rec[3]; }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long tlT1 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_sched_barrier(0);
                tlNodeWait += tlT1 - tlT0;
                if (step) tv.node_process(S, q0, q1, q2, q3, stack, PATHS_BLOCK, ovf, cNode);
                RT_TL_END(tlNode);
            }
#else
            if (step) tv.node_step(S, stack, PATHS_BLOCK, ovf, cNode);
#endif
            // wants_test() and can_step() of every lane, from three compares and scalar logic
            const unsigned long long mNone = ballot64(tv.cur == REF_NONE);
            const unsigned long long mFull = ballot64(tv.npend > FastTrav<COUNT>::PEND_MAX - 4);
            const unsigned long long mPend = ballot64(tv.npend > 0);
            const unsigned long long waiting = mPend & (mNone | mFull);
            const unsigned long long stepping = ~(mNone | mFull);
            if (waiting != 0ull && ((int)__popcll(waiting) >= S.leafBatch || stepping == 0ull)) {
#if defined(RT_EXP_PROFILE) && !defined(RT_EXP_TOPK) && !defined(RT_EXP_EMPTY)
                xTest += 1ull + ((unsigned long long)__popcll(ballot64(tv.npend > 0)) << 32);
#endif
                RT_TL_BEGIN();
                if (tv.npend > 0) tv.test_step(S, stack, PATHS_BLOCK, cTri);
                RT_TL_END(tlTest);
            }
        } else {
            if (have) {
                if (tv.at_interior()) tv.node_step(S, stack, PATHS_BLOCK, cNode);
                else if (tv.at_leaf()) tv.leaf_step(S, stack, PATHS_BLOCK, cTri);
            }
        }
    }
#ifdef RT_EXP_TIMELINE
    if (S.timeline) {
        uint32_t r = cRays, rt2 = tlRetrace;
        for (int off = 32; off > 0; off >>= 1) { r += __shfl_down(r, off); rt2 += __shfl_down(rt2, off); }
        if (lane == 0) {
            unsigned long long *o = S.timeline + (size_t)(blockIdx.x * (PATHS_BLOCK / 64) + (threadIdx.x >> 6)) * 10;
            o[0] = tlStart; o[1] = wall_clock64(); o[2] = r | ((unsigned long long)rt2 << 32); o[3] = tlIter; o[4] = tlExhausted;
            o[5] = tlGo; o[6] = tlNode; o[7] = tlTest; o[8] = tlNodeWait; o[9] = 0;
        }
    }
#endif
    flush_counters(wc, cRays, cNode, cTri, cHit);
#ifdef RT_EXP_PROFILE
    // experiment builds only: wave-level block counts (low word) and active lanes (high word)
    if (lane == 0) {
        atomicAdd(&wc->node_tests, xIter & 0xFFFFFFFFull); atomicAdd(&wc->node_tests, (xIter >> 32) << 32);
        atomicAdd(&wc->tri_tests, xNode & 0xFFFFFFFFull); atomicAdd(&wc->tri_tests, (xNode >> 32) << 32);
        atomicAdd(&wc->tri_hits, xTest & 0xFFFFFFFFull); atomicAdd(&wc->tri_hits, (xTest >> 32) << 32);
        atomicAdd(&wc->retraced, xGo & 0xFFFFFFFFull); atomicAdd(&wc->retraced, (xGo >> 32) << 32);
    }
#endif
}

// frag:697 (fp32 history) and the plain sum for the `nframes` samples of a batch, in frame order.
__global__ __launch_bounds__(BLOCK) void k_accumulate_batch(const float *__restrict__ cur, float *__restrict__ sum,
                                                            float *__restrict__ mean, const size_t nfloats,
                                                            const int firstLoopNum, const int nframes, const int history8) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nfloats) return;
    float m = mean[i], s = sum[i];
    for (int f = 0; f < nframes; f++) {
        const int loopNum = firstLoopNum + f;
        const float n = (float)loopNum;
        const float wa = 1.0f / n, wb = (float)(loopNum - 1) / n;
        const float c = cur[(size_t)f * nfloats + i];
        m = rtm::madd(wa, c, wb * m);
        if (history8) m = store_unorm8(m);
        s += c;
    }
    mean[i] = m;
    sum[i] = s;
}

// Device arithmetic exposed for the primitive-parity test: out[0..4] per element =
// a/b, sqrt(|a|), fma(a,b,a), sin and cos of |b| (taken as 0 outside [0, 2*pi]).
__global__ void k_selftest(const float *__restrict__ a, const float *__restrict__ b, const long long n, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    out[5 * i + 0] = x / y;
    out[5 * i + 1] = __builtin_sqrtf(fabsf(x));
    out[5 * i + 2] = rtm::madd(x, y, x);
    float th = fabsf(y);
    if (!(th <= 6.2831855f)) th = 0.0f;     // the sampler's argument never leaves [0, 2*pi]
    float s, c;
    rtm::sincos_2pi_range(th, s, c);
    out[5 * i + 3] = s;
    out[5 * i + 4] = c;
}

// Staging kernels of rt_trace_rays_noalloc: caller rays (n*3 floats each) into the queue layout
// the extend kernels read, with the queue's control words cleared and shard 0 holding all n
// rays; and the results back into the caller's arrays (-1 for a miss).
__global__ __launch_bounds__(BLOCK) void k_stage_rays(const float *__restrict__ origins, const float *__restrict__ dirs, const uint32_t n,
                                                     const uint32_t cap, float *__restrict__ soa, uint32_t *__restrict__ ctl, const uint32_t ctlWords) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < ctlWords) ctl[i] = (i == 0u) ? n : 0u;
    if (i >= cap) return;
    for (uint32_t c = 0; c < 3u; c++) {
        soa[(size_t)c * cap + i] = (i < n) ? origins[3u * i + c] : 0.0f;
        soa[(size_t)(3u + c) * cap + i] = (i < n) ? dirs[3u * i + c] : 0.0f;
    }
}

__global__ __launch_bounds__(BLOCK) void k_unstage_hits(const float *__restrict__ hitT, const uint32_t *__restrict__ hitTri, const uint32_t n,
                                                       float *__restrict__ tOut, int32_t *__restrict__ idOut) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const bool hit = hitTri[i] != REF_NONE;
    tOut[i] = hit ? hitT[i] : -1.0f;
    idOut[i] = hit ? (int32_t)hitTri[i] : -1;
}

}  // namespace

// Working memory of rt_trace_rays_noalloc, one copy per GPU the library runs on, held until the
// process ends: the ray queue (6 words per ray), the hit records (2 words per ray) and the
// queue's control words.  Host code never uses this name as an address: rt_create looks the
// device address up once per device (hipGetSymbolAddress) and the entry point passes that
// pointer to the kernels.
constexpr uint32_t NOALLOC_CTL_WORDS = 2 * NSHARD * CTL_STRIDE;     // fill counts, then dequeue heads, of one sharded queue
constexpr size_t NOALLOC_SCRATCH_WORDS = (size_t)8 * RT_NOALLOC_MAX_RAYS + NOALLOC_CTL_WORDS;
static __device__ __align__(256) uint32_t scratch[NOALLOC_SCRATCH_WORDS];

// ====================================================================== host side

constexpr int NSLOT = 4;

struct RtContext {
    int device = 0;
    int width = 0, height = 0;
    hipStream_t stream = nullptr;
    RtOptions opt{};
    std::string err;

    // scene
    float4 *dNodes = nullptr, *dTris = nullptr, *dMats = nullptr;
    float4 *dVNrm = nullptr;            // vertex normals, only when the scene was uploaded with shading_normals set
    float4 *dUVs = nullptr;             // texture coordinates, only when the scene was uploaded with albedo_texture set
    uint8_t *dTexels = nullptr;         // rt_set_albedo_texture
    int texW = 0, texH = 0;
    float4 *dANodes = nullptr;                                  // search index: nodes
    uint32_t *dAOvf[NSLOT] = {};                               // overflow of the deferred-node stack, one per launch slot
    int accelLdsEntries = 0;    // RT_AMD_ACCEL_LDS_ENTRIES; 0 = by the path kernel's occupancy (index_lds_entries_for)
    int64_t nTris = 0, nInterior = 0, nIndexNodes = 0;
    int indexStackNeed = 0;
    bool boxesExact = true;         // the uploaded boxes are the nested min/max boxes the proof step needs
    SceneD scene{};
    bool haveScene = false;
    bool indexOnDevice = false;
    bool indexFellBack = false;     // the device build of the index failed and the host built it

    FrameD frame{};
    bool haveCamera = false;
    uint32_t sobol[RT_SOBOL_WORDS];
    uint32_t *dSobolV = nullptr;    // this context's direction numbers (setSobol, sobol.h:27-34)

    // images
    float *dSum = nullptr, *dSumOwned = nullptr, *dMean = nullptr, *dCur = nullptr;
    // wavefront state
    float *dPathMem = nullptr;      // 2 x 11 arrays of P words
    PathSoA q[2]{};
    float *dHitT = nullptr; uint32_t *dHitTri = nullptr;
    uint32_t *dQueueCtl = nullptr;  // fill counts and dequeue heads of the sharded queues
    uint32_t queueCap = 0;          // slots per shard (multiple of 64)
    // experiment (RT_AMD_WAVEFRONT_FRAMES): queues, hit records and sample images for wfFrames frames at a time
    int wfFrames = 1, wfAllocFrames = 0;
    float *wfPathMem = nullptr, *wfHitT = nullptr, *wfCur = nullptr, *wfSobTable = nullptr; uint32_t *wfHitTri = nullptr;
    PathSoA wfQ[2];
    int extendBlocksPerCU = 7;
    float *dRandOrigin = nullptr; int randCap = 0;
    // persistent-path pipeline: one sample image per frame of a batch, and the item counter
    // Launch slots, used in rotation by consecutive batches.  With RtOptions.overlap each slot has
    // its own stream, so that the next batches' path kernels fill the GPU while the last paths of
    // the previous ones finish.  Measured per wave on the 1.31M-triangle scene: once the work
    // counter runs out a wave lives another 0.23 ms (median) but a few dozen of the 7168 waves
    // another 0.9-1.5 ms (paths whose four rays each take some 75 node steps, ~2 us per step on a
    // nearly empty chip), and a one-frame launch also starts with 88 % of its items handed out in
    // the first chunks; a launch therefore ends with ~1 ms in which most of the chip idles.
    // The accumulate kernels are chained by events: the running mean depends on frame order.
    // Two slots: with four in rotation a one-frame call pattern ran 6 % slower and an eight-frame one 5 %.
    struct Slot {
        hipStream_t stream = nullptr;       // non-blocking; created on first use
        float *dCurBatch = nullptr; int curBatchFrames = 0;     // sample images allocated so far (grown on demand)
        float *dSobTable = nullptr;         // k_sobol_table output for one batch
        uint32_t *dWorkHead = nullptr;
        hipEvent_t accDone = nullptr;       // recorded after the slot's accumulate kernel
        bool used = false;
    } slot[NSLOT];
    size_t ovfBytes = 0;
    int nextSlot = 0;
    int lastSlot = -1;              // slot whose accumulate kernel was enqueued last
    // RtOptions.coalesce: frames of small rt_render_frames calls that wait for a launch of their own
    std::vector<float> heldFrames;  // their randOrigin values; LoopNum heldFirst, heldFirst + 1, ...
    int heldFirst = 0;
    hipEvent_t readEvent = nullptr; // end of the last rt_read_mean_async copy
    bool readPending = false;
    bool sumPtrGivenOut = false;    // rt_sum_device_ptr was called: the caller may order device work after render calls
    bool inputDirty = false;        // work enqueued on `stream` (reset, restore) that the slots' streams must wait for
    hipEvent_t inputEvent = nullptr;
    int nSlots = 0;             // launch slots in use (RtOptions.overlap): 0 = by the size of the batch, else 2..NSLOT
    int gridDiv = 0;            // experiment (RT_AMD_GRID_DIV): overlapped launches use 1/gridDiv of the resident grid
    int pathsOcc = 0;           // waves per SIMD the path kernel is compiled for (5, 6 or 7); 0 = by scene size
    int pathsLeafBatch = 8;     // lanes waiting for a triangle pass that trigger one in the path kernel
    WorkCounters *dCounters = nullptr;
    uint64_t frames = 0;
    int numCU = 256;

    // timing of the dominant kernel
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pendingEvents;
    std::vector<hipEvent_t> freeEvents;
    double kernelMs = 0.0; int64_t kernelLaunches = 0;
};

namespace {

// Device address of `scratch` on each GPU (every GPU holds its own copy at its own address).
constexpr int MAX_DEVICES = 64;
uint32_t *g_scratchPtr[MAX_DEVICES] = {};
std::mutex g_scratchMutex;

// Looks the address up once per device; `device` must be the current device.
hipError_t lookup_scratch(int device) {
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (g_scratchPtr[device]) return hipSuccess;
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratchPtr[device] = (uint32_t *)p;
    return hipSuccess;
}

// Paths of the HIP runtime copies mapped into this process, when there is more than one.
std::string loaded_hip_runtimes() {
    std::vector<std::string> found;
    dl_iterate_phdr([](struct dl_phdr_info *info, size_t, void *data) -> int {
        if (info->dlpi_name && std::strstr(info->dlpi_name, "libamdhip64"))
            static_cast<std::vector<std::string> *>(data)->push_back(info->dlpi_name);
        return 0;
    }, &found);
    if (found.size() < 2) return std::string();
    std::string out;
    for (const auto &f : found) out += (out.empty() ? "" : ", ") + f;
    return out;
}

int flush_pending(RtContext *ctx);      // launches the frames RtOptions.coalesce holds back (defined with rt_render_frames)

int fail(RtContext *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    rt::set_global_error(msg);
    return code;
}

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((ctx), RT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

// fill counts of queues 0..B and dequeue heads of queues 0..B-1, NSHARD each, one per 128-byte line
constexpr int QCTL_COUNTS = (RT_MAX_BOUNCES + 1) * NSHARD * CTL_STRIDE;
constexpr int QCTL_HEADS = RT_MAX_BOUNCES * NSHARD * CTL_STRIDE;
constexpr int QCTL_WORDS = QCTL_COUNTS + QCTL_HEADS;

hipEvent_t take_event(RtContext *ctx) {
    if (!ctx->freeEvents.empty()) { hipEvent_t e = ctx->freeEvents.back(); ctx->freeEvents.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void give_back_events(RtContext *ctx, hipEvent_t a, hipEvent_t b) {
    if (a) ctx->freeEvents.push_back(a);
    if (b) ctx->freeEvents.push_back(b);
}

int resolve_events(RtContext *ctx) {
    int rc = RT_OK;
    for (auto &pr : ctx->pendingEvents) {
        // a pair that a failed render call left half recorded has no time to give
        float ms = 0.0f;
        hipError_t e = hipEventSynchronize(pr.second);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, pr.first, pr.second);
        if (e == hipSuccess) { ctx->kernelMs += (double)ms; ctx->kernelLaunches++; }
        else if (e != hipErrorInvalidHandle && e != hipErrorInvalidResourceHandle && e != hipErrorNotReady)
            rc = fail(ctx, RT_E_HIP, std::string("rt_amd: kernel timing: ") + hipGetErrorString(e));
        else (void)hipGetLastError();
        ctx->freeEvents.push_back(pr.first); ctx->freeEvents.push_back(pr.second);
    }
    ctx->pendingEvents.clear();
    return rc;
}

// fn(begin, end) over [0, n) on up to rt::host_threads() host threads (the upload's per-record conversions).
template <typename Fn>
void parallel_for(int64_t n, Fn fn) {
    const int64_t minChunk = 32768;
    int threads = (int)std::min<int64_t>(rt::host_threads(), (n + minChunk - 1) / minChunk);
    if (threads <= 1) { fn((int64_t)0, n); return; }
    std::vector<std::thread> pool;
    const int64_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        const int64_t b = t * per, e = std::min(n, b + per);
        if (b >= e) break;
        pool.emplace_back([=] { fn(b, e); });
    }
    for (auto &th : pool) th.join();
}

// Waits for everything this context has in flight: its own stream and the launch slots' streams.
hipError_t sync_all_streams(RtContext *ctx) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    for (auto &sl : ctx->slot)
        if (sl.stream) { const hipError_t e2 = hipStreamSynchronize(sl.stream); if (e == hipSuccess) e = e2; }
    return e;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Converts the reference's linear tree + triangle list into the device layout; validates the
// tree shape first so that a malformed tree is rejected on the host instead of faulting on the
// GPU.
int build_device_scene(RtContext *ctx, const RtTriangle *tris, int64_t n, const RtLinearBVHNode *nodes, int64_t m,
                       std::vector<float4> &hNodes, std::vector<float4> &hTris, std::vector<float4> &hMats, SceneD &S,
                       bool &boxesExact) {
    if (n <= 0 || m != 2 * n - 1) return fail(ctx, RT_E_INVALID, "scene upload: need n >= 1 triangles and exactly 2n-1 nodes");
    if (n > (int64_t)RT_MAX_TRIANGLES) return fail(ctx, RT_E_INVALID, "scene upload: more than RT_MAX_TRIANGLES (50 000 000) triangles");
    static_assert((uint64_t)RT_MAX_TRIANGLES * 80u < ((uint64_t)1 << 32), "the largest per-triangle record (80 bytes) must stay below 4 GiB");
    std::vector<uint32_t> interiorIdx((size_t)m, REF_NONE);
    std::vector<int32_t> depth((size_t)m, 0);
    std::vector<uint8_t> referenced((size_t)m, 0);
    std::vector<uint8_t> triSeen((size_t)n, 0);
    uint32_t nInterior = 0; int maxDepth = 0;
    referenced[0] = 1;
    for (int64_t k = 0; k < m; k++) {
        const RtLinearBVHNode &nd = nodes[k];
        if (!referenced[(size_t)k]) return fail(ctx, RT_E_INVALID, "scene upload: node " + std::to_string(k) + " is not reachable in pre-order");
        if (nd.triangleNum == 1) {
            if (nd.offsetOrRc < 0 || nd.offsetOrRc >= n || triSeen[(size_t)nd.offsetOrRc])
                return fail(ctx, RT_E_INVALID, "scene upload: leaf " + std::to_string(k) + " has a bad or repeated triangle index");
            triSeen[(size_t)nd.offsetOrRc] = 1;
        } else {
            const int64_t l = k + 1, r = nd.offsetOrRc;
            if (nd.triangleNum < 2 || r <= l || r >= m || nd.axis < 0 || nd.axis > 2 || referenced[(size_t)l] || referenced[(size_t)r])
                return fail(ctx, RT_E_INVALID, "scene upload: interior node " + std::to_string(k) + " is malformed");
            referenced[(size_t)l] = referenced[(size_t)r] = 1;
            depth[(size_t)l] = depth[(size_t)r] = depth[(size_t)k] + 1;
            if (depth[(size_t)l] > maxDepth) maxDepth = depth[(size_t)l];
            interiorIdx[(size_t)k] = nInterior++;
        }
    }
    if ((int64_t)nInterior != n - 1) return fail(ctx, RT_E_INVALID, "scene upload: tree does not have n-1 interior nodes");
    // The search index proves its results with one test of the triangle's own min/max box
    // (verify_reference_reaches), which stands for every box above it only when each leaf box
    // IS that min/max box and each parent box contains its children (BVH.h:190-197 builds them
    // so).  A tree with other boxes is still rendered, in the reference's order only.
    std::atomic<bool> exact{true};
    parallel_for(m, [&](int64_t kb, int64_t ke) {
        bool ok = true;
        for (int64_t k = kb; k < ke && ok; k++) {
            const RtLinearBVHNode &nd = nodes[k];
            if (nd.triangleNum == 1) {
                const RtTriangle &t = tris[nd.offsetOrRc];
                for (int c = 0; c < 3; c++) {
                    const float lo = std::fmin(std::fmin(t.v[0][c], t.v[1][c]), t.v[2][c]);
                    const float hi = std::fmax(std::fmax(t.v[0][c], t.v[1][c]), t.v[2][c]);
                    if (!(nd.minBound[c] == lo) || !(nd.maxBound[c] == hi)) ok = false;
                }
            } else {
                const RtLinearBVHNode *ch[2] = {&nodes[k + 1], &nodes[nd.offsetOrRc]};
                for (int s = 0; s < 2; s++)
                    for (int c = 0; c < 3; c++)
                        if (!(nd.minBound[c] <= ch[s]->minBound[c]) || !(nd.maxBound[c] >= ch[s]->maxBound[c])) ok = false;
            }
        }
        if (!ok) exact = false;
    });
    boxesExact = exact.load();
    if (maxDepth > 60) return fail(ctx, RT_E_INVALID, "scene upload: tree deeper than 60 levels");
    auto ref_of = [&](int64_t k) -> uint32_t {
        return nodes[k].triangleNum == 1 ? (LEAF_BIT | (uint32_t)nodes[k].offsetOrRc) : interiorIdx[(size_t)k];
    };
    hNodes.resize((size_t)nInterior * 4);
    parallel_for(m, [&](int64_t kb, int64_t ke) {
        for (int64_t k = kb; k < ke; k++) {
            const RtLinearBVHNode &nd = nodes[k];
            if (nd.triangleNum == 1) continue;
            const RtLinearBVHNode &L = nodes[k + 1], &R = nodes[nd.offsetOrRc];
            float4 *o = &hNodes[(size_t)interiorIdx[(size_t)k] * 4];
            o[0] = make_float4(L.minBound[0], L.minBound[1], L.minBound[2], L.maxBound[0]);
            o[1] = make_float4(L.maxBound[1], L.maxBound[2], R.minBound[0], R.minBound[1]);
            o[2] = make_float4(R.minBound[2], R.maxBound[0], R.maxBound[1], R.maxBound[2]);
            uint32_t w[4] = {ref_of(k + 1), ref_of(nd.offsetOrRc), (uint32_t)nd.axis, 0u};
            std::memcpy(&o[3], w, 16);
        }
    });
    hTris.resize((size_t)n * TRI_RECORD); hMats.resize((size_t)n * 5);
    parallel_for(n, [&](int64_t ib, int64_t ie) {
    for (int64_t i = ib; i < ie; i++) {
        const RtTriangle &t = tris[i];
        const vec3 v0 = rtm::mk(t.v[0][0], t.v[0][1], t.v[0][2]);
        const vec3 v1 = rtm::mk(t.v[1][0], t.v[1][1], t.v[1][2]);
        const vec3 v2 = rtm::mk(t.v[2][0], t.v[2][1], t.v[2][2]);
        const vec3 N = rtm::normalize(rtm::cross(rtm::sub(v1, v0), rtm::sub(v2, v0)));      // frag:257-259
        hTris[(size_t)i * TRI_RECORD + 0] = make_float4(v0.x, v0.y, v0.z, v1.x);
        hTris[(size_t)i * TRI_RECORD + 1] = make_float4(v1.y, v1.z, v2.x, v2.y);
        hTris[(size_t)i * TRI_RECORD + 2] = make_float4(v2.z, N.x, N.y, N.z);
        // shading record: materials, the shading normal of getTriangleNormal (frag:296-300; the
        // operand order differs from N above) and its tangent frame (frag:494-500)
        const vec3 nS = rtm::normalize(rtm::cross(rtm::sub(v2, v0), rtm::sub(v1, v0)));
        vec3 tg, bt;
        tangent_frame(nS, tg, bt);
        hMats[(size_t)i * 5 + 0] = make_float4(t.ka[0], t.ka[1], t.ka[2], t.kd[0]);
        hMats[(size_t)i * 5 + 1] = make_float4(t.kd[1], t.kd[2], t.ks[0], t.ks[1]);
        hMats[(size_t)i * 5 + 2] = make_float4(t.ks[2], nS.x, nS.y, nS.z);
        hMats[(size_t)i * 5 + 3] = make_float4(tg.x, tg.y, tg.z, bt.x);
        hMats[(size_t)i * 5 + 4] = make_float4(bt.y, bt.z, 0.0f, 0.0f);
    }
    });
    S = SceneD{};
    float maxAbs = 0.0f;
    for (int c = 0; c < 3; c++) {
        S.rootMin[c] = nodes[0].minBound[c]; S.rootMax[c] = nodes[0].maxBound[c];
        maxAbs = std::fmax(maxAbs, std::fmax(std::fabs(S.rootMin[c]), std::fabs(S.rootMax[c])));
    }
    S.rootRef = (n == 1) ? (LEAF_BIT | 0u) : 0u;
    if (n == 1 && nodes[0].offsetOrRc != 0) return fail(ctx, RT_E_INVALID, "scene upload: single leaf must reference triangle 0");
    S.cullEps = maxAbs * (1.0f / 1024.0f);
    S.stackEntries = maxDepth + 1;
    S.refill = 32;
    S.leafBatch = 1;
    if (const char *env = getenv("RT_AMD_LEAF_BATCH")) { const int v = atoi(env); if (v >= 1 && v <= 64) S.leafBatch = v; }
    if (const char *env = getenv("RT_AMD_REFILL")) { const int v = atoi(env); if (v >= 1 && v <= 64) S.refill = v; }
    return RT_OK;
}

// Most frames one launch of the path kernel may render at this image size.
int paths_frames_cap(const RtContext *ctx) {
    int bitsX = 1, bitsY = 1;
    while ((1 << bitsX) < ctx->width) bitsX++;
    while ((1 << bitsY) < ctx->height) bitsY++;
    const size_t perFrame = (size_t)ctx->width * ctx->height * 3 * sizeof(float);
    const size_t tiles = (size_t)((ctx->width + 7) / 8) * ((ctx->height + 7) / 8);
    int g = (int)std::min<size_t>(PATHS_MAX_FRAMES, std::max<size_t>(1, ((size_t)2 << 30) / perFrame));
    g = std::min(g, (int)std::min<uint32_t>(64u, 1u << (29 - bitsX - bitsY)));      // frame slot, py and px share the lane's 32-bit pixel word (three bits to spare)
    g = (int)std::min<size_t>((size_t)g, std::max<size_t>(1, ((size_t)1 << 31) / (tiles * 64 + 1)));
    if (const char *env = getenv("RT_AMD_BATCH_FRAMES")) { const int v = atoi(env); if (v >= 1 && v < g) g = v; }
    return g;
}

// Waves per SIMD the path kernel runs at for a scene of n triangles.  More waves hide more of the
// wait for the slowest lane of each node or triangle fetch; the registers this leaves (64 at 8
// waves, 72 at 7) cost spills in the finish block only.  Measured 8 against 7 waves (6 on the demo
// scene): 1.31M triangles +1 %, 262k at 4K +1.5 %, the 32-triangle demo scene +2 %, 10.5M -1.3 %.
int paths_occupancy_for(const RtContext *ctx, int64_t n) {
    return ctx->pathsOcc ? std::min(std::max(ctx->pathsOcc, 5), 8) : (n >= ((int64_t)4 << 20) ? 7 : 8);
}
int paths_occupancy(const RtContext *ctx) { return paths_occupancy_for(ctx, ctx->nTris); }
// Deferred-node slots per lane in LDS: with the 6 queue entries, a block of 256 lanes may take 1/8
// resp. 1/7 of a CU's 160 KB (allocated in units of 1280 bytes) for 8 resp. 7 waves per SIMD.
int index_lds_entries_for(const RtContext *ctx, int64_t n) {
    if (ctx->accelLdsEntries > 0) return ctx->accelLdsEntries;
    return paths_occupancy_for(ctx, n) >= 8 ? 14 : 16;
}

size_t stack_bytes(const RtContext *ctx) { return (size_t)ctx->scene.stackEntries * BLOCK * sizeof(uint32_t); }
// triangle queue + deferred nodes of the index search (the in-place exact re-trace uses the global overflow column)
size_t fast_stack_bytes(const RtContext *ctx) { return (size_t)(ctx->scene.aldsEntries + FAST_QUEUE_ENTRIES) * BLOCK * sizeof(uint32_t); }

}  // namespace

extern "C" {

int rt_default_options(RtOptions *out) {
    if (!out) return fail(nullptr, RT_E_INVALID, "rt_default_options: null pointer");
    out->max_bounces = 4;
    out->pipeline = RT_PIPELINE_PERSISTENT;
    out->cull = 1;
    out->count_work = 0;
    out->accel = 1;
    out->history8 = 0;
    out->overlap = 1;
    out->index_build = RT_INDEX_BUILD_AUTO;
    out->shading_normals = 0;
    out->albedo_texture = 0;
    out->coalesce = 1;
    return RT_OK;
}

int rt_create(RtContext **out, int width, int height, int device) {
    if (!out) return fail(nullptr, RT_E_INVALID, "rt_create: null pointer");
    *out = nullptr;
    if (width <= 0 || height <= 0 || width > 16384 || height > 16384)
        return fail(nullptr, RT_E_INVALID, "rt_create: width/height must be in 1..16384");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        // A process that holds two copies of the HIP runtime (an application that bundles its own,
        // loaded after this library bound to the system's) shows exactly this: the second copy
        // sees no GPU.  Say so instead of "no device".
        const std::string copies = loaded_hip_runtimes();
        if (!copies.empty())
            return fail(nullptr, RT_E_HIP, "rt_create: no HIP device visible, and this process holds more than one HIP runtime (" + copies +
                        "): load the runtime the application bundles before librt_amd.so, so that both use one copy");
        return fail(nullptr, RT_E_HIP, "rt_create: no HIP device available (this library has no CPU path)");
    }
    if (device < 0 || device >= ndev) return fail(nullptr, RT_E_INVALID, "rt_create: device index out of range");
    RtContext *ctx = new (std::nothrow) RtContext;
    if (!ctx) return fail(nullptr, RT_E_NOMEM, "rt_create: out of memory");
    ctx->device = device; ctx->width = width; ctx->height = height;
    rt_default_options(&ctx->opt);
    rt_sobol_default_table(ctx->sobol);
    hipError_t e = hipSetDevice(device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) ctx->numCU = prop.multiProcessorCount;
    if (e == hipSuccess) e = lookup_scratch(device);
    const size_t img = (size_t)width * height * 3 * sizeof(float);
    // queue capacity: 8x8 tiles dealt round-robin to the shards
    const size_t tilesTotal = (size_t)((width + 7) / 8) * ((height + 7) / 8);
    ctx->queueCap = (uint32_t)(((tilesTotal + NSHARD - 1) / NSHARD) * 64);
    const size_t P = (size_t)ctx->queueCap * NSHARD;     // slots per queue
    if (e == hipSuccess) e = hipMalloc(&ctx->dSumOwned, img);
    if (e == hipSuccess) e = hipMalloc(&ctx->dMean, img);
    if (e == hipSuccess) e = hipMalloc(&ctx->dCur, img);
    if (e == hipSuccess) e = hipMalloc(&ctx->dPathMem, 2 * 11 * P * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&ctx->dHitT, P * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&ctx->dHitTri, P * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&ctx->dQueueCtl, QCTL_WORDS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&ctx->dCounters, sizeof(WorkCounters));
    if (e == hipSuccess) e = hipMalloc(&ctx->dSobolV, sizeof ctx->sobol);
    if (e == hipSuccess) e = hipMemcpy(ctx->dSobolV, ctx->sobol, sizeof ctx->sobol, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        std::string msg = std::string("rt_create: ") + hipGetErrorString(e);
        rt_destroy(ctx);
        return fail(nullptr, RT_E_HIP, msg);
    }
    ctx->dSum = ctx->dSumOwned;
    if (const char *env = getenv("RT_AMD_EXTEND_BLOCKS_PER_CU")) {
        const int v = atoi(env);
        if (v >= 1 && v <= 8) ctx->extendBlocksPerCU = v;
    }
    if (const char *env = getenv("RT_AMD_PATHS_OCC")) ctx->pathsOcc = atoi(env);
    if (const char *env = getenv("RT_AMD_WAVEFRONT_FRAMES")) { const int v = atoi(env); if (v >= 1 && v <= PATHS_MAX_FRAMES) ctx->wfFrames = v; }
    if (const char *env = getenv("RT_AMD_SLOTS")) { const int v = atoi(env); if (v >= 2 && v <= NSLOT) ctx->nSlots = v; }
    if (const char *env = getenv("RT_AMD_GRID_DIV")) { const int v = atoi(env); if (v >= 1 && v <= 8) ctx->gridDiv = v; }
    if (const char *env = getenv("RT_AMD_PATHS_LEAF_BATCH")) { const int v = atoi(env); if (v >= 1 && v <= 64) ctx->pathsLeafBatch = v; }
    if (const char *env = getenv("RT_AMD_ACCEL_LDS_ENTRIES")) {
        const int v = atoi(env);
        if (v >= 1 && v <= 64) ctx->accelLdsEntries = v;
    }
    for (int b = 0; b < 2; b++) {
        float *base = ctx->dPathMem + (size_t)b * 11 * P;
        PathSoA &q = ctx->q[b];
        q.ox = base; q.oy = base + P; q.oz = base + 2 * P;
        q.dx = base + 3 * P; q.dy = base + 4 * P; q.dz = base + 5 * P;
        q.tr = base + 6 * P; q.tg = base + 7 * P; q.tb = base + 8 * P;
        q.seed = (uint32_t *)(base + 9 * P); q.pix = (uint32_t *)(base + 10 * P);
    }
    ctx->frame.width = width; ctx->frame.height = height;
    ctx->frame.maxBounces = ctx->opt.max_bounces;
    int rc = rt_reset(ctx);
    if (rc != RT_OK) { std::string msg = ctx->err; rt_destroy(ctx); return fail(nullptr, rc, msg); }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(RtContext *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    (void)sync_all_streams(ctx);        // a failed render call may have left kernels on a slot's stream that the context's stream never waited for
    for (auto &pr : ctx->pendingEvents) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto e : ctx->freeEvents) hipEventDestroy(e);
    hipFree(ctx->dNodes); hipFree(ctx->dTris); hipFree(ctx->dMats); hipFree(ctx->dVNrm);
    hipFree(ctx->dUVs); hipFree(ctx->dTexels);
    hipFree(ctx->dANodes);
    for (auto &p : ctx->dAOvf) hipFree(p);
    hipFree(ctx->dSumOwned); hipFree(ctx->dMean); hipFree(ctx->dCur);
    hipFree(ctx->dPathMem); hipFree(ctx->dHitT); hipFree(ctx->dHitTri);
    hipFree(ctx->wfPathMem); hipFree(ctx->wfHitT); hipFree(ctx->wfHitTri); hipFree(ctx->wfCur); hipFree(ctx->wfSobTable);
    hipFree(ctx->dQueueCtl); hipFree(ctx->dCounters); hipFree(ctx->dRandOrigin); hipFree(ctx->dSobolV);
    for (auto &sl : ctx->slot) {
        if (sl.stream) { hipStreamSynchronize(sl.stream); hipStreamDestroy(sl.stream); }
        if (sl.accDone) hipEventDestroy(sl.accDone);
        hipFree(sl.dCurBatch); hipFree(sl.dWorkHead); hipFree(sl.dSobTable);
    }
    if (ctx->inputEvent) hipEventDestroy(ctx->inputEvent);
    if (ctx->readEvent) hipEventDestroy(ctx->readEvent);
    delete ctx;
}

int rt_set_stream(RtContext *ctx, void *hip_stream) {
    if (!ctx) return fail(nullptr, RT_E_INVALID, "rt_set_stream: null context");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sync_all_streams(ctx));
    ctx->stream = (hipStream_t)hip_stream;
    ctx->inputDirty = true;
    return RT_OK;
}

int rt_set_options(RtContext *ctx, const RtOptions *opt) {
    if (!ctx || !opt) return fail(ctx, RT_E_INVALID, "rt_set_options: null pointer");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    if (opt->max_bounces < 1 || opt->max_bounces > RT_MAX_BOUNCES) return fail(ctx, RT_E_INVALID, "rt_set_options: max_bounces must be in 1..8");
    if (opt->pipeline != RT_PIPELINE_WAVEFRONT && opt->pipeline != RT_PIPELINE_MEGAKERNEL && opt->pipeline != RT_PIPELINE_PERSISTENT) return fail(ctx, RT_E_INVALID, "rt_set_options: unknown pipeline");
    if (opt->index_build != RT_INDEX_BUILD_HOST && opt->index_build != RT_INDEX_BUILD_GPU && opt->index_build != RT_INDEX_BUILD_AUTO) return fail(ctx, RT_E_INVALID, "rt_set_options: unknown index_build");
    if (opt->shading_normals != 0 && ctx->haveScene && !ctx->dVNrm)
        return fail(ctx, RT_E_STATE, "rt_set_options: shading_normals needs a scene uploaded with the option already set");
    if (opt->albedo_texture != 0 && ctx->haveScene && !ctx->dUVs)
        return fail(ctx, RT_E_STATE, "rt_set_options: albedo_texture needs a scene uploaded with the option already set");
    ctx->opt = *opt;
    ctx->frame.maxBounces = opt->max_bounces;
    return RT_OK;
}

int rt_scene_upload(RtContext *ctx, const RtTriangle *tris, int64_t n, const RtLinearBVHNode *nodes, int64_t m) {
    if (!ctx || !tris || !nodes) return fail(ctx, RT_E_INVALID, "rt_scene_upload: null pointer");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    std::vector<float4> hNodes, hTris, hMats; SceneD S;
    bool boxesExact = true;
    const double tStart = now_ms();
    int rc = build_device_scene(ctx, tris, n, nodes, m, hNodes, hTris, hMats, S, boxesExact);
    if (rc != RT_OK) return rc;
    const double tConverted = now_ms();
    // search index for the extend stage (skipped when RT_AMD_NO_ACCEL is set): built here on the
    // host (binned SAH), or further down on the device (RtOptions.index_build)
    rt::Accel accel;
    bool haveAccel = false;
    const bool wantIndex = boxesExact && !getenv("RT_AMD_NO_ACCEL");
    bool gpuBuild = wantIndex && (ctx->opt.index_build == RT_INDEX_BUILD_GPU ||
                                  (ctx->opt.index_build == RT_INDEX_BUILD_AUTO && n >= RT_INDEX_AUTO_GPU_TRIANGLES));
    auto host_build = [&]() {
        const float maxAbs = S.cullEps * 1024.0f;
        if (rt::build_accel(tris, n, maxAbs, 1, accel) == RT_OK && accel.directLeaves && accel.wideStack <= 160) {
            haveAccel = true;
            if (getenv("RT_AMD_VERBOSE"))
                fprintf(stderr, "rt_amd: search index: %zu binary / %zu 4-wide nodes, depth %d / %d; reference tree height %d\n",
                        accel.nodes.size(), accel.wide.size(), accel.maxDepth, accel.wideDepth, S.stackEntries - 1);
        }
    };
    if (wantIndex && !gpuBuild) host_build();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sync_all_streams(ctx));
    hipFree(ctx->dNodes); hipFree(ctx->dTris); hipFree(ctx->dMats); hipFree(ctx->dANodes); hipFree(ctx->dVNrm); hipFree(ctx->dUVs);
    ctx->dVNrm = nullptr; ctx->dUVs = nullptr;
    for (auto &p : ctx->dAOvf) { hipFree(p); p = nullptr; }
    ctx->dNodes = ctx->dTris = ctx->dMats = ctx->dANodes = nullptr; ctx->haveScene = false;
    const size_t nodeBytes = std::max<size_t>(hNodes.size(), 4) * sizeof(float4);
    HIP_TRY(ctx, hipMalloc(&ctx->dNodes, nodeBytes));
    HIP_TRY(ctx, hipMalloc(&ctx->dTris, hTris.size() * sizeof(float4)));
    HIP_TRY(ctx, hipMalloc(&ctx->dMats, hMats.size() * sizeof(float4)));
    if (!hNodes.empty()) HIP_TRY(ctx, hipMemcpy(ctx->dNodes, hNodes.data(), hNodes.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->dTris, hTris.data(), hTris.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->dMats, hMats.data(), hMats.size() * sizeof(float4), hipMemcpyHostToDevice));
    S.nodes = ctx->dNodes; S.tris = ctx->dTris; S.mats = ctx->dMats;
    if (ctx->opt.shading_normals) {
        std::vector<float4> hVN((size_t)n * 3);
        parallel_for(n, [&](int64_t ib, int64_t ie) {
            for (int64_t i = ib; i < ie; i++) {
                const RtTriangle &t = tris[i];
                hVN[(size_t)i * 3 + 0] = make_float4(t.n[0][0], t.n[0][1], t.n[0][2], t.n[1][0]);
                hVN[(size_t)i * 3 + 1] = make_float4(t.n[1][1], t.n[1][2], t.n[2][0], t.n[2][1]);
                hVN[(size_t)i * 3 + 2] = make_float4(t.n[2][2], 0.0f, 0.0f, 0.0f);
            }
        });
        HIP_TRY(ctx, hipMalloc(&ctx->dVNrm, hVN.size() * sizeof(float4)));
        HIP_TRY(ctx, hipMemcpy(ctx->dVNrm, hVN.data(), hVN.size() * sizeof(float4), hipMemcpyHostToDevice));
    }
    if (ctx->opt.albedo_texture) {
        std::vector<float4> hUV((size_t)n * 2);
        parallel_for(n, [&](int64_t ib, int64_t ie) {
            for (int64_t i = ib; i < ie; i++) {
                const RtTriangle &t = tris[i];
                hUV[(size_t)i * 2 + 0] = make_float4(t.tex[0][0], t.tex[0][1], t.tex[1][0], t.tex[1][1]);
                hUV[(size_t)i * 2 + 1] = make_float4(t.tex[2][0], t.tex[2][1], 0.0f, 0.0f);
            }
        });
        HIP_TRY(ctx, hipMalloc(&ctx->dUVs, hUV.size() * sizeof(float4)));
        HIP_TRY(ctx, hipMemcpy(ctx->dUVs, hUV.data(), hUV.size() * sizeof(float4), hipMemcpyHostToDevice));
    }
    S.anodes = nullptr; S.arootRef = REF_NONE; S.aovf = nullptr; S.aovfStride = 0; S.aovfRetrace = 0; S.aldsEntries = 1;
    uint32_t indexNodes = 0; int indexStackNeed = 0;
    rt::GpuIndex gi;
    bool haveGpuIndex = false;
    bool indexFellBack = false;
    if (gpuBuild) {
        std::string gerr;
        int grc;
        if (getenv("RT_AMD_TEST_FAIL_GPU_INDEX")) { grc = RT_E_NOMEM; gerr = "GPU index build: failure requested by RT_AMD_TEST_FAIL_GPU_INDEX"; }   // test hook
        else grc = rt::build_index_gpu(ctx->dTris, n, S.rootMin, S.rootMax, S.cullEps * 1024.0f, ctx->stream, gi, gerr);
        if (grc == RT_OK) haveGpuIndex = true;
        else {
            // The device build did not get its working memory (about 236 bytes per triangle) or gave
            // up on the tree's shape: the host builds the index instead, whichever mode asked for the
            // device build.  A HIP error that stays is reported by the calls that follow.
            if (getenv("RT_AMD_VERBOSE")) fprintf(stderr, "rt_amd: device index build failed (%s), building on the host\n", gerr.c_str());
            (void)hipGetLastError();
            gpuBuild = false;
            indexFellBack = true;
            host_build();
        }
    }
    if (haveAccel) {
        const size_t ab = std::max<size_t>(accel.wide.size(), 1) * sizeof(rt::WideNode);
        HIP_TRY(ctx, hipMalloc(&ctx->dANodes, ab));
        if (!accel.wide.empty()) HIP_TRY(ctx, hipMemcpy(ctx->dANodes, accel.wide.data(), accel.wide.size() * sizeof(rt::WideNode), hipMemcpyHostToDevice));
        S.arootRef = accel.wide.empty() ? accel.rootRef : accel.wideRootRef;
        for (int c = 0; c < 3; c++) { S.arootMin[c] = accel.rootMin[c]; S.arootMax[c] = accel.rootMax[c]; }
        indexNodes = (uint32_t)accel.wide.size();
        indexStackNeed = std::max(accel.wideStack, 1);
    } else if (haveGpuIndex) {
        if (3 * gi.wideDepth + 1 <= 160) {
            haveAccel = true;
            ctx->dANodes = (float4 *)gi.wide;
            if (!ctx->dANodes) HIP_TRY(ctx, hipMalloc(&ctx->dANodes, sizeof(rt::WideNode)));    // one-triangle scene: never read
            S.arootRef = gi.rootRef;
            for (int c = 0; c < 3; c++) { S.arootMin[c] = gi.rootMin[c]; S.arootMax[c] = gi.rootMax[c]; }
            indexNodes = gi.wideCount;
            indexStackNeed = std::max(3 * gi.wideDepth + 1, 1);
            if (getenv("RT_AMD_VERBOSE"))
                fprintf(stderr, "rt_amd: search index built on the GPU: %u 4-wide nodes, %d levels\n", gi.wideCount, gi.wideDepth);
        } else {
            hipFree(gi.wide);       // deeper than the traversal stack allows: render in the reference's order
        }
    }
    if (haveAccel) {
        S.anodes = ctx->dANodes;
        S.originLimit = 4.0f * (S.cullEps * 1024.0f);
        // deferred-node slots: the first ones in LDS, the rest (rarely reached) in global memory
        const int need = indexStackNeed;
        S.aldsEntries = std::min(need, index_lds_entries_for(ctx, n));
        S.aovfStride = (uint32_t)ctx->numCU * 8u * BLOCK;
        // overflow column per thread: the deeper deferred nodes, or the whole stack of an exact re-trace
        S.aovfRetrace = (uint32_t)std::max(need - S.aldsEntries, 0);
        const size_t ovfEntries = (size_t)(S.aovfRetrace + std::max(S.stackEntries, 1)) * S.aovfStride;
        HIP_TRY(ctx, hipMalloc(&ctx->dAOvf[0], ovfEntries * sizeof(uint32_t)));
        ctx->ovfBytes = ovfEntries * sizeof(uint32_t);      // the other slots' columns are allocated when a slot is first used
        S.aovf = ctx->dAOvf[0];
    }
#ifdef RT_EXP_TIMELINE
    {
        static unsigned long long *tl = nullptr;
        if (!tl) HIP_TRY(ctx, hipMalloc(&tl, (size_t)ctx->numCU * 8 * 4 * 10 * sizeof(unsigned long long)));
        S.timeline = tl;
    }
#endif
    ctx->scene = S; ctx->nTris = n; ctx->nInterior = n - 1; ctx->haveScene = true;
    ctx->boxesExact = boxesExact;
    if (getenv("RT_AMD_VERBOSE"))
        fprintf(stderr, "rt_amd: upload of %lld triangles: host conversion %.1f ms, index + copies %.1f ms\n", (long long)n,
                tConverted - tStart, now_ms() - tConverted);
    ctx->nIndexNodes = haveAccel ? (int64_t)indexNodes : 0;
    ctx->indexOnDevice = haveAccel && gpuBuild;
    ctx->indexFellBack = indexFellBack;
    ctx->indexStackNeed = haveAccel ? indexStackNeed : 0;
    return RT_OK;
}

int rt_scene_upload_textures(RtContext *ctx, const float *tex_triangle, int64_t n, const float *tex_bvh, int64_t m) {
    if (!ctx || !tex_triangle || !tex_bvh) return fail(ctx, RT_E_INVALID, "rt_scene_upload_textures: null pointer");
    if (n <= 0 || m <= 0) return fail(ctx, RT_E_INVALID, "rt_scene_upload_textures: empty scene");
    std::vector<RtTriangle> tris((size_t)n); std::vector<RtLinearBVHNode> nodes((size_t)m);
    for (int64_t i = 0; i < n; i++) {           // record layout of ObjectTexture.h:86-109
        const float *r = tex_triangle + i * 33; RtTriangle &t = tris[(size_t)i];
        std::memset(&t, 0, sizeof t);
        for (int k = 0; k < 3; k++) {
            for (int c = 0; c < 3; c++) { t.v[k][c] = r[k * 8 + c]; t.n[k][c] = r[k * 8 + 3 + c]; }
            t.tex[k][0] = r[k * 8 + 6]; t.tex[k][1] = r[k * 8 + 7];
        }
        for (int c = 0; c < 3; c++) { t.ka[c] = r[24 + c]; t.kd[c] = r[27 + c]; t.ks[c] = r[30 + c]; }
    }
    for (int64_t i = 0; i < m; i++) {           // ObjectTexture.h:144-154; ints recovered as int(f + 0.5), frag:180-182
        const float *r = tex_bvh + i * 9; RtLinearBVHNode &nd = nodes[(size_t)i];
        for (int c = 0; c < 3; c++) { nd.maxBound[c] = r[c]; nd.minBound[c] = r[3 + c]; }
        nd.triangleNum = (int32_t)(r[6] + 0.5f);
        nd.offsetOrRc = (int32_t)(r[7] + 0.5f);
        nd.axis = (nd.triangleNum == 1) ? -1 : (int32_t)(r[8] + 0.5f);
    }
    return rt_scene_upload(ctx, tris.data(), n, nodes.data(), m);
}

int rt_set_albedo_texture(RtContext *ctx, const uint8_t *rgb, int width, int height) {
    if (!ctx) return fail(nullptr, RT_E_INVALID, "rt_set_albedo_texture: null context");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    if (rgb && (width < 1 || height < 1 || width > 16384 || height > 16384))
        return fail(ctx, RT_E_INVALID, "rt_set_albedo_texture: width/height must be in 1..16384");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // kernels of earlier render calls may still read the old image
    HIP_TRY(ctx, sync_all_streams(ctx));
    hipFree(ctx->dTexels); ctx->dTexels = nullptr; ctx->texW = ctx->texH = 0;
    if (!rgb) return RT_OK;
    const size_t bytes = (size_t)width * (size_t)height * 3;
    HIP_TRY(ctx, hipMalloc(&ctx->dTexels, bytes));
    HIP_TRY(ctx, hipMemcpy(ctx->dTexels, rgb, bytes, hipMemcpyHostToDevice));
    ctx->texW = width; ctx->texH = height;
    return RT_OK;
}

int rt_set_sobol(RtContext *ctx, const uint32_t v[RT_SOBOL_WORDS]) {
    if (!ctx || !v) return fail(ctx, RT_E_INVALID, "rt_set_sobol: null pointer");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    std::memcpy(ctx->sobol, v, sizeof ctx->sobol);
    // stream ordered: frames already queued keep the old table
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dSobolV, ctx->sobol, sizeof ctx->sobol, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_set_spheres(RtContext *ctx, const RtSphere *spheres, int count) {
    if (!ctx || count < 0 || count > RT_MAX_SPHERES || (count > 0 && !spheres)) return fail(ctx, RT_E_INVALID, "rt_set_spheres: count must be in 0..10");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    ctx->frame.sphereNum = count;
    for (int i = 0; i < count; i++) ctx->frame.sph[i] = spheres[i];
    return RT_OK;
}

int rt_set_camera(RtContext *ctx, const RtCamera *cam) {
    if (!ctx || !cam) return fail(ctx, RT_E_INVALID, "rt_set_camera: null pointer");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    ctx->frame.cam = *cam;
    ctx->haveCamera = true;
    return RT_OK;
}

int rt_reset(RtContext *ctx) {
    if (!ctx) return fail(nullptr, RT_E_INVALID, "rt_reset: null context");
    ctx->heldFrames.clear();        // frames not launched yet would be cleared with the rest
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t img = (size_t)ctx->width * ctx->height * 3 * sizeof(float);
    HIP_TRY(ctx, hipMemsetAsync(ctx->dSum, 0, img, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->dMean, 0, img, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->dCur, 0, img, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->dCounters, 0, sizeof(WorkCounters), ctx->stream));
    ctx->inputDirty = true;
    int rc = resolve_events(ctx);
    ctx->kernelMs = 0.0; ctx->kernelLaunches = 0;
    ctx->frames = 0;
    return rc;
}

int rt_synchronize(RtContext *ctx) {
    if (!ctx) return fail(nullptr, RT_E_INVALID, "rt_synchronize: null context");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

}  // extern "C"

namespace {

int render_now(RtContext *ctx, int first_loopnum, int count, const float *rand_origin);

// Launches the frames that earlier calls of rt_render_frames left waiting (RtOptions.coalesce).
// Every entry point that reads or changes what a frame depends on calls this first.
int flush_pending(RtContext *ctx) {
    if (ctx->heldFrames.empty()) return RT_OK;
    std::vector<float> ro;
    ro.swap(ctx->heldFrames);           // render_now may fail: the frames are gone either way, as after any failed call
    return render_now(ctx, ctx->heldFirst, (int)ro.size(), ro.data());
}

// Batches of this context that the GPU has not finished yet.
int batches_in_flight(RtContext *ctx) {
    int n = 0;
    for (auto &sl : ctx->slot)
        if (sl.used && sl.accDone && hipEventQuery(sl.accDone) == hipErrorNotReady) n++;
    (void)hipGetLastError();
    return n;
}

}  // namespace

extern "C" {

int rt_render_frames(RtContext *ctx, int first_loopnum, int count, const float *rand_origin) {
    if (!ctx || !rand_origin) return fail(ctx, RT_E_INVALID, "rt_render_frames: null pointer");
    if (count <= 0 || first_loopnum < 1 || (int64_t)first_loopnum + count > (int64_t)1 << 30) return fail(ctx, RT_E_INVALID, "rt_render_frames: need first_loopnum >= 1 and count >= 1");
    if (!ctx->haveScene) return fail(ctx, RT_E_STATE, "rt_render_frames: no scene uploaded");
    if (!ctx->haveCamera) return fail(ctx, RT_E_STATE, "rt_render_frames: no camera set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // RtOptions.coalesce: while two batches are still running, the frames of small calls wait on the
    // host and go out together.  A launch of one 1080p frame spends a fifth of its paths in waves
    // that only drain (nothing refills their lanes), a launch of eight frames a fortieth; the GPU
    // is busy either way, so the wait costs nothing.  Only where the caller cannot tell the
    // difference: not with a stream of its own, a bound sum buffer or the sum buffer's address in
    // its hands, all of which let it order device work after "the frames of the calls so far".
    const int cap = paths_frames_cap(ctx);
    const bool mayHold = ctx->opt.coalesce != 0 && ctx->opt.pipeline == RT_PIPELINE_PERSISTENT && ctx->opt.overlap != 0 &&
                         ctx->stream == nullptr && ctx->dSum == ctx->dSumOwned && !ctx->sumPtrGivenOut && count < cap;
    if (!mayHold) {
        const int rc = flush_pending(ctx);
        if (rc != RT_OK) return rc;
        return render_now(ctx, first_loopnum, count, rand_origin);
    }
    if (!ctx->heldFrames.empty() && (first_loopnum != ctx->heldFirst + (int)ctx->heldFrames.size() || (int)ctx->heldFrames.size() + count > cap)) {
        const int rc = flush_pending(ctx);
        if (rc != RT_OK) return rc;
    }
    if (ctx->heldFrames.empty()) ctx->heldFirst = first_loopnum;
    ctx->heldFrames.insert(ctx->heldFrames.end(), rand_origin, rand_origin + count);
    if ((int)ctx->heldFrames.size() >= cap || batches_in_flight(ctx) < 2) return flush_pending(ctx);
    return RT_OK;
}

}  // extern "C"

namespace {

int render_now(RtContext *ctx, int first_loopnum, int count, const float *rand_origin) {
    int rc = RT_OK;
    if (ctx->pendingEvents.size() > 8192) { rc = resolve_events(ctx); if (rc != RT_OK) return rc; }
    const FrameD &F = ctx->frame;
    SceneD S = ctx->scene;
    const bool vnrm = ctx->opt.shading_normals != 0, tex = ctx->opt.albedo_texture != 0;
    if (vnrm && !ctx->dVNrm) return fail(ctx, RT_E_STATE, "rt_render_frames: shading_normals needs a scene uploaded with the option already set");
    if (tex && !ctx->dUVs) return fail(ctx, RT_E_STATE, "rt_render_frames: albedo_texture needs a scene uploaded with the option already set");
    if (tex && !ctx->dTexels) return fail(ctx, RT_E_STATE, "rt_render_frames: albedo_texture is set but rt_set_albedo_texture gave no image");
    S.vnrm = vnrm ? ctx->dVNrm : nullptr;
    S.uvs = tex ? ctx->dUVs : nullptr;
    S.texels = tex ? ctx->dTexels : nullptr;
    S.texW = ctx->texW; S.texH = ctx->texH;
    const bool ext = vnrm || tex;       // the kernels with the shading extensions
    const int tiles = ((ctx->width + 7) / 8) * ((ctx->height + 7) / 8);
    const int tileBlocks = (tiles + BLOCK / 64 - 1) / (BLOCK / 64);
    const size_t lds = stack_bytes(ctx);
    const bool cull = ctx->opt.cull != 0, cnt = ctx->opt.count_work != 0;
    hipStream_t st = ctx->stream;

    if (ctx->opt.pipeline == RT_PIPELINE_MEGAKERNEL) {
        if (count > ctx->randCap) {
            HIP_TRY(ctx, hipStreamSynchronize(st));
            hipFree(ctx->dRandOrigin); ctx->dRandOrigin = nullptr; ctx->randCap = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->dRandOrigin, (size_t)count * sizeof(float)));
            ctx->randCap = count;
        }
        // pageable host memory: the copy has returned from the caller's buffer when this call returns
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dRandOrigin, rand_origin, (size_t)count * sizeof(float), hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        hipEvent_t e0 = take_event(ctx), e1 = take_event(ctx);
        if (!e0 || !e1) { give_back_events(ctx, e0, e1); return fail(ctx, RT_E_HIP, "rt_render_frames: hipEventCreate failed"); }
        ctx->pendingEvents.emplace_back(e0, e1);
        HIP_TRY(ctx, hipEventRecord(e0, st));
#define LAUNCH_MEGA(C, N) hipLaunchKernelGGL((k_megakernel<C, N>), dim3(tileBlocks), dim3(BLOCK), lds, st, S, F, first_loopnum, count, ctx->dRandOrigin, ctx->dSobolV, ctx->dSum, ctx->dMean, ctx->opt.history8, ctx->dCounters)
        if (ext) {     // shading extensions: the counting instantiations only
            if (cull) hipLaunchKernelGGL((k_megakernel<true, true, true>), dim3(tileBlocks), dim3(BLOCK), lds, st, S, F, first_loopnum, count, ctx->dRandOrigin, ctx->dSobolV, ctx->dSum, ctx->dMean, ctx->opt.history8, ctx->dCounters);
            else hipLaunchKernelGGL((k_megakernel<false, true, true>), dim3(tileBlocks), dim3(BLOCK), lds, st, S, F, first_loopnum, count, ctx->dRandOrigin, ctx->dSobolV, ctx->dSum, ctx->dMean, ctx->opt.history8, ctx->dCounters);
        }
        else if (cull) { if (cnt) LAUNCH_MEGA(true, true); else LAUNCH_MEGA(true, false); }
        else { if (cnt) LAUNCH_MEGA(false, true); else LAUNCH_MEGA(false, false); }
#undef LAUNCH_MEGA
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(e1, st));
        ctx->frames += (uint64_t)count;
        ctx->inputDirty = true;         // the accumulators were written on the context's stream
        return RT_OK;
    }

    if (ctx->opt.pipeline == RT_PIPELINE_PERSISTENT) {
        const size_t nfloats = (size_t)ctx->width * ctx->height * 3;
        int bitsX = 1, bitsY = 1;       // bits of the largest pixel coordinates
        while ((1 << bitsX) < ctx->width) bitsX++;
        while ((1 << bitsY) < ctx->height) bitsY++;
        // frames per launch: up to 64, within 2 GiB of sample images, 2^31 items and the slot bits of
        // the pixel word; the sample images are allocated for the largest batch seen so far
        const int cap = paths_frames_cap(ctx);
        const bool overlap = ctx->opt.overlap != 0;
        const bool fast = ctx->opt.accel != 0 && S.anodes != nullptr;
        // one block per resident slot: 4 waves per block, so waves per SIMD = blocks per CU
        const int occ = ext ? 5 : paths_occupancy(ctx);
        int blocks = ctx->numCU * (fast ? occ : 5) * (BLOCK / PATHS_BLOCK);
        if (overlap && ctx->gridDiv > 1) blocks = std::max(blocks / ctx->gridDiv, 1);
        const int accBlocks = (int)((nfloats + BLOCK - 1) / BLOCK);
        if (overlap) {
            // Everything enqueued on the context's stream before this call - by the caller (a collective
            // on the sum buffer, a memset of it) or by this library (a reset, another pipeline's frames,
            // rt_trace_rays_noalloc, which shares slot 0's overflow columns) - is marked here.  The
            // accumulate kernels of this call always wait for the mark: they are the only kernels
            // that touch memory the caller can see (the sum buffer may be the caller's).  The path
            // kernels wait for it only when the library itself put work there (inputDirty); making
            // them wait every time would serialise the batches, since the context's stream waits for
            // every batch's accumulate kernel.
            if (!ctx->inputEvent) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->inputEvent, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventRecord(ctx->inputEvent, st));
            if (ctx->inputDirty) {
                for (auto &o : ctx->slot)
                    if (o.stream) HIP_TRY(ctx, hipStreamWaitEvent(o.stream, ctx->inputEvent, 0));
            }
        }
        // batches of equal size (64 frames where a launch holds 21 are four launches of 16, not 21 + 21 + 21 + 1)
        const int nBatches = (count + cap - 1) / cap;
        const int perBatch = (count + nBatches - 1) / nBatches;
        for (int off = 0; off < count; off += perBatch) {
            BatchD B;
            B.firstLoopNum = first_loopnum + off;
            B.nframes = std::min(perBatch, count - off);
            B.bitsX = bitsX; B.bitsY = bitsY;
            for (int f = 0; f < PATHS_MAX_FRAMES; f++) B.randOrigin[f] = f < B.nframes ? rand_origin[off + f] : 0.0f;
            // Launch slots: a third one pays for small batches, whose launches are mostly tail (one
            // frame per call at 1080p +8..20 %, two +2..5 %); from about 5M work items on it costs
            // (four frames -1..-4 %, eight -6 %, one frame at 4K -1 %): the blocks of two waiting
            // kernels then share the CUs that come free, and both end later.
            const int nsl = ctx->nSlots ? ctx->nSlots : ((size_t)B.nframes * ctx->width * ctx->height <= ((size_t)5 << 20) ? 3 : 2);
            const int si = overlap ? ctx->nextSlot : 0;
            if (overlap) ctx->nextSlot = (si + 1) % nsl;
            RtContext::Slot &sl = ctx->slot[si];
            if (fast && !ctx->dAOvf[si]) HIP_TRY(ctx, hipMalloc(&ctx->dAOvf[si], ctx->ovfBytes));
            if (overlap && !sl.stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
            if (!sl.accDone) HIP_TRY(ctx, hipEventCreateWithFlags(&sl.accDone, hipEventDisableTiming));
            hipStream_t ps = overlap ? sl.stream : st;
            if (B.nframes > sl.curBatchFrames) {
                HIP_TRY(ctx, hipStreamSynchronize(ps));
                hipFree(sl.dCurBatch); sl.dCurBatch = nullptr; sl.curBatchFrames = 0;
                HIP_TRY(ctx, hipMalloc(&sl.dCurBatch, (size_t)B.nframes * nfloats * sizeof(float)));
                sl.curBatchFrames = B.nframes;
            }
            if (!sl.dWorkHead) HIP_TRY(ctx, hipMalloc(&sl.dWorkHead, 128));
            if (!sl.dSobTable) HIP_TRY(ctx, hipMalloc(&sl.dSobTable, (size_t)PATHS_MAX_FRAMES * RT_MAX_BOUNCES * 2 * sizeof(float)));
            if (overlap && ctx->inputDirty) {
                // a slot whose stream was only created just now (above) has not seen the mark yet
                HIP_TRY(ctx, hipStreamWaitEvent(ps, ctx->inputEvent, 0));
            }
            SceneD Sp = S;
            Sp.leafBatch = ctx->pathsLeafBatch;
            Sp.aovf = ctx->dAOvf[si];       // concurrent launches must not share overflow columns
            const size_t ldsP = (fast ? fast_stack_bytes(ctx) : lds) * PATHS_BLOCK / BLOCK;
            const int sobWords = B.nframes * F.maxBounces * 2;
            hipLaunchKernelGGL(k_sobol_table, dim3((sobWords + 255) / 256), dim3(256), 0, ps, B, F.maxBounces, ctx->dSobolV, sl.dSobTable);
            HIP_TRY(ctx, hipMemsetAsync(sl.dWorkHead, 0, sizeof(uint32_t), ps));
            hipEvent_t e0 = take_event(ctx), e1 = take_event(ctx);
            if (!e0 || !e1) { give_back_events(ctx, e0, e1); return fail(ctx, RT_E_HIP, "rt_render_frames: hipEventCreate failed"); }
            ctx->pendingEvents.emplace_back(e0, e1);    // owned by the context from here on, whatever fails below
            HIP_TRY(ctx, hipEventRecord(e0, ps));
#define LAUNCH_PATHS(A, C, N, O) hipLaunchKernelGGL((k_paths<A, C, N, O>), dim3(blocks), dim3(PATHS_BLOCK), ldsP, ps, Sp, F, B, sl.dSobTable, sl.dWorkHead, sl.dCurBatch, ctx->dCounters)
#define LAUNCH_PATHS_VN(A, C) hipLaunchKernelGGL((k_paths<A, C, true, 5, true>), dim3(blocks), dim3(PATHS_BLOCK), ldsP, ps, Sp, F, B, sl.dSobTable, sl.dWorkHead, sl.dCurBatch, ctx->dCounters)
            if (ext) {     // shading extensions: the counting instantiations only
                if (fast) LAUNCH_PATHS_VN(true, true); else if (cull) LAUNCH_PATHS_VN(false, true); else LAUNCH_PATHS_VN(false, false);
            }
#undef LAUNCH_PATHS_VN
            else if (fast) {
                if (cnt) LAUNCH_PATHS(true, true, true, 5);
                else if (occ == 5) LAUNCH_PATHS(true, true, false, 5);
                else if (occ == 7) LAUNCH_PATHS(true, true, false, 7);
                else if (occ == 8) LAUNCH_PATHS(true, true, false, 8);
                else LAUNCH_PATHS(true, true, false, 6);
            }
            else if (cull) { if (cnt) LAUNCH_PATHS(false, true, true, 5); else LAUNCH_PATHS(false, true, false, 5); }
            else { if (cnt) LAUNCH_PATHS(false, false, true, 5); else LAUNCH_PATHS(false, false, false, 5); }
#undef LAUNCH_PATHS
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipEventRecord(e1, ps));
            // the running mean is folded in frame order: after the previous batch's accumulate kernel,
            // and after whatever was on the context's stream when the call began
            if (overlap && ctx->lastSlot >= 0 && ctx->lastSlot != si)
                HIP_TRY(ctx, hipStreamWaitEvent(ps, ctx->slot[ctx->lastSlot].accDone, 0));
            if (overlap) HIP_TRY(ctx, hipStreamWaitEvent(ps, ctx->inputEvent, 0));
            hipLaunchKernelGGL(k_accumulate_batch, dim3(accBlocks), dim3(BLOCK), 0, ps, sl.dCurBatch, ctx->dSum, ctx->dMean, nfloats,
                               B.firstLoopNum, B.nframes, ctx->opt.history8);
            HIP_TRY(ctx, hipGetLastError());
            if (overlap) {
                HIP_TRY(ctx, hipEventRecord(sl.accDone, ps));
                sl.used = true;
                ctx->lastSlot = si;
                // whatever follows on the context's stream (reads, resets, the caller's own work) sees this batch
                HIP_TRY(ctx, hipStreamWaitEvent(st, sl.accDone, 0));
            }
        }
        ctx->frames += (uint64_t)count;
        // overlapped: the slots have seen the mark; otherwise this call itself worked on the context's stream
        ctx->inputDirty = !overlap;
        return RT_OK;
    }

    // wavefront pipeline: one frame at a time, or (experiment, RT_AMD_WAVEFRONT_FRAMES=n) n frames share the
    // queues, so that every launch ends with the longest ray of n frames instead of one
    int wbitsX = 1, wbitsY = 1;
    while ((1 << wbitsX) < ctx->width) wbitsX++;
    while ((1 << wbitsY) < ctx->height) wbitsY++;
    const int wf = std::max(1, std::min(ctx->wfFrames, paths_frames_cap(ctx)));
    const int extendBlocks = ctx->numCU * ctx->extendBlocksPerCU;   // persistent waves
    const size_t nfloats = (size_t)ctx->width * ctx->height * 3;
    uint32_t cap = ctx->queueCap;
    const PathSoA *wq = ctx->q;
    float *hitT = ctx->dHitT; uint32_t *hitTri = ctx->dHitTri; float *curImg = ctx->dCur; float *sobTable = nullptr;
    if (wf > 1) {
        if ((uint64_t)ctx->queueCap * (uint64_t)wf * NSHARD >= ((uint64_t)1 << 31)) return fail(ctx, RT_E_INVALID, "rt_render_frames: RT_AMD_WAVEFRONT_FRAMES too large for this image size");
        if (ctx->wfAllocFrames != wf) {
            HIP_TRY(ctx, hipStreamSynchronize(st));
            hipFree(ctx->wfPathMem); hipFree(ctx->wfHitT); hipFree(ctx->wfHitTri); hipFree(ctx->wfCur); hipFree(ctx->wfSobTable);
            ctx->wfPathMem = nullptr; ctx->wfHitT = nullptr; ctx->wfHitTri = nullptr; ctx->wfCur = nullptr; ctx->wfSobTable = nullptr; ctx->wfAllocFrames = 0;
            const size_t P = (size_t)ctx->queueCap * wf * NSHARD;
            HIP_TRY(ctx, hipMalloc(&ctx->wfPathMem, 2 * 11 * P * sizeof(float)));
            HIP_TRY(ctx, hipMalloc(&ctx->wfHitT, P * sizeof(float)));
            HIP_TRY(ctx, hipMalloc(&ctx->wfHitTri, P * sizeof(uint32_t)));
            HIP_TRY(ctx, hipMalloc(&ctx->wfCur, (size_t)wf * nfloats * sizeof(float)));
            HIP_TRY(ctx, hipMalloc(&ctx->wfSobTable, (size_t)PATHS_MAX_FRAMES * RT_MAX_BOUNCES * 2 * sizeof(float)));
            for (int b = 0; b < 2; b++) {
                float *base = ctx->wfPathMem + (size_t)b * 11 * P;
                PathSoA &q = ctx->wfQ[b];
                q.ox = base; q.oy = base + P; q.oz = base + 2 * P;
                q.dx = base + 3 * P; q.dy = base + 4 * P; q.dz = base + 5 * P;
                q.tr = base + 6 * P; q.tg = base + 7 * P; q.tb = base + 8 * P;
                q.seed = (uint32_t *)(base + 9 * P); q.pix = (uint32_t *)(base + 10 * P);
            }
            ctx->wfAllocFrames = wf;
        }
        cap = ctx->queueCap * (uint32_t)wf;
        wq = ctx->wfQ; hitT = ctx->wfHitT; hitTri = ctx->wfHitTri; curImg = ctx->wfCur; sobTable = ctx->wfSobTable;
    }
    const int shadeBlocks = (int)std::min<size_t>(((size_t)cap * NSHARD + BLOCK - 1) / BLOCK, (size_t)ctx->numCU * 32);
    const int accBlocks = (int)((nfloats + BLOCK - 1) / BLOCK);
    uint32_t *counts = ctx->dQueueCtl, *heads = ctx->dQueueCtl + QCTL_COUNTS;
    const int qstride = NSHARD * CTL_STRIDE;
    const bool useAccel = ctx->opt.accel != 0 && S.anodes != nullptr;
    const size_t ldsFast = fast_stack_bytes(ctx);
    ctx->inputDirty = true;             // accumulators and slot 0's overflow columns are used on the context's stream
    for (int f0 = 0; f0 < count; f0 += wf) {
        const int nf = std::min(wf, count - f0);
        const int loopNum = first_loopnum + f0;
        HIP_TRY(ctx, hipMemsetAsync(ctx->dQueueCtl, 0, QCTL_WORDS * sizeof(uint32_t), st));
        if (wf > 1) {
            BatchD B;
            B.firstLoopNum = loopNum; B.nframes = nf; B.bitsX = wbitsX; B.bitsY = wbitsY;
            for (int k = 0; k < PATHS_MAX_FRAMES; k++) B.randOrigin[k] = 0.0f;
            const int sobWords = nf * F.maxBounces * 2;
            hipLaunchKernelGGL(k_sobol_table, dim3((sobWords + 255) / 256), dim3(256), 0, st, B, F.maxBounces, ctx->dSobolV, sobTable);
        }
        for (int k = 0; k < nf; k++)
            hipLaunchKernelGGL(k_generate, dim3(tileBlocks), dim3(BLOCK), 0, st, F, rand_origin[f0 + k], wq[0], cap, counts, curImg + (size_t)k * nfloats,
                               (uint32_t)k, wbitsX, wbitsY);
        for (int b = 0; b < F.maxBounces; b++) {
            const PathSoA &qin = wq[b & 1], &qout = wq[(b + 1) & 1];
            hipEvent_t e0 = take_event(ctx), e1 = take_event(ctx);
            if (!e0 || !e1) { give_back_events(ctx, e0, e1); return fail(ctx, RT_E_HIP, "rt_render_frames: hipEventCreate failed"); }
            ctx->pendingEvents.emplace_back(e0, e1);
            HIP_TRY(ctx, hipEventRecord(e0, st));
            if (useAccel) {
                if (cnt) hipLaunchKernelGGL((k_extend_fast<true>), dim3(extendBlocks), dim3(BLOCK), ldsFast, st, S, qin, cap, counts + b * qstride, heads + b * qstride, hitT, hitTri, ctx->dCounters);
                else hipLaunchKernelGGL((k_extend_fast<false>), dim3(extendBlocks), dim3(BLOCK), ldsFast, st, S, qin, cap, counts + b * qstride, heads + b * qstride, hitT, hitTri, ctx->dCounters);
            } else {
#define LAUNCH_EXT(C, N) hipLaunchKernelGGL((k_extend<C, N>), dim3(extendBlocks), dim3(BLOCK), lds, st, S, qin, cap, counts + b * qstride, heads + b * qstride, hitT, hitTri, ctx->dCounters)
                if (cull) { if (cnt) LAUNCH_EXT(true, true); else LAUNCH_EXT(true, false); }
                else { if (cnt) LAUNCH_EXT(false, true); else LAUNCH_EXT(false, false); }
#undef LAUNCH_EXT
            }
            HIP_TRY(ctx, hipEventRecord(e1, st));
            if (ext) hipLaunchKernelGGL(k_shade<true>, dim3(shadeBlocks), dim3(BLOCK), 0, st, S, F, b, (uint32_t)loopNum, ctx->dSobolV, qin, cap, counts + b * qstride,
                                         hitT, hitTri, qout, counts + (b + 1) * qstride, curImg, wbitsX, wbitsY, (const float *)sobTable, (uint32_t)nfloats);
            else hipLaunchKernelGGL(k_shade<false>, dim3(shadeBlocks), dim3(BLOCK), 0, st, S, F, b, (uint32_t)loopNum, ctx->dSobolV, qin, cap, counts + b * qstride,
                                    hitT, hitTri, qout, counts + (b + 1) * qstride, curImg, wbitsX, wbitsY, (const float *)sobTable, (uint32_t)nfloats);
        }
        if (wf > 1) {
            hipLaunchKernelGGL(k_count_queue_rays, dim3(1), dim3(1), 0, st, counts, F.maxBounces, ctx->dCounters);
            hipLaunchKernelGGL(k_accumulate_batch, dim3(accBlocks), dim3(BLOCK), 0, st, curImg, ctx->dSum, ctx->dMean, nfloats, loopNum, nf, ctx->opt.history8);
        } else {
            hipLaunchKernelGGL(k_accumulate, dim3(accBlocks), dim3(BLOCK), 0, st, curImg, ctx->dSum, ctx->dMean, nfloats, loopNum,
                               counts, F.maxBounces, ctx->opt.history8, ctx->dCounters);
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    ctx->frames += (uint64_t)count;
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_read_mean(RtContext *ctx, float *rgb) {
    if (!ctx || !rgb) return fail(ctx, RT_E_INVALID, "rt_read_mean: null pointer");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(rgb, ctx->dMean, (size_t)ctx->width * ctx->height * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_read_mean_async(RtContext *ctx, float *rgb) {
    if (!ctx || !rgb) return fail(ctx, RT_E_INVALID, "rt_read_mean_async: null pointer");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->readEvent) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->readEvent, hipEventDisableTiming));
    // On the context's stream: after the accumulation of every frame rendered so far, and before that of
    // the next call (whose accumulate kernels wait for what is on this stream when the call begins), while
    // the next call's path tracing, which does not touch the image, runs beside the copy.
    HIP_TRY(ctx, hipMemcpyAsync(rgb, ctx->dMean, (size_t)ctx->width * ctx->height * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->readEvent, ctx->stream));
    ctx->readPending = true;
    return RT_OK;
}

int rt_read_wait(RtContext *ctx) {
    if (!ctx) return fail(nullptr, RT_E_INVALID, "rt_read_wait: null context");
    if (!ctx->readPending) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->readEvent));
    ctx->readPending = false;
    return RT_OK;
}

int rt_mean_device_ptr(RtContext *ctx, void **dev_ptr) {
    if (!ctx || !dev_ptr) return fail(ctx, RT_E_INVALID, "rt_mean_device_ptr: null pointer");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    ctx->sumPtrGivenOut = true;     // the caller can now order device work after render calls: every call launches its frames at once
    *dev_ptr = ctx->dMean;
    return RT_OK;
}

void *rt_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void rt_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int rt_read_sum(RtContext *ctx, float *rgb) {
    if (!ctx || !rgb) return fail(ctx, RT_E_INVALID, "rt_read_sum: null pointer");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(rgb, ctx->dSum, (size_t)ctx->width * ctx->height * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_restore(RtContext *ctx, const float *mean_rgb, const float *sum_rgb, uint64_t frames) {
    if (!ctx) return fail(nullptr, RT_E_INVALID, "rt_restore: null context");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    if (frames >= ((uint64_t)1 << 30)) return fail(ctx, RT_E_INVALID, "rt_restore: frame count out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t img = (size_t)ctx->width * ctx->height * 3 * sizeof(float);
    if (mean_rgb) HIP_TRY(ctx, hipMemcpyAsync(ctx->dMean, mean_rgb, img, hipMemcpyHostToDevice, ctx->stream));
    if (sum_rgb) HIP_TRY(ctx, hipMemcpyAsync(ctx->dSum, sum_rgb, img, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->inputDirty = true;
    ctx->frames = frames;
    return RT_OK;
}

int rt_sum_device_ptr(RtContext *ctx, void **dev_ptr) {
    if (!ctx || !dev_ptr) return fail(ctx, RT_E_INVALID, "rt_sum_device_ptr: null pointer");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    ctx->sumPtrGivenOut = true;     // from now on every call launches its frames before it returns
    *dev_ptr = ctx->dSum;
    return RT_OK;
}

int rt_bind_sum_buffer(RtContext *ctx, float *dev_ptr) {
    if (!ctx) return fail(nullptr, RT_E_INVALID, "rt_bind_sum_buffer: null context");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->dSum = dev_ptr ? dev_ptr : ctx->dSumOwned;
    return RT_OK;
}

int rt_get_stats(RtContext *ctx, RtStats *out) {
    if (!ctx || !out) return fail(ctx, RT_E_INVALID, "rt_get_stats: null pointer");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    WorkCounters wc;
    HIP_TRY(ctx, hipMemcpyAsync(&wc, ctx->dCounters, sizeof wc, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    out->rays = wc.rays; out->node_tests = wc.node_tests; out->tri_tests = wc.tri_tests; out->tri_hits = wc.tri_hits;
    out->frames = ctx->frames;
    out->retraced = wc.retraced;
    return RT_OK;
}

int rt_get_scene_info(RtContext *ctx, RtSceneInfo *out) {
    if (!ctx || !out) return fail(ctx, RT_E_INVALID, "rt_get_scene_info: null pointer");
    if (!ctx->haveScene) return fail(ctx, RT_E_STATE, "rt_get_scene_info: no scene uploaded");
    std::memset(out, 0, sizeof *out);
    out->triangles = ctx->nTris;
    out->index_nodes = ctx->nIndexNodes;
    out->tree_height = ctx->scene.stackEntries - 1;
    out->index_stack_need = ctx->indexStackNeed;
    out->index_stack_lds = ctx->scene.anodes ? ctx->scene.aldsEntries : 0;
    out->index_enabled = ctx->scene.anodes != nullptr;
    out->paths_waves_per_simd = (ctx->opt.accel != 0 && ctx->scene.anodes) ? paths_occupancy(ctx) : 5;
    out->frames_per_launch = paths_frames_cap(ctx);
    out->index_built_on_device = ctx->indexOnDevice ? 1 : 0;
    out->index_build_fell_back = ctx->indexFellBack ? 1 : 0;
    return RT_OK;
}

int rt_selftest_exact_math(RtContext *ctx, uint64_t mismatches[4]) {
    if (!ctx || !mismatches) return fail(ctx, RT_E_INVALID, "rt_selftest_exact_math: null pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long *d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, 4 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_selftest_exact_math, dim3(ctx->numCU * 8), dim3(256), 0, ctx->stream, d);
        e = hipGetLastError();
    }
    unsigned long long h[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(ctx, RT_E_HIP, std::string("rt_selftest_exact_math: ") + hipGetErrorString(e));
    for (int k = 0; k < 4; k++) mismatches[k] = h[k];
    return RT_OK;
}

int rt_get_kernel_time(RtContext *ctx, double *total_ms, int64_t *launches) {
    if (!ctx || !total_ms || !launches) return fail(ctx, RT_E_INVALID, "rt_get_kernel_time: null pointer");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int rc = resolve_events(ctx);
    *total_ms = ctx->kernelMs; *launches = ctx->kernelLaunches;
    return rc;
}

int rt_trace_rays(RtContext *ctx, const float *origins, const float *dirs, int64_t n, float *t_out, int32_t *id_out) {
    if (!ctx || !origins || !dirs || !t_out || !id_out || n < 0) return fail(ctx, RT_E_INVALID, "rt_trace_rays: bad argument");
    if (!ctx->haveScene) return fail(ctx, RT_E_STATE, "rt_trace_rays: no scene uploaded");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    if (n == 0) return RT_OK;
    if (n > (int64_t)1 << 26) return fail(ctx, RT_E_INVALID, "rt_trace_rays: more than 2^26 rays per call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // The rays go through the production extend kernels: one queue whose shard 0 holds them all.
    const uint32_t cap = (uint32_t)((n + 63) / 64 * 64);
    std::vector<float> soa((size_t)cap * 6, 0.0f);
    for (int64_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            soa[(size_t)c * cap + i] = origins[3 * i + c];
            soa[(size_t)(3 + c) * cap + i] = dirs[3 * i + c];
        }
    float *dRays = nullptr, *dT = nullptr; uint32_t *dI = nullptr, *dCtl = nullptr;
    std::vector<uint32_t> ctl(QCTL_WORDS, 0u);
    ctl[0] = (uint32_t)n;
    std::vector<float> hT((size_t)n); std::vector<uint32_t> hI((size_t)n);
    hipError_t e = hipMalloc(&dRays, soa.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&dT, (size_t)cap * 4);
    if (e == hipSuccess) e = hipMalloc(&dI, (size_t)cap * 4);
    if (e == hipSuccess) e = hipMalloc(&dCtl, QCTL_WORDS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(dRays, soa.data(), soa.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dCtl, ctl.data(), QCTL_WORDS * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        PathSoA q{};
        q.ox = dRays; q.oy = dRays + cap; q.oz = dRays + 2 * (size_t)cap;
        q.dx = dRays + 3 * (size_t)cap; q.dy = dRays + 4 * (size_t)cap; q.dz = dRays + 5 * (size_t)cap;
        const SceneD &S = ctx->scene;
        const size_t lds = stack_bytes(ctx);
        const int blocks = (int)std::min<int64_t>((n + BLOCK - 1) / BLOCK, (int64_t)ctx->numCU * ctx->extendBlocksPerCU);
        uint32_t *counts = dCtl, *heads = dCtl + QCTL_COUNTS;
        if (ctx->opt.accel != 0 && S.anodes != nullptr) {
            hipLaunchKernelGGL((k_extend_fast<false>), dim3(blocks), dim3(BLOCK), fast_stack_bytes(ctx), ctx->stream, S, q, cap, counts, heads, dT, dI, ctx->dCounters);
        } else if (ctx->opt.cull) {
            hipLaunchKernelGGL((k_extend<true, false>), dim3(blocks), dim3(BLOCK), lds, ctx->stream, S, q, cap, counts, heads, dT, dI, ctx->dCounters);
        } else {
            hipLaunchKernelGGL((k_extend<false, false>), dim3(blocks), dim3(BLOCK), lds, ctx->stream, S, q, cap, counts, heads, dT, dI, ctx->dCounters);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(hT.data(), dT, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(hI.data(), dI, (size_t)n * 4, hipMemcpyDeviceToHost);
    hipFree(dRays); hipFree(dT); hipFree(dI); hipFree(dCtl);
    if (e != hipSuccess) return fail(ctx, RT_E_HIP, std::string("rt_trace_rays: ") + hipGetErrorString(e));
    for (int64_t i = 0; i < n; i++) {
        const bool hit = hI[(size_t)i] != REF_NONE;
        t_out[i] = hit ? hT[(size_t)i] : -1.0f;
        id_out[i] = hit ? (int32_t)hI[(size_t)i] : -1;
    }
    return RT_OK;
}

// rt_trace_rays for callers that cannot allocate.  Rays and results are device arrays of the
// caller; everything in between lives in this GPU's `scratch` array, reached through the pointer
// rt_create looked up.  Nothing here allocates or frees, on the host or on the device, and the
// call does not wait for the GPU.
int rt_trace_rays_noalloc(RtContext *ctx, const float *d_origins, const float *d_dirs, int64_t n, float *d_t_out, int32_t *d_id_out) {
    if (!ctx || !d_origins || !d_dirs || !d_t_out || !d_id_out || n < 0) return fail(ctx, RT_E_INVALID, "rt_trace_rays_noalloc: bad argument");
    if (n > (int64_t)RT_NOALLOC_MAX_RAYS) return fail(ctx, RT_E_INVALID, "rt_trace_rays_noalloc: more than RT_NOALLOC_MAX_RAYS (262144) rays per call");
    if (!ctx->haveScene) return fail(ctx, RT_E_STATE, "rt_trace_rays_noalloc: no scene uploaded");
    { const int frc = flush_pending(ctx); if (frc != RT_OK) return frc; }
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t *base = (ctx->device >= 0 && ctx->device < MAX_DEVICES) ? g_scratchPtr[ctx->device] : nullptr;
    if (!base) return fail(ctx, RT_E_STATE, "rt_trace_rays_noalloc: working memory of this device was not initialised");
    const uint32_t cap = (uint32_t)((n + 63) / 64 * 64);
    float *soa = (float *)base;                                     // 6 * cap words
    float *hitT = (float *)(base + (size_t)6 * RT_NOALLOC_MAX_RAYS);
    uint32_t *hitTri = base + (size_t)7 * RT_NOALLOC_MAX_RAYS;
    uint32_t *ctl = base + (size_t)8 * RT_NOALLOC_MAX_RAYS;         // cleared by k_stage_rays on every call
    uint32_t *counts = ctl, *heads = ctl + NSHARD * CTL_STRIDE;
    PathSoA q{};
    q.ox = soa; q.oy = soa + cap; q.oz = soa + 2 * (size_t)cap;
    q.dx = soa + 3 * (size_t)cap; q.dy = soa + 4 * (size_t)cap; q.dz = soa + 5 * (size_t)cap;
    const SceneD &S = ctx->scene;
    hipStream_t st = ctx->stream;
    const int stageBlocks = (int)((std::max<uint32_t>(cap, NOALLOC_CTL_WORDS) + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_stage_rays, dim3(stageBlocks), dim3(BLOCK), 0, st, d_origins, d_dirs, (uint32_t)n, cap, soa, ctl, NOALLOC_CTL_WORDS);
    const int blocks = (int)std::min<int64_t>((n + BLOCK - 1) / BLOCK, (int64_t)ctx->numCU * ctx->extendBlocksPerCU);
    if (ctx->opt.accel != 0 && S.anodes != nullptr) {
        hipLaunchKernelGGL((k_extend_fast<false>), dim3(blocks), dim3(BLOCK), fast_stack_bytes(ctx), st, S, q, cap, counts, heads, hitT, hitTri, ctx->dCounters);
    } else if (ctx->opt.cull) {
        hipLaunchKernelGGL((k_extend<true, false>), dim3(blocks), dim3(BLOCK), stack_bytes(ctx), st, S, q, cap, counts, heads, hitT, hitTri, ctx->dCounters);
    } else {
        hipLaunchKernelGGL((k_extend<false, false>), dim3(blocks), dim3(BLOCK), stack_bytes(ctx), st, S, q, cap, counts, heads, hitT, hitTri, ctx->dCounters);
    }
    hipLaunchKernelGGL(k_unstage_hits, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, hitT, hitTri, (uint32_t)n, d_t_out, d_id_out);
    HIP_TRY(ctx, hipGetLastError());
    // the extend kernel keeps the stack of an exact re-trace in slot 0's overflow columns: the next
    // overlapped batch on that slot has to come after it
    ctx->inputDirty = true;
    return RT_OK;
}

int rt_selftest_device_sort(RtContext *ctx, uint64_t *keys, uint32_t *vals, int64_t n) {
    if (!ctx || (n > 0 && (!keys || !vals))) return fail(ctx, RT_E_INVALID, "rt_selftest_device_sort: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    const int rc = rt::selftest_device_sort(keys, vals, n, ctx->stream, err);
    return rc == RT_OK ? RT_OK : fail(ctx, rc, "rt_selftest_device_sort: " + err);
}

int rt_selftest_device_scan(RtContext *ctx, uint64_t *values, int64_t n) {
    if (!ctx || (n > 0 && !values)) return fail(ctx, RT_E_INVALID, "rt_selftest_device_scan: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    const int rc = rt::selftest_device_scan(values, n, ctx->stream, err);
    return rc == RT_OK ? RT_OK : fail(ctx, rc, "rt_selftest_device_scan: " + err);
}

// Not part of the reference boundary: runs the device's divide / sqrt / fma / sincos on
// caller data so that a test can compare them with the host's (out has 5*n floats).
int rt_selftest_arith(RtContext *ctx, const float *a, const float *b, int64_t n, float *out) {
    if (!ctx || !a || !b || !out || n <= 0 || n > (int64_t)1 << 26) return fail(ctx, RT_E_INVALID, "rt_selftest_arith: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *dA = nullptr, *dB = nullptr, *dOut = nullptr;
    hipError_t e = hipMalloc(&dA, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&dB, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&dOut, (size_t)n * 20);
    if (e == hipSuccess) e = hipMemcpy(dA, a, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dB, b, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dA, dB, (long long)n, dOut);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out, dOut, (size_t)n * 20, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dOut);
    if (e != hipSuccess) return fail(ctx, RT_E_HIP, std::string("rt_selftest_arith: ") + hipGetErrorString(e));
    return RT_OK;
}

#ifdef RT_EXP_TIMELINE
// experiment builds: copies the per-wave records of the last path-kernel launch (5 words per wave)
int rt_debug_timeline(RtContext *ctx, unsigned long long *out, int64_t waves) {
    if (!ctx || !out || !ctx->scene.timeline) return RT_E_INVALID;
    hipDeviceSynchronize();
    return hipMemcpy(out, ctx->scene.timeline, (size_t)waves * 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? RT_OK : RT_E_HIP;
}
#endif

const char *rt_last_error(const RtContext *ctx) { return ctx ? ctx->err.c_str() : rt::global_error(); }

}  // extern "C"
