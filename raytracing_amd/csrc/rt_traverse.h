// Closest-hit search of one ray as per-lane state machines (device code, gfx950):
//   Trav      the loop of hitBVH (frag:380-421) over the reference's tree in the reference's order
//   FastTrav  the same answer through the library's 4-wide search index, plus the proof that the
//             reference's traversal reaches the triangle found (verify_reference_reaches)
// Included by rt_device.hip only; compile with -ffp-contract=off (rt_math.h).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "rt_kernels.h"

namespace rtd {

constexpr float HIT_MIN_INIT = 100000.0f;   // frag:633,689

// Record `i` of an array of `bytes`-byte records: a 32-bit byte offset from the (wave-uniform) base,
// so that the load takes the base from scalar registers and one vector register of offset instead
// of a 64-bit vector address (two to three vector instructions less per record; measured +0.4 %
// on 1.3M triangles, +1.7 % on 262k at 4K).  Valid because rt_scene_upload refuses scenes whose
// record arrays would reach 4 GiB (RT_MAX_TRIANGLES).
#define RT_REC(base, i, bytes) ((const float4 *)((const char *)(base) + (uint32_t)((uint32_t)(i) * (bytes))))
constexpr int FAST_QUEUE_ENTRIES = 6;        // FastTrav::PEND_MAX: LDS entries per lane ahead of the deferred nodes

// IntersectBound, frag:309-331, evaluated without early exits (the values computed after a
// failed test are unused, so the outcome is the same).  near[] receives the three entry
// distances for the conservative culling bound.
__device__ __forceinline__ bool slab_test(float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                          const vec3 o, const vec3 inv, bool nx, bool ny, bool nz,
                                          float &nearx, float &neary, float &nearz) {
    float tMin = ((nx ? mxx : mnx) - o.x) * inv.x;
    float tMax = ((nx ? mnx : mxx) - o.x) * inv.x;
    const float tyMin = ((ny ? mxy : mny) - o.y) * inv.y;
    const float tyMax = ((ny ? mny : mxy) - o.y) * inv.y;
    nearx = tMin; neary = tyMin;
    const bool fail1 = (tMin > tyMax) || (tyMin > tMax);
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    const float tzMin = ((nz ? mxz : mnz) - o.z) * inv.z;
    const float tzMax = ((nz ? mnz : mxz) - o.z) * inv.z;
    nearz = tzMin;
    const bool fail2 = (tMin > tzMax) || (tzMin > tMax);
    if (tzMax < tMax) tMax = tzMax;
    return !fail1 && !fail2 && (tMax > 0.0f);
}

// hitTriangle, frag:255-278, with N and D taken from the triangle record.
__device__ __forceinline__ float triangle_test(const float4 q0, const float4 q1, const float4 q2,
                                               const vec3 o, const vec3 d) {
    const vec3 v0 = rtm::mk(q0.x, q0.y, q0.z);
    const vec3 v1 = rtm::mk(q0.w, q1.x, q1.y);
    const vec3 v2 = rtm::mk(q1.z, q1.w, q2.x);
    const vec3 N = rtm::mk(q2.y, q2.z, q2.w);
    const float D = -rtm::dot(N, v0);            // frag:262; three instructions instead of a fourth load of the record
    const float nd = rtm::dot(N, d);
    // The shader returns early after each failed test (frag:264-275); evaluating everything and
    // combining the five rejections gives the same value (what follows a rejection is unused
    // there, and nothing here traps), without divergent branches and with the whole record
    // fetched in one go.
    const float t = -(rtm::dot(N, o) + D) / nd;
    const vec3 pHit = rtm::along(o, t, d);
    const float e0 = rtm::dot(N, rtm::cross(rtm::sub(v1, v0), rtm::sub(pHit, v0)));
    const float e1 = rtm::dot(N, rtm::cross(rtm::sub(v2, v1), rtm::sub(pHit, v1)));
    const float e2 = rtm::dot(N, rtm::cross(rtm::sub(v0, v2), rtm::sub(pHit, v2)));
    const bool reject = (nd == 0.0f) | (t < 0.0f) | (e0 < 0.0f) | (e1 < 0.0f) | (e2 < 0.0f);
    return reject ? -1.0f : t - 0.00001f;
}

// The traversal loop of hitBVH, frag:380-421, as a per-lane state machine.  It visits children
// in the reference's order (near child by dirIsNeg[axis], far child deferred on a LIFO stack)
// and keeps the closest hit with the same strict comparison, so ties resolve as in the
// reference.  `cur` is the next thing to process: an interior node index, LEAF_BIT|triangle,
// or REF_NONE when the ray is finished.
// stack: this lane's LDS column, entry i at stack[i*stride].
// CULL: additionally skip a box when a lower bound of its entry distance, computed against
// the box grown by eps, lies beyond the closest hit so far; no triangle in such a box can
// pass `dis_t < ray.hitMin` (frag:396), so the result is unchanged.
template <bool CULL, bool COUNT>
struct Trav {
    vec3 o, d, inv;
    float ex, ey, ez, hitCull;
    float hitMin;
    uint32_t best;
    uint32_t cur;
    int sp;
    bool nx, ny, nz;

    __device__ __forceinline__ void begin(const SceneD &S, const vec3 o_, const vec3 d_, uint32_t &nNode) {
        o = o_; d = d_;
        hitMin = HIT_MIN_INIT;
        best = REF_NONE;
        inv = rtm::recip3_exact(d);                                 // frag:376
        nx = inv.x < 0.0f; ny = inv.y < 0.0f; nz = inv.z < 0.0f;    // frag:378
        ex = ey = ez = hitCull = 0.0f;
        if (CULL) {
            const float oM = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z)));
            const float eps = rtm::madd(oM, 1.0f / 1024.0f, S.cullEps);
            ex = eps * fabsf(inv.x); ey = eps * fabsf(inv.y); ez = eps * fabsf(inv.z);
            hitCull = rtm::madd(hitMin, 1.0001f, 1.0e-4f);
        }
        sp = 0;
        cur = S.rootRef;
        if (cur != REF_NONE) {
            float a, b, c;
            const bool pass = slab_test(S.rootMin[0], S.rootMin[1], S.rootMin[2], S.rootMax[0], S.rootMax[1], S.rootMax[2],
                                        o, inv, nx, ny, nz, a, b, c);
            if (COUNT) nNode++;
            if (!pass) cur = REF_NONE;
        }
    }

    __device__ __forceinline__ bool at_interior() const { return (int32_t)cur >= 0; }          // not LEAF, not NONE
    __device__ __forceinline__ bool at_leaf() const { return (cur & LEAF_BIT) && cur != REF_NONE; }
    __device__ __forceinline__ bool done() const { return cur == REF_NONE; }

    __device__ __forceinline__ void pop(const uint32_t *stack, const int stride) {
        cur = REF_NONE;
        if (sp > 0) cur = stack[(--sp) * stride];
    }

    // cur is an interior node: test both children, descend / defer / pop (frag:386-420)
    __device__ __forceinline__ void node_step(const SceneD &S, uint32_t *stack, const int stride, uint32_t &nNode) {
        const float4 *rec = RT_REC(S.nodes, cur, 64u);
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
        float lnx, lny, lnz, rnx, rny, rnz;
        bool passL = slab_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, nx, ny, nz, lnx, lny, lnz);
        bool passR = slab_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, nx, ny, nz, rnx, rny, rnz);
        if (COUNT) nNode += 2;
        if (CULL) {
            const float lb = fmaxf(lnx - ex, fmaxf(lny - ey, lnz - ez));
            const float rb = fmaxf(rnx - ex, fmaxf(rny - ey, rnz - ez));
            if (lb > hitCull) passL = false;
            if (rb > hitCull) passR = false;
        }
        const uint32_t left = __float_as_uint(q3.x), right = __float_as_uint(q3.y), axis = __float_as_uint(q3.z);
        const bool negAxis = (axis == 0u) ? nx : ((axis == 1u) ? ny : nz);      // frag:407
        const uint32_t nearRef = negAxis ? right : left;
        const uint32_t farRef = negAxis ? left : right;
        const bool nearPass = negAxis ? passR : passL;
        const bool farPass = negAxis ? passL : passR;
        if (nearPass) {
            cur = nearRef;
            if (farPass) stack[(sp++) * stride] = farRef;
        } else if (farPass) {
            cur = farRef;
        } else {
            pop(stack, stride);
        }
    }

    // cur is a leaf: test its triangle, keep it if closer (frag:390-403), then pop
    __device__ __forceinline__ void leaf_step(const SceneD &S, const uint32_t *stack, const int stride, uint32_t &nTri) {
        const uint32_t ti = cur & ~LEAF_BIT;
        const float4 *rec = RT_REC(S.tris, ti, 48u);
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        const float dis = triangle_test(q0, q1, q2, o, d);
        if (COUNT) nTri++;
        if (dis > 0.0f && dis < hitMin) {           // frag:396
            hitMin = dis;
            best = ti;
            if (CULL) hitCull = rtm::madd(hitMin, 1.0001f, 1.0e-4f);
        }
        pop(stack, stride);
    }
};

// Every lane of a wave that is still working is either at an interior node or at a leaf.
// Measured on MI355X: making lanes wait at a leaf so that triangle tests run in batches
// (while-while traversal) is slower at every batch size (1.22 -> 0.66 Grays/s from batch 1 to
// 64 on the 1.31M-triangle scene); the cost is per wave iteration, not per issued instruction.

template <bool CULL, bool COUNT>
__device__ __forceinline__ void wave_advance(Trav<CULL, COUNT> &tv, const SceneD &S, uint32_t *stack, const int stride,
                                             uint32_t &nNode, uint32_t &nTri) {
    // every working lane advances by one step per iteration
    if (tv.at_interior()) tv.node_step(S, stack, stride, nNode);
    else if (tv.at_leaf()) tv.leaf_step(S, stack, stride, nTri);
}

// Runs the rays of the calling lanes to completion (all lanes leave together).
template <bool CULL, bool COUNT>
__device__ __forceinline__ void closest_triangle(const SceneD &S, const vec3 o, const vec3 d,
                                                 uint32_t *stack, const int stride,
                                                 float &tOut, uint32_t &triOut,
                                                 uint32_t &nNode, uint32_t &nTri) {
    Trav<CULL, COUNT> tv;
    tv.begin(S, o, d, nNode);
    while (true) {
        const unsigned long long mI = ballot64(tv.at_interior());
        const unsigned long long mL = ballot64(tv.at_leaf());
        if ((mI | mL) == 0ull) break;
        wave_advance(tv, S, stack, stride, nNode, nTri);
    }
    tOut = tv.hitMin;
    triOut = tv.best;
}

// ---------------------------------------------------------------- search index traversal

__device__ __forceinline__ bool finite3(const vec3 v) {
    return (fabsf(v.x) < INFINITY) && (fabsf(v.y) < INFINITY) && (fabsf(v.z) < INFINITY);
}

//
// Finds the closest triangle that hitTriangle accepts, through the SAH index.  The index boxes
// are grown (rt_accel.cpp), the box test below treats NaN as "inside", and a box is only
// skipped when it starts beyond the closest hit so far by a margin, so no triangle that could
// be accepted at a distance <= the final one is missed.  `tie` reports that two different
// triangles were accepted at exactly the final distance (the reference keeps whichever it
// meets first, which this search cannot tell).
template <bool COUNT>
struct FastTrav {
    vec3 o, d, inv;
    float hitMin, hitCull;
    uint32_t best, cur;         // cur: interior node of the index, or REF_NONE when nothing is left
    int sp;
    bool tie;
    // Triangles met but not yet tested.  Leaves of the index hold one triangle each and are
    // named by their parent, so a node step never has to stop at a leaf: it queues the
    // triangle and goes on.  The wave tests queued triangles together (one per lane per
    // pass); testing them one lane at a time costs a full pass of the long triangle code at a
    // few percent lane utilisation.  The search keeps every candidate whatever the order of
    // the tests.  The queue is the first PEND_MAX entries of the lane's LDS column, the
    // deferred nodes follow it.
    int npend;
#ifdef RT_EXP_EMPTY
    bool dbgWouldHit;           // experiment builds: the last node step had a child that only the known closest hit rules out
#endif

    static constexpr int PEND_MAX = FAST_QUEUE_ENTRIES;     // a node step may queue up to 4 triangles

    // rays this search may be used for: finite 1/d and an origin within the range the growth
    // of the index boxes was sized for
    template <typename SceneT>
    __device__ __forceinline__ bool usable(const SceneT &S) const {
        const float oM = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z)));
        return finite3(inv) && (oM <= S.originLimit);
    }
    __device__ __forceinline__ void queue_triangle(uint32_t *col, const int stride, uint32_t ti) {
        col[npend * stride] = ti;
        npend++;
    }
    __device__ __forceinline__ uint32_t unqueue_triangle(const uint32_t *col, const int stride) {
        npend--;
        return col[npend * stride] & ~LEAF_BIT;
    }
    // deferred nodes: LDS column of this lane first, global overflow beyond S.aldsEntries
    __device__ __forceinline__ void push(const SceneD &S, uint32_t *col, const int stride, uint32_t *ovf, uint32_t v) {
        if (sp < S.aldsEntries) col[(PEND_MAX + sp) * stride] = v;
        else ovf[(size_t)(sp - S.aldsEntries) * S.aovfStride] = v;
        sp++;
    }
    __device__ __forceinline__ uint32_t pop(const SceneD &S, const uint32_t *col, const int stride, const uint32_t *ovf) {
        sp--;
        // the empty asm keeps the LDS read and the (rare) global read apart; merged, they become one
        // flat load of a selected address
        uint32_t v;
        if (sp < S.aldsEntries) {
            v = col[(PEND_MAX + sp) * stride];
            asm volatile("" : "+v"(v));
        } else {
            v = ovf[(size_t)(sp - S.aldsEntries) * S.aovfStride];
        }
        return v;
    }

    template <typename SceneT>
    __device__ __forceinline__ void begin(const SceneT &S, uint32_t *col, const int stride, const vec3 o_, const vec3 d_, uint32_t &nNode) {
        o = o_; d = d_;
        inv = rtm::recip3_exact(d);
        hitMin = HIT_MIN_INIT;
        hitCull = rtm::madd(hitMin, 1.0001f, 1.0e-4f);
        best = REF_NONE; tie = false; sp = 0; npend = 0;
        // root box (not quantised)
        const float x0 = (S.arootMin[0] - o.x) * inv.x, x1 = (S.arootMax[0] - o.x) * inv.x;
        const float y0 = (S.arootMin[1] - o.y) * inv.y, y1 = (S.arootMax[1] - o.y) * inv.y;
        const float z0 = (S.arootMin[2] - o.z) * inv.z, z1 = (S.arootMax[2] - o.z) * inv.z;
        const float tn = fmaxf(fminf(x0, x1), fmaxf(fminf(y0, y1), fminf(z0, z1)));
        const float tf = fminf(fmaxf(x0, x1), fminf(fmaxf(y0, y1), fmaxf(z0, z1)));
        if (COUNT) nNode++;
        cur = REF_NONE;
        // a ray with a zero, NaN or infinite direction component is not searched here at all
        // (with NaN every box would count as hit); the caller sends it to the exact path
        if (!(tn > tf) && !(tf < 0.0f) && !(tn > hitCull) && usable(S)) {
            if (S.arootRef & LEAF_BIT) queue_triangle(col, stride, S.arootRef);     // one-triangle scene
            else cur = S.arootRef;
        }
    }
    __device__ __forceinline__ bool can_step() const { return cur != REF_NONE && npend <= PEND_MAX - 4; }
    __device__ __forceinline__ bool wants_test() const { return npend > 0 && (cur == REF_NONE || npend > PEND_MAX - 4); }
    __device__ __forceinline__ bool finished() const { return cur == REF_NONE && npend == 0; }

    // cur is a 4-wide node: decode and test its (up to) four child boxes; leaf children that
    // are hit are queued, interior children are entered nearest first, the others deferred.
    // With 1/d finite, t along axis k at quantised coordinate q is q*(scale_k/d_k) + (origin_k - o_k)/d_k.
    // A child counts as hit unless its box provably is not: max(entry, 0) > min(exit, hitCull)
    // (a NaN entry counts as 0, a NaN exit as hitCull).
    __device__ __forceinline__ void node_step(const SceneD &S, uint32_t *col, const int stride, uint32_t *ovf, uint32_t &nNode) {
        const float4 *rec = RT_REC(S.anodes, cur, 64u);
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
        node_process(S, q0, q1, q2, q3, col, stride, ovf, nNode);
    }
    // the step proper, for the node whose record is q0..q3
    __device__ __forceinline__ void node_process(const SceneD &S, const float4 q0, const float4 q1, const float4 q2, const float4 q3,
                                                 uint32_t *col, const int stride, uint32_t *ovf, uint32_t &nNode) {
        const float Ax = q0.w * inv.x, Bx = (q0.x - o.x) * inv.x;
        const float Ay = q3.z * inv.y, By = (q0.y - o.y) * inv.y;
        const float Az = q3.w * inv.z, Bz = (q0.z - o.z) * inv.z;
        const uint32_t lox = __float_as_uint(q2.x), loy = __float_as_uint(q2.y), loz = __float_as_uint(q2.z);
        const uint32_t hix = __float_as_uint(q2.w), hiy = __float_as_uint(q3.x), hiz = __float_as_uint(q3.y);
        const bool ngx = inv.x < 0.0f, ngy = inv.y < 0.0f, ngz = inv.z < 0.0f;
        const uint32_t nrx = ngx ? hix : lox, frx = ngx ? lox : hix;
        const uint32_t nry = ngy ? hiy : loy, fry = ngy ? loy : hiy;
        const uint32_t nrz = ngz ? hiz : loz, frz = ngz ? loz : hiz;
        uint32_t r0 = __float_as_uint(q1.x), r1 = __float_as_uint(q1.y), r2 = __float_as_uint(q1.z), r3 = __float_as_uint(q1.w);
        // per child: clamped entry distance and whether the box is hit.  An empty slot (REF_NONE) has an
        // inverted box, but a ray for which that box still counts as hit (infinite or NaN distances) must
        // not follow it: as a signed number REF_NONE is -1, interior refs are >= 0, leaf refs < -1.
#define RT_CHILD(sh, k, hit) \
        const float k = fmaxf(fmaxf(rtm::madd((float)((nrx >> sh) & 0xFFu), Ax, Bx), fmaxf(rtm::madd((float)((nry >> sh) & 0xFFu), Ay, By), \
                                    rtm::madd((float)((nrz >> sh) & 0xFFu), Az, Bz))), 0.0f); \
        const bool hit = !(k > fminf(fminf(rtm::madd((float)((frx >> sh) & 0xFFu), Ax, Bx), fminf(rtm::madd((float)((fry >> sh) & 0xFFu), Ay, By), \
                                    rtm::madd((float)((frz >> sh) & 0xFFu), Az, Bz))), hitCull));
        RT_CHILD(0, t0, hit0) RT_CHILD(8, t1, hit1) RT_CHILD(16, t2, hit2) RT_CHILD(24, t3, hit3)
#undef RT_CHILD
#ifdef RT_EXP_EMPTY
        // experiment: would a child be hit if the closest hit so far were not known (hitCull = +inf)?
#define RT_CHILD_NC(sh, k) !(k > fminf(rtm::madd((float)((frx >> sh) & 0xFFu), Ax, Bx), fminf(rtm::madd((float)((fry >> sh) & 0xFFu), Ay, By), \
                                    rtm::madd((float)((frz >> sh) & 0xFFu), Az, Bz))))
        dbgWouldHit = (RT_CHILD_NC(0, t0) && r0 != REF_NONE) || (RT_CHILD_NC(8, t1) && r1 != REF_NONE) || (RT_CHILD_NC(16, t2) && r2 != REF_NONE) || (RT_CHILD_NC(24, t3) && r3 != REF_NONE);
#undef RT_CHILD_NC
#endif
        if (COUNT) nNode += 4;
        // hit leaf children are queued for the triangle pass (stored with the leaf bit, removed when taken out)
        // Without branches: every child is written at the slot the next queued one would take, so a
        // child that is not queued is overwritten by the next one that is, or stays above the top
        // (a step is only taken with npend <= PEND_MAX - 4, so all four slots exist).
        {
            const int c0 = (hit0 && (int32_t)r0 < -1) ? 1 : 0, c1 = (hit1 && (int32_t)r1 < -1) ? 1 : 0;
            const int c2 = (hit2 && (int32_t)r2 < -1) ? 1 : 0, c3 = (hit3 && (int32_t)r3 < -1) ? 1 : 0;
            uint32_t *q = col + npend * stride;
            q[0] = r0; q += c0 * stride;
            q[0] = r1; q += c1 * stride;
            q[0] = r2; q += c2 * stride;
            q[0] = r3;
            npend += c0 + c1 + c2 + c3;
        }
        // hit interior children, farthest first (the others sort last)
        const bool in0 = hit0 && (int32_t)r0 >= 0, in1 = hit1 && (int32_t)r1 >= 0;
        const bool in2 = hit2 && (int32_t)r2 >= 0, in3 = hit3 && (int32_t)r3 >= 0;
        float k0 = in0 ? t0 : -INFINITY, k1 = in1 ? t1 : -INFINITY, k2 = in2 ? t2 : -INFINITY, k3 = in3 ? t3 : -INFINITY;
        const int m = (in0 ? 1 : 0) + (in1 ? 1 : 0) + (in2 ? 1 : 0) + (in3 ? 1 : 0);
#define RT_CSWAP(ka, ra, kb, rb) { const bool sw = kb > ka; const float tk = sw ? kb : ka; kb = sw ? ka : kb; ka = tk; \
                                   const uint32_t tr = sw ? rb : ra; rb = sw ? ra : rb; ra = tr; }
        RT_CSWAP(k0, r0, k1, r1) RT_CSWAP(k2, r2, k3, r3) RT_CSWAP(k0, r0, k2, r2) RT_CSWAP(k1, r1, k3, r3) RT_CSWAP(k1, r1, k2, r2)
#undef RT_CSWAP
        if (m == 0) {
            cur = (sp > 0) ? pop(S, col, stride, ovf) : REF_NONE;
        } else {
            // r0..r(m-2) are deferred (the nearest of them on top), r(m-1) is entered now
            if (sp + 3 <= S.aldsEntries) {
                // room for three: store them all, the ones beyond m-1 lie above the new top
                uint32_t *dst = col + (PEND_MAX + sp) * stride;
                dst[0] = r0; dst[stride] = r1; dst[2 * stride] = r2;
                sp += m - 1;
            } else {
                if (m > 1) push(S, col, stride, ovf, r0);
                if (m > 2) push(S, col, stride, ovf, r1);
                if (m > 3) push(S, col, stride, ovf, r2);
            }
            cur = (m == 1) ? r0 : ((m == 2) ? r1 : ((m == 3) ? r2 : r3));
        }
    }

    // tests one queued triangle with the reference's hitTriangle arithmetic
    __device__ __forceinline__ void test_step(const SceneD &S, const uint32_t *col, const int stride, uint32_t &nTri) {
        const uint32_t ti = unqueue_triangle(col, stride);
        const float4 *rec = RT_REC(S.tris, ti, 48u);
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        const float dis = triangle_test(q0, q1, q2, o, d);
        if (COUNT) nTri++;
        if (dis > 0.0f) {
            if (dis < hitMin) {
                hitMin = dis; best = ti; tie = false;
                hitCull = rtm::madd(hitMin, 1.0001f, 1.0e-4f);
            } else if (dis == hitMin && ti != best) {
                tie = true;
            }
        }
    }
};

// The reference reaches triangle `tri` iff every box on the path from the root to its leaf
// passes IntersectBound.  Those boxes are nested (each is the exact min/max of the triangles
// below it), and for a ray with finite 1/d the entry distance (b_near - o)*inv can only
// decrease and the exit distance only increase when a box grows, because fp32 subtraction and
// multiplication are monotonic.  IntersectBound passes exactly when every entry distance is
// <= every other axis' exit distance and all exit distances are > 0, so a pass at the leaf
// implies a pass at every ancestor: testing the leaf's own box (the min/max of the triangle's
// vertices, BVH.h:42-54) is enough.
template <typename SceneT>
__device__ __forceinline__ bool verify_reference_reaches(const SceneT &S, const vec3 o, const vec3 inv, uint32_t tri) {
    const float4 *rec = RT_REC(S.tris, tri, 48u);
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
    const float mnx = fminf(fminf(q0.x, q0.w), q1.z), mxx = fmaxf(fmaxf(q0.x, q0.w), q1.z);
    const float mny = fminf(fminf(q0.y, q1.x), q1.w), mxy = fmaxf(fmaxf(q0.y, q1.x), q1.w);
    const float mnz = fminf(fminf(q0.z, q1.y), q2.x), mxz = fmaxf(fmaxf(q0.z, q1.y), q2.x);
    float a, b, c;
    return slab_test(mnx, mny, mnz, mxx, mxy, mxz, o, inv, inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f, a, b, c);
}

}  // namespace rtd
