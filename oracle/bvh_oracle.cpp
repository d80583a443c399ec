/*
 * bvh_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's BVH builder, flattener and texture packer:
 *   BVH.h:26-57    Triangle (144 B), getMin/getMax/getCenter
 *   BVH.h:81-96    BVHNode, calDimension
 *   BVH.h:117-156  BVHTree::build
 *   BVH.h:188-264  BVHTree::split  (median split with std::nth_element, BVH.h:222)
 *   BVH.h:157-185, 225-256  class Bound and the sweep split the author left commented out
 *                  ("SAH"), offered as a second build mode
 *   BVH.h:267-287  buildLinerTree / addLinearNode
 *   ObjectTexture.h:69-182  getTexture(BVHTree&, ...)  (33 floats/triangle, 9 floats/node)
 *   Scene.h:54-65  pushModel's scale-then-translate
 *
 * The reference itself cannot be compiled in this environment (glm, glad, Assimp and
 * stb_image headers are absent and Shader::setUint is undefined), so this file follows
 * BVH.h statement by statement instead.  The one library routine that decides the node
 * order, std::nth_element, is called here exactly as the reference calls it (same element
 * type size class, same comparator), so the permutation is libstdc++'s own.
 *
 * PARITY STATUS: pinned by the known answers recorded from the reference builder for its
 * bundled scene (tests/golden/demo_bvh_known_answers.json); the shipped builder
 * (raytracing_amd/csrc/rt_bvh.cpp) is compared against this file bit for bit.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct OTriangle {              // BVH.h:26-33, sizeof == 144
    float v[3][3];
    float n[3][3];
    float tex[3][2];
    float ka[3], kd[3], ks[3];
    float center[3];
};
static_assert(sizeof(OTriangle) == 144, "Triangle layout");

struct OLinearNode {            // BVH.h:99-105, sizeof == 36
    float minBound[3], maxBound[3];
    int triangleNum, offsetOrRc, axis;
};
static_assert(sizeof(OLinearNode) == 36, "LinearBVHNode layout");

struct ONode {                  // BVH.h:81-88
    float minBound[3], maxBound[3];
    int dimension = -1;
    int triangleNum = 0;
    int index = 0;
    ONode *lc = nullptr, *rc = nullptr;
};

// BVH.h:42-54 -- std::min/std::max folds in this exact order
inline void tri_min(const OTriangle &t, float out[3]) {
    for (int k = 0; k < 3; k++) out[k] = std::min(std::min(t.v[0][k], t.v[1][k]), t.v[2][k]);
}
inline void tri_max(const OTriangle &t, float out[3]) {
    for (int k = 0; k < 3; k++) out[k] = std::max(std::max(t.v[0][k], t.v[1][k]), t.v[2][k]);
}

// class Bound, BVH.h:157-185.  getArea() is the reference's own formula, `dif.y + dif.z` where a
// surface area would have dif.y * dif.z (BVH.h:181-184); the factor 2.0 is a double, applied to
// the float sum and rounded back to float (exact).
struct OBound {
    float maxP[3], minP[3];
    static OBound of(const OTriangle &t) { OBound b; tri_max(t, b.maxP); tri_min(t, b.minP); return b; }
    OBound unite(const OBound &o) const {
        OBound r;
        for (int k = 0; k < 3; k++) { r.maxP[k] = std::max(maxP[k], o.maxP[k]); r.minP[k] = std::min(minP[k], o.minP[k]); }
        return r;
    }
    float area() const {
        const float dx = maxP[0] - minP[0], dy = maxP[1] - minP[1], dz = maxP[2] - minP[2];
        const float s = dx * dy + dx * dz + dy + dz;
        return (float)(2.0 * (double)s);
    }
};

struct Builder {
    OTriangle *tris;
    int n;
    bool sah = false;           // the block the author left commented out (BVH.h:225-256) instead of the median split
    int size = 0;
    std::vector<ONode *> pool;
    OLinearNode *linear = nullptr;
    int linearCnt = 0;

    ONode *newNode() { ONode *p = new ONode; pool.push_back(p); return p; }

    // BVH.h:188-264
    void split(ONode *now, OTriangle *begin, OTriangle *end) {
        tri_max(*begin, now->maxBound);
        tri_min(*begin, now->minBound);
        now->triangleNum = 1;
        for (OTriangle *it = begin + 1; it != end; it++) {
            float mx[3], mn[3];
            tri_max(*it, mx); tri_min(*it, mn);
            for (int k = 0; k < 3; k++) {
                now->maxBound[k] = std::max(now->maxBound[k], mx[k]);   // getMaxBound(a,b)=max(a,b)
                now->minBound[k] = std::min(now->minBound[k], mn[k]);
            }
            now->triangleNum++;
        }
        if (begin + 1 == end) {
            now->triangleNum = 1;
            now->index = (int)(begin - tris);
            return;
        }
        // calDimension, BVH.h:90-95
        float dx = now->maxBound[0] - now->minBound[0];
        float dy = now->maxBound[1] - now->minBound[1];
        float dz = now->maxBound[2] - now->minBound[2];
        if (dx > dy && dx > dz) now->dimension = 0;
        else if (dy > dz) now->dimension = 1;
        else now->dimension = 2;

        int mid = now->triangleNum / 2;
        if (!sah) {
            int dim = now->dimension;
            std::nth_element(begin, begin + mid, end,
                             [dim](const OTriangle &a, const OTriangle &b) { return a.center[dim] < b.center[dim]; });
        } else {
            // BVH.h:225-256 ("SAH" in the author's comment) in place of BVH.h:219-222: sort along each
            // axis, sweep prefix / suffix boxes, keep the split with the smallest areaLeft + areaRight
            // (no triangle counts in the cost), first one found on ties; then sort along the winner.
            int cnt = (int)(end - begin);
            int axis = 0;
            mid = 0;
            float area = 10000000000.0f;
            for (int i = 0; i < 3; i++) {
                std::sort(begin, end, [i](const OTriangle &a, const OTriangle &b) { return a.center[i] < b.center[i]; });
                std::vector<OBound> boundLeft(cnt), boundRight(cnt);
                boundLeft[0] = OBound::of(*begin);
                boundRight[cnt - 1] = OBound::of(*(end - 1));
                for (int j = 1; j < cnt; j++) {
                    boundLeft[j] = boundLeft[j - 1].unite(OBound::of(*(begin + j)));
                    boundRight[cnt - 1 - j] = boundRight[cnt - 1 - j + 1].unite(OBound::of(*(end - 1 - j)));
                }
                for (int j = 0; j < cnt - 1; j++) {
                    float area_t = boundLeft[j].area() + boundRight[j + 1].area();
                    if (area_t < area) { area = area_t; axis = i; mid = j; }
                }
            }
            std::sort(begin, end, [axis](const OTriangle &a, const OTriangle &b) { return a.center[axis] < b.center[axis]; });
            mid++;
            now->dimension = axis;
        }
        now->lc = newNode(); size++;
        split(now->lc, begin, begin + mid);
        now->rc = newNode(); size++;
        split(now->rc, begin + mid, end);
    }

    // BVH.h:273-287
    void addLinearNode(ONode *now) {
        int k = linearCnt++;
        for (int c = 0; c < 3; c++) {
            linear[k].maxBound[c] = now->maxBound[c];
            linear[k].minBound[c] = now->minBound[c];
        }
        linear[k].triangleNum = now->triangleNum;
        linear[k].offsetOrRc = now->index;
        linear[k].axis = now->dimension;
        if (linear[k].triangleNum == 1) return;
        if (now->lc) addLinearNode(now->lc);
        linear[k].offsetOrRc = linearCnt;
        if (now->rc) addLinearNode(now->rc);
    }
};

}  // namespace

extern "C" {

int bvh_oracle_sizeof_triangle() { return (int)sizeof(OTriangle); }
int bvh_oracle_sizeof_node() { return (int)sizeof(OLinearNode); }

// Scene.h:57-60: Position = Position * scale + translation (two roundings), in place.
void bvh_oracle_push_model(float *positions, int nverts, const float scale[3], const float trans[3]) {
    for (int i = 0; i < nverts; i++)
        for (int k = 0; k < 3; k++) {
            volatile float p = positions[3 * i + k] * scale[k];
            positions[3 * i + k] = p + trans[k];
        }
}

// BVH.h:117-156 + 267-271.  tris: n records whose v/n/tex/ka/kd/ks are set (center is
// computed here, BVH.h:36-39); permuted in place.  nodes: room for 2n-1 records.
// Returns the node count (BVHTree::size), or 0 for an empty scene (the reference would
// dereference triangles[0]).
static int build_common(void *tris_v, int n, void *nodes_v, bool sah);

int bvh_oracle_build(void *tris_v, int n, void *nodes_v) { return build_common(tris_v, n, nodes_v, false); }

// The same with the split rule of BVH.h:225-256 (see Builder::split).  std::sort is called as the
// reference would call it, so the order among equal centres is libstdc++'s.  The cost has no
// triangle counts, so trees can be very deep and the build quadratic: small scenes only.
int bvh_oracle_build_sah(void *tris_v, int n, void *nodes_v) { return build_common(tris_v, n, nodes_v, true); }

// std::sort alone on records of the reference's size (keys[i] plays center[dim]).
void bvh_oracle_sort(float *keys, int *ids, int n) {
    struct Rec { float key; int id; char pad[136]; };
    std::vector<Rec> v(n);
    for (int i = 0; i < n; i++) { v[i].key = keys[i]; v[i].id = ids[i]; }
    std::sort(v.begin(), v.end(), [](const Rec &a, const Rec &b) { return a.key < b.key; });
    for (int i = 0; i < n; i++) { keys[i] = v[i].key; ids[i] = v[i].id; }
}

static int build_common(void *tris_v, int n, void *nodes_v, bool sah) {
    OTriangle *tris = (OTriangle *)tris_v;
    if (n <= 0) return 0;
    for (int i = 0; i < n; i++) {
        float mx[3], mn[3];
        tri_max(tris[i], mx); tri_min(tris[i], mn);
        for (int k = 0; k < 3; k++) {
            float c = mx[k] + mn[k];
            tris[i].center[k] = c / 2.0f;
        }
    }
    Builder b; b.tris = tris; b.n = n; b.sah = sah;
    ONode *root = b.newNode();
    b.size = 1;
    b.split(root, tris, tris + n);
    b.linear = (OLinearNode *)nodes_v;
    b.addLinearNode(root);
    for (ONode *p : b.pool) delete p;
    return b.size;
}

// Texture dimensions exactly as ObjectTexture.h:79-80 / 138-139 compute them.
void bvh_oracle_tex_dims(int count, int per, int *x_out, int *y_out) {
    int y = (int)ceilf((float)sqrt((double)(count * per)));
    int x = (y + per - 1) / per;
    *x_out = x; *y_out = y;
}

// ObjectTexture.h:82-115.  out has x*y*33 floats; texels past the last triangle are
// uninitialised heap in the reference and are set to NaN here so that a read shows up.
void bvh_oracle_pack_triangles(const void *tris_v, int n, float *out, int x, int y) {
    const OTriangle *tris = (const OTriangle *)tris_v;
    size_t total = (size_t)x * y * 33;
    for (size_t i = 0; i < total; i++) out[i] = std::numeric_limits<float>::quiet_NaN();
    int cnt = 0;
    for (int i = 0; i < y && cnt < n; i++) {
        for (int j = 0; j < x && cnt < n; j++) {
            float *o = out + (size_t)i * x * 33 + (size_t)j * 33;
            for (int k = 0; k < 3; k++) {
                for (int c = 0; c < 3; c++) o[k * 8 + c] = tris[cnt].v[k][c];
                for (int c = 0; c < 3; c++) o[k * 8 + 3 + c] = tris[cnt].n[k][c];
                for (int c = 0; c < 2; c++) o[k * 8 + 6 + c] = tris[cnt].tex[k][c];
            }
            for (int c = 0; c < 3; c++) {
                o[24 + c] = tris[cnt].ka[c];
                o[27 + c] = tris[cnt].kd[c];
                o[30 + c] = tris[cnt].ks[c];
            }
            cnt++;
        }
    }
}

// ObjectTexture.h:140-161 (max before min; ints stored as floats)
void bvh_oracle_pack_nodes(const void *nodes_v, int m, float *out, int x, int y) {
    const OLinearNode *nodes = (const OLinearNode *)nodes_v;
    size_t total = (size_t)x * y * 9;
    for (size_t i = 0; i < total; i++) out[i] = std::numeric_limits<float>::quiet_NaN();
    int cnt = 0;
    for (int i = 0; i < y && cnt < m; i++) {
        for (int j = 0; j < x && cnt < m; j++) {
            float *o = out + (size_t)i * x * 9 + (size_t)j * 9;
            for (int c = 0; c < 3; c++) {
                o[c] = nodes[cnt].maxBound[c];
                o[3 + c] = nodes[cnt].minBound[c];
            }
            o[6] = (float)nodes[cnt].triangleNum;
            o[7] = (float)nodes[cnt].offsetOrRc;
            o[8] = (float)nodes[cnt].axis;
            cnt++;
        }
    }
}

// Thin wrapper so tests can fuzz a selection routine against libstdc++'s nth_element on
// records of the reference's size: keys[i] plays center[dim]; ids are permuted alongside.
void bvh_oracle_nth_element(float *keys, int *ids, int n, int nth) {
    struct Rec { float key; int id; char pad[136]; };
    std::vector<Rec> v(n);
    for (int i = 0; i < n; i++) { v[i].key = keys[i]; v[i].id = ids[i]; }
    std::nth_element(v.begin(), v.begin() + nth, v.end(),
                     [](const Rec &a, const Rec &b) { return a.key < b.key; });
    for (int i = 0; i < n; i++) { keys[i] = v[i].key; ids[i] = v[i].id; }
}

}  // extern "C"
