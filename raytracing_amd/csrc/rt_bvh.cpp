// Host-side BVH construction with the reference's exact node order, and the reference's
// texture wire format.
//
// Reference behaviour reproduced (file:line in the reference source directory):
//   BVH.h:117-156   BVHTree::build   - triangle centres, root
//   BVH.h:188-264   BVHTree::split   - bounds of the range, longest-axis rule, median
//                                      selection with std::nth_element (BVH.h:222)
//   BVH.h:267-287   buildLinerTree/addLinearNode - pre-order flattening
//   ObjectTexture.h:69-182 getTexture - 33 floats per triangle, 9 per node
//
// Design: the recursion works on 40-byte selection records (centre, id, triangle bounds)
// instead of the 144-byte triangles, writes nodes straight into their pre-order slots (a
// range of c triangles always occupies 2c-1 consecutive nodes, so the right child of node k
// whose left range has `mid` triangles is k + 2*mid), runs the two halves of the top levels
// on separate threads, and permutes the triangle array once at the end.  The order of the
// result depends only on the outcomes of the comparisons made by the selection routine, so
// `select_nth` below re-implements libstdc++'s introselect (GCC 11 bits/stl_algo.h,
// bits/stl_heap.h) operation for operation; it must not be replaced by std::nth_element,
// whose permutation is implementation-defined.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "rt_internal.h"

namespace {

struct Rec {
    float c[3];     // Triangle::center
    uint32_t id;    // index into the caller's array
    float mn[3];    // Triangle::getMin()
    float mx[3];    // Triangle::getMax()
};

struct Less {
    int dim;
    bool operator()(const Rec &a, const Rec &b) const { return a.c[dim] < b.c[dim]; }
};

inline void swap_rec(Rec *a, Rec *b) { Rec t = *a; *a = *b; *b = t; }

// ---- libstdc++ introselect, restated on Rec* ---------------------------------------

void move_median_to_first(Rec *result, Rec *a, Rec *b, Rec *c, Less lt) {
    if (lt(*a, *b)) {
        if (lt(*b, *c)) swap_rec(result, b);
        else if (lt(*a, *c)) swap_rec(result, c);
        else swap_rec(result, a);
    } else if (lt(*a, *c)) swap_rec(result, a);
    else if (lt(*b, *c)) swap_rec(result, c);
    else swap_rec(result, b);
}

Rec *unguarded_partition(Rec *first, Rec *last, Rec *pivot, Less lt) {
    while (true) {
        while (lt(*first, *pivot)) ++first;
        --last;
        while (lt(*pivot, *last)) --last;
        if (!(first < last)) return first;
        swap_rec(first, last);
        ++first;
    }
}

Rec *partition_pivot(Rec *first, Rec *last, Less lt) {
    Rec *mid = first + (last - first) / 2;
    move_median_to_first(first, first + 1, mid, last - 1, lt);
    return unguarded_partition(first + 1, last, first, lt);
}

void push_heap_hole(Rec *first, int64_t hole, int64_t top, const Rec &value, Less lt) {
    int64_t parent = (hole - 1) / 2;
    while (hole > top && lt(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

void adjust_heap(Rec *first, int64_t hole, int64_t len, const Rec &value, Less lt) {
    const int64_t top = hole;
    int64_t second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (lt(first[second], first[second - 1])) second--;
        first[hole] = first[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    push_heap_hole(first, hole, top, value, lt);
}

void make_heap(Rec *first, Rec *last, Less lt) {
    if (last - first < 2) return;
    const int64_t len = last - first;
    int64_t parent = (len - 2) / 2;
    while (true) {
        Rec value = first[parent];
        adjust_heap(first, parent, len, value, lt);
        if (parent == 0) return;
        parent--;
    }
}

void heap_select(Rec *first, Rec *middle, Rec *last, Less lt) {
    make_heap(first, middle, lt);
    for (Rec *i = middle; i < last; ++i)
        if (lt(*i, *first)) {
            Rec value = *i;         // __pop_heap(first, middle, i)
            *i = *first;
            adjust_heap(first, 0, middle - first, value, lt);
        }
}

void insertion_sort(Rec *first, Rec *last, Less lt) {
    if (first == last) return;
    for (Rec *i = first + 1; i != last; ++i) {
        if (lt(*i, *first)) {
            Rec val = *i;
            std::memmove(first + 1, first, (size_t)(i - first) * sizeof(Rec));
            *first = val;
        } else {
            Rec val = *i;           // __unguarded_linear_insert
            Rec *hole = i;
            Rec *next = i - 1;
            while (lt(val, *next)) {
                *hole = *next;
                hole = next;
                --next;
            }
            *hole = val;
        }
    }
}

inline int floor_log2(int64_t n) {
    int k = 0;
    while (n > 1) { n >>= 1; k++; }
    return k;
}

void select_nth(Rec *first, Rec *nth, Rec *last, Less lt) {
    if (first == last || nth == last) return;
    int64_t depth_limit = (int64_t)floor_log2(last - first) * 2;
    while (last - first > 3) {
        if (depth_limit == 0) {
            heap_select(first, nth + 1, last, lt);
            swap_rec(first, nth);
            return;
        }
        --depth_limit;
        Rec *cut = partition_pivot(first, last, lt);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertion_sort(first, last, lt);
}

// ---- libstdc++ std::sort (introsort + final insertion sort), restated on Rec* ----------

void sort_heap(Rec *first, Rec *last, Less lt) {
    while (last - first > 1) {
        --last;
        Rec value = *last;          // __pop_heap(first, last, last)
        *last = *first;
        adjust_heap(first, 0, last - first, value, lt);
    }
}

void unguarded_linear_insert(Rec *last, Less lt) {
    Rec val = *last;
    Rec *next = last - 1;
    while (lt(val, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

void introsort_loop(Rec *first, Rec *last, int64_t depth_limit, Less lt) {
    while (last - first > 16) {
        if (depth_limit == 0) {     // __partial_sort(first, last, last)
            heap_select(first, last, last, lt);
            sort_heap(first, last, lt);
            return;
        }
        --depth_limit;
        Rec *cut = partition_pivot(first, last, lt);
        introsort_loop(cut, last, depth_limit, lt);
        last = cut;
    }
}

void sort_recs(Rec *first, Rec *last, Less lt) {
    if (first == last) return;
    introsort_loop(first, last, (int64_t)floor_log2(last - first) * 2, lt);
    if (last - first > 16) {        // __final_insertion_sort
        insertion_sort(first, first + 16, lt);
        for (Rec *i = first + 16; i != last; ++i) unguarded_linear_insert(i, lt);
    } else {
        insertion_sort(first, last, lt);
    }
}

// ---- the recursion of BVHTree::split ------------------------------------------------

// class Bound of BVH.h:157-185 with its getArea() as written (`dif.y + dif.z`, not a product)
struct SweepBox {
    float mx[3], mn[3];
    void set(const Rec &r) { for (int c = 0; c < 3; c++) { mx[c] = r.mx[c]; mn[c] = r.mn[c]; } }
    void unite(const SweepBox &a, const Rec &r) {
        for (int c = 0; c < 3; c++) { mx[c] = (a.mx[c] < r.mx[c]) ? r.mx[c] : a.mx[c]; mn[c] = (r.mn[c] < a.mn[c]) ? r.mn[c] : a.mn[c]; }
    }
    float area() const {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        const float s = dx * dy + dx * dz + dy + dz;
        return (float)(2.0 * (double)s);
    }
};

struct Build {
    Rec *recs;
    RtLinearBVHNode *nodes;
    bool sweep = false;             // split rule of BVH.h:225-256 instead of the median
    std::vector<SweepBox> left, right;

    void range(int64_t begin, int64_t end, int64_t k, int fork_levels) {
        RtLinearBVHNode &nd = nodes[k];
        // BVH.h:190-197: fold in range order, max(a,b) = a<b ? b : a, min(a,b) = b<a ? b : a
        float mx[3] = {recs[begin].mx[0], recs[begin].mx[1], recs[begin].mx[2]};
        float mn[3] = {recs[begin].mn[0], recs[begin].mn[1], recs[begin].mn[2]};
        for (int64_t i = begin + 1; i < end; i++) {
            const Rec &r = recs[i];
            for (int c = 0; c < 3; c++) {
                if (mx[c] < r.mx[c]) mx[c] = r.mx[c];
                if (r.mn[c] < mn[c]) mn[c] = r.mn[c];
            }
        }
        for (int c = 0; c < 3; c++) { nd.maxBound[c] = mx[c]; nd.minBound[c] = mn[c]; }
        const int64_t count = end - begin;
        if (count == 1) {                   // BVH.h:207-211
            nd.triangleNum = 1;
            nd.offsetOrRc = (int32_t)begin;
            nd.axis = -1;
            return;
        }
        // BVHNode::calDimension, BVH.h:90-95
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        int dim;
        if (dx > dy && dx > dz) dim = 0;
        else if (dy > dz) dim = 1;
        else dim = 2;
        int64_t mid = count / 2;            // BVH.h:219
        if (!sweep) {
            select_nth(recs + begin, recs + begin + mid, recs + end, Less{dim});
        } else {
            // BVH.h:225-256: sort along each axis, sweep prefix and suffix boxes, keep the split
            // with the smallest areaLeft + areaRight (first one on ties), sort along its axis
            int axis = 0;
            mid = 0;
            float area = 10000000000.0f;
            Rec *first = recs + begin, *last = recs + end;
            for (int i = 0; i < 3; i++) {
                sort_recs(first, last, Less{i});
                left[0].set(first[0]);
                right[(size_t)count - 1].set(first[count - 1]);
                for (int64_t j = 1; j < count; j++) {
                    left[(size_t)j].unite(left[(size_t)j - 1], first[j]);
                    right[(size_t)(count - 1 - j)].unite(right[(size_t)(count - j)], first[count - 1 - j]);
                }
                for (int64_t j = 0; j < count - 1; j++) {
                    const float area_t = left[(size_t)j].area() + right[(size_t)j + 1].area();
                    if (area_t < area) { area = area_t; axis = i; mid = j; }
                }
            }
            sort_recs(first, last, Less{axis});
            mid++;
            dim = axis;
        }
        nd.triangleNum = (int32_t)count;
        nd.offsetOrRc = (int32_t)(k + 2 * mid);
        nd.axis = dim;
        if (fork_levels > 0 && count > 65536 && !sweep) {
            std::thread left([=] { range(begin, begin + mid, k + 1, fork_levels - 1); });
            range(begin + mid, end, k + 2 * mid, fork_levels - 1);
            left.join();
        } else {
            range(begin, begin + mid, k + 1, 0);
            range(begin + mid, end, k + 2 * mid, 0);
        }
    }
};

inline float min3(float a, float b, float c) {     // min(min(a,b),c) with std::min's tie rule
    float m = (b < a) ? b : a;
    return (c < m) ? c : m;
}
inline float max3(float a, float b, float c) {
    float m = (a < b) ? b : a;
    return (m < c) ? c : m;
}

}  // namespace

extern "C" {

static int build_tree(RtTriangle *tris, int64_t n, RtLinearBVHNode *nodes, int64_t node_capacity, int64_t *node_count, bool sweep);

int rt_bvh_build(RtTriangle *tris, int64_t n, RtLinearBVHNode *nodes, int64_t node_capacity, int64_t *node_count) {
    return build_tree(tris, n, nodes, node_capacity, node_count, false);
}

int rt_bvh_build_sweep(RtTriangle *tris, int64_t n, RtLinearBVHNode *nodes, int64_t node_capacity, int64_t *node_count) {
    return build_tree(tris, n, nodes, node_capacity, node_count, true);
}

static int build_tree(RtTriangle *tris, int64_t n, RtLinearBVHNode *nodes, int64_t node_capacity,
                      int64_t *node_count, bool sweep) {
    if (!tris || !nodes || n <= 0) { rt::set_global_error("rt_bvh_build: empty scene or null pointer"); return RT_E_INVALID; }
    if (n > (int64_t)1 << 30) { rt::set_global_error("rt_bvh_build: more than 2^30 triangles"); return RT_E_INVALID; }
    if (node_capacity < 2 * n - 1) { rt::set_global_error("rt_bvh_build: node_capacity < 2n-1"); return RT_E_INVALID; }
    if (sweep && n > RT_SWEEP_MAX_TRIANGLES) {
        // its trees can be as deep as they are large (the cost ignores triangle counts) and every node sorts its range four times
        rt::set_global_error("rt_bvh_build_sweep: more than RT_SWEEP_MAX_TRIANGLES (16384) triangles");
        return RT_E_INVALID;
    }
    std::vector<Rec> recs((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        Rec &r = recs[(size_t)i];
        const RtTriangle &t = tris[i];
        r.id = (uint32_t)i;
        for (int c = 0; c < 3; c++) {
            r.mn[c] = min3(t.v[0][c], t.v[1][c], t.v[2][c]);   // BVH.h:42-47
            r.mx[c] = max3(t.v[0][c], t.v[1][c], t.v[2][c]);   // BVH.h:49-54
            float s = r.mx[c] + r.mn[c];                       // BVH.h:36-39
            r.c[c] = s / 2.0f;
        }
    }
    Build b;
    b.recs = recs.data(); b.nodes = nodes; b.sweep = sweep;
    if (sweep) { b.left.resize((size_t)n); b.right.resize((size_t)n); }
    b.range(0, n, 0, sweep ? 0 : std::min(rt::host_fork_levels(), 3));
    // permute the caller's triangles into tree order and store their centres
    std::vector<RtTriangle> tmp(tris, tris + n);
    for (int64_t i = 0; i < n; i++) {
        const Rec &r = recs[(size_t)i];
        tris[i] = tmp[r.id];
        for (int c = 0; c < 3; c++) tris[i].center[c] = r.c[c];
    }
    if (node_count) *node_count = 2 * n - 1;
    return RT_OK;
}

// Diagnostics: the selection routine alone, on keys with ids carried along, so that a test can
// compare its permutation with std::nth_element's on the same input.
int rt_selftest_nth_element(float *keys, int32_t *ids, int64_t n, int64_t nth) {
    if (!keys || !ids || n < 0 || nth < 0 || nth > n) { rt::set_global_error("rt_selftest_nth_element: bad argument"); return RT_E_INVALID; }
    std::vector<Rec> v((size_t)n);
    for (int64_t i = 0; i < n; i++) { v[(size_t)i] = Rec{}; v[(size_t)i].c[0] = keys[i]; v[(size_t)i].id = (uint32_t)ids[i]; }
    select_nth(v.data(), v.data() + nth, v.data() + n, Less{0});
    for (int64_t i = 0; i < n; i++) { keys[i] = v[(size_t)i].c[0]; ids[i] = (int32_t)v[(size_t)i].id; }
    return RT_OK;
}

int rt_selftest_sort(float *keys, int32_t *ids, int64_t n) {
    if (!keys || !ids || n < 0) { rt::set_global_error("rt_selftest_sort: bad argument"); return RT_E_INVALID; }
    std::vector<Rec> v((size_t)n);
    for (int64_t i = 0; i < n; i++) { v[(size_t)i] = Rec{}; v[(size_t)i].c[0] = keys[i]; v[(size_t)i].id = (uint32_t)ids[i]; }
    sort_recs(v.data(), v.data() + n, Less{0});
    for (int64_t i = 0; i < n; i++) { keys[i] = v[(size_t)i].c[0]; ids[i] = (int32_t)v[(size_t)i].id; }
    return RT_OK;
}

int rt_texture_dims(int64_t count, int per, int *x, int *y) {
    if (count < 0 || per <= 0 || !x || !y) { rt::set_global_error("rt_texture_dims: bad argument"); return RT_E_INVALID; }
    // int y = ceilf(sqrt(count * per)); int x = (y + per - 1) / per;   ObjectTexture.h:79-80
    int yy = (int)ceilf((float)std::sqrt((double)(count * per)));
    *y = yy;
    *x = (yy + per - 1) / per;
    return RT_OK;
}

int rt_pack_triangle_texture(const RtTriangle *tris, int64_t n, float *out, int x, int y) {
    if (!tris || !out || n < 0 || (int64_t)x * y < n) { rt::set_global_error("rt_pack_triangle_texture: bad argument"); return RT_E_INVALID; }
    std::memset(out, 0, (size_t)x * y * 33 * sizeof(float));
    for (int64_t i = 0; i < n; i++) {
        float *o = out + i * 33;        // rows are x*33 wide and contiguous: record i starts at i*33
        const RtTriangle &t = tris[i];
        for (int k = 0; k < 3; k++) {
            o[k * 8 + 0] = t.v[k][0]; o[k * 8 + 1] = t.v[k][1]; o[k * 8 + 2] = t.v[k][2];
            o[k * 8 + 3] = t.n[k][0]; o[k * 8 + 4] = t.n[k][1]; o[k * 8 + 5] = t.n[k][2];
            o[k * 8 + 6] = t.tex[k][0]; o[k * 8 + 7] = t.tex[k][1];
        }
        for (int c = 0; c < 3; c++) { o[24 + c] = t.ka[c]; o[27 + c] = t.kd[c]; o[30 + c] = t.ks[c]; }
    }
    return RT_OK;
}

int rt_pack_bvh_texture(const RtLinearBVHNode *nodes, int64_t m, float *out, int x, int y) {
    if (!nodes || !out || m < 0 || (int64_t)x * y < m) { rt::set_global_error("rt_pack_bvh_texture: bad argument"); return RT_E_INVALID; }
    std::memset(out, 0, (size_t)x * y * 9 * sizeof(float));
    for (int64_t i = 0; i < m; i++) {
        float *o = out + i * 9;
        const RtLinearBVHNode &nd = nodes[i];
        for (int c = 0; c < 3; c++) { o[c] = nd.maxBound[c]; o[3 + c] = nd.minBound[c]; }
        o[6] = (float)nd.triangleNum;
        o[7] = (float)nd.offsetOrRc;
        o[8] = (float)nd.axis;
    }
    return RT_OK;
}

}  // extern "C"
