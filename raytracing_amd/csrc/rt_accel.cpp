// Search index for the extend stage: a binned-SAH BVH over the uploaded triangles.
//
// The reference's own tree (BVH.h:188-264) is a median split on centroids with one triangle per
// leaf.  A handful of large triangles (the walls of the reference's room) make every subtree
// that contains one of them as large as the room, so a ray has to test ~190 boxes in a
// 1.3M-triangle scene although it touches only a few leaves.  The image does not depend on how
// the candidate triangles are FOUND, only on which triangle the reference would keep:
// hitBVH keeps the accepted triangle (hitTriangle, frag:255-278) with the smallest distance
// among those whose leaf it reaches, the first one met winning an exact tie (frag:396).
// The extend stage therefore searches this index for the closest accepted triangle, and then
// proves that the reference reaches it (one exact box test, see verify_reference_reaches in
// rt_device.hip); rays for which the proof fails are re-traced in the reference's order.
//
// Boxes here are grown by eps so that the search is conservative: a point accepted by
// hitTriangle lies within ~3e-6 * scale of the triangle (plane residual of the computed t plus
// the slop of the three edge tests), and eps = 2^-18 * scale (3.8e-6) times a shape factor that
// grows for thin triangles, whose computed normal is less accurate.  The outward rounding of the
// 8-bit child boxes adds more on top (a grid step is 1/255 of a node's extent).  Measured with
// tools/eps_sweep.sh: the parity tests (rays and images against the CPU checker at full size) still
// pass with 2^-25 and fail with 2^-27.  A smaller growth is fewer box hits: 2^-15 -> 2^-18 is
// +1 % rays per second on 1.3M triangles of 8e-3 scene units and +4 % on 10M of 3e-3.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "rt_accel.h"
#include "rt_internal.h"

namespace rt {

namespace {

constexpr int MAX_BINS = 64;     // measured: 16 -> 64 bins is 5 % fewer box tests per ray, more gives nothing

struct Prim {               // the builder partitions these records themselves, so every scan is sequential
    float mn[3], mx[3], c[3];
    uint32_t id;            // triangle number
};

struct Box {
    float mn[3], mx[3];
    void reset() { for (int k = 0; k < 3; k++) { mn[k] = INFINITY; mx[k] = -INFINITY; } }
    void grow(const float *a, const float *b) {
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], a[k]); mx[k] = std::max(mx[k], b[k]); }
    }
    void grow(const Box &o) { grow(o.mn, o.mx); }
    float half_area() const {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return (dx < 0.0f) ? 0.0f : dx * dy + dy * dz + dz * dx;
    }
};

struct TmpNode {
    Box lbox, rbox;
    uint32_t left, right;       // child refs: LEAF encodings or TmpNode indices
};

struct Builder {
    int leafMax = 1;
    Prim *prims;
    std::vector<TmpNode> nodes;
    std::atomic<uint32_t> next{0};

    static uint32_t leaf_ref(int64_t begin, int64_t count) {
        return ACCEL_LEAF_BIT | ((uint32_t)(count - 1) << 28) | (uint32_t)begin;
    }

    Box bounds(int64_t begin, int64_t end) const {
        Box b; b.reset();
        for (int64_t i = begin; i < end; i++) b.grow(prims[i].mn, prims[i].mx);
        return b;
    }

    // Triangles much larger than the rest (the walls of a room around a dense mesh) are split off
    // first, by size class: a split along an axis cannot tell an enclosing triangle from the ones
    // it encloses, and every subtree that keeps one gets a box as large as the room.  The classes
    // are a quarter, a sixteenth and a sixty-fourth of the scene's largest extent; a class is only
    // split off while it is the small part (at most a tenth of the range).
    uint32_t build_by_size(int64_t begin, int64_t end, Box &box, float sceneExtent, int cls) {
        const int64_t count = end - begin;
        if (cls >= 3 || count < 64) return build(begin, end, box, host_fork_levels());
        const float thr = sceneExtent * (cls == 0 ? 0.25f : (cls == 1 ? 0.0625f : 0.015625f));
        Prim *first = prims + begin, *last = prims + end;
        Prim *m = std::partition(first, last, [&](const Prim &p) {
            return std::max(p.mx[0] - p.mn[0], std::max(p.mx[1] - p.mn[1], p.mx[2] - p.mn[2])) >= thr;
        });
        const int64_t nLarge = m - first;
        if (nLarge == 0 || nLarge * 10 > count) return build_by_size(begin, end, box, sceneExtent, cls + 1);
        const uint32_t me = next.fetch_add(1);
        TmpNode nd;
        nd.left = build(begin, begin + nLarge, nd.lbox, 0);
        nd.right = build_by_size(begin + nLarge, end, nd.rbox, sceneExtent, cls + 1);
        nodes[me] = nd;
        box = nd.lbox; box.grow(nd.rbox);
        return me;
    }

    struct AxisBins { Box bb[MAX_BINS]; int64_t bc[MAX_BINS]; };
    static constexpr int64_t PAR_COUNT = (int64_t)1 << 18;     // ranges from this size on are scanned on several threads

    static int scan_threads(int64_t count) {
        if (count < PAR_COUNT) return 1;
        const int hw = host_threads();
        return (int)std::min<int64_t>(hw, count / (PAR_COUNT / 4));
    }
    template <typename Fn>
    static void scan_chunks(int64_t begin, int64_t end, int threads, Fn fn) {     // fn(thread, chunk begin, chunk end)
        if (threads <= 1) { fn(0, begin, end); return; }
        std::vector<std::thread> pool;
        const int64_t per = (end - begin + threads - 1) / threads;
        for (int t = 0; t < threads; t++) {
            const int64_t b = begin + t * per, e = std::min(end, b + per);
            if (b >= e) break;
            pool.emplace_back([=] { fn(t, b, e); });
        }
        for (auto &th : pool) th.join();
    }
    // box of the triangles and box of their centres (min/max are exact, so the split of the range
    // over threads does not change the result)
    void bounds2(int64_t begin, int64_t end, Box &box, Box &cb) const {
        const int T = scan_threads(end - begin);
        if (T <= 1) {
            box.reset(); cb.reset();
            for (int64_t i = begin; i < end; i++) { const Prim &p = prims[i]; box.grow(p.mn, p.mx); cb.grow(p.c, p.c); }
            return;
        }
        std::vector<Box> pb((size_t)T), pc((size_t)T);
        for (int t = 0; t < T; t++) { pb[(size_t)t].reset(); pc[(size_t)t].reset(); }
        scan_chunks(begin, end, T, [&](int t, int64_t b, int64_t e) {
            Box lb, lc; lb.reset(); lc.reset();
            for (int64_t i = b; i < e; i++) { const Prim &p = prims[i]; lb.grow(p.mn, p.mx); lc.grow(p.c, p.c); }
            pb[(size_t)t] = lb; pc[(size_t)t] = lc;
        });
        box.reset(); cb.reset();
        for (int t = 0; t < T; t++) { box.grow(pb[(size_t)t]); cb.grow(pc[(size_t)t]); }
    }
    // the bins of the three axes in one pass
    void bin_all(int64_t begin, int64_t end, const Box &cb, int NBINS, AxisBins out[3]) const {
        float lo[3], scale[3]; bool use[3];
        for (int ax = 0; ax < 3; ax++) {
            const float ext = cb.mx[ax] - cb.mn[ax];
            use[ax] = ext > 0.0f; lo[ax] = cb.mn[ax]; scale[ax] = use[ax] ? (float)NBINS / ext : 0.0f;
        }
        const int T = scan_threads(end - begin);
        std::vector<AxisBins> part((size_t)(T > 1 ? 3 * T : 0));
        auto clear = [&](AxisBins *a) { for (int ax = 0; ax < 3; ax++) for (int b = 0; b < NBINS; b++) { a[ax].bb[b].reset(); a[ax].bc[b] = 0; } };
        clear(out);
        scan_chunks(begin, end, T, [&](int t, int64_t b0, int64_t e0) {
            AxisBins *dst = (T > 1) ? &part[(size_t)3 * t] : out;
            if (T > 1) clear(dst);
            for (int64_t i = b0; i < e0; i++) {
                const Prim &p = prims[i];
                for (int ax = 0; ax < 3; ax++) {
                    if (!use[ax]) continue;
                    int b = (int)((p.c[ax] - lo[ax]) * scale[ax]);
                    b = b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
                    dst[ax].bb[b].grow(p.mn, p.mx); dst[ax].bc[b]++;
                }
            }
        });
        if (T > 1)
            for (int t = 0; t < T; t++)
                for (int ax = 0; ax < 3; ax++)
                    for (int b = 0; b < NBINS; b++) { out[ax].bb[b].grow(part[(size_t)3 * t + ax].bb[b]); out[ax].bc[b] += part[(size_t)3 * t + ax].bc[b]; }
    }

    // Returns the ref of the subtree over idx[begin,end) and its box.
    uint32_t build(int64_t begin, int64_t end, Box &box, int fork) {
        const int64_t count = end - begin;
        if (count <= 1) { box = bounds(begin, end); return leaf_ref(begin, count); }
        if (count == 2 && leafMax == 1) {       // nothing to choose: two leaves
            const uint32_t me = next.fetch_add(1);
            TmpNode nd;
            nd.lbox.reset(); nd.lbox.grow(prims[begin].mn, prims[begin].mx);
            nd.rbox.reset(); nd.rbox.grow(prims[begin + 1].mn, prims[begin + 1].mx);
            nd.left = leaf_ref(begin, 1); nd.right = leaf_ref(begin + 1, 1);
            nodes[me] = nd;
            box = nd.lbox; box.grow(nd.rbox);
            return me;
        }
        Box cb;
        bounds2(begin, end, box, cb);
        // binned SAH over the three axes; small nodes need fewer bins to tell all their splits apart
        const int NBINS = (int)std::min<int64_t>(MAX_BINS, std::max<int64_t>(8, 4 * count));
        float bestCost = INFINITY; int bestAxis = -1, bestSplit = -1;
        AxisBins bins[3];
        bin_all(begin, end, cb, NBINS, bins);
        for (int ax = 0; ax < 3; ax++) {
            if (!(cb.mx[ax] - cb.mn[ax] > 0.0f)) continue;
            const Box *bb = bins[ax].bb; const int64_t *bc = bins[ax].bc;
            float rightArea[MAX_BINS]; int64_t rightCount[MAX_BINS];
            Box acc; acc.reset(); int64_t cnt = 0;
            for (int b = NBINS - 1; b > 0; b--) { acc.grow(bb[b]); cnt += bc[b]; rightArea[b] = acc.half_area(); rightCount[b] = cnt; }
            acc.reset(); cnt = 0;
            for (int b = 0; b < NBINS - 1; b++) {
                acc.grow(bb[b]); cnt += bc[b];
                if (cnt == 0 || rightCount[b + 1] == 0) continue;
                const float cost = acc.half_area() * (float)cnt + rightArea[b + 1] * (float)rightCount[b + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestSplit = b; }
            }
        }
        const float parentArea = box.half_area();
        if (count <= leafMax) {
            const float leafCost = (float)count;
            const float splitCost = (bestAxis < 0 || !(parentArea > 0.0f)) ? INFINITY : 1.0f + bestCost / parentArea;
            if (leafCost <= splitCost) return leaf_ref(begin, count);
        }
        int64_t mid;
        if (bestAxis < 0) {
            mid = begin + count / 2;        // all centroids coincide: split by position in the list
        } else {
            const float lo = cb.mn[bestAxis], scale = (float)NBINS / (cb.mx[bestAxis] - cb.mn[bestAxis]);
            Prim *first = prims + begin, *last = prims + end;
            Prim *m = std::partition(first, last, [&](const Prim &p) {
                int b = (int)((p.c[bestAxis] - lo) * scale);
                b = b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
                return b <= bestSplit;
            });
            mid = begin + (m - first);
            if (mid == begin || mid == end) mid = begin + count / 2;
        }
        const uint32_t me = next.fetch_add(1);
        TmpNode nd;
        if (fork > 0 && count > 100000) {
            uint32_t l = 0; Box lb;
            std::thread t([&] { l = build(begin, mid, lb, fork - 1); });
            nd.right = build(mid, end, nd.rbox, fork - 1);
            t.join();
            nd.left = l; nd.lbox = lb;
        } else {
            nd.left = build(begin, mid, nd.lbox, 0);
            nd.right = build(mid, end, nd.rbox, 0);
        }
        nodes[me] = nd;
        return me;
    }
};


// fn(begin, end) over [0, n) on up to host_threads() threads
template <typename Fn>
void par_for(int64_t n, Fn fn) {
    const int64_t minChunk = 16384;
    const int threads = (int)std::min<int64_t>(host_threads(), (n + minChunk - 1) / minChunk);
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

// Collapses the binary tree into 4-wide nodes: a wide node starts from the two children of a
// binary node and repeatedly replaces its largest interior child by that child's two children.
// First a serial walk that fixes the shape and the depth-first numbering (slot 0's subtree follows
// its parent), then the boxes are quantised on all cores.
void collapse_wide(const Builder &b, uint32_t root, Accel &out) {
    struct Shape { uint32_t ref[4]; uint32_t from[4]; uint8_t right[4]; uint8_t n; };     // child c: ref, and the binary node + side holding its box
    struct Item { uint32_t old; uint32_t parent; int slot; int depth; };
    auto box_of = [&](uint32_t from, uint8_t right) -> const Box & { return right ? b.nodes[from].rbox : b.nodes[from].lbox; };
    // Which binary nodes become wide nodes is chosen so that the summed surface area of all wide
    // nodes - the expected number of wide nodes a random ray visits, four box tests each - is
    // smallest (the dynamic programme of Ylitie, Karras and Laine 2017 for wide BVHs, here with
    // leaves that cost nothing because a triangle is named by its parent's slot):
    //   cost[n][i]  smallest cost of the subtree of binary node n cut into at most i+1 subtrees
    //               that hang in slots of one wide node above it (i = 0, 1, 2)
    //   cost[n][0] = area(n) + min over k of cost[L][k-1] + cost[R][3-k]      n itself is a wide node
    //   cost[n][i] = min(cost[n][i-1], min over k of cost[L][k-1] + cost[R][i-k])
    // Children carry larger indices than their parents (a node takes its index before its
    // children are built), so one pass from the last node to the first fills the table.
    const uint32_t nBin = b.next.load();
    // Measured on MI355X: a fifth fewer nodes and 1-3 % fewer box tests per ray, but not faster (1.3M
    // triangles +0.7 %, 262k at 4K -2 %, 10.5M -0.7 %, 32 triangles -7 %), so the default stays the
    // largest-child-first collapse below; RT_AMD_COLLAPSE_DP=1 selects this one.
    const bool useDp = getenv("RT_AMD_COLLAPSE_DP") != nullptr;
    struct Dp { float c[3]; uint8_t k[3]; };       // k[j-2]: slots given to the left child when the node is cut into j = 2, 3, 4
    std::vector<Dp> dp(useDp ? nBin : 0);
    if (useDp) {
        auto cost_of = [&](uint32_t ref, int i) -> float { return (ref & ACCEL_LEAF_BIT) ? 0.0f : dp[ref].c[i]; };
        for (uint32_t n = nBin; n-- > 0;) {
            const TmpNode &nd = b.nodes[n];
            Box nb = nd.lbox; nb.grow(nd.rbox);
            float dist[3]; uint8_t kk[3];
            for (int j = 2; j <= 4; j++) {
                float best = INFINITY; uint8_t bk = 1;
                for (int k = 1; k < j; k++) {
                    const float c = cost_of(nd.left, std::min(k, 3) - 1) + cost_of(nd.right, std::min(j - k, 3) - 1);
                    if (c < best) { best = c; bk = (uint8_t)k; }
                }
                dist[j - 2] = best; kk[j - 2] = bk;
            }
            Dp &d = dp[n];
            d.k[0] = kk[0]; d.k[1] = kk[1]; d.k[2] = kk[2];
            d.c[0] = nb.half_area() + dist[2];
            d.c[1] = std::min(d.c[0], dist[0]);
            d.c[2] = std::min(d.c[1], dist[1]);
        }
    }
    // the (up to) four children of the wide node that stands for binary node `old`; interior refs are binary indices
    struct Cut { uint32_t ref, from; uint8_t right; };
    auto shape_dp = [&](uint32_t old) {
        Shape sh;
        int n = 0;
        // expands (ref, j): the subtree `ref` as at most j slots
        Cut stack[8]; int jj[8]; int sp = 0;
        const TmpNode &root = b.nodes[old];
        const int kRoot = dp[old].k[2];
        stack[sp] = {root.right, old, 1}; jj[sp++] = 4 - kRoot;
        stack[sp] = {root.left, old, 0}; jj[sp++] = kRoot;
        while (sp > 0) {
            const Cut c = stack[--sp]; int j = jj[sp];
            bool slot = (c.ref & ACCEL_LEAF_BIT) || j <= 1;
            if (!slot) {
                // fewer pieces when more do not pay: cost[.][j-1] was taken from a smaller cut
                const Dp &d = dp[c.ref];
                while (j > 1 && d.c[j - 1] == d.c[j - 2]) j--;
                if (j <= 1) slot = true;
                else {
                    const int k = d.k[j - 2];
                    stack[sp] = {b.nodes[c.ref].right, c.ref, 1}; jj[sp++] = j - k;
                    stack[sp] = {b.nodes[c.ref].left, c.ref, 0}; jj[sp++] = k;
                }
            }
            if (slot) { sh.ref[n] = c.ref; sh.from[n] = c.from; sh.right[n] = c.right; n++; }
        }
        for (int c = n; c < 4; c++) { sh.ref[c] = 0xFFFFFFFFu; sh.from[c] = 0; sh.right[c] = 0; }
        sh.n = (uint8_t)n;
        return sh;
    };
    auto shape_greedy = [&](uint32_t old) {
        Shape sh;
        int n = 2;
        sh.ref[0] = b.nodes[old].left; sh.from[0] = old; sh.right[0] = 0;
        sh.ref[1] = b.nodes[old].right; sh.from[1] = old; sh.right[1] = 1;
        while (n < 4) {
            int pick = -1; float area = -1.0f;
            for (int c = 0; c < n; c++)
                if (!(sh.ref[c] & ACCEL_LEAF_BIT)) { const float a = box_of(sh.from[c], sh.right[c]).half_area(); if (a > area) { area = a; pick = c; } }
            if (pick < 0) break;
            const uint32_t src = sh.ref[pick];
            sh.ref[pick] = b.nodes[src].left; sh.from[pick] = src; sh.right[pick] = 0;
            sh.ref[n] = b.nodes[src].right; sh.from[n] = src; sh.right[n] = 1;
            n++;
        }
        for (int c = n; c < 4; c++) { sh.ref[c] = 0xFFFFFFFFu; sh.from[c] = 0; sh.right[c] = 0; }
        sh.n = (uint8_t)n;
        return sh;
    };
    auto shape_of = [&](uint32_t old) { return useDp ? shape_dp(old) : shape_greedy(old); };
    // The top of the tree level by level, until there are enough subtrees to hand one to each thread ...
    std::vector<Shape> shape;
    std::vector<Item> frontier{{root, 0xFFFFFFFFu, 0, 1}};
    int depthTop = 0;
    while (!frontier.empty() && frontier.size() < 256) {
        std::vector<Item> next;
        for (const Item &it : frontier) {
            const uint32_t me = (uint32_t)shape.size();
            shape.push_back(shape_of(it.old));
            if (it.parent != 0xFFFFFFFFu) shape[it.parent].ref[it.slot] = me;
            if (it.depth > depthTop) depthTop = it.depth;
            for (int c = 0; c < shape[me].n; c++)
                if (!(shape[me].ref[c] & ACCEL_LEAF_BIT)) next.push_back({shape[me].ref[c], me, c, it.depth + 1});
        }
        frontier.swap(next);
    }
    // ... then every subtree depth first (slot 0's subtree follows its parent), numbered from 0 within itself
    const size_t nSub = frontier.size();
    std::vector<std::vector<Shape>> sub(nSub);
    std::vector<int> subDepth(nSub, 0);
    std::atomic<size_t> nextSub{0};
    auto worker = [&]() {
        std::vector<Item> stack;
        for (size_t j = nextSub.fetch_add(1); j < nSub; j = nextSub.fetch_add(1)) {
            std::vector<Shape> &loc = sub[j];
            int deepest = 0;
            stack.clear();
            stack.push_back({frontier[j].old, 0xFFFFFFFFu, 0, frontier[j].depth});
            while (!stack.empty()) {
                const Item it = stack.back(); stack.pop_back();
                const uint32_t me = (uint32_t)loc.size();
                loc.push_back(shape_of(it.old));
                if (it.parent != 0xFFFFFFFFu) loc[it.parent].ref[it.slot] = me;
                if (it.depth > deepest) deepest = it.depth;
                for (int c = loc[me].n - 1; c >= 0; c--)
                    if (!(loc[me].ref[c] & ACCEL_LEAF_BIT)) stack.push_back({loc[me].ref[c], me, c, it.depth + 1});
            }
            subDepth[j] = deepest;
        }
    };
    {
        const int threads = (int)std::min<size_t>((size_t)host_threads(), std::max<size_t>(nSub, 1));
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
    }
    // one array: the top, then the subtrees in frontier order; local numbers become global ones
    std::vector<size_t> start(nSub + 1, shape.size());
    for (size_t j = 0; j < nSub; j++) start[j + 1] = start[j] + sub[j].size();
    const size_t nTop = shape.size();
    shape.resize(start[nSub]);
    out.wideDepth = depthTop;
    for (size_t j = 0; j < nSub; j++) {
        shape[frontier[j].parent].ref[frontier[j].slot] = (uint32_t)start[j];
        out.wideDepth = std::max(out.wideDepth, subDepth[j]);
    }
    par_for((int64_t)nSub, [&](int64_t jb, int64_t je) {
        for (int64_t j = jb; j < je; j++) {
            const uint32_t off = (uint32_t)start[(size_t)j];
            for (size_t k = 0; k < sub[(size_t)j].size(); k++) {
                Shape sh = sub[(size_t)j][k];
                for (int c = 0; c < sh.n; c++)
                    if (!(sh.ref[c] & ACCEL_LEAF_BIT)) sh.ref[c] += off;
                shape[off + k] = sh;
            }
            std::vector<Shape>().swap(sub[(size_t)j]);
        }
    });
    (void)nTop;
    out.wide.assign(shape.size(), WideNode());
    par_for((int64_t)shape.size(), [&](int64_t wb, int64_t we) {
        for (int64_t me = wb; me < we; me++) {
            const Shape &sh = shape[(size_t)me];
            const int n = sh.n;
            Box nb; nb.reset();
            for (int c = 0; c < n; c++) nb.grow(box_of(sh.from[c], sh.right[c]));
            WideNode w;
            std::memset(&w, 0, sizeof w);
            double scale[3];
            for (int k = 0; k < 3; k++) {
                w.origin[k] = nb.mn[k];
                const double ext = (double)nb.mx[k] - (double)nb.mn[k];
                int e = (ext > 0.0) ? (int)std::ceil(std::log2(ext / 255.0)) : -100;
                e = std::min(std::max(e, -100), 100);
                while (255.0 * std::ldexp(1.0, e) < ext) e++;
                scale[k] = std::ldexp(1.0, e);
                (k == 0 ? w.scaleX : (k == 1 ? w.scaleY : w.scaleZ)) = (float)scale[k];
            }
            for (int c = 0; c < 4; c++) {
                if (c >= n) {
                    w.ref[c] = 0xFFFFFFFFu;
                    for (int k = 0; k < 3; k++) { w.qlo[k] |= 255u << (8 * c); }
                    continue;
                }
                const Box &cb = box_of(sh.from[c], sh.right[c]);
                for (int k = 0; k < 3; k++) {
                    double lo = std::floor(((double)cb.mn[k] - (double)nb.mn[k]) / scale[k]);
                    double hi = std::ceil(((double)cb.mx[k] - (double)nb.mn[k]) / scale[k]);
                    lo = std::min(std::max(lo, 0.0), 255.0); hi = std::min(std::max(hi, 0.0), 255.0);
                    w.qlo[k] |= (uint32_t)lo << (8 * c);
                    w.qhi[k] |= (uint32_t)hi << (8 * c);
                }
                if (sh.ref[c] & ACCEL_LEAF_BIT) w.ref[c] = ACCEL_LEAF_BIT | out.triIdx[sh.ref[c] & 0x0FFFFFFFu];
                else w.ref[c] = sh.ref[c];
            }
            out.wide[(size_t)me] = w;
        }
    });
    out.wideRootRef = 0;
    out.wideStack = 3 * out.wideDepth + 1;
    if (getenv("RT_AMD_VERBOSE")) {
        // summed area of the wide nodes over the root's: wide nodes a random ray through the scene visits
        double sum = 0.0, rootArea = 0.0;
        for (size_t me = 0; me < shape.size(); me++) {
            Box nb; nb.reset();
            for (int c = 0; c < shape[me].n; c++) nb.grow(box_of(shape[me].from[c], shape[me].right[c]));
            if (me == 0) rootArea = nb.half_area();
            sum += nb.half_area();
        }
        fprintf(stderr, "rt_amd: 4-wide collapse (%s): %zu nodes, depth %d, expected visits per ray through the root box %.2f\n",
                useDp ? "least summed area" : "largest child first", shape.size(), out.wideDepth, rootArea > 0.0 ? sum / rootArea : 0.0);
    }
}

}  // namespace

int index_eps_log2() {
    if (const char *env = getenv("RT_AMD_INDEX_EPS_LOG2")) { const int v = atoi(env); if (v <= -8 && v >= -40) return v; }
    return INDEX_EPS_LOG2;
}

int build_accel(const RtTriangle *tris, int64_t n, float sceneMaxAbs, int leafMax, Accel &out) {
    out = Accel();
    if (n <= 0 || n >= ((int64_t)1 << 28)) return RT_E_INVALID;
    const bool verbose = getenv("RT_AMD_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tStart = now();
    std::vector<Prim> prims((size_t)n);
    const double epsBase = std::ldexp((double)sceneMaxAbs, index_eps_log2());
    par_for(n, [&](int64_t ib, int64_t ie) {
    for (int64_t i = ib; i < ie; i++) {
        const RtTriangle &t = tris[i];
        Prim &p = prims[(size_t)i];
        // shape factor |A||B| / |A x B| (1 for a right-angled corner, large for slivers)
        double A[3], B[3];
        for (int k = 0; k < 3; k++) { A[k] = (double)t.v[1][k] - t.v[0][k]; B[k] = (double)t.v[2][k] - t.v[0][k]; }
        const double cx = A[1] * B[2] - A[2] * B[1], cy = A[2] * B[0] - A[0] * B[2], cz = A[0] * B[1] - A[1] * B[0];
        const double cr = std::sqrt(cx * cx + cy * cy + cz * cz);
        const double ab = std::sqrt((A[0] * A[0] + A[1] * A[1] + A[2] * A[2]) * (B[0] * B[0] + B[1] * B[1] + B[2] * B[2]));
        double f = (cr > 0.0 && std::isfinite(ab / cr)) ? ab / cr : 1.0;
        f = std::min(std::max(f, 1.0), 4096.0);
        const float eps = (float)(epsBase * f);
        for (int k = 0; k < 3; k++) {
            const float lo = std::min(std::min(t.v[0][k], t.v[1][k]), t.v[2][k]);
            const float hi = std::max(std::max(t.v[0][k], t.v[1][k]), t.v[2][k]);
            p.mn[k] = lo - eps; p.mx[k] = hi + eps;
            p.c[k] = 0.5f * lo + 0.5f * hi;
        }
        p.id = (uint32_t)i;
    }
    });
    const double tPrims = now();
    Builder b;
    b.leafMax = std::min(std::max(leafMax, 1), 8);
    b.prims = prims.data();
    b.nodes.resize((size_t)std::max<int64_t>(n - 1, 1));
    Box rootBox;
    uint32_t root;
    if (getenv("RT_AMD_NO_SIZE_CLASSES")) {
        root = b.build(0, n, rootBox, host_fork_levels());
    } else {
        Box all = b.bounds(0, n);
        const float ext = std::max(all.mx[0] - all.mn[0], std::max(all.mx[1] - all.mn[1], all.mx[2] - all.mn[2]));
        root = b.build_by_size(0, n, rootBox, ext, 0);
    }
    for (int k = 0; k < 3; k++) { out.rootMin[k] = rootBox.mn[k]; out.rootMax[k] = rootBox.mx[k]; }
    out.triIdx.resize((size_t)n);       // triangle numbers in leaf order
    par_for(n, [&](int64_t ib, int64_t ie) { for (int64_t i = ib; i < ie; i++) out.triIdx[(size_t)i] = prims[(size_t)i].id; });
    const double tBuilt = now();
    const uint32_t used = b.next.load();
    out.maxDepth = 0;
    out.directLeaves = (b.leafMax == 1);
    if (root & ACCEL_LEAF_BIT) { out.rootRef = out.directLeaves ? (ACCEL_LEAF_BIT | out.triIdx[root & 0x0FFFFFFFu]) : root; return RT_OK; }
    if (out.directLeaves) {
        // one triangle per leaf (what the GPU kernels read): only the 4-wide form is kept
        out.rootRef = 0;
        collapse_wide(b, root, out);
        if (verbose)
            fprintf(stderr, "rt_amd: host index build: boxes %.0f ms, SAH tree %.0f ms, 4-wide collapse %.0f ms\n",
                    tPrims - tStart, tBuilt - tPrims, now() - tBuilt);
        return RT_OK;
    }
    // Binary form, relaid in depth-first order (a left child follows its parent).
    out.nodes.resize(used);
    struct Item { uint32_t old; uint32_t parentNew; int side; int depth; };
    std::vector<Item> stack;
    stack.push_back({root, 0xFFFFFFFFu, 0, 1});
    uint32_t nextNew = 0;
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        const uint32_t me = nextNew++;
        const TmpNode &src = b.nodes[it.old];
        AccelNode &dst = out.nodes[me];
        for (int k = 0; k < 3; k++) {
            dst.lmin[k] = src.lbox.mn[k]; dst.lmax[k] = src.lbox.mx[k];
            dst.rmin[k] = src.rbox.mn[k]; dst.rmax[k] = src.rbox.mx[k];
        }
        dst.left = src.left; dst.right = src.right; dst.pad0 = dst.pad1 = 0;
        if (b.leafMax == 1) {       // one triangle per leaf: the ref names the triangle itself
            if (dst.left & ACCEL_LEAF_BIT) dst.left = ACCEL_LEAF_BIT | out.triIdx[dst.left & 0x0FFFFFFFu];
            if (dst.right & ACCEL_LEAF_BIT) dst.right = ACCEL_LEAF_BIT | out.triIdx[dst.right & 0x0FFFFFFFu];
        }
        if (it.parentNew != 0xFFFFFFFFu) {
            if (it.side == 0) out.nodes[it.parentNew].left = me; else out.nodes[it.parentNew].right = me;
        }
        if (it.depth > out.maxDepth) out.maxDepth = it.depth;
        // push right first so that the left subtree is laid out immediately after its parent
        if (!(src.right & ACCEL_LEAF_BIT)) stack.push_back({src.right, me, 1, it.depth + 1});
        if (!(src.left & ACCEL_LEAF_BIT)) stack.push_back({src.left, me, 0, it.depth + 1});
    }
    out.rootRef = 0;
    return RT_OK;
}

}  // namespace rt

// Diagnostics (include/rt_amd.h): the host build of the search index alone, no GPU involved.
extern "C" int rt_selftest_index_build(const RtTriangle *tris, int64_t n, float scene_max_abs, void *nodes_out,
                                       int64_t node_capacity, int64_t *node_count, double stats[4]) {
    if (!tris || n <= 0 || !node_count) { rt::set_global_error("rt_selftest_index_build: bad argument"); return RT_E_INVALID; }
    const double t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    rt::Accel a;
    const int rc = rt::build_accel(tris, n, scene_max_abs, 1, a);
    if (rc != RT_OK) { rt::set_global_error("rt_selftest_index_build: build failed"); return rc; }
    const double t1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    *node_count = (int64_t)a.wide.size();
    if (nodes_out) {
        if (node_capacity < (int64_t)a.wide.size()) { rt::set_global_error("rt_selftest_index_build: node_capacity too small"); return RT_E_INVALID; }
        std::memcpy(nodes_out, a.wide.data(), a.wide.size() * sizeof(rt::WideNode));
    }
    if (stats) {
        // summed half-area of the decoded node boxes over the root node's: wide nodes a random ray through the root visits
        double sum = 0.0, root = 0.0;
        for (size_t i = 0; i < a.wide.size(); i++) {
            const rt::WideNode &w = a.wide[i];
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            const float sc[3] = {w.scaleX, w.scaleY, w.scaleZ};
            for (int c = 0; c < 4; c++) {
                if (w.ref[c] == 0xFFFFFFFFu) continue;
                for (int k = 0; k < 3; k++) {
                    lo[k] = std::min(lo[k], (double)w.origin[k] + (double)((w.qlo[k] >> (8 * c)) & 255u) * sc[k]);
                    hi[k] = std::max(hi[k], (double)w.origin[k] + (double)((w.qhi[k] >> (8 * c)) & 255u) * sc[k]);
                }
            }
            const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            const double ar = dx * dy + dy * dz + dz * dx;
            if (i == 0) root = ar;
            sum += ar;
        }
        stats[0] = t1 - t0; stats[1] = (double)a.wideDepth; stats[2] = root > 0.0 ? sum / root : 0.0; stats[3] = (double)a.wideStack;
    }
    return RT_OK;
}
