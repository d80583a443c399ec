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
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "rt_accel.h"

namespace rt {

namespace {

constexpr int MAX_BINS = 64;     // measured: 16 -> 64 bins is 5 % fewer box tests per ray, more gives nothing

struct Prim {
    float mn[3], mx[3], c[3];
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
    const Prim *prims;
    uint32_t *idx;
    std::vector<TmpNode> nodes;
    std::atomic<uint32_t> next{0};

    static uint32_t leaf_ref(int64_t begin, int64_t count) {
        return ACCEL_LEAF_BIT | ((uint32_t)(count - 1) << 28) | (uint32_t)begin;
    }

    Box bounds(int64_t begin, int64_t end) const {
        Box b; b.reset();
        for (int64_t i = begin; i < end; i++) b.grow(prims[idx[i]].mn, prims[idx[i]].mx);
        return b;
    }

    // Triangles much larger than the rest (the walls of a room around a dense mesh) are split off
    // first, by size class: a split along an axis cannot tell an enclosing triangle from the ones
    // it encloses, and every subtree that keeps one gets a box as large as the room.  The classes
    // are a quarter, a sixteenth and a sixty-fourth of the scene's largest extent; a class is only
    // split off while it is the small part (at most a tenth of the range).
    uint32_t build_by_size(int64_t begin, int64_t end, Box &box, float sceneExtent, int cls) {
        const int64_t count = end - begin;
        if (cls >= 3 || count < 64) return build(begin, end, box, 3);
        const float thr = sceneExtent * (cls == 0 ? 0.25f : (cls == 1 ? 0.0625f : 0.015625f));
        uint32_t *first = idx + begin, *last = idx + end;
        uint32_t *m = std::partition(first, last, [&](uint32_t id) {
            const Prim &p = prims[id];
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

    // Returns the ref of the subtree over idx[begin,end) and its box.
    uint32_t build(int64_t begin, int64_t end, Box &box, int fork) {
        const int64_t count = end - begin;
        box = bounds(begin, end);
        if (count <= 1) return leaf_ref(begin, count);
        Box cb; cb.reset();
        for (int64_t i = begin; i < end; i++) cb.grow(prims[idx[i]].c, prims[idx[i]].c);
        // binned SAH over the three axes; small nodes need fewer bins to tell all their splits apart
        const int NBINS = (int)std::min<int64_t>(MAX_BINS, std::max<int64_t>(8, 4 * count));
        float bestCost = INFINITY; int bestAxis = -1, bestSplit = -1;
        for (int ax = 0; ax < 3; ax++) {
            const float lo = cb.mn[ax], ext = cb.mx[ax] - cb.mn[ax];
            if (!(ext > 0.0f)) continue;
            Box bb[MAX_BINS]; int64_t bc[MAX_BINS];
            for (int b = 0; b < NBINS; b++) { bb[b].reset(); bc[b] = 0; }
            const float scale = (float)NBINS / ext;
            for (int64_t i = begin; i < end; i++) {
                const Prim &p = prims[idx[i]];
                int b = (int)((p.c[ax] - lo) * scale);
                b = b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
                bb[b].grow(p.mn, p.mx); bc[b]++;
            }
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
            uint32_t *first = idx + begin, *last = idx + end;
            uint32_t *m = std::partition(first, last, [&](uint32_t id) {
                int b = (int)((prims[id].c[bestAxis] - lo) * scale);
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


// Collapses the binary tree into 4-wide nodes: a wide node starts from the two children of a
// binary node and repeatedly replaces its largest interior child by that child's two children.
void collapse_wide(const Builder &b, uint32_t root, Accel &out) {
    struct Child { uint32_t ref; Box box; };
    struct Item { uint32_t old; uint32_t parent; int slot; int depth; };
    std::vector<Item> stack;
    stack.push_back({root, 0xFFFFFFFFu, 0, 1});
    out.wide.clear();
    out.wide.reserve(b.nodes.size() / 2 + 1);
    out.wideDepth = 0;
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        const uint32_t me = (uint32_t)out.wide.size();
        out.wide.emplace_back();
        if (it.parent != 0xFFFFFFFFu) out.wide[it.parent].ref[it.slot] = me;
        if (it.depth > out.wideDepth) out.wideDepth = it.depth;
        Child ch[4]; int n = 2;
        ch[0] = {b.nodes[it.old].left, b.nodes[it.old].lbox};
        ch[1] = {b.nodes[it.old].right, b.nodes[it.old].rbox};
        while (n < 4) {
            int pick = -1; float area = -1.0f;
            for (int c = 0; c < n; c++)
                if (!(ch[c].ref & ACCEL_LEAF_BIT) && ch[c].box.half_area() > area) { area = ch[c].box.half_area(); pick = c; }
            if (pick < 0) break;
            const TmpNode &src = b.nodes[ch[pick].ref];
            ch[pick] = {src.left, src.lbox};
            ch[n++] = {src.right, src.rbox};
        }
        Box nb; nb.reset();
        for (int c = 0; c < n; c++) nb.grow(ch[c].box);
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
            for (int k = 0; k < 3; k++) {
                double lo = std::floor(((double)ch[c].box.mn[k] - (double)nb.mn[k]) / scale[k]);
                double hi = std::ceil(((double)ch[c].box.mx[k] - (double)nb.mn[k]) / scale[k]);
                lo = std::min(std::max(lo, 0.0), 255.0); hi = std::min(std::max(hi, 0.0), 255.0);
                w.qlo[k] |= (uint32_t)lo << (8 * c);
                w.qhi[k] |= (uint32_t)hi << (8 * c);
            }
            if (ch[c].ref & ACCEL_LEAF_BIT) w.ref[c] = ACCEL_LEAF_BIT | out.triIdx[ch[c].ref & 0x0FFFFFFFu];
            else w.ref[c] = ch[c].ref;      // binary index for now; patched when the child is emitted
        }
        out.wide[me] = w;
        // children are emitted depth first; slot 0's subtree follows its parent
        for (int c = n - 1; c >= 0; c--)
            if (!(ch[c].ref & ACCEL_LEAF_BIT)) stack.push_back({ch[c].ref, me, c, it.depth + 1});
    }
    out.wideRootRef = 0;
    out.wideStack = 3 * out.wideDepth + 1;
}

}  // namespace

int index_eps_log2() {
    if (const char *env = getenv("RT_AMD_INDEX_EPS_LOG2")) { const int v = atoi(env); if (v <= -8 && v >= -40) return v; }
    return INDEX_EPS_LOG2;
}

int build_accel(const RtTriangle *tris, int64_t n, float sceneMaxAbs, int leafMax, Accel &out) {
    out = Accel();
    if (n <= 0 || n >= ((int64_t)1 << 28)) return RT_E_INVALID;
    std::vector<Prim> prims((size_t)n);
    const double epsBase = std::ldexp((double)sceneMaxAbs, index_eps_log2());
    for (int64_t i = 0; i < n; i++) {
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
    }
    out.triIdx.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) out.triIdx[(size_t)i] = (uint32_t)i;
    Builder b;
    b.leafMax = std::min(std::max(leafMax, 1), 8);
    b.prims = prims.data(); b.idx = out.triIdx.data();
    b.nodes.resize((size_t)std::max<int64_t>(n - 1, 1));
    Box rootBox;
    uint32_t root;
    if (getenv("RT_AMD_NO_SIZE_CLASSES")) {
        root = b.build(0, n, rootBox, 3);
    } else {
        Box all = b.bounds(0, n);
        const float ext = std::max(all.mx[0] - all.mn[0], std::max(all.mx[1] - all.mn[1], all.mx[2] - all.mn[2]));
        root = b.build_by_size(0, n, rootBox, ext, 0);
    }
    for (int k = 0; k < 3; k++) { out.rootMin[k] = rootBox.mn[k]; out.rootMax[k] = rootBox.mx[k]; }
    const uint32_t used = b.next.load();
    // Relayout in depth-first order (a left child follows its parent) and measure the depth.
    out.nodes.resize(used);
    out.maxDepth = 0;
    out.directLeaves = (b.leafMax == 1);
    if (root & ACCEL_LEAF_BIT) { out.rootRef = out.directLeaves ? (ACCEL_LEAF_BIT | out.triIdx[root & 0x0FFFFFFFu]) : root; return RT_OK; }
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
    if (out.directLeaves) collapse_wide(b, root, out);
    return RT_OK;
}

}  // namespace rt
