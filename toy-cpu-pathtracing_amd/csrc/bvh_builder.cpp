// Sweep-SAH BVH2 builder for the flattened render-space triangle soup.
//
// Not a restatement of the reference's builder (scene/src/bvh.rs:92-230 builds a two-level tree with
// O(n^2) split evaluation and stores one box per node): the GPU wants ONE flat tree, boxes of both
// children in the parent (one 64 B fetch decides both subtrees), leaves of <= 4 leaf-ordered
// triangles, and a bounded depth so the per-lane LDS stack cannot overflow.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <numeric>

#include "scene.hpp"

namespace pt {
namespace {

struct Box {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const BuildTri& t) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], t.lo[a]); hi[a] = std::max(hi[a], t.hi[a]); } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float area() const {
        float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (d[0] < 0) return 0.0f;
        return 2.0f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
};

// SAH cost ratio and leaf size: defaults measured on MI355X; env overrides exist for sweeps only
// With per-lane triangle tests a watertight test cost about two node steps and 2-triangle leaves were best; since leaves only
// queue (triangle, ray) pairs that the whole wave tests densely, a triangle is cheaper than a node step: tools/sah_sweep.sh on
// scenes 3 / 17 / 0 puts the plateau at 0.6-1.0 with leaves of up to 4 (+1.2 % over 2.0 / 2).
static float COST_TRAVERSE = 1.0f, COST_TRI = 0.8f;
static int LEAF_MAX = 4;

struct Builder {
    const std::vector<BuildTri>& tris;
    BvhOut& out;
    std::vector<uint32_t> idx;
    std::vector<float> right_area;
    int max_depth = 0;

    // builds the subtree over idx[begin,end); returns the child link and its box
    int32_t build(uint32_t begin, uint32_t end, int depth, Box* box_out) {
        uint32_t n = end - begin;
        Box box;
        for (uint32_t i = begin; i < end; ++i) box.grow(tris[idx[i]]);
        *box_out = box;
        max_depth = std::max(max_depth, depth);
        auto make_leaf_here = [&]() {
            uint32_t first = (uint32_t)out.order.size();
            for (uint32_t i = begin; i < end; ++i) out.order.push_back(idx[i]);
            return make_leaf(first, n);
        };
        if (n == 1) return make_leaf_here();
        float leaf_cost = COST_TRI * (float)n;
        float best_cost = FLT_MAX; int best_axis = -1; uint32_t best_split = 0;
        // Depth bound (the LDS traversal stack holds STACK_DEPTH entries): a node with `levels_left` split levels below it can
        // always be finished by object-median splits if n <= MAX_LEAF_TRIS << levels_left; SAH may split freely only while both
        // children are sure to stay inside that bound, otherwise the split is the median.
        const int levels_left = MAX_BUILD_DEPTH - depth;
        const bool must_leaf = levels_left <= 0;
        const bool force_median = !must_leaf && (uint64_t)n > ((uint64_t)MAX_LEAF_TRIS << std::min(40, levels_left - 1));
        if (must_leaf) return make_leaf_here();   // n <= MAX_LEAF_TRIS by the bound above (build_bvh checks the root)
        if (!force_median) {
            float inv_area = 1.0f / std::max(box.area(), 1e-30f);
            for (int axis = 0; axis < 3; ++axis) {
                std::sort(idx.begin() + begin, idx.begin() + end, [&](uint32_t a, uint32_t b) {
                    float ca = tris[a].c[axis], cb = tris[b].c[axis];
                    return ca < cb || (ca == cb && a < b);
                });
                Box r;
                for (uint32_t i = end; i-- > begin + 1;) { r.grow(tris[idx[i]]); right_area[i] = r.area(); }
                Box l;
                for (uint32_t i = begin + 1; i < end; ++i) {
                    l.grow(tris[idx[i - 1]]);
                    float cost = COST_TRAVERSE + COST_TRI * inv_area * (l.area() * (float)(i - begin) + right_area[i] * (float)(end - i));
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = i; }
                }
            }
        }
        if (!force_median && n <= (uint32_t)LEAF_MAX && leaf_cost <= best_cost) return make_leaf_here();
        if (force_median || best_axis < 0) {
            // balanced object-median split on the widest axis bounds the remaining depth by log2(n)
            int axis = 0; float ext = -1;
            for (int a = 0; a < 3; ++a) if (box.hi[a] - box.lo[a] > ext) { ext = box.hi[a] - box.lo[a]; axis = a; }
            best_axis = axis; best_split = begin + n / 2;
        }
        std::sort(idx.begin() + begin, idx.begin() + end, [&](uint32_t a, uint32_t b) {
            float ca = tris[a].c[best_axis], cb = tris[b].c[best_axis];
            return ca < cb || (ca == cb && a < b);
        });
        uint32_t node = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        Box b0, b1;
        int32_t c0 = build(begin, best_split, depth + 1, &b0);
        int32_t c1 = build(best_split, end, depth + 1, &b1);
        DevNode& nd = out.nodes[node];
        nd.bx[0] = b0.lo[0]; nd.bx[1] = b1.lo[0]; nd.bx[2] = b0.hi[0]; nd.bx[3] = b1.hi[0];
        nd.by[0] = b0.lo[1]; nd.by[1] = b1.lo[1]; nd.by[2] = b0.hi[1]; nd.by[3] = b1.hi[1];
        nd.bz[0] = b0.lo[2]; nd.bz[1] = b1.lo[2]; nd.bz[2] = b0.hi[2]; nd.bz[3] = b1.hi[2];
        nd.child[0] = c0; nd.child[1] = c1; nd.pad[0] = nd.pad[1] = 0;
        return (int32_t)node;
    }
};

}  // namespace

void bvh_build_config(float* cost_traverse, float* cost_tri, int* leaf_max) {
    if (const char* e = getenv("MI355PT_BVH_COST_TRI")) COST_TRI = (float)atof(e);
    if (const char* e = getenv("MI355PT_BVH_LEAF")) LEAF_MAX = std::max(1, std::min(atoi(e), MAX_LEAF_TRIS));
    *cost_traverse = COST_TRAVERSE; *cost_tri = COST_TRI; *leaf_max = LEAF_MAX;
}

void build_bvh(const std::vector<BuildTri>& tris, BvhOut* out) {
    { float a, b; int c; bvh_build_config(&a, &b, &c); }
    out->nodes.clear(); out->order.clear();
    Builder b{tris, *out, {}, {}};
    b.idx.resize(tris.size());
    std::iota(b.idx.begin(), b.idx.end(), 0u);
    b.right_area.assign(tris.size() + 1, 0.0f);
    out->nodes.reserve(tris.size());
    out->order.reserve(tris.size());
    Box root_box;
    out->root = b.build(0, (uint32_t)tris.size(), 0, &root_box);
    out->max_depth = b.max_depth;
    if (out->nodes.empty()) {
        // a scene of <= MAX_LEAF_TRIS triangles: wrap the single leaf in a node so traversal has a root node
        DevNode nd{};
        nd.bx[0] = root_box.lo[0]; nd.bx[2] = root_box.hi[0]; nd.by[0] = root_box.lo[1]; nd.by[2] = root_box.hi[1];
        nd.bz[0] = root_box.lo[2]; nd.bz[2] = root_box.hi[2];
        // empty second child: a point box at +FLT_MAX is missed by the slab test for every finite ray
        nd.bx[1] = nd.by[1] = nd.bz[1] = FLT_MAX; nd.bx[3] = nd.by[3] = nd.bz[3] = FLT_MAX;
        nd.child[0] = out->root; nd.child[1] = out->root;
        out->nodes.push_back(nd);
        out->root = 0;
    }
}

}  // namespace pt
