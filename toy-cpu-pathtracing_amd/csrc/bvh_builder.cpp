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
// scenes 3 / 17 / 0 puts the plateau at 0.6-1.0 with leaves of up to 4 (+1.2 % over 2.0 / 2).  Re-swept with the 4-wide tree and the merged
// traversal (round 2, scenes 3 / 8 / 15 / 17): leaves of up to 3 (1 844 / 1 608 / 977 / 1 185 Msamples/s against 1 823 / 1 591 / 977 / 1 184 with 4 and
// 1 840 / 1 606 / 969 / 1 181 with 2; single-triangle leaves lose 10 %); forced median splits may still make leaves of MAX_LEAF_TRIS.
static float COST_TRAVERSE = 1.0f, COST_TRI = 0.8f;
static int LEAF_MAX = 3;

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
#ifdef MI355PT_TUNING   // SAH sweeps (tools/sah_sweep.sh): not in the shipped library
    if (const char* e = getenv("MI355PT_BVH_COST_TRI")) COST_TRI = (float)atof(e);
    if (const char* e = getenv("MI355PT_BVH_LEAF")) LEAF_MAX = std::max(1, std::min(atoi(e), MAX_LEAF_TRIS));
#endif
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

// ---------------------------------------------------------------------------------------------------------------------------------
// BVH2 -> BVH4 (layout.hpp DevNode4).  `budget` = stack slots still free below this node; expanding a child into its two children makes the
// node one wider, i.e. costs one more slot for every subtree below it, and is allowed only while each member's all-binary height still fits.
namespace {
struct Collapse4 {
    const std::vector<DevNode>& n2;
    std::vector<DevNode4>& out;
    std::vector<int> height;      // all-binary height of each BVH2 node's subtree (leaf = 0)
    int height_of(int32_t link) const { return link < 0 ? 0 : height[(size_t)link]; }
    int compute_height(int32_t link) {
        if (link < 0) return 0;
        int h = 1 + std::max(compute_height(n2[(size_t)link].child[0]), compute_height(n2[(size_t)link].child[1]));
        height[(size_t)link] = h;
        return h;
    }
    struct Member { float lo[3], hi[3]; int32_t link; };
    static Member member(const DevNode& n, int c) {
        return Member{{n.bx[c], n.by[c], n.bz[c]}, {n.bx[2 + c], n.by[2 + c], n.bz[2 + c]}, n.child[c]};
    }
    static float area(const Member& m) {
        float d[3] = {m.hi[0] - m.lo[0], m.hi[1] - m.lo[1], m.hi[2] - m.lo[2]};
        return d[0] < 0 ? 0.0f : 2.0f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
    int32_t emit(int32_t link2, int budget) {
        if (link2 < 0) return link2;
        std::vector<Member> m = {member(n2[(size_t)link2], 0), member(n2[(size_t)link2], 1)};
        while (m.size() < 4) {
            // widest-area internal member whose expansion keeps every member inside the slots that remain with one more sibling pending
            int best = -1;
            for (size_t i = 0; i < m.size(); ++i) {
                if (m[i].link < 0) continue;
                const DevNode& e = n2[(size_t)m[i].link];
                bool fits = true;
                const int left = budget - (int)m.size();                 // slots below a node of m.size() + 1 children
                for (size_t j = 0; j < m.size() && fits; ++j) if (j != i && height_of(m[j].link) > left) fits = false;
                if (height_of(e.child[0]) > left || height_of(e.child[1]) > left) fits = false;
                if (fits && (best < 0 || area(m[i]) > area(m[(size_t)best]))) best = (int)i;
            }
            if (best < 0) break;
            const DevNode& e = n2[(size_t)m[(size_t)best].link];
            Member a = member(e, 0), b = member(e, 1);
            m[(size_t)best] = a; m.push_back(b);
        }
        const size_t idx = out.size();
        out.emplace_back();
        const int below = budget - ((int)m.size() - 1);
        int32_t links[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < m.size(); ++i) links[i] = emit(m[i].link, below);
        DevNode4& d = out[idx];
        for (int c = 0; c < 4; ++c) {
            const bool used = (size_t)c < m.size();
            // an unused slot is a point box at +FLT_MAX: the slab test misses it for every finite ray (an INVERTED infinite box would be hit by
            // every ray: min(lo, hi) = -inf, max = +inf), like the empty second child of the wrapped single-leaf root in bvh_builder.cpp
            d.lox[c] = used ? m[(size_t)c].lo[0] : FLT_MAX; d.loy[c] = used ? m[(size_t)c].lo[1] : FLT_MAX; d.loz[c] = used ? m[(size_t)c].lo[2] : FLT_MAX;
            d.hix[c] = used ? m[(size_t)c].hi[0] : FLT_MAX; d.hiy[c] = used ? m[(size_t)c].hi[1] : FLT_MAX; d.hiz[c] = used ? m[(size_t)c].hi[2] : FLT_MAX;
            d.child[c] = used ? links[c] : 0; d.pad[c] = 0;
        }
        return (int32_t)idx;
    }
};

// Cost-optimal collapse under the stack bound (dynamic programming, after Ylitie et al. 2017, sec. 4.1, with the budget as a third index).
// Cost = sum of the surface areas of the 4-wide nodes' boxes (the expected number of node visits; a constant per node on top of the area was
// swept and only loses); leaves are fixed.  b = stack slots free on
// arrival at a node: a node with k children needs k - 1 of them and leaves b - (k - 1) to every child.
//   root(n, b)    = A(n) + min over k in {2, 3, 4}, k - 1 <= b, and splits j + (k - j) of pack(left, j, b') + pack(right, k - j, b'),  b' = b - (k - 1)
//   pack(n, i, b') = least cost of hanging n's subtree under the current node as at most i of its children:
//                    leaf: 0;  i = 1: root(n, b');  else min(pack(n, i - 1, b'), min over j of pack(left, j, b') + pack(right, i - j, b'))
struct CollapseDP {
    const std::vector<DevNode>& n2;
    std::vector<DevNode4>& out;
    static constexpr int NB = STACK_DEPTH;                     // budgets 0 .. STACK_DEPTH - 1
    std::vector<float> area, c_root, c_pack;                   // c_root[n][b], c_pack[n][i - 1][b] (i = 1..3)
    std::vector<uint8_t> k_root, j_root, ch_pack;              // choices: k and the left share j of a root; for pack: 0 = use i - 1, else left share j
    static constexpr float INF = 3.0e38f;
    float& R(int32_t n, int b) { return c_root[(size_t)n * NB + (size_t)b]; }
    float& P(int32_t n, int i, int b) { return c_pack[((size_t)n * 3 + (size_t)(i - 1)) * NB + (size_t)b]; }
    uint8_t& PC(int32_t n, int i, int b) { return ch_pack[((size_t)n * 3 + (size_t)(i - 1)) * NB + (size_t)b]; }
    float pack(int32_t link, int i, int b) { return link < 0 ? 0.0f : P(link, i, b); }
    void solve(int32_t root) {
        const size_t n = n2.size();
        area.assign(n, 0.0f); c_root.assign(n * NB, INF); c_pack.assign(n * 3 * NB, INF);
        k_root.assign(n * NB, 0); j_root.assign(n * NB, 0); ch_pack.assign(n * 3 * NB, 0);
        // post-order without recursion limits: children have larger construction indices than their parents in both builders? not guaranteed
        std::vector<int32_t> order; order.reserve(n);
        std::vector<int32_t> st{root};
        while (!st.empty()) { int32_t v = st.back(); st.pop_back(); order.push_back(v); for (int c = 0; c < 2; ++c) if (n2[(size_t)v].child[c] >= 0) st.push_back(n2[(size_t)v].child[c]); }
        for (size_t q = order.size(); q-- > 0;) {
            const int32_t v = order[q];
            const DevNode& nd = n2[(size_t)v];
            const float lo[3] = {std::min(nd.bx[0], nd.bx[1]), std::min(nd.by[0], nd.by[1]), std::min(nd.bz[0], nd.bz[1])};
            const float hi[3] = {std::max(nd.bx[2], nd.bx[3]), std::max(nd.by[2], nd.by[3]), std::max(nd.bz[2], nd.bz[3])};
            const float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
            area[(size_t)v] = 2.0f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
            const int32_t l = nd.child[0], r = nd.child[1];
            for (int b = 0; b < NB; ++b) {                      // this node as the root of a 4-wide node arriving with b free slots
                float best = INF; int bk = 0, bj = 0;
                for (int k = 2; k <= 4 && k - 1 <= b; ++k) {
                    const int bb = b - (k - 1);
                    for (int j = 1; j < k; ++j) {
                        if (j > 3 || k - j > 3) continue;
                        const float c = pack(l, j, bb) + pack(r, k - j, bb);
                        if (c < best) { best = c; bk = k; bj = j; }
                    }
                }
                if (best < INF) { R(v, b) = area[(size_t)v] + best; k_root[(size_t)v * NB + (size_t)b] = (uint8_t)bk; j_root[(size_t)v * NB + (size_t)b] = (uint8_t)bj; }
            }
            for (int b = 0; b < NB; ++b) {                      // this node's subtree hung under a node whose children have b free slots
                P(v, 1, b) = R(v, b); PC(v, 1, b) = 0;
                for (int i = 2; i <= 3; ++i) {
                    float best = P(v, i - 1, b); int bj = 0;
                    for (int j = 1; j < i; ++j) {
                        const float c = pack(l, j, b) + pack(r, i - j, b);
                        if (c < best) { best = c; bj = j; }
                    }
                    P(v, i, b) = best; PC(v, i, b) = (uint8_t)bj;
                }
            }
        }
    }
    using Member = Collapse4::Member;
    void collect(const Member& m, int i, int b, std::vector<Member>& into) {
        if (m.link < 0 || i == 1) { into.push_back(m); return; }
        const int j = PC(m.link, i, b);
        if (j == 0) { collect(m, i - 1, b, into); return; }
        const DevNode& e = n2[(size_t)m.link];
        collect(Collapse4::member(e, 0), j, b, into);
        collect(Collapse4::member(e, 1), i - j, b, into);
    }
    int32_t emit(int32_t link2, int b) {
        if (link2 < 0) return link2;
        const int k = k_root[(size_t)link2 * NB + (size_t)b], j = j_root[(size_t)link2 * NB + (size_t)b];
        const DevNode& e = n2[(size_t)link2];
        std::vector<Member> m;
        collect(Collapse4::member(e, 0), j, b - (k - 1), m);
        collect(Collapse4::member(e, 1), k - j, b - (k - 1), m);
        const size_t idx = out.size();
        out.emplace_back();
        int32_t links[4] = {0, 0, 0, 0};
        for (size_t c = 0; c < m.size(); ++c) links[c] = emit(m[c].link, b - (k - 1));
        DevNode4& d = out[idx];
        for (int c = 0; c < 4; ++c) {
            const bool used = (size_t)c < m.size();
            d.lox[c] = used ? m[(size_t)c].lo[0] : FLT_MAX; d.loy[c] = used ? m[(size_t)c].lo[1] : FLT_MAX; d.loz[c] = used ? m[(size_t)c].lo[2] : FLT_MAX;
            d.hix[c] = used ? m[(size_t)c].hi[0] : FLT_MAX; d.hiy[c] = used ? m[(size_t)c].hi[1] : FLT_MAX; d.hiz[c] = used ? m[(size_t)c].hi[2] : FLT_MAX;
            d.child[c] = used ? links[c] : 0; d.pad[c] = 0;
        }
        return (int32_t)idx;
    }
};
}  // namespace

bool collapse_bvh4(const std::vector<DevNode>& nodes2, int32_t root2, size_t n_tris, std::vector<DevNode4>* nodes4, int32_t* root4, int* max_stack,
                   std::string* err, const char** method) {
    nodes4->clear();
    nodes4->reserve(nodes2.size() / 2 + 1);
    Collapse4 col{nodes2, *nodes4, std::vector<int>(nodes2.size(), 0)};
    col.compute_height(root2);
    if (root2 < 0 || col.height_of(root2) >= STACK_DEPTH) { *err = "BVH deeper than the traversal stack"; return false; }
    // the cost-optimal collapse where its tables fit (24 budgets x 4 entries per binary node), the greedy one otherwise and for the wrapped
    // single-leaf root; MI355PT_BVH_COLLAPSE=greedy|dp in a -DMI355PT_TUNING build for A/B runs
    bool use_dp = nodes2.size() >= 2 && nodes2.size() <= 1200000;
#ifdef MI355PT_TUNING
    if (const char* e = getenv("MI355PT_BVH_COLLAPSE")) use_dp = use_dp && std::string(e) != "greedy";
#endif
    if (use_dp) {
        // the tables take ~504 B per binary node (0.6 GB at the cutoff): a host that cannot spare them gets the greedy collapse, not an
        // exception through the C ABI
        try {
            CollapseDP dp{nodes2, *nodes4};
            dp.solve(root2);
            if (dp.R(root2, STACK_DEPTH - 1) < CollapseDP::INF) *root4 = dp.emit(root2, STACK_DEPTH - 1);
            else use_dp = false;
        } catch (const std::bad_alloc&) { use_dp = false; }
    }
    if (!use_dp) { nodes4->clear(); *root4 = col.emit(root2, STACK_DEPTH - 1); }
    if (method) *method = use_dp ? "dp" : "greedy";
    // the guarantees the kernel relies on, checked on the tree that is uploaded: links in range, no cycle, every triangle in exactly one
    // leaf, worst-case pending siblings along any path (= the per-lane LDS stack need) below STACK_DEPTH
    size_t tris_seen = 0, visited = 0;
    bool ok = *root4 >= 0 && (size_t)*root4 < nodes4->size();
    int worst = 0;
    std::vector<std::pair<int32_t, int>> todo;            // (node, stack slots in use on arrival)
    if (ok) todo.push_back({*root4, 0});
    while (ok && !todo.empty()) {
        const int32_t ni = todo.back().first; const int used = todo.back().second; todo.pop_back();
        if (++visited > nodes4->size()) { ok = false; break; }
        const DevNode4& nd = (*nodes4)[(size_t)ni];
        int k = 0;
        for (int c = 0; c < 4; ++c) if (nd.lox[c] <= nd.hix[c] && nd.lox[c] < FLT_MAX) ++k;
        if (k < 1 || used + (k - 1) >= STACK_DEPTH) { ok = false; break; }
        worst = std::max(worst, used + (k - 1));
        for (int c = 0; c < 4; ++c) {
            if (!(nd.lox[c] <= nd.hix[c] && nd.lox[c] < FLT_MAX)) continue;
            const int32_t l = nd.child[c];
            if (l < 0) { const uint32_t f = leaf_first(l), n = leaf_count(l); if ((size_t)f + n > n_tris) ok = false; tris_seen += n; }
            else if ((size_t)l >= nodes4->size()) ok = false;
            else todo.push_back({l, used + (k - 1)});
        }
    }
    if (!ok || tris_seen != n_tris) { *err = "internal error: collapsed BVH failed validation"; return false; }
    *max_stack = worst;
    return true;
}

}  // namespace pt
