// ORACLE — TEST INFRASTRUCTURE ONLY.
// Restatement of scene/src/bvh.rs: full-sweep SAH build (:92-230), DFS flatten with
// inline leaf items (:234-295), recursive closest-hit that visits both children and
// never tightens t_max (:344-444) and recursive any-hit (:447-520).
//
// Two deliberate accelerations that do not change results:
//  * build: the reference recomputes both partitions' bounds for every split position
//    (O(n^2)); here prefix/suffix bounds give the same min/max values exactly.
//  * traversal `shrink` flag (fast mode): the second subtree / later items are
//    searched with t_max = best t so far.  Box and triangle tests only *reject* with
//    t_max and ties (t == best) are still found and resolved by the same `<` rules,
//    so the returned hit is identical to faithful mode (tests/test_oracle.py checks).
#pragma once
#include <functional>
#include <vector>
#include "o_math.hpp"

namespace oracle {

struct TraversalCounters {
    uint64_t nodes = 0;      // box tests
    uint64_t items = 0;      // leaf item tests
};

struct BvhNode {
    // kind: 0 = Node{bounds, second_offset}, 1 = Leaf{bounds, item_count}, 2 = Item{item}
    uint32_t kind;
    Bounds bounds;
    uint32_t value;
};

template <typename BoundsFn>
struct BvhBuilder {
    const BoundsFn& bounds_of;
    std::vector<BvhNode>& nodes;

    Bounds list_bounds(const std::vector<uint32_t>& items) const {          // :73-79
        Bounds b = bounds_of(items[0]);
        for (size_t i = 1; i < items.size(); ++i) b = b.merge(bounds_of(items[i]));
        return b;
    }

    // returns false when no split beats the leaf cost
    void build(const std::vector<uint32_t>& items, const Bounds& bounds) {   // :161-230 + flatten :234-295
        if (items.size() > 1) {
            float min_cost = 1.0f * (float)items.size();
            bool have = false;
            std::vector<uint32_t> best_first, best_second;
            float parent_area = bounds.area();
            for (int axis = 0; axis < 3; ++axis) {
                std::vector<uint32_t> sorted = items;
                std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b) {
                    return bounds_of(a).center()[axis] < bounds_of(b).center()[axis];
                });
                size_t n = sorted.size();
                std::vector<Bounds> pre(n), suf(n);
                pre[0] = bounds_of(sorted[0]);
                for (size_t i = 1; i < n; ++i) pre[i] = pre[i - 1].merge(bounds_of(sorted[i]));
                // suffix bounds must merge in the same left-to-right order as the reference;
                // min/max are exact and associative so any order gives identical values.
                suf[n - 1] = bounds_of(sorted[n - 1]);
                for (size_t i = n - 1; i-- > 0;) suf[i] = bounds_of(sorted[i]).merge(suf[i + 1]);
                float axis_min = std::numeric_limits<float>::infinity();
                size_t axis_i = 0;
                for (size_t i = 1; i < n; ++i) {
                    float cost = 1.0f + 1.0f * pre[i - 1].area() / parent_area * (float)i +
                                 1.0f * suf[i].area() / parent_area * (float)(n - i);
                    if (cost < axis_min) { axis_min = cost; axis_i = i; }
                }
                if (axis_min < min_cost) {
                    min_cost = axis_min; have = true;
                    best_first.assign(sorted.begin(), sorted.begin() + axis_i);
                    best_second.assign(sorted.begin() + axis_i, sorted.end());
                }
            }
            if (have) {
                size_t node_index = nodes.size();
                nodes.push_back(BvhNode{0, bounds, 0});
                build(best_first, list_bounds(best_first));
                nodes[node_index].value = (uint32_t)(nodes.size() - node_index);
                build(best_second, list_bounds(best_second));
                return;
            }
        }
        nodes.push_back(BvhNode{1, bounds, (uint32_t)items.size()});
        for (uint32_t it : items) nodes.push_back(BvhNode{2, Bounds{}, it});
    }
};

struct Bvh {
    std::vector<BvhNode> nodes;

    template <typename BoundsFn>
    static Bvh build(uint32_t n_items, const BoundsFn& bounds_of) {            // :325-332, :161-167
        Bvh bvh;
        std::vector<uint32_t> items(n_items);
        for (uint32_t i = 0; i < n_items; ++i) items[i] = i;
        BvhBuilder<BoundsFn> b{bounds_of, bvh.nodes};
        b.build(items, b.list_bounds(items));
        return bvh;
    }
    Bounds bounds() const { return nodes[0].bounds; }

    // ItemFn: bool(uint32_t item, const Ray&, float t_max, float* t_hit, Hit* hit)
    template <typename Hit, typename ItemFn>
    bool traverse_closest(size_t index, const Ray& ray, float t_max, V3 inv_dir, const ItemFn& item_fn, bool shrink,
                          float* t_out, Hit* hit_out, TraversalCounters* ctr) const {
        const BvhNode& nd = nodes[index];
        if (ctr) ctr->nodes++;
        if (!bounds_intersect(nd.bounds, ray, t_max, inv_dir)) return false;
        if (nd.kind == 0) {
            float t1, t2; Hit h1, h2;
            bool f1 = traverse_closest(index + 1, ray, t_max, inv_dir, item_fn, shrink, &t1, &h1, ctr);
            float t_max2 = (shrink && f1) ? std::min(t_max, t1) : t_max;
            bool f2 = traverse_closest(index + nd.value, ray, t_max2, inv_dir, item_fn, shrink, &t2, &h2, ctr);
            if (f1 && f2) {                                                     // :381-388
                if (t1 < t2) { *t_out = t1; *hit_out = h1; } else { *t_out = t2; *hit_out = h2; }
                return true;
            } else if (f1) { *t_out = t1; *hit_out = h1; return true; }
            else if (f2) { *t_out = t2; *hit_out = h2; return true; }
            return false;
        }
        bool found = false; float best_t = 0; Hit best{};
        float cur_max = t_max;
        for (uint32_t i = 1; i <= nd.value; ++i) {                              // :397-426
            float t; Hit h;
            if (ctr) ctr->items++;
            if (item_fn(nodes[index + i].value, ray, cur_max, &t, &h)) {
                if (!found || t < best_t) { found = true; best_t = t; best = h; }
                // ties must stay findable for the strict '<' above => inclusive t_max, which
                // the triangle test already is (t_scaled > t_max*det rejects only beyond).
                if (shrink) cur_max = std::min(cur_max, best_t);
            }
        }
        if (found) { *t_out = best_t; *hit_out = best; }
        return found;
    }

    template <typename Hit, typename ItemFn>
    bool intersect(const Ray& ray, float t_max, const ItemFn& item_fn, bool shrink, Hit* hit, TraversalCounters* ctr) const {
        V3 inv_dir{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};              // :433
        float t;
        return traverse_closest<Hit>(0, ray, t_max, inv_dir, item_fn, shrink, &t, hit, ctr);
    }

    // PFn: bool(uint32_t item, const Ray&, float t_max)
    template <typename PFn>
    bool traverse_any(size_t index, const Ray& ray, float t_max, V3 inv_dir, const PFn& item_fn, TraversalCounters* ctr) const {
        const BvhNode& nd = nodes[index];
        if (ctr) ctr->nodes++;
        if (!bounds_intersect(nd.bounds, ray, t_max, inv_dir)) return false;
        if (nd.kind == 0) {
            if (traverse_any(index + 1, ray, t_max, inv_dir, item_fn, ctr)) return true;
            return traverse_any(index + nd.value, ray, t_max, inv_dir, item_fn, ctr);
        }
        for (uint32_t i = 1; i <= nd.value; ++i) {
            if (ctr) ctr->items++;
            if (item_fn(nodes[index + i].value, ray, t_max)) return true;
        }
        return false;
    }
    template <typename PFn>
    bool intersect_p(const Ray& ray, float t_max, const PFn& item_fn, TraversalCounters* ctr) const {
        V3 inv_dir{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
        return traverse_any(0, ray, t_max, inv_dir, item_fn, ctr);
    }
};

}  // namespace oracle
