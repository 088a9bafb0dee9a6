// GPU BVH2 builder for gfx950: breadth-first binned SAH over the flattened render-space triangle soup.
//
// Replaces, for large scenes, the host sweep-SAH builder (bvh_builder.cpp), which itself stands where the reference
// has Bvh::build (scene/src/bvh.rs:92-230, O(n^2) split evaluation per level).  Same output contract as build_bvh():
// DevNode records holding BOTH child boxes, leaves of <= LEAF_MAX (4) leaf-ordered triangles, depth <= MAX_BUILD_DEPTH
// so the per-lane LDS traversal stack (STACK_DEPTH) cannot overflow, and a deterministic result.
//
// One level of the tree per round of launches.  Every open node ("work range") owns a contiguous range of the
// triangle permutation, so a leaf's triangles are simply its range and no leaf allocator is needed:
//   bin      every triangle adds its box, its centroid and a count to one of 32 bins of its range on each of the three
//            axes (u32 min/max/add atomics on order-preserving float keys; the range that owns the first
//            triangle of a workgroup is accumulated in LDS first: the top levels would otherwise serialise on
//            32 x 13 addresses)
//   split    one lane per range sweeps the 3 x 31 bin boundaries for the SAH minimum, or decides for a leaf
//   emit     after an exclusive scan over the ranges: DevNode records (breadth-first numbering), parent links,
//            the child ranges of the next level with their centroid bounds (merged from the bins)
//   scatter  stable partition of the permutation inside each range (exclusive scan of the "goes right" flags)
// Ranges that may not split freely any more (count > 2^(levels left), or all centroids equal) bin by RANK in the
// permutation and split in the middle; the permutation starts in Morton order and partitions are stable, so a rank
// split is still a spatial one.  That is the same depth rule as the host builder's forced object-median split.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "scene.hpp"

namespace pt {
namespace {

constexpr int NBINS = 32;
constexpr int ROWS = 13;                 // count, box lo xyz, box hi xyz, centroid lo xyz, centroid hi xyz
constexpr int AXIS_WORDS = ROWS * NBINS; // 208 u32 per work range and binning axis
constexpr int BIN_WORDS = 3 * AXIS_WORDS;  // x, y and z binned at once: the SAH sweep picks the best of 93 planes
constexpr int BLOCK = 256;
constexpr int ITEMS = 4;                 // triangles per thread in the per-triangle kernels
constexpr size_t BIN_BUFFER_RANGES = 1u << 15;   // 164 MB of bins

enum : uint32_t { MODE_SPATIAL = 0, MODE_RANK = 1 };
enum : uint32_t { DEC_LEAF = 0, DEC_SPLIT = 1 };

struct Work {
    uint32_t begin, end;
    int32_t parent;      // node whose child[slot] points here; -1: the root
    uint32_t slot;
    uint32_t mode, pad;
    float cmin[3], scale[3];   // bin on axis a = (c[a] - cmin[a]) * scale[a]; scale 0: axis without extent
};
struct Decision {
    uint32_t kind, n_left, split_bin, axis;
    uint32_t child_open[2];       // child becomes a work range of the next level (else it is a leaf already)
    float box[2][6];              // lo xyz, hi xyz
    float cbox[2][6];
};
struct Config { float cost_traverse, cost_tri; uint32_t leaf_max; };

__device__ __forceinline__ uint32_t fkey(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fval(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }
// rows 1-3 and 7-9 are minima, 4-6 and 10-12 maxima, row 0 a sum
__device__ __forceinline__ bool row_is_min(int row) { return (row >= 1 && row <= 3) || (row >= 7 && row <= 9); }
__device__ __forceinline__ uint32_t row_identity(int row) { return row_is_min(row) ? 0xffffffffu : 0u; }

__device__ __forceinline__ float box_area(const float* b) {
    float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    if (!(dx >= 0.0f)) return 0.0f;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}
__device__ __forceinline__ void box_reset(float* b) { b[0] = b[1] = b[2] = FLT_MAX; b[3] = b[4] = b[5] = -FLT_MAX; }
__device__ __forceinline__ void box_merge(float* b, const float* o) {
    for (int a = 0; a < 3; ++a) { b[a] = fminf(b[a], o[a]); b[3 + a] = fmaxf(b[3 + a], o[3 + a]); }
}

// Depth bound, the same rule as the host builder: a range with `levels_left` split levels below it can always be finished by
// median splits if n <= MAX_LEAF_TRIS << levels_left; SAH may choose freely only while both children stay inside that bound.
__device__ __forceinline__ bool depth_forces_median(uint32_t n, uint32_t depth) {
    const int levels_left = MAX_BUILD_DEPTH - (int)depth;
    if (levels_left <= 0) return false;                          // the range becomes a leaf
    return (uint64_t)n > ((uint64_t)MAX_LEAF_TRIS << (levels_left - 1));
}

__device__ __forceinline__ uint32_t bin_of(const Work& wk, const BuildTri& t, uint32_t i, uint32_t axis) {
    if (wk.mode == MODE_RANK) return (uint32_t)(((uint64_t)(i - wk.begin) * NBINS) / (uint64_t)(wk.end - wk.begin));
    float x = (t.c[axis] - wk.cmin[axis]) * wk.scale[axis];
    return (uint32_t)(int)fminf(fmaxf(x, 0.0f), (float)(NBINS - 1));   // fmaxf(NaN, 0) = 0
}

// ---------------------------------------------------------------- scene bounds + Morton codes
__global__ __launch_bounds__(BLOCK) void k_centroid_bounds(const BuildTri* __restrict__ tris, uint32_t n, uint32_t* __restrict__ out6) {
    __shared__ uint32_t s[6];
    if (threadIdx.x < 6) s[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
    __syncthreads();
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) { float c = tris[i].c[a]; lo[a] = fminf(lo[a], c); hi[a] = fmaxf(hi[a], c); }
    for (int a = 0; a < 3; ++a) { atomicMin(&s[a], fkey(lo[a])); atomicMax(&s[3 + a], fkey(hi[a])); }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&out6[threadIdx.x], s[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&out6[threadIdx.x], s[threadIdx.x]);
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) {   // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(BLOCK) void k_morton(const BuildTri* __restrict__ tris, uint32_t n, const uint32_t* __restrict__ bounds6,
                                                  uint32_t* __restrict__ codes, uint32_t* __restrict__ idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        float lo = fval(bounds6[a]), hi = fval(bounds6[3 + a]);
        float ext = hi - lo;
        float x = ext > 1e-20f ? (tris[i].c[a] - lo) * (1023.999f / ext) : 0.0f;
        q[a] = (uint32_t)(int)fminf(fmaxf(x, 0.0f), 1023.0f);
    }
    codes[i] = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
    idx[i] = i;
}

// the root range: all triangles, centroid bounds from k_centroid_bounds
__device__ __forceinline__ void open_range(Work* w, uint32_t begin, uint32_t end, int32_t parent, uint32_t slot, const float* cbox, uint32_t child_depth) {
    w->begin = begin; w->end = end; w->parent = parent; w->slot = slot;
    float ext_max = 0.0f;
    for (uint32_t a = 0; a < 3; ++a) {
        float e = cbox[3 + a] - cbox[a];
        w->cmin[a] = cbox[a];
        w->scale[a] = e > 1e-20f ? ((float)NBINS * (1.0f - 1e-6f)) / e : 0.0f;
        ext_max = fmaxf(ext_max, e);
    }
    uint32_t n = end - begin;
    bool rank = depth_forces_median(n, child_depth) || !(ext_max > 1e-20f);
    w->mode = rank ? MODE_RANK : MODE_SPATIAL; w->pad = 0;
}
__global__ void k_root(Work* work, const uint32_t* bounds6, uint32_t n) {
    float cb[6];
    for (int a = 0; a < 6; ++a) cb[a] = fval(bounds6[a]);
    open_range(&work[0], 0, n, -1, 0, cb, 0);
}

// ---------------------------------------------------------------- per level
__global__ __launch_bounds__(BLOCK) void k_init_bins(uint32_t* __restrict__ bins, uint32_t n_work) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_work * (uint32_t)BIN_WORDS) return;
    bins[i] = row_identity((int)((i % AXIS_WORDS) / NBINS));
}

template <bool LDS>
__device__ __forceinline__ void bin_add(uint32_t* b, uint32_t bin, const BuildTri& t) {
    atomicAdd(&b[bin], 1u);
    for (int a = 0; a < 3; ++a) {
        atomicMin(&b[(1 + a) * NBINS + bin], fkey(t.lo[a]));
        atomicMax(&b[(4 + a) * NBINS + bin], fkey(t.hi[a]));
        uint32_t ck = fkey(t.c[a]);
        atomicMin(&b[(7 + a) * NBINS + bin], ck);
        atomicMax(&b[(10 + a) * NBINS + bin], ck);
    }
}

__global__ __launch_bounds__(BLOCK) void k_bin(const BuildTri* __restrict__ tris, const uint32_t* __restrict__ idx,
                                               const int32_t* __restrict__ work_of, const Work* __restrict__ work,
                                               uint32_t* __restrict__ bins, uint32_t n, uint32_t w_lo, uint32_t w_hi) {
    // bins holds the ranges [w_lo, w_hi) of this pass (a level wider than the bin buffer takes several passes)
    __shared__ uint32_t s_bins[BIN_WORDS];
    __shared__ int32_t s_w0;
    const uint32_t base = blockIdx.x * (BLOCK * ITEMS);
    if (threadIdx.x == 0) {
        int32_t w = work_of[base];   // base < n by the grid size
        s_w0 = (w >= (int32_t)w_lo && w < (int32_t)w_hi) ? w : -1;
    }
    for (uint32_t k = threadIdx.x; k < BIN_WORDS; k += BLOCK) s_bins[k] = row_identity((int)((k % AXIS_WORDS) / NBINS));
    __syncthreads();
    const int32_t w0 = s_w0;
    for (int k = 0; k < ITEMS; ++k) {
        uint32_t i = base + k * BLOCK + threadIdx.x;
        if (i >= n) break;
        int32_t w = work_of[i];
        if (w < (int32_t)w_lo || w >= (int32_t)w_hi) continue;
        Work wk = work[w];
        BuildTri t = tris[idx[i]];
        const uint32_t n_axes = wk.mode == MODE_RANK ? 1u : 3u;   // rank bins live in the x slot
        for (uint32_t a = 0; a < n_axes; ++a) {
            if (wk.mode == MODE_SPATIAL && wk.scale[a] == 0.0f) continue;
            uint32_t b = bin_of(wk, t, i, a);
            if (w == w0) bin_add<true>(s_bins + a * AXIS_WORDS, b, t);
            else bin_add<false>(bins + (size_t)(w - w_lo) * BIN_WORDS + a * AXIS_WORDS, b, t);
        }
    }
    __syncthreads();
    if (w0 >= 0) {
        for (uint32_t k = threadIdx.x; k < BIN_WORDS; k += BLOCK) {
            int row = (int)((k % AXIS_WORDS) / NBINS);
            uint32_t v = s_bins[k];
            if (v != row_identity(row)) {
                uint32_t* g = bins + (size_t)(w0 - w_lo) * BIN_WORDS + k;
                if (row == 0) atomicAdd(g, v);
                else if (row_is_min(row)) atomicMin(g, v);
                else atomicMax(g, v);
            }
        }
    }
}

// SAH over the 15 bin boundaries of one range (or the forced middle split of a rank-binned one)
__global__ __launch_bounds__(BLOCK) void k_split(const Work* __restrict__ work, const uint32_t* __restrict__ bins, Decision* __restrict__ dec,
                                                 uint64_t* __restrict__ alloc, uint32_t w_lo, uint32_t w_hi, uint32_t depth, Config cfg,
                                                 uint32_t* __restrict__ error) {
    uint32_t w = w_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= w_hi) return;
    const Work wk = work[w];
    const uint32_t n = wk.end - wk.begin;
    const bool must_leaf = (int)depth >= MAX_BUILD_DEPTH;       // n <= MAX_LEAF_TRIS here, by the bound in depth_forces_median
    const bool force_median = depth_forces_median(n, depth);
    float best_cost = FLT_MAX; int best = -1; uint32_t best_axis = 0;
    const uint32_t n_axes = wk.mode == MODE_RANK ? 1u : 3u;
    for (uint32_t axis = 0; axis < n_axes; ++axis) {
        if (wk.mode == MODE_SPATIAL && wk.scale[axis] == 0.0f) continue;
        const uint32_t* b = bins + (size_t)(w - w_lo) * BIN_WORDS + axis * AXIS_WORDS;
        uint32_t cnt[NBINS];
        float right_area[NBINS];   // area of bins [s, NBINS)
        uint32_t right_cnt[NBINS];
        float acc[6];
        box_reset(acc);
        uint32_t c = 0;
        for (int s = NBINS - 1; s >= 0; --s) {
            cnt[s] = b[s];
            if (cnt[s]) {
                float bb[6];
                for (int a = 0; a < 3; ++a) { bb[a] = fval(b[(1 + a) * NBINS + s]); bb[3 + a] = fval(b[(4 + a) * NBINS + s]); }
                box_merge(acc, bb);
                c += cnt[s];
            }
            right_area[s] = box_area(acc); right_cnt[s] = c;
        }
        if (c != n) atomicOr(error, 1u);   // every triangle of the range must have been binned
        const float inv_area = 1.0f / fmaxf(right_area[0], 1e-30f);
        box_reset(acc);
        uint32_t lc = 0;
        for (int s = 1; s < NBINS; ++s) {
            if (cnt[s - 1]) {
                float bb[6];
                for (int a = 0; a < 3; ++a) { bb[a] = fval(b[(1 + a) * NBINS + s - 1]); bb[3 + a] = fval(b[(4 + a) * NBINS + s - 1]); }
                box_merge(acc, bb);
                lc += cnt[s - 1];
            }
            if (lc == 0 || right_cnt[s] == 0) continue;
            if (wk.mode == MODE_RANK && s != NBINS / 2) continue;
            float cost = cfg.cost_traverse + cfg.cost_tri * inv_area * (box_area(acc) * (float)lc + right_area[s] * (float)right_cnt[s]);
            if (cost < best_cost) { best_cost = cost; best = s; best_axis = axis; }
        }
    }
    Decision d;
    memset(&d, 0, sizeof(d));
    const float leaf_cost = cfg.cost_tri * (float)n;
    if (must_leaf || (!force_median && n <= cfg.leaf_max && leaf_cost <= best_cost)) {
        d.kind = DEC_LEAF;
        dec[w] = d;
        alloc[w] = 0;
        return;
    }
    if (best < 0) {   // cannot happen: spatial ranges have both end bins occupied, rank ranges have n >= 2
        atomicOr(error, 2u);
        d.kind = DEC_LEAF; dec[w] = d; alloc[w] = 0;
        return;
    }
    d.kind = DEC_SPLIT; d.split_bin = (uint32_t)best; d.axis = best_axis;
    for (int side = 0; side < 2; ++side) { box_reset(d.box[side]); box_reset(d.cbox[side]); }
    uint32_t nl = 0;
    const uint32_t* b = bins + (size_t)(w - w_lo) * BIN_WORDS + best_axis * AXIS_WORDS;
    for (int s = 0; s < NBINS; ++s) {
        const uint32_t cnt_s = b[s];
        if (!cnt_s) continue;
        int side = s >= best;
        float bb[6], cb[6];
        for (int a = 0; a < 3; ++a) {
            bb[a] = fval(b[(1 + a) * NBINS + s]); bb[3 + a] = fval(b[(4 + a) * NBINS + s]);
            cb[a] = fval(b[(7 + a) * NBINS + s]); cb[3 + a] = fval(b[(10 + a) * NBINS + s]);
        }
        box_merge(d.box[side], bb); box_merge(d.cbox[side], cb);
        if (!side) nl += cnt_s;
    }
    d.n_left = nl;
    d.child_open[0] = nl > 1; d.child_open[1] = (n - nl) > 1;
    dec[w] = d;
    alloc[w] = 1ull | ((uint64_t)(d.child_open[0] + d.child_open[1]) << 32);   // low: DevNode records, high: ranges of the next level
}

__global__ __launch_bounds__(BLOCK) void k_emit(const Work* __restrict__ work, const Decision* __restrict__ dec, const uint64_t* __restrict__ alloc_scan,
                                                DevNode* __restrict__ nodes, Work* __restrict__ next_work, uint32_t n_work, uint32_t node_base,
                                                uint32_t depth, uint32_t* __restrict__ root_link) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_work) return;
    const Work wk = work[w];
    const Decision d = dec[w];
    int32_t link;
    if (d.kind == DEC_LEAF) {
        link = make_leaf(wk.begin, wk.end - wk.begin);
    } else {
        const uint64_t a = alloc_scan[w];
        const uint32_t node = node_base + (uint32_t)(a & 0xffffffffu);
        uint32_t next = (uint32_t)(a >> 32);
        link = (int32_t)node;
        DevNode nd;
        nd.bx[0] = d.box[0][0]; nd.bx[1] = d.box[1][0]; nd.bx[2] = d.box[0][3]; nd.bx[3] = d.box[1][3];
        nd.by[0] = d.box[0][1]; nd.by[1] = d.box[1][1]; nd.by[2] = d.box[0][4]; nd.by[3] = d.box[1][4];
        nd.bz[0] = d.box[0][2]; nd.bz[1] = d.box[1][2]; nd.bz[2] = d.box[0][5]; nd.bz[3] = d.box[1][5];
        nd.pad[0] = nd.pad[1] = 0;
        const uint32_t mid = wk.begin + d.n_left;
        for (uint32_t side = 0; side < 2; ++side) {
            uint32_t cb = side ? mid : wk.begin, ce = side ? wk.end : mid;
            if (d.child_open[side]) {
                open_range(&next_work[next], cb, ce, (int32_t)node, side, d.cbox[side], depth + 1);
                nd.child[side] = 0;   // patched by the child's own emit on the next level
                ++next;
            } else {
                nd.child[side] = make_leaf(cb, 1);
            }
        }
        // the children write their links into child[] on the next level: store the boxes now, the links of open children later
        nodes[node] = nd;
    }
    if (wk.parent < 0) *root_link = (uint32_t)link;
    else nodes[wk.parent].child[wk.slot] = link;
}

__global__ __launch_bounds__(BLOCK) void k_flags(const BuildTri* __restrict__ tris, const uint32_t* __restrict__ idx, const int32_t* __restrict__ work_of,
                                                 const Work* __restrict__ work, const Decision* __restrict__ dec, uint32_t* __restrict__ flags, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t w = work_of[i];
    uint32_t f = 0;
    if (w >= 0 && dec[w].kind == DEC_SPLIT) {
        Work wk = work[w];
        f = bin_of(wk, tris[idx[i]], i, dec[w].axis) >= dec[w].split_bin;
    }
    flags[i] = f;
}

__global__ __launch_bounds__(BLOCK) void k_scatter(const uint32_t* __restrict__ idx, const int32_t* __restrict__ work_of, const Work* __restrict__ work,
                                                   const Decision* __restrict__ dec, const uint64_t* __restrict__ alloc_scan,
                                                   const uint32_t* __restrict__ flags, const uint32_t* __restrict__ flag_scan,
                                                   uint32_t* __restrict__ idx_out, int32_t* __restrict__ work_of_out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t w = work_of[i];
    if (w < 0 || dec[w].kind != DEC_SPLIT) { idx_out[i] = idx[i]; work_of_out[i] = -1; return; }
    const Work wk = work[w];
    const uint32_t rights_before = flag_scan[i] - flag_scan[wk.begin];
    const uint32_t side = flags[i];
    const uint32_t n_left = dec[w].n_left;
    const uint32_t pos = side ? wk.begin + n_left + rights_before : i - rights_before;
    const uint32_t open0 = dec[w].child_open[0], open1 = dec[w].child_open[1];
    const uint32_t next = (uint32_t)(alloc_scan[w] >> 32);
    idx_out[pos] = idx[i];
    work_of_out[pos] = side ? (open1 ? (int32_t)(next + open0) : -1) : (open0 ? (int32_t)next : -1);
}

__global__ __launch_bounds__(BLOCK) void k_fill_i32(int32_t* p, int32_t v, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct DeviceArena {
    std::vector<void*> ptrs;
    ~DeviceArena() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T> T* alloc(size_t count, bool* ok) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) { *ok = false; return nullptr; }
        ptrs.push_back(p);
        return (T*)p;
    }
};

inline unsigned blocks_for(size_t n, unsigned per_block) { return (unsigned)((n + per_block - 1) / per_block); }

}  // namespace

#define GPU_CHECK(expr)                                                                                      \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) { *err = std::string("GPU BVH build: ") + #expr + ": " + hipGetErrorString(e_); return false; } \
    } while (0)

bool build_bvh_gpu(const std::vector<BuildTri>& tris, BvhOut* out, double* device_ms, std::string* err) {
    const uint32_t n = (uint32_t)tris.size();
    float cost_traverse, cost_tri; int leaf_max;
    bvh_build_config(&cost_traverse, &cost_tri, &leaf_max);
    if (n < 8) { *err = "GPU BVH build: fewer than 8 triangles (use the host builder)"; return false; }
    if ((uint64_t)n > ((uint64_t)MAX_LEAF_TRIS << MAX_BUILD_DEPTH)) { *err = "GPU BVH build: too many triangles for the depth bound"; return false; }
    Config cfg{cost_traverse, cost_tri, (uint32_t)leaf_max};

    bool ok = true;
    DeviceArena mem;
    const size_t max_work = (size_t)n / 2 + 1;
    BuildTri* d_tris = mem.alloc<BuildTri>(n, &ok);
    uint32_t* d_idx[2] = {mem.alloc<uint32_t>(n, &ok), mem.alloc<uint32_t>(n, &ok)};
    int32_t* d_work_of[2] = {mem.alloc<int32_t>(n, &ok), mem.alloc<int32_t>(n, &ok)};
    uint32_t* d_codes[2] = {mem.alloc<uint32_t>(n, &ok), mem.alloc<uint32_t>(n, &ok)};   // reused as flags / flag scan after the sort
    Work* d_work[2] = {mem.alloc<Work>(max_work, &ok), mem.alloc<Work>(max_work, &ok)};
    Decision* d_dec = mem.alloc<Decision>(max_work, &ok);
    uint64_t* d_alloc = mem.alloc<uint64_t>(max_work + 1, &ok);
    uint64_t* d_alloc_scan = mem.alloc<uint64_t>(max_work + 1, &ok);
    // 4.9 KB of bins per open range: a fixed buffer, levels with more ranges than it holds are binned in several passes
    const uint32_t bin_ranges = (uint32_t)std::min<size_t>(max_work, BIN_BUFFER_RANGES);
    uint32_t* d_bins = mem.alloc<uint32_t>((size_t)bin_ranges * BIN_WORDS, &ok);
    DevNode* d_nodes = mem.alloc<DevNode>(n, &ok);
    uint32_t* d_small = mem.alloc<uint32_t>(16, &ok);   // [0..5] centroid bounds keys, [6] root link, [7] error flags
    if (!ok) { *err = "GPU BVH build: out of device memory"; return false; }

    size_t tmp_bytes = 0, need = 0;
    GPU_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, d_codes[0], d_codes[1], d_idx[0], d_idx[1], (int)n, 0, 30, (hipStream_t)0));
    tmp_bytes = std::max(tmp_bytes, need);
    GPU_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_codes[0], d_codes[1], (int)n, (hipStream_t)0));
    tmp_bytes = std::max(tmp_bytes, need);
    GPU_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_alloc, d_alloc_scan, (int)(max_work + 1), (hipStream_t)0));
    tmp_bytes = std::max(tmp_bytes, need);
    void* d_tmp = mem.alloc<uint8_t>(tmp_bytes, &ok);
    if (!ok) { *err = "GPU BVH build: out of device memory"; return false; }

    hipEvent_t ev0, ev1;
    GPU_CHECK(hipEventCreate(&ev0)); GPU_CHECK(hipEventCreate(&ev1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } ev_guard{ev0, ev1};
    GPU_CHECK(hipMemcpy(d_tris, tris.data(), (size_t)n * sizeof(BuildTri), hipMemcpyHostToDevice));
    GPU_CHECK(hipEventRecord(ev0, 0));

    // scene centroid bounds, Morton order
    const uint32_t small_init[16] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0};
    GPU_CHECK(hipMemcpy(d_small, small_init, sizeof(small_init), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_centroid_bounds, dim3(std::min(blocks_for(n, BLOCK), 1024u)), dim3(BLOCK), 0, 0, d_tris, n, d_small);
    hipLaunchKernelGGL(k_morton, dim3(blocks_for(n, BLOCK)), dim3(BLOCK), 0, 0, d_tris, n, d_small, d_codes[0], d_idx[1]);
    need = tmp_bytes;
    GPU_CHECK(hipcub::DeviceRadixSort::SortPairs(d_tmp, need, d_codes[0], d_codes[1], d_idx[1], d_idx[0], (int)n, 0, 30, (hipStream_t)0));
    hipLaunchKernelGGL(k_root, dim3(1), dim3(1), 0, 0, d_work[0], d_small, n);
    hipLaunchKernelGGL(k_fill_i32, dim3(blocks_for(n, BLOCK)), dim3(BLOCK), 0, 0, d_work_of[0], 0, n);
    uint32_t* d_flags = d_codes[0];
    uint32_t* d_flag_scan = d_codes[1];

    uint32_t n_work = 1, n_nodes = 0, depth = 0;
    int cur = 0, max_depth = 0;
    while (n_work > 0) {
        if (depth >= (uint32_t)STACK_DEPTH) { *err = "GPU BVH build: tree deeper than the traversal stack"; return false; }
        for (uint32_t w_lo = 0; w_lo < n_work; w_lo += bin_ranges) {
            const uint32_t w_hi = std::min(n_work, w_lo + bin_ranges);
            hipLaunchKernelGGL(k_init_bins, dim3(blocks_for((size_t)(w_hi - w_lo) * BIN_WORDS, BLOCK)), dim3(BLOCK), 0, 0, d_bins, w_hi - w_lo);
            hipLaunchKernelGGL(k_bin, dim3(blocks_for(n, BLOCK * ITEMS)), dim3(BLOCK), 0, 0, d_tris, d_idx[cur], d_work_of[cur], d_work[cur], d_bins, n, w_lo, w_hi);
            hipLaunchKernelGGL(k_split, dim3(blocks_for(w_hi - w_lo, BLOCK)), dim3(BLOCK), 0, 0, d_work[cur], d_bins, d_dec, d_alloc, w_lo, w_hi, depth, cfg,
                               d_small + 7);
        }
        GPU_CHECK(hipMemsetAsync(d_alloc + n_work, 0, sizeof(uint64_t), 0));
        need = tmp_bytes;
        GPU_CHECK(hipcub::DeviceScan::ExclusiveSum(d_tmp, need, d_alloc, d_alloc_scan, (int)(n_work + 1), (hipStream_t)0));
        hipLaunchKernelGGL(k_emit, dim3(blocks_for(n_work, BLOCK)), dim3(BLOCK), 0, 0, d_work[cur], d_dec, d_alloc_scan, d_nodes, d_work[cur ^ 1], n_work,
                           n_nodes, depth, d_small + 6);
        hipLaunchKernelGGL(k_flags, dim3(blocks_for(n, BLOCK)), dim3(BLOCK), 0, 0, d_tris, d_idx[cur], d_work_of[cur], d_work[cur], d_dec, d_flags, n);
        need = tmp_bytes;
        GPU_CHECK(hipcub::DeviceScan::ExclusiveSum(d_tmp, need, d_flags, d_flag_scan, (int)n, (hipStream_t)0));
        hipLaunchKernelGGL(k_scatter, dim3(blocks_for(n, BLOCK)), dim3(BLOCK), 0, 0, d_idx[cur], d_work_of[cur], d_work[cur], d_dec, d_alloc_scan, d_flags,
                           d_flag_scan, d_idx[cur ^ 1], d_work_of[cur ^ 1], n);
        uint64_t totals = 0;
        GPU_CHECK(hipMemcpy(&totals, d_alloc_scan + n_work, sizeof(totals), hipMemcpyDeviceToHost));   // synchronises the level
        const uint32_t new_nodes = (uint32_t)(totals & 0xffffffffu), next_work = (uint32_t)(totals >> 32);
        if (new_nodes > 0) max_depth = (int)depth + 1;
        if ((size_t)n_nodes + new_nodes > n || next_work > max_work) { *err = "GPU BVH build: internal count overflow"; return false; }
        n_nodes += new_nodes;
        n_work = next_work;
        cur ^= 1;
        ++depth;
    }
    GPU_CHECK(hipEventRecord(ev1, 0));
    GPU_CHECK(hipEventSynchronize(ev1));
    float ms = 0.0f;
    GPU_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
    if (device_ms) *device_ms = ms;

    uint32_t small[16];
    GPU_CHECK(hipMemcpy(small, d_small, sizeof(small), hipMemcpyDeviceToHost));
    if (small[7] != 0) { *err = "GPU BVH build: inconsistent bins (flags " + std::to_string(small[7]) + ")"; return false; }
    if (n_nodes == 0 || (int32_t)small[6] != 0) { *err = "GPU BVH build: no root node"; return false; }
    out->nodes.resize(n_nodes);
    out->order.resize(n);
    GPU_CHECK(hipMemcpy(out->nodes.data(), d_nodes, (size_t)n_nodes * sizeof(DevNode), hipMemcpyDeviceToHost));
    GPU_CHECK(hipMemcpy(out->order.data(), d_idx[cur], (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    out->root = 0;
    out->max_depth = max_depth;
    return true;
}

}  // namespace pt
