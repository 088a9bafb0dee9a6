// The sample-loop kernel template (see pt_kernels.hip for the design notes).  A header so that the specialisations by
// MODE are compiled in separate translation units (pt_kernels.hip: generic + instrumented + probe; pt_kernels_mis.hip,
// pt_kernels_nee.hip), in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.hpp"
#include "pt_device.hpp"
#include "pt_path.hpp"

namespace pt {

#ifndef PT_MIN_WAVES_CC
#define PT_MIN_WAVES_CC 3
#endif
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 4   // 128 VGPRs: measured +26 % over the unconstrained 220-VGPR build (latency hiding beats the spills)
#endif
// work item -> lane assignment
struct LaneJob { uint32_t px, py, s_cur, s_end; bool valid; };
PT_DEV LaneJob lane_job(uint32_t work, uint32_t lane, const DevCamera& cam, const DevParams& prm) {
    LaneJob j{0, 0, 0, 0, false};
    // work = ((tile * blocks per tile) + block) * chunks + chunk; lanes >= 4^b own no pixel of the block
    const uint32_t b = prm.block_log2, bside = 1u << b;
    uint32_t item = work / prm.chunks, chunk = work % prm.chunks;
    uint32_t tile_k = item >> (6u - 2u * b), blk = item & ((64u >> (2u * b)) - 1u);
    uint32_t tile = prm.shard_index + tile_k * prm.shard_count;
    uint32_t tx = tile % prm.tiles_x, ty = tile / prm.tiles_x;
    uint32_t bx = blk & ((8u >> b) - 1u), by = blk >> (3u - b);
    j.px = tx * 8 + bx * bside + (lane & (bside - 1u)); j.py = ty * 8 + by * bside + ((lane >> b) & (bside - 1u));
    j.valid = lane < (1u << (2u * b)) && j.px < cam.width && j.py < cam.height;
    j.s_cur = prm.sample_begin + chunk * prm.chunk_size;
    j.s_end = min(j.s_cur + prm.chunk_size, prm.sample_end);
    return j;
}

PT_DEV void flush_stats(DevStats* stats, const StatCounters& st) {
    atomicAdd(&stats->samples, (unsigned long long)st.samples);
    atomicAdd(&stats->closest_rays, (unsigned long long)st.closest_rays);
    atomicAdd(&stats->shadow_rays, (unsigned long long)st.shadow_rays);
    atomicAdd(&stats->nodes_closest, (unsigned long long)st.nodes_closest);
    atomicAdd(&stats->tris_closest, (unsigned long long)st.tris_closest);
    atomicAdd(&stats->nodes_shadow, (unsigned long long)st.nodes_shadow);
    atomicAdd(&stats->tris_shadow, (unsigned long long)st.tris_shadow);
    atomicAdd(&stats->closest_hits, (unsigned long long)st.closest_hits);
    atomicAdd(&stats->bounces, (unsigned long long)st.bounces);
    atomicAdd(&stats->spectrum_evals, (unsigned long long)st.spectrum_evals);
    atomicAdd(&stats->textured_lookups, (unsigned long long)st.textured_lookups);
    for (int i = 0; i < 8; ++i) if (st.w[i]) atomicAdd(&stats->wave_steps[i], (unsigned long long)st.w[i]);
    for (int i = 0; i < 16; ++i) if (st.hist[i]) atomicAdd(&stats->busy_hist[i >> 3][i & 7], (unsigned long long)st.hist[i]);
    for (int i = 0; i < 4; ++i) if (st.dv[i]) atomicAdd(&stats->divergence[i], (unsigned long long)st.dv[i]);
    if (st.ties) atomicAdd(&stats->phase_cycles[9], (unsigned long long)st.ties | ((unsigned long long)st.ties_differ << 32));
}

#ifndef PT_ANY_DEFERRED
#define PT_ANY_DEFERRED 1
#endif
// Wave priority by stage (s_setprio; the SIMD's issue arbitration goes by priority, then age): a wave in a traversal is a chain of dependent
// round trips and loses nothing by yielding the issue port, a wave in the shading stage has independent work to issue — shading and hand-out
// run at priority 3, traversals at 0: +0.9...+1.3 % (scenes 3 / 0 / 8 / 17; 0.0 on 15 / 19); the other way round -1.1 %.
#ifndef PT_PRIO_TRAV
#define PT_PRIO_TRAV 2             // 0: no priorities, 1: traversals high, 2: traversals low
#endif
#if PT_PRIO_TRAV == 1
#define PT_PRIO_TRAV_ENTER __builtin_amdgcn_s_setprio(3)
#define PT_PRIO_TRAV_EXIT __builtin_amdgcn_s_setprio(0)
#elif PT_PRIO_TRAV == 2
#define PT_PRIO_TRAV_ENTER __builtin_amdgcn_s_setprio(0)
#define PT_PRIO_TRAV_EXIT __builtin_amdgcn_s_setprio(3)
#else
#define PT_PRIO_TRAV_ENTER ((void)0)
#define PT_PRIO_TRAV_EXIT ((void)0)
#endif
// ONE cooperative traversal per iteration for the next closest-hit rays AND the light connections of the vertex just shaded (trace_pair_coop,
// pt_device.hpp): a wave's step count is bounded by its deepest ray, not by the ray count, so the two traversals together cost ~20 node
// steps instead of ~18 + ~14.  The price is the pending connection's 11 registers across one more stage, and a heavier loop where there
// are no connections at all.  Measured per kernel (same box):
//   kernels without the clearcoat code: +2.4...+5 % (scene 3 1 716 -> 1 757, scene 0 1 814 -> 1 906, scene 8 1 525 -> 1 574, scene 10 1 519 -> 1 583);
//   clearcoat kernels (3 waves per SIMD), with the LDS parking and the sinking in place: MIS +1.7 / +5.5 % (scenes 15 / 19); NEE +2.2 % (scene 17,
//   C5's kernel: 1 242 -> 1 270), +2.2 / +3.6 % (scenes 16 / 20), +1.5 % (scene 19) — but -1.1 / -2.1 % in the clearcoat + texture set under NEE or
//   the generic mode (scenes 15 / 18), which keeps two traversals;
//   the plain path tracer (strategy pt: no connections exist) loses 9 % with it (scene 3 2 790 vs 2 557): it has its own specialisation
//   (MODE_PT, pt_kernels_pt.hip) without it; choosing inside one kernel at run time costs both sides (-5 % / -8 %: measured);
//   generic-mode clearcoat kernels (random sampler with NEE / MIS): not measured, two traversals as before.
#ifndef PT_MERGED_TRAVERSAL
#define PT_MERGED_TRAVERSAL 1      // 0: never, 1: as measured (above), 2: every kernel
#endif
enum : uint32_t { MODE_GENERIC = 0, MODE_MIS_SOBOL = 1, MODE_NEE_SOBOL = 2, MODE_PT = 3 };
template <uint32_t FEAT, uint32_t MODE> constexpr bool merged_traversal() {
    if (PT_MERGED_TRAVERSAL != 1) return PT_MERGED_TRAVERSAL == 2;
    if (MODE == MODE_PT) return false;
    if ((FEAT & FEAT_CC) == 0u || MODE == MODE_MIS_SOBOL) return true;
    return MODE == MODE_NEE_SOBOL && FEAT != (FEAT_CC | FEAT_TEX);
}
#ifndef PT_CLOSEST_COOP
#define PT_CLOSEST_COOP 1      // needs PT_ANY_DEFERRED (shares its LDS ring)
#endif
// Straggler carry-over of the merged traversal (trace_pair_coop, pt_device.hpp): the traversal returns when at most this many lanes are still
// walking; the owners of the unfinished rays sit out one shading stage and their rays walk on beside the next iteration's.  0: off.
#ifndef PT_CARRY_MAX
#define PT_CARRY_MAX 0
#endif
template <uint32_t FEAT, uint32_t MODE> constexpr int carry_max() { return PT_CARRY_MAX; }
// MATERIAL SORT BETWEEN BOUNCES, wave-local (PT_DEFER).  A wave shades all its lanes together, so an iteration in which a few lanes hit
// the hero pays the hero's whole BSDF branch on top of the room's: measured 19-52 k of ~270-300 k cycles per iteration wherever the scene
// has a second material class, i.e. in 84-93 % of the iterations for 8-13 % of the lanes.  With PT_DEFER a lane whose closest
// hit lies on a DEFERRED class (the clearcoat material in the kernels that have it, the dielectrics in theirs) does not shade: it writes its
// path and the hit to the wave's own queue in global memory (128 B per path, a ring of 128 entries per resident wave; L2-resident) and is
// FREE — it starts a new camera path at the top of the next iteration like a lane whose path ended, so no lane idles for the sort.  When at
// least PT_DEFER_MIN paths wait (or the work item has no new paths left), the free lanes of the next iteration take queued paths instead of new
// ones and the wave shades them together: the minority branch runs once for ~30 lanes instead of in every iteration for ~6.  A path only
// ever waits in the queue of the wave that owns its pixel's LDS film tile, the order of pushes and pops is a function of the wave's own
// deterministic schedule, and a sample's arithmetic does not depend on when it is shaded: frames stay bit-identical from run to run and
// sample-for-sample equal to the oracle's.  (Moving paths BETWEEN waves — the usual wavefront formulation — needs 9 KB of LDS per queue or
// float atomics on the film; sorting inside the wave's own time line needs neither.)
#ifndef PT_DEFER
#define PT_DEFER 2             // 0: off; 1: queued paths are taken at the top of an iteration (they sit out its traversal: +4 % scene 17, -5...-9 % dielectrics);
#endif                         // 2: taken after the shading stage and shaded at once in a second pass (+11...+34 %, dielectrics +-0)
#ifndef PT_DEFER_MIN
#define PT_DEFER_MIN 56         // paths waiting before the queue is shaded (20 / 28 / 40 / 48 / 60 measured: scene 17 1 497 / 1 512 / 1 523 / 1 528 / 1 529)
#endif
constexpr uint32_t DEFER_RING = 128u;           // entries per wave: a drain starts at PT_DEFER_MIN waiting paths and one iteration adds at most 64
constexpr uint32_t DEFER_F4 = 8u;               // float4 per entry
#ifndef PT_DEFER_TEX
#define PT_DEFER_TEX 1      // textured materials are a class of their own in the kernels without the clearcoat code
#endif
template <uint32_t FEAT, uint32_t MODE> constexpr uint32_t defer_classes() {    // bit c: sort class c (MT_* | 8 if the material has a spectrum texture, DevTri::pad[0]) is deferred
    if (PT_DEFER == 0 || !merged_traversal<FEAT, MODE>()) return 0u;
    // the clearcoat material wherever it exists (its branch is the longest: +12 % scene 17, +22 % scene 19, +34 % scene 15); in the kernels
    // without it, the materials with a spectrum texture (bilinear fetches + rgb2spec cells; C2's hero: +0.6 %, scene 4 +0.8 %; a class for
    // every textured material lost 4 % on the normal-map-only hero of scene 5).  The dielectrics alone measured +-0 (their branch is short)
    // and are not deferred.
    if ((FEAT & FEAT_CC) != 0u) return (1u << MT_CLEARCOAT) | (1u << (MT_CLEARCOAT | 8u));
    if (PT_DEFER_TEX != 0 && (FEAT & FEAT_TEX) != 0u) return 0xff00u;
    return 0u;
}
// TAIL QUEUE (PT_TAILQ): the same idea for EVERY path.  A third of the lanes that enter the shading stage end their path in its front
// (emission, roulette) and used to idle through BSDF sampling and the light connection — 28 % of a wave's time at 65 % of its lanes.  Now the
// shading stage is two: (1) the FRONT of the vertex for every lane that traced (emission with its weight, throughput, roulette, depth:
// shade_vertex_head<PHASE 1>); the paths that go on are written to the wave's queue (path + hit, the 128-byte record of PT_DEFER) and ALL
// lanes are free; (2) when 64 paths wait, one pass takes them into the free lanes and runs the back of the vertex (the surface again from
// the hit, frames, the BSDF's draws: <PHASE 2>) and the tail — BSDF sample, light connection — for a FULL wave.  Lanes left free start new
// camera paths as before.  In the clearcoat kernels the paths whose hit is on the clearcoat material have a queue of their own, so a pass
// shades one class (this subsumes PT_DEFER there); when a work item has nothing new left, whatever waits shares the passes.  A sample's
// arithmetic and its sampler dimensions are what they were — the record carries the path between the two halves of ITS vertex —, the
// queues belong to the wave that owns the pixels, the schedule is the wave's own: frames bit-identical from run to run, 166 GPU tests
// unchanged.  Same-box A/B (round 3): **C2 2 088 -> 2 312 Msamples/s (+10.7 %), C3 +11.9 %, C4 +13.2 %, C5 1 574 -> 1 893 (+20 %)**, scene 19
// +37 %, scene 15 +22 %, scenes 0 / 5 +10 / +12 %.  Measured on the way: a threshold of 48 instead of 64 waiting paths gives +7 % instead of
// +10.7 % (passes not full); a second queue for every class but plain Lambert in the kernels WITHOUT clearcoat loses everything again
// (a class that is a tenth of the hits leaves up to 63 paths to be bounced out in sparse passes at the end of every work item); writing
// the pass with the back of the head and the tail in two divergent regions instead of one costs those kernels 12 % (ShadeCtx across a
// re-convergence point at 128 VGPRs, as round 1 found for the fused shader).
#ifndef PT_TAILQ
#define PT_TAILQ 1
#endif
#ifndef PT_TAILQ_MIN
#define PT_TAILQ_MIN 64
#endif
#ifndef PT_TAILQ_CC
#define PT_TAILQ_CC 1        // the clearcoat kernels too (two queues: one per sort class)
#endif
template <uint32_t FEAT, uint32_t MODE> constexpr bool tail_queue() { return PT_TAILQ != 0 && merged_traversal<FEAT, MODE>() && ((FEAT & FEAT_CC) == 0u || PT_TAILQ_CC != 0); }
constexpr uint32_t QUEUE_RING = (PT_TAILQ != 0) ? 256u : DEFER_RING;     // entries per wave and queue
constexpr uint32_t QUEUE_MAX = (PT_TAILQ != 0) ? 2u : 1u;               // queues per wave (the tail queue keeps one per sort class)
// the queue's records are written once and read once: with PT_TQ_NT their stores / loads carry the non-temporal hint, so that they do not
// push the BVH out of the L2 (L2 hit rate 0.98 before the queues, 0.89 with them).  Measured: -4 % (C2 2 578 -> 2 481, C4 -5.6 %): the
// records then come back from HBM instead of the L2 / Infinity Cache; off
#ifndef PT_TQ_NT
#define PT_TQ_NT 0
#endif
typedef float pt_v4f __attribute__((ext_vector_type(4)));
#if PT_TQ_NT
#define PT_TQ_ST(p, v) do { const float4 _v = (v); pt_v4f _w = {_v.x, _v.y, _v.z, _v.w}; __builtin_nontemporal_store(_w, (pt_v4f*)(p)); } while (0)
#define PT_TQ_LD(p) ([&]() { const pt_v4f _w = __builtin_nontemporal_load((const pt_v4f*)(p)); return make_float4(_w.x, _w.y, _w.z, _w.w); }())
#else
#define PT_TQ_ST(p, v) (*(p) = (v))
#define PT_TQ_LD(p) (*(p))
#endif
static_assert(PT_TAILQ == 0 || QUEUE_RING == 256u, "the tail queue's entry index keeps the queue number in bit 8");
constexpr uint32_t TQ_F4 = 5u;                                           // float4 per record of the tail queue (80 B; PT_DEFER's record: 8)
constexpr size_t defer_bytes_per_wave() { return (PT_DEFER || PT_TAILQ) ? (size_t)QUEUE_MAX * QUEUE_RING * DEFER_F4 * 16u : 0u; }
// MODE compiles the renderer strategy and the sampler in (MODE_GENERIC reads them from DevParams): the branches on
// prm.strategy / the sampler mode fold away, worth +2.5 % on C2 (MIS + Sobol), +1.3 % on C5 (NEE + Sobol).
// Which tree the cooperative traversals walk (both are on the device; the plain traversals of the probes and of the canonical-count
// mode walk the BVH2).  The 4-wide tree halves the dependent node round trips per ray (18 -> ~10 wave steps per closest-hit trace): +3...4.5 %
// on every kernel, and +4.7 % on the kernel specialised for textured Lambert scenes (C2's) once that kernel stopped spilling around the wider
// step (round 2: machine LICM off, see the Makefile; before that the same tree cost it 4 % and it kept the BVH2).  PT_WIDE_BVH=0 builds the
// BVH2 form for A/B runs.
#ifndef PT_WIDE_BVH
#define PT_WIDE_BVH 1
#endif
template <uint32_t FEAT> constexpr bool wide_bvh() { return PT_WIDE_BVH != 0; }
// Waves per SIMD of the clearcoat kernels.  Round 2: 3 (168 VGPRs) with the spawning sample's record parked in LDS.  Since the material sort
// moved most clearcoat shading into dedicated passes (PT_DEFER), the common iteration is the room's: the kernels whose clearcoat code is not
// ALSO carrying the texture code gain from a fourth wave (128 VGPRs, no parking: the LDS is needed for 16 waves) — scene 17 NEE (C5) +1.8 %,
// scene 19 (all-features kernel) +3.6 %, scenes 16 / 17 MIS +0 ... 0.5 %; the clearcoat + texture set loses 5.8 % (scene 15: 276 B of scratch) and stays at 3.
#ifndef PT_CC_WAVES_SEL
#define PT_CC_WAVES_SEL 1
#endif
template <uint32_t FEAT> constexpr int kernel_min_waves() {
    if ((FEAT & FEAT_CC) == 0u) return PT_MIN_WAVES;
    if (PT_CC_WAVES_SEL != 0 && FEAT != (FEAT_CC | FEAT_TEX)) return 4;
    return PT_MIN_WAVES_CC;
}
template <bool STATS, uint32_t FEAT, uint32_t MODE = MODE_GENERIC>
__global__ __launch_bounds__(64, kernel_min_waves<FEAT>()) void pt_kernel(DevScene sc, DevCamera cam, DevParams prm_in, const uint64_t* __restrict__ dim_hash_tab,
                                                float* __restrict__ accum, float* __restrict__ partial, unsigned* __restrict__ work_counter,
                                                DevStats* __restrict__ stats, PathOut pout, float4* __restrict__ defer_buf) {
    __shared__ uint32_t s_stack[STACK_DEPTH * 64];
#ifndef PT_FILM_PIX
#define PT_FILM_PIX 64
#endif
    __shared__ float s_film[PT_FILM_PIX * 3];                 // the work item's 8x8 film tile
    __shared__ uint32_t s_hi[SOBOL_HI_DIMS];
    __shared__ uint32_t s_p6[SOBOL_HI_DIMS];
    __shared__ unsigned s_work;
#ifndef PT_PARK_LDS
#define PT_PARK_LDS 1
#endif
    // The clearcoat kernels run 12 waves per CU, so each has 3.4 KB of LDS the 16-wave kernels do not: the record of the BSDF sample that
    // spawned the ray in flight (f, pdf, the vertex left: 8 dwords per lane, read only at the start of the next vertex's shading) waits there
    // during the traversals instead of in registers the allocator would spill to scratch
    constexpr bool PARK = PT_PARK_LDS != 0 && (FEAT & FEAT_CC) != 0u && kernel_min_waves<FEAT>() <= 3;
    __shared__ float s_park[PARK ? 8 * 64 : 1];
    __shared__ uint8_t s_perm[96];
#if PT_ANY_DEFERRED
    __shared__ uint32_t s_ring[ANY_RING];
    __shared__ uint32_t s_occl[2];
    __shared__ uint32_t s_pair[64];
    const AnyLds any_lds{s_ring, s_occl, s_pair};
#if PT_CLOSEST_COOP
    __shared__ unsigned long long s_best[64];
    const ClosestLds closest_lds{s_ring, s_best, s_pair};
    __shared__ uint32_t s_infl[2];
    const PairLds pair_lds{s_ring, s_best, s_occl, s_pair, s_infl};
#endif
#endif
    DevParams prm = prm_in;
    if constexpr (MODE == MODE_MIS_SOBOL) { prm.strategy = 2u; prm.sampler = 1u; }
    if constexpr (MODE == MODE_NEE_SOBOL) { prm.strategy = 1u; prm.sampler = 1u; }
    if constexpr (MODE == MODE_PT) prm.strategy = 0u;                      // either sampler
    const uint32_t lane = threadIdx.x;
    auto park = [&](Path& Q, bool mine = true) {      // `mine`: this lane's record is stored (a second shading pass only stores the lanes it shaded)
        if constexpr (PARK) {
            if (mine) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s_park[i * 64 + lane] = Q.pf[i];
                s_park[4 * 64 + lane] = Q.p_pdf; s_park[5 * 64 + lane] = Q.prev_pos.x; s_park[6 * 64 + lane] = Q.prev_pos.y; s_park[7 * 64 + lane] = Q.prev_pos.z;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) Q.pf[i] = 0.0f;
            Q.p_pdf = 0.0f; Q.prev_pos = mk3(0.0f, 0.0f, 0.0f);
        }
    };
    auto unpark = [&](Path& Q) {
        if constexpr (PARK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) Q.pf[i] = s_park[i * 64 + lane];
            Q.p_pdf = s_park[4 * 64 + lane]; Q.prev_pos = mk3(s_park[5 * 64 + lane], s_park[6 * 64 + lane], s_park[7 * 64 + lane]);
        }
    };
    uint32_t* stack = s_stack + lane;
    // murmur(dimension, seed) comes straight from its 1 KB global table (L1-resident): the LDS it used holds the tile's film
    for (uint32_t k = lane; k < 96u; k += 64u) s_perm[k] = (uint8_t)((perm_packed(k >> 2) >> (2u * (k & 3u))) & 3u);
#if PT_ZNODES_LDS
    if constexpr ((FEAT & (FEAT_TEX | FEAT_EMTEX | FEAT_ENV)) != 0u) s_znodes[lane] = sc.z_nodes[lane];   // rgb2spec_lookup's z search (pt_device.hpp)
#endif
    __syncthreads();
    SamplerCtx sctx{prm.sampler, prm.seed, prm.log2_spp, prm.n_base4_digits, cam.width, dim_hash_tab, nullptr, 0u, 0u, nullptr, s_perm};
    StatCounters st{};
    unsigned long long tp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dvc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // mi355pt_stats.divergence[4..11]
    unsigned long long t_loop0 = 0;
    if (STATS) t_loop0 = __builtin_amdgcn_s_memtime();

    for (;;) {
        if (lane == 0) s_work = atomicAdd(work_counter, 1u);
        __syncthreads();
        const uint32_t work = s_work;
        __syncthreads();
        if (work >= prm.n_work) break;
        // this lane's own pixel of the tile (film write-back) and the work item's wave-uniform sample range
        const LaneJob job = lane_job(work, lane, cam, prm);
        const LaneJob job0 = lane_job(work, 0u, cam, prm);
        const uint32_t blk_log2 = prm.block_log2, blk_mask = (1u << blk_log2) - 1u;
        const uint32_t s_prefix = prm.sample_prefix_digits;
        // (the tables hold the permuted prefix in 27 bits per entry: a launch shape whose prefix is wider hashes every digit instead)
        const uint32_t hi_first_w = sobol_hi_first(prm.log2_spp, blk_log2) - s_prefix, hi_shift_w = 2u * hi_first_w - (prm.log2_spp & 1u);
        if (prm.sampler == 1u && hi_first_w < prm.n_base4_digits && hi_first_w >= 3u &&
            2u * prm.n_base4_digits - (prm.log2_spp & 1u) <= hi_shift_w + 27u) {
            // block-uniform Sobol digit prefixes: lane d computes dimension d for this block (lane 0's pixel is the block origin).
            // Single-pixel items over an aligned 4^m block of sample indices: the sample digits above m are part of the prefix.
            sctx.hi_first = sobol_hi_first(prm.log2_spp, blk_log2) - s_prefix;
            sctx.hi_shift = 2u * sctx.hi_first - (prm.log2_spp & 1u);
            const uint32_t tile_m = (encode_morton2_u32(job0.px, job0.py) << prm.log2_spp) | (s_prefix ? job0.s_cur : 0u);
            for (uint32_t dmn = lane; dmn < (uint32_t)SOBOL_HI_DIMS; dmn += 64) {
                uint32_t e = (uint32_t)(sobol_tile_hi_digits(tile_m, dmn, prm.log2_spp, prm.n_base4_digits, sctx.hi_first) >> sctx.hi_shift);   // <= 26 bits: the Morton index is a u32 and hi_shift >= 6
                const uint64_t prefix = (uint64_t)tile_m >> sctx.hi_shift;                 // the digits above digit hi_first-1
                e |= sobol_perm_index(prefix, dmn) << 27;
                uint32_t e6 = 0;
                for (uint32_t v7 = 0; v7 < 4u; ++v7) e6 |= sobol_perm_index((prefix << 2) | v7, dmn) << (5u * v7);
                s_hi[dmn] = e; s_p6[dmn] = e6;
            }
            sctx.hi_lds = s_hi; sctx.p6_lds = s_p6;
        }
        // The work item's paths form a pool of (pixel, sample) pairs, sample-major.  A lane whose path ended takes the next pair,
        // whichever pixel of the tile it belongs to: no lane idles while another still has samples of "its" pixel to do.  The
        // tile's film lives in LDS (ds_add_f32); the hand-out order is a function of the wave's own deterministic schedule.
        if (lane < PT_FILM_PIX) { s_film[3 * lane] = 0.0f; s_film[3 * lane + 1] = 0.0f; s_film[3 * lane + 2] = 0.0f; }
        __syncthreads();
        const uint32_t n_s = job0.s_end > job0.s_cur ? job0.s_end - job0.s_cur : 0u;
        const uint32_t pool_size = n_s << (2u * blk_log2);
        uint32_t pool_next = 0u;                                   // wave-uniform
        uint32_t my_pix = lane;
        Path P{};
        park(P);
        bool active = false;
        constexpr bool MERGED = merged_traversal<FEAT, MODE>();
        constexpr int CARRY = MERGED ? carry_max<FEAT, MODE>() : 0;
        ShadowReq sh{};                                            // merged form: the light connection of the vertex just shaded, traced together with the NEXT closest-hit ray
        // merged form: a path that ended with a light connection pending stays for one more iteration (`dying`) in which only the connection
        // is traced, instead of a separate any-hit traversal at the end of the iteration (rare; keeps one traversal instance in the kernel);
        // `susp` (carry-over): this lane's closest-hit ray is still in flight, the lane sits out the shading stage
        bool dying = false, susp = false;
        CarryState carry{0ull};
        constexpr bool TAILQ = !STATS && tail_queue<FEAT, MODE>();       // (the instrumented kernels keep the plain shading stage)
        constexpr uint32_t DEFER = TAILQ ? 0u : defer_classes<FEAT, MODE>();
        float4* const q_base = (DEFER != 0u || TAILQ) ? defer_buf + (size_t)blockIdx.x * (QUEUE_MAX * QUEUE_RING * DEFER_F4) : nullptr;     // this wave's queue(s)
        uint32_t q_head = 0u, q_tail = 0u;                         // wave-uniform; the queue is empty between work items
        // tail queue: a second queue for the paths whose hit is on a sort class of its own (defer_classes), so that a pass shades one class
        // (the clearcoat material in the kernels that have it: its branch is the long one.  A second queue for every class but plain Lambert
        // was measured on the kernels without clearcoat and LOSES — scene 3 2 312 -> 2 063, scene 8 2 097 -> 1 932: a class that is a tenth of
        // the hits fills its queue every ~16 iterations and leaves up to 63 paths to be bounced out in sparse passes when a work item ends)
        constexpr uint32_t TQ_CLASSES = (TAILQ && (FEAT & FEAT_CC) != 0u) ? ((1u << MT_CLEARCOAT) | (1u << (MT_CLEARCOAT | 8u))) : 0u;
        uint32_t q2_head = 0u, q2_tail = 0u;
        // entries per ring: with one queue at most 127 paths ever wait (a pass starts at 64, an iteration adds at most 64 and every lane is
        // free after the front), with two at most 191 in both together — the smaller ring keeps the records closer to the L2
#ifndef PT_TAILQ_RING1
#define PT_TAILQ_RING1 128u
#endif
        constexpr uint32_t QR = (TQ_CLASSES != 0u) ? QUEUE_RING : PT_TAILQ_RING1;
        while (true) {
            unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, tsa = 0, tsb = 0;
            uint32_t bsdf_classes = 0u;
            if (STATS) ts0 = __builtin_amdgcn_s_memtime();
            // deferral queue: when enough paths wait (or nothing new is left to start), this iteration's free lanes take them
            bool popped = false, drain = false;
            uint32_t pop_e = 0u;
            if constexpr (DEFER != 0u) {
                const uint32_t q_count = q_tail - q_head;
                drain = PT_DEFER == 1 && !(STATS && prm.stats_mode == 1u) && (q_count >= (uint32_t)PT_DEFER_MIN || (pool_next >= pool_size && q_count != 0u));
                if (drain) {
                    const unsigned long long m_free = __ballot(!active);
                    const uint32_t r = rank_below(m_free);
                    if (!active && r < q_count) { popped = true; active = true; pop_e = (q_head + r) & (DEFER_RING - 1u); }   // its record is read after the traversal
                    q_head += min((uint32_t)__popcll(m_free), q_count);
                }
            }
            const unsigned long long m_needy = __ballot(!active);
            if (m_needy != 0ull && pool_next < pool_size) {
                const uint32_t idx = pool_next + rank_below(m_needy);
                if (!active && idx < pool_size) {
                    const uint32_t pix = idx & ((1u << (2u * blk_log2)) - 1u);
                    const uint32_t px = job0.px + (pix & blk_mask), py = job0.py + (pix >> blk_log2);
                    const uint32_t smp_i = job0.s_cur + (idx >> (2u * blk_log2));
                    const bool valid = px < cam.width && py < cam.height;
                    if (valid) { active = true; my_pix = pix; regen_path<STATS>(P, sctx, cam, px, py, smp_i, st); }
                }
                pool_next = min(pool_next + (uint32_t)__popcll(m_needy), pool_size);
            }
            if (!__any(active)) {
                if (pool_next >= pool_size && q_tail == q_head && q2_tail == q2_head) break;
                // (PT_DEFER 2 takes queued paths AFTER the shading stage: with nothing left to start, an iteration without rays still has to get there)
                if (!(((DEFER != 0u && PT_DEFER == 2) || TAILQ) && pool_next >= pool_size)) continue;
            }
            if (STATS) { ts1 = __builtin_amdgcn_s_memtime(); if (lane == 0) st.w[4]++; if (active) st.w[5]++; }
            Hit hit{};
            bool got = false;
            const bool canonical = STATS && prm.stats_mode == 1u;                    // plain per-lane traversals in the reference's order (step counts)
            if (canonical) { if (active) got = trace_closest<STATS>(sc, P.ro, P.rd, 3.402823466e+38f, stack, hit, st); }
            else if constexpr (MERGED) {
                // ONE traversal for this iteration's closest-hit rays and the light connections the previous shading left pending
                bool occluded = false;
                if (STATS && sh.on) st.w[6]++;
                PT_PRIO_TRAV_ENTER;
                trace_pair_coop<STATS, wide_bvh<FEAT>(), CARRY>(sc, P.ro, P.rd, active && !dying && !popped, sh.o, sh.d, sh.t, sh.on, stack, lane, pair_lds, hit, got, occluded, st, &carry, susp, &susp);
                PT_PRIO_TRAV_EXIT;
                if (sh.on && !occluded) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) P.L[i] = P.L[i] + sh.c[i];
                }
                sh = ShadowReq{};      // consumed: every field dead from here on, for every lane — none of them is carried through the shading stage (+2 % on scenes 0 / 8)
            }
            else { PT_PRIO_TRAV_ENTER; got = trace_closest_coop<STATS, wide_bvh<FEAT>()>(sc, P.ro, P.rd, active, stack, lane, closest_lds, hit, st); PT_PRIO_TRAV_EXIT; }
            if (STATS) {
                // material divergence of the shading stage (mi355pt_stats.divergence): classes among the lanes that shade a surface
                const uint32_t mclass = (active && got) ? sc.materials[__float_as_uint(((const float4*)(sc.shade + hit.tri))[4].z)].type : 8u;
                uint32_t classes = 0u, largest = 0u, lanes = 0u;
                for (uint32_t c = 0u; c < 8u; ++c) {
                    const uint32_t n = (uint32_t)__popcll(__ballot(mclass == c));
                    classes += n ? 1u : 0u; largest = max(largest, n); lanes += n;
                }
                if (lane == 0 && lanes) { st.dv[0]++; st.dv[1] += classes; st.dv[2] += lanes; st.dv[3] += largest; }
                bsdf_classes = classes - (__ballot(mclass == MT_EMISSIVE) != 0ull ? 1u : 0u);
                ts2 = __builtin_amdgcn_s_memtime();
            }
            auto q_store = [&](uint32_t e, const Path& Q, const Hit& h, uint32_t pix) {
                float4* r = q_base + (size_t)e * DEFER_F4;
                const uint32_t fl = (Q.wl.term ? 1u : 0u) | (Q.from_camera ? 2u : 0u) | (Q.prev_spec ? 4u : 0u) | ((Q.depth & 255u) << 8) | (pix << 16);
                r[0] = make_float4(__uint_as_float(Q.smp.morton), __uint_as_float(Q.smp.dimension), __uint_as_float(Q.smp.rkey_lo), __uint_as_float(Q.smp.rkey_hi));
                r[1] = make_float4(Q.wl.lam0, __uint_as_float(fl), Q.T[0], Q.T[1]);
                r[2] = make_float4(Q.T[2], Q.T[3], Q.L[0], Q.L[1]);
                r[3] = make_float4(Q.L[2], Q.L[3], Q.rd.x, Q.rd.y);
                r[4] = make_float4(Q.rd.z, Q.pf[0], Q.pf[1], Q.pf[2]);
                r[5] = make_float4(Q.pf[3], Q.p_pdf, Q.prev_pos.x, Q.prev_pos.y);
                r[6] = make_float4(Q.prev_pos.z, h.t, h.b0, h.b1);
                r[7] = make_float4(h.b2, __uint_as_float(h.tri), __uint_as_float(h.mclass), 0.0f);
            };
            auto q_load = [&](uint32_t e, Path& Q, Hit& h, uint32_t& pix) {
                const float4* r = q_base + (size_t)e * DEFER_F4;
                const float4 a = r[0], b = r[1], c = r[2], d = r[3], e4 = r[4], f = r[5], g = r[6], hh = r[7];
                const uint32_t fl = __float_as_uint(b.y);
                Q.smp.morton = __float_as_uint(a.x); Q.smp.dimension = __float_as_uint(a.y); Q.smp.rkey_lo = __float_as_uint(a.z); Q.smp.rkey_hi = __float_as_uint(a.w);
                Q.wl.lam0 = b.x; Q.wl.term = (fl & 1u) != 0u; Q.from_camera = (fl & 2u) != 0u; Q.prev_spec = (fl & 4u) != 0u; Q.depth = (fl >> 8) & 255u; pix = fl >> 16;
                Q.T[0] = b.z; Q.T[1] = b.w; Q.T[2] = c.x; Q.T[3] = c.y; Q.L[0] = c.z; Q.L[1] = c.w; Q.L[2] = d.x; Q.L[3] = d.y;
                Q.rd = mk3(d.z, d.w, e4.x); Q.ro = mk3(0.0f, 0.0f, 0.0f);
                Q.pf[0] = e4.y; Q.pf[1] = e4.z; Q.pf[2] = e4.w; Q.pf[3] = f.x; Q.p_pdf = f.y; Q.prev_pos = mk3(f.z, f.w, g.x);
                h.t = g.y; h.b0 = g.z; h.b1 = g.w; h.b2 = hh.x; h.tri = __float_as_uint(hh.y); h.mclass = __float_as_uint(hh.z);
            };
            // the tail queue's record: what the back of the vertex still needs once the front has run — the spawning sample's f, pdf and the
            // vertex left are consumed by the front, from_camera / prev_spec are rewritten by the tail, the hit's t is never read: 20 dwords in
            // 5 float4 = 80 B.  The record's size is the queue's price: 128 -> 96 B was worth +6.5 % on C2 (the queues stream through L2 / HBM)
            // Layout: FIELD-major inside a ring (float4 k of entry e at ring base + k * ring + e): the entries of one push / pop are
            // consecutive, so each of the five store / load instructions of a wave covers 64 x 16 B = eight whole 128-byte lines instead of a
            // sixth of 64 different ones (+1 ... 2 % over the record-major layout; a 128-entry ring where one queue suffices +0.2 ... 0.8 %)
            auto tq_store = [&](uint32_t e, const Path& Q, const Hit& h, uint32_t pix) {
                float4* r = q_base + (size_t)(e >> 8) * (TQ_F4 * QUEUE_RING) + (e & (QR - 1u));
                const uint32_t fl = (Q.wl.term ? 1u : 0u) | ((Q.depth & 1023u) << 1) | ((pix & 63u) << 11) | (Q.smp.dimension << 17);   // (max_depth <= 1000: api.cpp check_args)
                PT_TQ_ST(r + 0u * QR, make_float4(__uint_as_float(Q.smp.morton), __uint_as_float(fl), Q.wl.lam0, __uint_as_float(h.tri)));
                PT_TQ_ST(r + 1u * QR, make_float4(Q.T[0], Q.T[1], Q.T[2], Q.T[3]));
                PT_TQ_ST(r + 2u * QR, make_float4(Q.L[0], Q.L[1], Q.L[2], Q.L[3]));
                PT_TQ_ST(r + 3u * QR, make_float4(Q.rd.x, Q.rd.y, Q.rd.z, h.b0));
                PT_TQ_ST(r + 4u * QR, make_float4(h.b1, h.b2, __uint_as_float(Q.smp.rkey_lo), __uint_as_float(Q.smp.rkey_hi)));
            };
            auto tq_load = [&](uint32_t e, Path& Q, Hit& h, uint32_t& pix) {
                const float4* r = q_base + (size_t)(e >> 8) * (TQ_F4 * QUEUE_RING) + (e & (QR - 1u));
                const float4 a = PT_TQ_LD(r + 0u * QR), b = PT_TQ_LD(r + 1u * QR), c = PT_TQ_LD(r + 2u * QR), d = PT_TQ_LD(r + 3u * QR), e4 = PT_TQ_LD(r + 4u * QR);
                const uint32_t fl = __float_as_uint(a.y);
                Q.smp.morton = __float_as_uint(a.x); Q.smp.dimension = fl >> 17; Q.smp.rkey_lo = __float_as_uint(e4.z); Q.smp.rkey_hi = __float_as_uint(e4.w);
                Q.wl.lam0 = a.z; Q.wl.term = (fl & 1u) != 0u; Q.depth = (fl >> 1) & 1023u; pix = (fl >> 11) & 63u;
                Q.from_camera = false; Q.prev_spec = false;
                Q.T[0] = b.x; Q.T[1] = b.y; Q.T[2] = b.z; Q.T[3] = b.w; Q.L[0] = c.x; Q.L[1] = c.y; Q.L[2] = c.z; Q.L[3] = c.w;
                Q.rd = mk3(d.x, d.y, d.z); Q.ro = mk3(0.0f, 0.0f, 0.0f);
                h.b0 = d.w; h.b1 = e4.x; h.b2 = e4.y; h.tri = __float_as_uint(a.w); h.mclass = 0u; h.t = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) Q.pf[i] = 0.0f;
                Q.p_pdf = 0.0f; Q.prev_pos = mk3(0.0f, 0.0f, 0.0f);
            };
            auto finish_path = [&]() {            // Sensor::add_sample of a finished path into the work item's LDS film tile (+ the per-sample log)
                if (pout.L != nullptr) {
                    const uint32_t px = job0.px + (my_pix & blk_mask), py = job0.py + (my_pix >> blk_log2);
                    const uint32_t tile_k = (work / prm.chunks) >> (6u - 2u * blk_log2);
                    const uint32_t smp_i = P.smp.morton & ((1u << prm.log2_spp) - 1u);
                    sample_log(P, pout, ((size_t)tile_k * 64u + ((py & 7u) * 8u + (px & 7u))) * pout.n_s + (smp_i - pout.s_base));
                }
                float r, g, b;
                film_rgb(P, sc, prm, r, g, b);
                atomicAdd(&s_film[3 * my_pix], r); atomicAdd(&s_film[3 * my_pix + 1], g); atomicAdd(&s_film[3 * my_pix + 2], b);
            };
            if constexpr (TAILQ) {
                // front of the vertex for every lane that traced: does the path go on?
                bool end_path = dying;                         // (a dying path's last connection has just been resolved)
                const bool front = active && !dying && !susp;
                unpark(P);
                {
                    ShadeCtx C0;
                    C0.cont = false;
                    if (front) end_path = shade_vertex_head<STATS, FEAT, 1>(P, sc, prm, sctx, got, hit, sh, st, tsa, C0);
                }
                dying = false;
                // the paths that go on wait in the queue of their sort class; their lanes are free
                const bool go_on = front && !end_path;
                const bool cls2 = TQ_CLASSES != 0u && ((TQ_CLASSES >> (hit.mclass & 31u)) & 1u) != 0u;
                const unsigned long long m_on1 = __ballot(go_on && !cls2), m_on2 = TQ_CLASSES != 0u ? __ballot(go_on && cls2) : 0ull;
                if ((m_on1 | m_on2) != 0ull) {
                    if (go_on) {
                        const uint32_t e = cls2 ? QUEUE_RING + ((q2_tail + rank_below(m_on2)) & (QR - 1u)) : ((q_tail + rank_below(m_on1)) & (QR - 1u));
                        tq_store(e, P, hit, my_pix);
                        active = false;
                    }
                    q_tail += (uint32_t)__popcll(m_on1); q2_tail += (uint32_t)__popcll(m_on2);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                }
                if (active && end_path) { finish_path(); active = false; }
                park(P, false);                                 // (nothing of these lanes' records is needed any more: the queue has them)
                if (STATS) ts3 = ts4 = __builtin_amdgcn_s_memtime();
                // the back of the vertex and its tail for a full wave of queued paths — of ONE class while new paths still arrive (the class
                // with its own queue first); when the work item has nothing new left, whatever waits in either queue shares the passes
                const bool draining = pool_next >= pool_size;
                const uint32_t c2 = q2_tail - q2_head, c1 = q_tail - q_head;
                const bool full2 = TQ_CLASSES != 0u && c2 >= (uint32_t)PT_TAILQ_MIN, full1 = c1 >= (uint32_t)PT_TAILQ_MIN;
                if (full2 || full1 || (draining && (c1 | c2) != 0u)) {
                    const unsigned long long m_free = __ballot(!active);
                    const uint32_t n_free = (uint32_t)__popcll(m_free), r = rank_below(m_free);
                    // lanes 0 .. n2-1 of the free lanes take from queue 2, the next n1 from queue 1
                    const uint32_t n2 = (full2 || (draining && !full1)) ? min(n_free, c2) : 0u;
                    const uint32_t n1 = (!full2 || draining) ? min(n_free - n2, c1) : 0u;
                    const bool take2 = !active && r < n2, take1 = !active && !take2 && r - n2 < n1;
                    const bool take = take1 || take2;
                    const uint32_t e = take2 ? QUEUE_RING + ((q2_head + r) & (QR - 1u)) : ((q_head + (r - n2)) & (QR - 1u));
                    q2_head += n2; q_head += n1;
                    bool ep = false;
                    if constexpr ((FEAT & FEAT_CC) != 0u) {
                        ShadeCtx C;
                        C.cont = false; C.need_cc = false; C.cc_fc = 0.0f; C.cc_alpha_c = 0.0f; C.cc_r0c = 0.0f; C.wo_nm = mk3(0, 0, 1); C.mc_key = 0ull;
                        if (take) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            tq_load(e, P, hit, my_pix);
                            active = true;
                            shade_vertex_head<STATS, FEAT, 2>(P, sc, prm, sctx, true, hit, sh, st, tsa, C);
                        }
                        // the coat's 64-sample directional albedo, estimated by the whole wave for the lanes that need it
                        const bool want_mc = take && C.cont && C.need_cc;
                        const float fc_mc = coat_directional_albedo_coop(want_mc, C.cc_alpha_c, C.cc_r0c, C.wo_nm, C.mc_key, lane);
                        if (want_mc) C.cc_fc = fc_mc;
                        if (take && C.cont) ep = shade_vertex_tail<STATS, FEAT>(P, sc, prm, sctx, sh, st, tsb, C);
                    } else if (take) {
                        // (ONE divergent region for the back of the head and the tail: ShadeCtx must not cross a re-convergence point in the
                        // kernels that run 4 waves per SIMD — the split form of this block cost them 12 %)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        tq_load(e, P, hit, my_pix);
                        active = true;
                        ShadeCtx C;
                        C.cont = false; C.need_cc = false; C.cc_fc = 0.0f;
                        shade_vertex_head<STATS, FEAT, 2>(P, sc, prm, sctx, true, hit, sh, st, tsa, C);
                        if (C.cont) ep = shade_vertex_tail<STATS, FEAT>(P, sc, prm, sctx, sh, st, tsb, C);
                    }
                    park(P, take);
                    if (sh.on && sh.c[0] == 0.0f && sh.c[1] == 0.0f && sh.c[2] == 0.0f && sh.c[3] == 0.0f) sh.on = false;
                    if (!active) sh.on = false;
                    if (take) { dying = sh.on && ep; if (dying) ep = false; }
                    if (take && ep) { finish_path(); active = false; }
                }
            } else
            // The shading stage.  PT_DEFER 2 runs it a second time in the iterations that shade the deferral queue: the lanes the first pass
            // freed (ended paths, deferred hits) take queued paths and shade them at once, so a queued path rejoins the NEXT traversal
            // with its next ray like everybody else (PT_DEFER 1 pops at the top of the iteration and lets those lanes sit out a traversal).
            for (int pass = 0;; ++pass) {
            const bool mine = pass == 0 || popped;     // the lanes this pass works on
            bool end_path = pass == 0 && MERGED && dying;           // a dying path's last connection has just been resolved
            if (pass == 0) {
                if constexpr (!MERGED) sh = ShadowReq{};
                unpark(P);
            }
            if constexpr (DEFER != 0u) {
                // queued paths join here: path state, hit and pixel of the lanes that popped (PT_DEFER 1: registers dead during the traversal)
                if (popped) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const float4* r = q_base + (size_t)pop_e * DEFER_F4;
                    const float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5], g = r[6], h = r[7];
                    const uint32_t fl = __float_as_uint(b.y);
                    P.smp.morton = __float_as_uint(a.x); P.smp.dimension = __float_as_uint(a.y); P.smp.rkey_lo = __float_as_uint(a.z); P.smp.rkey_hi = __float_as_uint(a.w);
                    P.wl.lam0 = b.x; P.wl.term = (fl & 1u) != 0u; P.from_camera = (fl & 2u) != 0u; P.prev_spec = (fl & 4u) != 0u; P.depth = (fl >> 8) & 255u; my_pix = fl >> 16;
                    P.T[0] = b.z; P.T[1] = b.w; P.T[2] = c.x; P.T[3] = c.y; P.L[0] = c.z; P.L[1] = c.w; P.L[2] = d.x; P.L[3] = d.y;
                    P.rd = mk3(d.z, d.w, e.x); P.ro = mk3(0.0f, 0.0f, 0.0f);
                    P.pf[0] = e.y; P.pf[1] = e.z; P.pf[2] = e.w; P.pf[3] = f.x; P.p_pdf = f.y; P.prev_pos = mk3(f.z, f.w, g.x);
                    hit.t = g.y; hit.b0 = g.z; hit.b1 = g.w; hit.b2 = h.x; hit.tri = __float_as_uint(h.y); hit.mclass = __float_as_uint(h.z);
                    got = true;
                }
                // and the lanes whose hit is on a deferred material leave (unless this is the iteration that shades the queue)
                const bool defer_now = pass == 0 && !drain && !(STATS && prm.stats_mode == 1u) && active && !dying && !susp && !popped && got && ((DEFER >> (hit.mclass & 31u)) & 1u) != 0u;
                const unsigned long long m_def = __ballot(defer_now);
                if (m_def != 0ull) {
                    if (defer_now) {
                        float4* r = q_base + (size_t)((q_tail + rank_below(m_def)) & (DEFER_RING - 1u)) * DEFER_F4;
                        const uint32_t fl = (P.wl.term ? 1u : 0u) | (P.from_camera ? 2u : 0u) | (P.prev_spec ? 4u : 0u) | ((P.depth & 255u) << 8) | (my_pix << 16);
                        r[0] = make_float4(__uint_as_float(P.smp.morton), __uint_as_float(P.smp.dimension), __uint_as_float(P.smp.rkey_lo), __uint_as_float(P.smp.rkey_hi));
                        r[1] = make_float4(P.wl.lam0, __uint_as_float(fl), P.T[0], P.T[1]);
                        r[2] = make_float4(P.T[2], P.T[3], P.L[0], P.L[1]);
                        r[3] = make_float4(P.L[2], P.L[3], P.rd.x, P.rd.y);
                        r[4] = make_float4(P.rd.z, P.pf[0], P.pf[1], P.pf[2]);
                        r[5] = make_float4(P.pf[3], P.p_pdf, P.prev_pos.x, P.prev_pos.y);
                        r[6] = make_float4(P.prev_pos.z, hit.t, hit.b0, hit.b1);
                        r[7] = make_float4(hit.b2, __uint_as_float(hit.tri), __uint_as_float(hit.mclass), 0.0f);
                        active = false;                                      // free: a new path (or a queued one) next
                    }
                    q_tail += (uint32_t)__popcll(m_def);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                }
            }
            const bool shade_now = mine && active && !(MERGED && (dying || susp));
            if constexpr ((FEAT & FEAT_CC) != 0u) {
                ShadeCtx C;
                C.cont = false; C.need_cc = false; C.cc_fc = 0.0f; C.cc_alpha_c = 0.0f; C.cc_r0c = 0.0f; C.wo_nm = mk3(0, 0, 1); C.mc_key = 0ull;
                if (shade_now) end_path = shade_vertex_head<STATS, FEAT>(P, sc, prm, sctx, got, hit, sh, st, tsa, C);
                // the coat's 64-sample directional albedo, estimated by the whole wave for the lanes that need it
                const bool want_mc = shade_now && C.cont && C.need_cc;
                const float fc_mc = coat_directional_albedo_coop(want_mc, C.cc_alpha_c, C.cc_r0c, C.wo_nm, C.mc_key, lane);
                if (want_mc) C.cc_fc = fc_mc;
                if (shade_now && C.cont) end_path = shade_vertex_tail<STATS, FEAT>(P, sc, prm, sctx, sh, st, tsb, C);
            } else {
                if (shade_now) end_path = shade_vertex<STATS, FEAT>(P, sc, prm, sctx, got, hit, sh, st, tsa, tsb);
            }
            park(P, mine);
            if (STATS) {
                ts3 = __builtin_amdgcn_s_memtime();
                // the stamps inside shade_vertex are taken by the lanes that reach them: make them wave-level (first lane that has one)
                unsigned long long ma = __ballot(tsa != 0ull), mb = __ballot(tsb != 0ull);
                if (ma) { int l = (int)__ffsll((long long)ma) - 1; tsa = ((unsigned long long)__shfl((uint32_t)(tsa >> 32), l) << 32) | __shfl((uint32_t)tsa, l); }
                if (mb) { int l = (int)__ffsll((long long)mb) - 1; tsb = ((unsigned long long)__shfl((uint32_t)(tsb >> 32), l) << 32) | __shfl((uint32_t)tsb, l); }
            }
            // a light connection whose contribution is exactly zero (light behind the surface, f == 0) cannot change L whatever the
            // visibility test says: the production path does not trace it (the canonical-count mode does, like the reference)
            if (!canonical && sh.on && sh.c[0] == 0.0f && sh.c[1] == 0.0f && sh.c[2] == 0.0f && sh.c[3] == 0.0f) sh.on = false;
            if (!active) sh.on = false;
            if (MERGED && !canonical) {
                // merged form: the connection of a CONTINUING path waits for the next iteration's traversal; a path that ends here with a
                // connection pending (a failed BSDF sample after the light was sampled: rare) lives on for that traversal alone
                if (mine) {
                    dying = sh.on && end_path;
                    if (dying) end_path = false;
                }
            } else {
                // two traversals per iteration — and everything in the canonical-count mode: the connection is traced now
                const bool now = sh.on;
                if (__any(now)) {
                    if (STATS && now) st.w[6]++;
                    bool occluded = false;
                    if (canonical) { if (now) occluded = trace_any<STATS>(sc, sh.o, sh.d, sh.t, stack, st); }
                    else { PT_PRIO_TRAV_ENTER; occluded = trace_any_deferred<STATS, wide_bvh<FEAT>()>(sc, sh.o, sh.d, sh.t, now, stack, lane, any_lds, st); PT_PRIO_TRAV_EXIT; }
                    if (now && !occluded) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) P.L[i] = P.L[i] + sh.c[i];
                    }
                    if (now) sh.on = false;
                }
            }
            if (STATS) ts4 = __builtin_amdgcn_s_memtime();
            if (mine && active && end_path) {
                if (pout.L != nullptr) {
                    // per-sample log (see PathOut): the sample index is the low log2(spp) bits of the lane's Morton index (spp a power of two)
                    const uint32_t px = job0.px + (my_pix & blk_mask), py = job0.py + (my_pix >> blk_log2);
                    const uint32_t tile_k = (work / prm.chunks) >> (6u - 2u * blk_log2);
                    const uint32_t smp_i = P.smp.morton & ((1u << prm.log2_spp) - 1u);
                    sample_log(P, pout, ((size_t)tile_k * 64u + ((py & 7u) * 8u + (px & 7u))) * pout.n_s + (smp_i - pout.s_base));
                }
                float r, g, b;
                film_rgb(P, sc, prm, r, g, b);
                atomicAdd(&s_film[3 * my_pix], r); atomicAdd(&s_film[3 * my_pix + 1], g); atomicAdd(&s_film[3 * my_pix + 2], b);
                active = false;
            }
            // a second pass for the deferral queue?
            if constexpr (DEFER != 0u && PT_DEFER == 2) {
                if (pass != 0 || (STATS && prm.stats_mode == 1u)) break;
                const uint32_t q_count = q_tail - q_head;
                if (!(q_count >= (uint32_t)PT_DEFER_MIN || (pool_next >= pool_size && q_count != 0u))) break;
                const unsigned long long m_free = __ballot(!active);
                if (m_free == 0ull) break;
                const uint32_t r = rank_below(m_free);
                popped = !active && r < q_count;
                if (popped) { active = true; dying = false; pop_e = (q_head + r) & (DEFER_RING - 1u); }
                q_head += min((uint32_t)__popcll(m_free), q_count);
            } else break;
            }
            if (STATS) {
                unsigned long long ts5 = __builtin_amdgcn_s_memtime();
                dvc[min(bsdf_classes, 3u)] += 1ull; dvc[4 + min(bsdf_classes, 3u)] += ts3 - ts2;
                tp[0] += ts1 - ts0; tp[1] += ts2 - ts1; tp[2] += ts3 - ts2; tp[3] += ts4 - ts3; tp[4] += ts5 - ts4;
                if (tsa) { tp[6] += tsa - ts2; if (tsb) { tp[7] += tsb - tsa; tp[8] += ts3 - tsb; } else tp[7] += ts3 - tsa; } else tp[6] += ts3 - ts2;
            }
        }
        __syncthreads();
        if (job.valid) {
            size_t o = ((size_t)job.py * cam.width + job.px) * 3;
            const float fr = s_film[3 * lane], fg = s_film[3 * lane + 1], fb = s_film[3 * lane + 2];
            if (prm.chunks == 1) { accum[o] += fr; accum[o + 1] += fg; accum[o + 2] += fb; }
            else {
                // the sample range of this tile is split over several work items: each writes its own slot, combine_kernel adds the
                // slots to the film in chunk order (no float atomics: frames stay bit-identical from run to run)
                // (slots are laid out per 8x8 tile and chunk whatever the block size, see combine_kernel)
                const uint32_t tile_k = (work / prm.chunks) >> (6u - 2u * blk_log2), chunk = work % prm.chunks;
                float* slot = partial + (((size_t)tile_k * prm.chunks + chunk) * 64u + ((job.py & 7u) * 8u + (job.px & 7u))) * 3u;
                slot[0] = fr; slot[1] = fg; slot[2] = fb;
            }
        }
        __syncthreads();
    }
    if (STATS && lane == 0) {
        tp[5] = __builtin_amdgcn_s_memtime() - t_loop0;
        for (int i = 0; i < 10; ++i) atomicAdd(&stats->phase_cycles[i], tp[i]);
        for (int i = 0; i < 8; ++i) atomicAdd(&stats->divergence[4 + i], dvc[i]);
    }
    if (STATS) flush_stats(stats, st);
}

// ---- launch of the production variants of one MODE: smallest compiled feature set covering `feat` ----
inline uint32_t pick_features(uint32_t feat) {
    const uint32_t sets[] = {0u, FEAT_TEX, FEAT_DIEL, FEAT_METAL, FEAT_DIEL | FEAT_ROUGH, FEAT_DELTA | FEAT_MLIGHT, FEAT_CC, FEAT_CC | FEAT_TEX, FEAT_STD & ~FEAT_CC, FEAT_STD, FEAT_ALL};
    for (uint32_t s : sets) if ((feat & ~s) == 0u) return s;
    return FEAT_ALL;
}
struct PtLaunchArgs {
    DevScene sc; DevCamera cam; DevParams prm; const uint64_t* d_hash; float* d_accum; float* d_partial; unsigned* d_counter; DevStats* d_stats;
    int grid; hipStream_t stream; PathOut pout; float4* d_defer;
};
// The feature sets in two classes, compiled in separate translation units with their own backend options (Makefile): the sets without the
// clearcoat code run at 4 waves per SIMD and gain from sinking / the AMDGPU pressure trackers, the clearcoat sets (3 waves per SIMD) lose.
#define PT_FOR_EACH_PLAIN_SET(X) X(0u) X(FEAT_TEX) X(FEAT_DIEL) X(FEAT_METAL) X(FEAT_DIEL | FEAT_ROUGH) X(FEAT_DELTA | FEAT_MLIGHT) X(FEAT_STD & ~FEAT_CC)
#define PT_FOR_EACH_CC_SET(X) X(FEAT_CC) X(FEAT_CC | FEAT_TEX) X(FEAT_STD) X(FEAT_ALL)
#define PT_FOR_EACH_FEATURE_SET(X) PT_FOR_EACH_PLAIN_SET(X) PT_FOR_EACH_CC_SET(X)
#define PT_CASE(F) case (F): hipLaunchKernelGGL((pt_kernel<false, (F), MODE>), dim3(a.grid), dim3(64), 0, a.stream, a.sc, a.cam, a.prm, a.d_hash, a.d_accum, a.d_partial, a.d_counter, a.d_stats, a.pout, a.d_defer); break;
template <uint32_t MODE>
void launch_pt_plain(const PtLaunchArgs& a, uint32_t feat) {
    switch (pick_features(feat)) { PT_FOR_EACH_PLAIN_SET(PT_CASE) default: break; }
}
template <uint32_t MODE>
void launch_pt_cc(const PtLaunchArgs& a, uint32_t feat) {
    switch (pick_features(feat)) { PT_FOR_EACH_CC_SET(PT_CASE) default: break; }
}
#undef PT_CASE
template <uint32_t MODE>
void launch_pt_mode(const PtLaunchArgs& a, uint32_t feat) {
    if (pick_features(feat) & FEAT_CC) launch_pt_cc<MODE>(a, feat); else launch_pt_plain<MODE>(a, feat);
}
// resident waves per CU of the EXACT instantiation a launch takes (launch bounds: 4 waves/SIMD, clearcoat variants 3; the register count and
// so the occupancy can differ between the MODE specialisations, which live in different translation units with their own backend flags):
// the persistent grid size.  Each translation unit answers for the kernels it holds.
#define PT_OCC_CASE(F) case (F): e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<false, (F), MODE>, 64, 0); break;
template <uint32_t MODE>
int occupancy_pt_plain(uint32_t feat) {
    int per_cu = 0;
    hipError_t e = hipErrorUnknown;
    switch (pick_features(feat)) { PT_FOR_EACH_PLAIN_SET(PT_OCC_CASE) default: break; }
    return (e == hipSuccess && per_cu > 0) ? per_cu : 8;
}
template <uint32_t MODE>
int occupancy_pt_cc(uint32_t feat) {
    int per_cu = 0;
    hipError_t e = hipErrorUnknown;
    switch (pick_features(feat)) { PT_FOR_EACH_CC_SET(PT_OCC_CASE) default: break; }
    return (e == hipSuccess && per_cu > 0) ? per_cu : 8;
}
#undef PT_OCC_CASE
template <uint32_t MODE>
int occupancy_pt_mode(uint32_t feat) { return (pick_features(feat) & FEAT_CC) ? occupancy_pt_cc<MODE>(feat) : occupancy_pt_plain<MODE>(feat); }
void launch_pt_mis_sobol(const PtLaunchArgs& a, uint32_t feat);      // pt_kernels_mis.hip (plain sets; forwards the clearcoat sets)
void launch_pt_mis_sobol_cc(const PtLaunchArgs& a, uint32_t feat);   // pt_kernels_mis_cc.hip
void launch_pt_nee_sobol(const PtLaunchArgs& a, uint32_t feat);      // pt_kernels_nee.hip
void launch_pt_nee_sobol_cc(const PtLaunchArgs& a, uint32_t feat);   // pt_kernels_nee_cc.hip
void launch_pt_strategy_pt(const PtLaunchArgs& a, uint32_t feat);    // pt_kernels_pt.hip (the plain path tracer, either sampler)
int occupancy_pt_mis_sobol(uint32_t feat);
int occupancy_pt_mis_sobol_cc(uint32_t feat);
int occupancy_pt_nee_sobol(uint32_t feat);
int occupancy_pt_nee_sobol_cc(uint32_t feat);
int occupancy_pt_strategy_pt(uint32_t feat);

}  // namespace pt
