// The sample loop of BaseSrgbRenderer::render (renderer/src/renderer/base_renderer.rs:146-280) as a
// persistent wave64 kernel for gfx950.
//
// Structure (MI355X-first, not the reference's depth-first per-pixel recursion):
//  * one wave owns an 8x8 pixel tile x sample range at a time; its paths form a pool of (pixel, sample) pairs handed out
//    sample-major to whichever lane needs a new path, so no lane idles while another still has samples of "its" pixel left
//    (lanes busy at shading 89 % -> 97 % at 64 samples per item, 71 % -> 81 % at 8).  The tile's film is 768 B of LDS updated
//    with ds_add_f32 and written back once per work item; the hand-out order is a function of the wave's own lock-step
//    schedule, so frames are bit-identical from run to run (when the sample range of a tile is split over several work items,
//    each writes its own slot and combine_kernel adds the slots in chunk order: no float atomics on the film);
//  * waves are persistent: they pull (tile, sample-chunk) work items from a global counter;
//  * path state never round-trips through HBM: every lane carries its path in registers and
//    regenerates the next sample of its pixel when the path ends (in-register compaction);
//  * inside a wave the path loop is a lock-step state machine: every iteration all 64 lanes trace one closest-hit
//    ray together, shade together, trace one shadow ray together; a lane whose path ended regenerates the next
//    sample of its pixel at the top of the next iteration.  Alternatives built and measured on MI355X:
//    a lane pool with vote-driven batched shading (516 vs 654 Msamples/s), one merged shadow+closest traversal loop
//    per iteration (670 vs 851), packed-f32 slab tests (-5 %): slower.  Persistent traversal with dynamic ray fetch
//    (probe_intersect_dyn_kernel below, wave-local ray batches): 1.16-1.55x on the stand-alone traversal kernel for
//    incoherent rays, 0.6x for coherent ones — not enough to pay for streaming path state through HBM;
//  * shadow rays use deferred, dense triangle tests (trace_any_deferred, pt_device.hpp): lanes only walk nodes and
//    queue (triangle, lane) pairs in an LDS ring, the wave tests 64 pairs at a time (any-hit is order independent).
//    Measured on C2: any-hit triangle steps per wave iteration 24 at 3.7 % lane use -> 1.8 at 72 %, +4-5 % Msamples/s;
//  * measured and dropped: two work items in flight per wave (a lane that finished its pixel starts the next tile's instead of
//    idling; lanes busy at shading 89 % -> 96 %, but the extra per-lane state spills and the hash table had to leave LDS:
//    -6 % at one GPU, +-0 on a 1/8 shard), a per-lane cache of the in-tile Sobol digits (not wave-uniform: -3 %);
//  * BVH traversal keeps the per-lane stack in LDS (stack[level][lane], conflict-free) and the
//    MurmurHash(dimension, seed) table of the Sobol sampler in LDS.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "layout.hpp"
#include "pt_device.hpp"
#include "pt_path.hpp"

namespace pt {

#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES_CC 3
#define PT_MIN_WAVES 4   // 128 VGPRs: measured +26 % over the unconstrained 220-VGPR build (latency hiding beats the spills)
#endif
// work item -> lane assignment
struct LaneJob { uint32_t px, py, s_cur, s_end; bool valid; };
template <bool PROBE>
PT_DEV LaneJob lane_job(uint32_t work, uint32_t lane, const DevCamera& cam, const DevParams& prm, const uint32_t* probe_xys, uint32_t n_probe) {
    LaneJob j{0, 0, 0, 0, false};
    if (PROBE) {
        uint32_t qi = work * 64 + lane;
        j.valid = qi < n_probe;
        if (j.valid) { j.px = probe_xys[3 * qi]; j.py = probe_xys[3 * qi + 1]; j.s_cur = probe_xys[3 * qi + 2]; j.s_end = j.s_cur + 1; }
    } else {
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
    }
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
}

#ifndef PT_ANY_DEFERRED
#define PT_ANY_DEFERRED 1
#endif
#ifndef PT_CLOSEST_COOP
#define PT_CLOSEST_COOP 1      // needs PT_ANY_DEFERRED (shares its LDS ring)
#endif
template <bool STATS, bool PROBE, uint32_t FEAT>
__global__ __launch_bounds__(64, ((FEAT & FEAT_CC) ? PT_MIN_WAVES_CC : PT_MIN_WAVES)) void pt_kernel(DevScene sc, DevCamera cam, DevParams prm, const uint64_t* __restrict__ dim_hash_tab,
                                                float* __restrict__ accum, float* __restrict__ partial, unsigned* __restrict__ work_counter,
                                                DevStats* __restrict__ stats, const uint32_t* __restrict__ probe_xys, uint32_t n_probe,
                                                PathOut pout) {
    __shared__ uint32_t s_stack[STACK_DEPTH * 64];
    __shared__ float s_film[64 * 3];                 // the work item's 8x8 film tile
    __shared__ uint32_t s_hi[SOBOL_HI_DIMS];
    __shared__ uint32_t s_p6[SOBOL_HI_DIMS];
    __shared__ unsigned s_work;
#if PT_ANY_DEFERRED
    __shared__ uint32_t s_ring[ANY_RING];
    __shared__ uint32_t s_occl[2];
    __shared__ uint32_t s_pair[64];
    const AnyLds any_lds{s_ring, s_occl, s_pair};
#if PT_CLOSEST_COOP
    __shared__ unsigned long long s_best[64];
    const ClosestLds closest_lds{s_ring, s_best, s_pair};
#endif
#endif
    const uint32_t lane = threadIdx.x;
    uint32_t* stack = s_stack + lane;
    // murmur(dimension, seed) comes straight from its 1 KB global table (L1-resident): the LDS it used holds the tile's film
    SamplerCtx sctx{prm.sampler, prm.seed, prm.log2_spp, prm.n_base4_digits, cam.width, dim_hash_tab, nullptr, 0u, 0u, nullptr};
    StatCounters st{};
    unsigned long long tp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_loop0 = 0;
    if (STATS) t_loop0 = __builtin_amdgcn_s_memtime();

    for (;;) {
        if (lane == 0) s_work = atomicAdd(work_counter, 1u);
        __syncthreads();
        const uint32_t work = s_work;
        __syncthreads();
        if (work >= prm.n_work) break;
        // this lane's own pixel of the tile (film write-back) and the work item's wave-uniform sample range
        const LaneJob job = lane_job<PROBE>(work, lane, cam, prm, probe_xys, n_probe);
        const LaneJob job0 = lane_job<PROBE>(work, 0u, cam, prm, probe_xys, n_probe);
        const uint32_t blk_log2 = PROBE ? 3u : prm.block_log2, blk_mask = (1u << blk_log2) - 1u;
        const uint32_t s_prefix = PROBE ? 0u : prm.sample_prefix_digits;
        if (!PROBE && prm.sampler == 1u && sobol_hi_first(prm.log2_spp, blk_log2) - s_prefix < prm.n_base4_digits) {
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
        s_film[3 * lane] = 0.0f; s_film[3 * lane + 1] = 0.0f; s_film[3 * lane + 2] = 0.0f;
        __syncthreads();
        const uint32_t n_s = PROBE ? 1u : (job0.s_end > job0.s_cur ? job0.s_end - job0.s_cur : 0u);
        const uint32_t pool_size = n_s << (2u * blk_log2);
        uint32_t pool_next = 0u;                                   // wave-uniform
        uint32_t my_pix = lane;
        Path P{};
        bool active = false;
        while (true) {
            unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, tsa = 0, tsb = 0;
            if (STATS) ts0 = __builtin_amdgcn_s_memtime();
            const unsigned long long m_needy = __ballot(!active);
            if (m_needy != 0ull && pool_next < pool_size) {
                const uint32_t idx = pool_next + (uint32_t)__popcll(m_needy & ((1ull << lane) - 1ull));
                if (!active && idx < pool_size) {
                    const uint32_t pix = idx & ((1u << (2u * blk_log2)) - 1u);
                    uint32_t px, py, smp_i; bool valid;
                    if (PROBE) {
                        const uint32_t qi = work * 64u + pix;
                        valid = qi < n_probe;
                        px = valid ? probe_xys[3 * qi] : 0u; py = valid ? probe_xys[3 * qi + 1] : 0u; smp_i = valid ? probe_xys[3 * qi + 2] : 0u;
                    } else {
                        px = job0.px + (pix & blk_mask); py = job0.py + (pix >> blk_log2);
                        smp_i = job0.s_cur + (idx >> (2u * blk_log2));
                        valid = px < cam.width && py < cam.height;
                    }
                    if (valid) { active = true; my_pix = pix; regen_path<STATS>(P, sctx, cam, px, py, smp_i, st); }
                }
                pool_next = min(pool_next + (uint32_t)__popcll(m_needy), pool_size);
            }
            if (!__any(active)) { if (pool_next >= pool_size) break; continue; }
            if (STATS) { ts1 = __builtin_amdgcn_s_memtime(); if (lane == 0) st.w[4]++; if (active) st.w[5]++; }
            Hit hit{};
            bool got = false;
#if PT_CLOSEST_COOP
            if (STATS && prm.stats_mode == 1u) { if (active) got = trace_closest<STATS>(sc, P.ro, P.rd, 3.402823466e+38f, stack, hit, st); }
            else got = trace_closest_coop<STATS>(sc, P.ro, P.rd, active, stack, lane, closest_lds, hit, st);
#else
            if (active) got = trace_closest<STATS>(sc, P.ro, P.rd, 3.402823466e+38f, stack, hit, st);
#endif
            if (STATS) ts2 = __builtin_amdgcn_s_memtime();
            bool end_path = false;
            ShadowReq sh{};
            if constexpr ((FEAT & FEAT_CC) != 0u) {
                ShadeCtx C;
                C.cont = false; C.need_cc = false; C.cc_fc = 0.0f; C.cc_alpha_c = 0.0f; C.cc_r0c = 0.0f; C.wo_nm = mk3(0, 0, 1); C.mc_key = 0ull;
                if (active) end_path = shade_vertex_a<STATS, FEAT>(P, sc, prm, sctx, got, hit, sh, st, tsa, C);
                // the coat's 64-sample directional albedo, estimated by the whole wave for the lanes that need it
                C.cc_fc = coat_directional_albedo_coop(active && C.cont && C.need_cc, C.cc_alpha_c, C.cc_r0c, C.wo_nm, C.mc_key, lane);
                if (active && C.cont) end_path = shade_vertex_b<STATS, FEAT>(P, sc, prm, sctx, sh, st, tsb, C);
            } else {
                if (active) end_path = shade_vertex<STATS, FEAT>(P, sc, prm, sctx, got, hit, sh, st, tsa, tsb);
            }
            if (STATS) {
                ts3 = __builtin_amdgcn_s_memtime();
                // the stamps inside shade_vertex are taken by the lanes that reach them: make them wave-level (first lane that has one)
                unsigned long long ma = __ballot(tsa != 0ull), mb = __ballot(tsb != 0ull);
                if (ma) { int l = (int)__ffsll((long long)ma) - 1; tsa = ((unsigned long long)__shfl((uint32_t)(tsa >> 32), l) << 32) | __shfl((uint32_t)tsa, l); }
                if (mb) { int l = (int)__ffsll((long long)mb) - 1; tsb = ((unsigned long long)__shfl((uint32_t)(tsb >> 32), l) << 32) | __shfl((uint32_t)tsb, l); }
            }
            // a light connection whose contribution is exactly zero (light behind the surface, f == 0) cannot change L whatever the
            // visibility test says: the production path does not trace it (the canonical-count mode does, like the reference)
            if (!(STATS && prm.stats_mode == 1u) && sh.on && sh.c[0] == 0.0f && sh.c[1] == 0.0f && sh.c[2] == 0.0f && sh.c[3] == 0.0f) sh.on = false;
#if PT_ANY_DEFERRED
            if (__any(sh.on)) {
                if (STATS && sh.on) st.w[6]++;
                bool occluded = false;
                if (STATS && prm.stats_mode == 1u) { if (sh.on) occluded = trace_any<STATS>(sc, sh.o, sh.d, sh.t, stack, st); }
                else occluded = trace_any_deferred<STATS>(sc, sh.o, sh.d, sh.t, sh.on, stack, lane, any_lds, st);
                if (sh.on && !occluded) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) P.L[i] = P.L[i] + sh.c[i];
                }
            }
#else
            if (sh.on) {
                if (STATS) st.w[6]++;
                bool occluded = trace_any<STATS>(sc, sh.o, sh.d, sh.t, stack, st);
                if (!occluded) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) P.L[i] = P.L[i] + sh.c[i];
                }
            }
#endif
            if (STATS) ts4 = __builtin_amdgcn_s_memtime();
            if (active && end_path) {
                if (PROBE) { float a0 = 0, a1 = 0, a2 = 0; film_add<true>(P, sc, prm, a0, a1, a2, pout, work * 64u + my_pix); }
                else {
                    float r, g, b;
                    film_rgb(P, sc, prm, r, g, b);
                    atomicAdd(&s_film[3 * my_pix], r); atomicAdd(&s_film[3 * my_pix + 1], g); atomicAdd(&s_film[3 * my_pix + 2], b);
                }
                active = false;
            }
            if (STATS) {
                unsigned long long ts5 = __builtin_amdgcn_s_memtime();
                tp[0] += ts1 - ts0; tp[1] += ts2 - ts1; tp[2] += ts3 - ts2; tp[3] += ts4 - ts3; tp[4] += ts5 - ts4;
                if (tsa) { tp[6] += tsa - ts2; if (tsb) { tp[7] += tsb - tsa; tp[8] += ts3 - tsb; } else tp[7] += ts3 - tsa; } else tp[6] += ts3 - ts2;
            }
        }
        __syncthreads();
        if (!PROBE && job.valid) {
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
    }
    if (STATS) flush_stats(stats, st);
}


// Sensor::to_rgb (sensor.rs:81-88) + ReinhardToneMap (tone_map.rs:20-28) + sRGB OETF (eotf.rs:54-61)
__global__ void resolve_kernel(const float* __restrict__ accum, uint32_t n_values, float inv_unused, uint32_t spp, float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t stride = gridDim.x * blockDim.x;
    for (; i < n_values; i += stride) {
        float c = accum[i] / (float)spp;
        c = fmaxf(c, 0.0f);
        c = c / (1.0f + c);
        out[i] = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    }
}

__global__ void probe_sobol_kernel(uint32_t width, uint32_t seed, uint32_t log2_spp, uint32_t nb4, const uint32_t* __restrict__ xys, uint32_t n,
                                   const uint8_t* __restrict__ pattern, uint32_t n_pat, uint32_t per, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // no LDS hash table here: dim_hash falls back to computing the MurmurHash when the pointer table is absent
    SamplerCtx c{1u, seed, log2_spp, nb4, width, nullptr, nullptr, 0u, 0u};
    Sampler s;
    sampler_start(s, c, xys[3 * i], xys[3 * i + 1], xys[3 * i + 2]);
    uint32_t* o = out + (size_t)i * per;
    for (uint32_t k = 0; k < n_pat; ++k) {
        if (pattern[k] == '2') {
            uint64_t si = sobol_sample_index(s.morton, s.dimension, log2_spp, nb4);
            s.dimension += 2;
            uint64_t h = murmur_dim_seed(s.dimension, seed);
            *o++ = fast_owen(sobol_dim0(si), (uint32_t)h);
            *o++ = fast_owen(sobol_dim1(si), (uint32_t)(h >> 32));
        } else {
            uint64_t si = sobol_sample_index(s.morton, s.dimension, log2_spp, nb4);
            s.dimension += 1;
            uint64_t h = murmur_dim_seed(s.dimension, seed);
            *o++ = fast_owen(sobol_dim0(si), (uint32_t)h);
        }
    }
}

__global__ __launch_bounds__(64) void probe_intersect_kernel(DevScene sc, const float* __restrict__ o, const float* __restrict__ d, uint32_t n,
                                                             float* __restrict__ out_t, uint32_t* __restrict__ out_inst, uint32_t* __restrict__ out_tri,
                                                             float* __restrict__ out_n) {
    __shared__ uint32_t s_stack[STACK_DEPTH * 64];
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    StatCounters st{};
    Hit h{};
    f3 ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    if (trace_closest<false>(sc, ro, rd, 3.402823466e+38f, s_stack + threadIdx.x, h, st)) {
        Surface sf = load_surface(sc, h);
        const float4* q = (const float4*)(sc.shade + h.tri);
        float4 e = q[4], f = q[5];
        out_t[i] = h.t; out_inst[i] = __float_as_uint(e.w); out_tri[i] = __float_as_uint(f.z);
        if (out_n) { out_n[3 * i] = sf.ng.x; out_n[3 * i + 1] = sf.ng.y; out_n[3 * i + 2] = sf.ng.z; }
    } else {
        out_t[i] = -1.0f; out_inst[i] = 0xffffffffu; out_tri[i] = 0xffffffffu;
        if (out_n) { out_n[3 * i] = 0; out_n[3 * i + 1] = 0; out_n[3 * i + 2] = 0; }
    }
}

__global__ __launch_bounds__(64) void probe_occluded_kernel(DevScene sc, const float* __restrict__ o, const float* __restrict__ d,
                                                            const float* __restrict__ tmax, uint32_t n, uint8_t* __restrict__ out) {
    __shared__ uint32_t s_stack[STACK_DEPTH * 64];
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    StatCounters st{};
    f3 ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    out[i] = trace_any<false>(sc, ro, rd, tmax[i], s_stack + threadIdx.x, st) ? 1 : 0;
}


// ---- experiment (MI355PT_TRAV=2, probe_intersect only): persistent traversal with dynamic ray fetch -----------------
// Waves own a contiguous batch of rays (one global atomic per DYN_BATCH rays); a lane whose ray finished takes the next
// ray of the batch, so the wave does not idle through the tail of its slowest ray.  Leaves are postponed until
// DYN_LEAF_MIN lanes hold one (or nobody has a node left).
#ifndef DYN_BATCH
#define DYN_BATCH 1024u
#endif
#ifndef DYN_LEAF_MIN
#define DYN_LEAF_MIN 16
#endif
#ifndef DYN_REFILL_MIN
#define DYN_REFILL_MIN 12
#endif
#ifndef DYN_WAVES
#define DYN_WAVES 8
#endif
__global__ __launch_bounds__(64, DYN_WAVES) void probe_intersect_dyn_kernel(DevScene sc, const float* __restrict__ o, const float* __restrict__ d, uint32_t n,
                                                                 float* __restrict__ out_t, uint32_t* __restrict__ out_inst,
                                                                 uint32_t* __restrict__ out_tri, unsigned* __restrict__ counter) {
    __shared__ uint32_t s_stack[STACK_DEPTH * 64];
    uint32_t* stack = s_stack + threadIdx.x;
    const uint32_t lane = threadIdx.x;
    bool have = false, pool_empty = false;
    uint32_t next = 0, end = 0;          // wave-uniform: the batch this wave is serving
    uint32_t ray = 0;
    f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1);
    RaySetup rs{};
    float t_best = 0.0f;
    int32_t cur = 0; int sp = 0;
    uint32_t best_tri = 0; bool found = false;
    for (;;) {
        // ---- refill: idle lanes take the next rays of the wave's batch ----
        unsigned long long idle = __ballot(!have);
        if (idle != 0ull) {
            if (next >= end && !pool_empty) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(counter, DYN_BATCH);
                base = __shfl(base, 0);
                if (base >= n) pool_empty = true; else { next = base; end = min(base + DYN_BATCH, n); }
            }
            uint32_t avail = end - next;
            if (avail != 0u) {
                uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (!have && rank < avail) {
                    uint32_t r = next + rank;
                    ray = r; have = true;
                    ro = mk3(o[3 * r], o[3 * r + 1], o[3 * r + 2]); rd = mk3(d[3 * r], d[3 * r + 1], d[3 * r + 2]);
                    rs = setup_ray(rd); t_best = 3.402823466e+38f; cur = sc.root; sp = 0; found = false;
                }
                next += min((uint32_t)__popcll(idle), avail);
            }
        }
        if (__ballot(have) == 0ull) { if (pool_empty || next >= end) { if (pool_empty) break; } continue; }
        // ---- traversal quantum ----
        for (int q = 0; q < 32; ++q) {
            bool fin = false;
            if (have && cur >= 0) {
                const float4* qn = (const float4*)(sc.nodes + cur);
                float4 nx = qn[0], ny = qn[1], nz = qn[2];
                int2 ch = *(const int2*)(qn + 3);
                float l0x = (nx.x - ro.x) * rs.inv.x, h0x = (nx.z - ro.x) * rs.inv.x;
                float l1x = (nx.y - ro.x) * rs.inv.x, h1x = (nx.w - ro.x) * rs.inv.x;
                float l0y = (ny.x - ro.y) * rs.inv.y, h0y = (ny.z - ro.y) * rs.inv.y;
                float l1y = (ny.y - ro.y) * rs.inv.y, h1y = (ny.w - ro.y) * rs.inv.y;
                float l0z = (nz.x - ro.z) * rs.inv.z, h0z = (nz.z - ro.z) * rs.inv.z;
                float l1z = (nz.y - ro.z) * rs.inv.z, h1z = (nz.w - ro.z) * rs.inv.z;
                float n0 = fmaxf(fmaxf(fminf(l0x, h0x), fminf(l0y, h0y)), fmaxf(fminf(l0z, h0z), 0.0f));
                float f0 = fminf(fminf(fmaxf(l0x, h0x), fmaxf(l0y, h0y)), fminf(fmaxf(l0z, h0z), t_best));
                float n1 = fmaxf(fmaxf(fminf(l1x, h1x), fminf(l1y, h1y)), fmaxf(fminf(l1z, h1z), 0.0f));
                float f1 = fminf(fminf(fmaxf(l1x, h1x), fmaxf(l1y, h1y)), fminf(fmaxf(l1z, h1z), t_best));
                bool hit0 = n0 <= f0, hit1 = n1 <= f1;
                if (hit0 && hit1) { bool first0 = n0 <= n1; stack[sp * 64] = (uint32_t)(first0 ? ch.y : ch.x); ++sp; cur = first0 ? ch.x : ch.y; }
                else if (hit0) cur = ch.x;
                else if (hit1) cur = ch.y;
                else if (sp == 0) fin = true;
                else { --sp; cur = (int32_t)stack[sp * 64]; }
            }
            const bool at_leaf = have && !fin && cur < 0;
            const unsigned long long m_leaf = __ballot(at_leaf);
            if (m_leaf != 0ull && (__popcll(m_leaf) >= DYN_LEAF_MIN || __ballot(have && !fin && cur >= 0) == 0ull)) {
                if (at_leaf) {
                    uint32_t first = leaf_first(cur), cnt = leaf_count(cur);
                    for (uint32_t i = 0; i < cnt; ++i) {
                        TriVerts tv = load_tri(sc.tris, first + i);
                        float t, b0, b1, b2;
                        if (intersect_triangle(ro, rd, rs.kx, rs.ky, rs.kz, rs.sx, rs.sy, rs.sz, t_best, tv, t, b0, b1, b2)) {
                            if (!found || t < t_best) { found = true; t_best = t; best_tri = first + i; }
                        }
                    }
                    if (sp == 0) fin = true; else { --sp; cur = (int32_t)stack[sp * 64]; }
                }
            }
            if (fin) {
                const float4* qs = (const float4*)(sc.shade + best_tri);
                out_t[ray] = found ? t_best : -1.0f;
                out_inst[ray] = found ? __float_as_uint(qs[4].w) : 0xffffffffu;
                out_tri[ray] = found ? __float_as_uint(qs[5].z) : 0xffffffffu;
                have = false;
            }
            const unsigned long long m_have = __ballot(have);
            if (m_have == 0ull) break;
            if (64 - __popcll(m_have) >= DYN_REFILL_MIN && !(pool_empty && next >= end)) break;   // refill
        }
    }
}

// adds the per-chunk film tiles of a split launch to the film, in chunk order (one thread per pixel of each tile of the shard)
__global__ void combine_kernel(DevCamera cam, DevParams prm, const float* __restrict__ partial, float* __restrict__ accum, uint32_t n_tiles) {
    const uint32_t tile_k = blockIdx.x, lane = threadIdx.x;
    if (tile_k >= n_tiles) return;
    const uint32_t tile = prm.shard_index + tile_k * prm.shard_count;
    const uint32_t px = (tile % prm.tiles_x) * 8u + (lane & 7u), py = (tile / prm.tiles_x) * 8u + (lane >> 3);
    if (px >= cam.width || py >= cam.height) return;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t c = 0; c < prm.chunks; ++c) {
        const float* slot = partial + (((size_t)tile_k * prm.chunks + c) * 64u + lane) * 3u;
        r += slot[0]; g += slot[1]; b += slot[2];
    }
    const size_t o = ((size_t)py * cam.width + px) * 3;
    accum[o] += r; accum[o + 1] += g; accum[o + 2] += b;
}

// ---------------------------------------------------------------------------------------------
// launch wrappers (host side, called from api.cpp)
// ---------------------------------------------------------------------------------------------
// smallest compiled specialisation covering `feat`
static uint32_t pick_features(uint32_t feat) {
    const uint32_t sets[] = {0u, FEAT_TEX, FEAT_DIEL, FEAT_METAL, FEAT_DIEL | FEAT_ROUGH, FEAT_CC, FEAT_CC | FEAT_TEX, FEAT_ALL & ~FEAT_CC, FEAT_ALL};
    for (uint32_t s : sets) if ((feat & ~s) == 0u) return s;
    return FEAT_ALL;
}
#define PT_LAUNCH(F) hipLaunchKernelGGL((pt_kernel<false, false, F>), dim3(grid), dim3(64), 0, stream, sc, cam, prm, d_hash, d_accum, d_partial, d_counter, d_stats, nullptr, 0u, po)
hipError_t launch_pt(const DevScene& sc, const DevCamera& cam, const DevParams& prm, const uint64_t* d_hash, float* d_accum, float* d_partial,
                     unsigned* d_counter, DevStats* d_stats, bool stats, uint32_t feat, int grid, hipStream_t stream) {
    PathOut po{nullptr, nullptr, nullptr};
    if (stats) {
        hipLaunchKernelGGL((pt_kernel<true, false, FEAT_ALL>), dim3(grid), dim3(64), 0, stream, sc, cam, prm, d_hash, d_accum, d_partial, d_counter, d_stats, nullptr, 0u, po);
    } else
    switch (pick_features(feat)) {
        case 0u: PT_LAUNCH(0u); break;
        case FEAT_TEX: PT_LAUNCH(FEAT_TEX); break;
        case FEAT_DIEL: PT_LAUNCH(FEAT_DIEL); break;
        case FEAT_METAL: PT_LAUNCH(FEAT_METAL); break;
        case FEAT_DIEL | FEAT_ROUGH: PT_LAUNCH(FEAT_DIEL | FEAT_ROUGH); break;
        case FEAT_CC: PT_LAUNCH(FEAT_CC); break;
        case FEAT_CC | FEAT_TEX: PT_LAUNCH(FEAT_CC | FEAT_TEX); break;
        case FEAT_ALL & ~FEAT_CC: PT_LAUNCH(FEAT_ALL & ~FEAT_CC); break;
        default: PT_LAUNCH(FEAT_ALL); break;
    }
    if (prm.chunks > 1) {
        const uint32_t n_tiles = (prm.n_work / prm.chunks) >> (6u - 2u * prm.block_log2);   // n_work = tiles * blocks per tile * chunks
        hipLaunchKernelGGL(combine_kernel, dim3(n_tiles), dim3(64), 0, stream, cam, prm, (const float*)d_partial, d_accum, n_tiles);
    }
    return hipGetLastError();
}
#undef PT_LAUNCH
hipError_t launch_probe_radiance(const DevScene& sc, const DevCamera& cam, const DevParams& prm, const uint64_t* d_hash, unsigned* d_counter,
                                 const uint32_t* d_xys, uint32_t n, float* d_L, float* d_lam, float* d_pdf, int grid, hipStream_t stream) {
    PathOut po{d_L, d_lam, d_pdf};
    hipLaunchKernelGGL((pt_kernel<false, true, FEAT_ALL>), dim3(grid), dim3(64), 0, stream, sc, cam, prm, d_hash, nullptr, nullptr, d_counter, nullptr, d_xys, n, po);
    return hipGetLastError();
}
hipError_t launch_resolve(const float* d_accum, uint32_t n_values, uint32_t spp, float* d_out, hipStream_t stream) {
    int grid = (int)std::min<uint32_t>((n_values + 255) / 256, 2048u);
    hipLaunchKernelGGL(resolve_kernel, dim3(grid), dim3(256), 0, stream, d_accum, n_values, 0.0f, spp, d_out);
    return hipGetLastError();
}
hipError_t launch_probe_sobol(uint32_t width, uint32_t seed, uint32_t log2_spp, uint32_t nb4, const uint32_t* d_xys, uint32_t n,
                              const uint8_t* d_pat, uint32_t n_pat, uint32_t per, uint32_t* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(probe_sobol_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, width, seed, log2_spp, nb4, d_xys, n, d_pat, n_pat, per, d_out);
    return hipGetLastError();
}
hipError_t launch_probe_intersect(const DevScene& sc, const float* o, const float* d, uint32_t n, float* t, uint32_t* inst, uint32_t* tri, float* nrm,
                                  hipStream_t stream) {
    const char* mode = getenv("MI355PT_TRAV");
    if (mode && mode[0] == '2') {
        static unsigned* d_ctr = nullptr;
        if (!d_ctr && hipMalloc((void**)&d_ctr, sizeof(unsigned)) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_ctr, 0, sizeof(unsigned), stream);
        if (nrm) (void)hipMemsetAsync(nrm, 0, sizeof(float) * 3 * (size_t)n, stream);
        int nb = 0, dev = 0; hipDeviceProp_t prop;
        (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&prop, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe_intersect_dyn_kernel, 64, 0);
        int grid = std::min<int>((int)((n + 63) / 64), nb * prop.multiProcessorCount);
        hipLaunchKernelGGL(probe_intersect_dyn_kernel, dim3(grid), dim3(64), 0, stream, sc, o, d, n, t, inst, tri, d_ctr);
        return hipGetLastError();
    }
    const char* lds = getenv("MI355PT_PROBE_LDS");   // experiment: extra dynamic LDS per wave to throttle occupancy
    hipLaunchKernelGGL(probe_intersect_kernel, dim3((n + 63) / 64), dim3(64), lds ? atoi(lds) : 0, stream, sc, o, d, n, t, inst, tri, nrm);
    return hipGetLastError();
}
hipError_t launch_probe_occluded(const DevScene& sc, const float* o, const float* d, const float* tmax, uint32_t n, uint8_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(probe_occluded_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, sc, o, d, tmax, n, out);
    return hipGetLastError();
}

}  // namespace pt

namespace pt {
uint64_t host_murmur_dim_seed(uint32_t dimension, uint32_t seed) { return murmur_dim_seed(dimension, seed); }
// resident 64-thread blocks (= waves) of the render kernel on the current device: the persistent grid size
// resident 64-thread blocks of the kernel specialisation that `feat` selects (the clearcoat kernels run 3 waves per SIMD, the
// others 4): the persistent grid size
int query_resident_waves(uint32_t feat) {
    int dev = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 2048;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2048;
    hipError_t e;
    switch (pick_features(feat)) {
        case 0u: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<false, false, 0u>, 64, 0); break;
        case FEAT_TEX: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<false, false, FEAT_TEX>, 64, 0); break;
        case FEAT_DIEL: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<false, false, FEAT_DIEL>, 64, 0); break;
        case FEAT_METAL: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<false, false, FEAT_METAL>, 64, 0); break;
        case FEAT_DIEL | FEAT_ROUGH: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<false, false, FEAT_DIEL | FEAT_ROUGH>, 64, 0); break;
        case FEAT_CC: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<false, false, FEAT_CC>, 64, 0); break;
        case FEAT_CC | FEAT_TEX: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<false, false, FEAT_CC | FEAT_TEX>, 64, 0); break;
        case FEAT_ALL & ~FEAT_CC: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<false, false, FEAT_ALL & ~FEAT_CC>, 64, 0); break;
        default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<false, false, FEAT_ALL>, 64, 0); break;
    }
    if (e != hipSuccess || per_cu <= 0) per_cu = 8;
    return prop.multiProcessorCount * per_cu;
}
}  // namespace pt
