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
//  * inside a wave the path loop is a lock-step state machine: every iteration all 64 lanes regenerate (those without a path),
//    trace together — ONE cooperative traversal for the closest-hit rays and the pending light connections —, run the FRONT
//    of their vertex together (emission, throughput, Russian roulette); a path that goes on is written to the wave's own queue
//    in global memory (80 B) and frees its lane, and whenever 64 paths wait one pass runs the rest of the vertex — surface,
//    frames, BSDF sample, light connection — for a FULL wave (round 3: the tail queue, pt_kernel.hpp PT_TAILQ; before it a third
//    of the lanes idled through that tail and a minority material's branch ran in every iteration).  Alternatives built and measured on MI355X:
//    a lane pool with vote-driven batched shading (516 vs 654 Msamples/s), one merged shadow+closest traversal loop
//    per iteration (670 vs 851), packed-f32 slab tests (-5 %): slower.  Persistent traversal with dynamic ray fetch
//    (experiments/probe_intersect_dyn.inc, wave-local ray batches): 1.16-1.55x on the stand-alone traversal kernel for
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

#include "pt_kernel.hpp"

namespace pt {

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
        Surface sf = load_surface(sc, h, rd);
        const float4* q = (const float4*)(sc.shade + h.tri);
        float4 e = q[4], f = q[5];
        out_t[i] = h.t; out_inst[i] = __float_as_uint(e.w); out_tri[i] = __float_as_uint(f.z);
        if (out_n) { out_n[3 * i] = sf.ng.x; out_n[3 * i + 1] = sf.ng.y; out_n[3 * i + 2] = sf.ng.z; }
    } else {
        out_t[i] = -1.0f; out_inst[i] = 0xffffffffu; out_tri[i] = 0xffffffffu;
        if (out_n) { out_n[3 * i] = 0; out_n[3 * i + 1] = 0; out_n[3 * i + 2] = 0; }
    }
}

// ref_sincosf (pt_device.hpp -> pt_libm.hpp) for the floats with bit patterns first + i * stride: mi355pt_probe_sincos compares them with the host's libm
__global__ void probe_sincos_kernel(uint32_t first, uint32_t stride, uint32_t n, float* __restrict__ out_s, float* __restrict__ out_c) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    ref_sincosf(__uint_as_float(first + i * stride), &s, &c);
    out_s[i] = s; out_c[i] = c;
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


#ifdef PT_EXPERIMENTS
#include "experiments/probe_intersect_dyn.inc"
#endif

// dst += src over a film (multi-device gather: the films of the other devices' tile shards, disjoint from this device's own)
__global__ void film_add_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 a = ((const float4*)dst)[i], b = ((const float4*)src)[i];
        ((float4*)dst)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// Multi-device gather (mi355pt_render_multi): a device's shard of the frame as a COMPACT film — its 8x8 tiles in shard order (tile k of the
// shard = frame tile shard_index + k * shard_count), 64 pixels x 3 floats each, pixels outside the frame 0 — so that 1/N of the film
// crosses xGMI per peer instead of the whole frame; and the inverse on the gathering device (the shards' tiles are disjoint: plain stores).
__global__ void film_pack_kernel(const float* __restrict__ film, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t shard_index, uint32_t shard_count,
                                 uint32_t n_tiles, float* __restrict__ packed) {
    const uint32_t k = blockIdx.x, lane = threadIdx.x;
    if (k >= n_tiles) return;
    const uint32_t tile = shard_index + k * shard_count;
    const uint32_t px = (tile % tiles_x) * 8u + (lane & 7u), py = (tile / tiles_x) * 8u + (lane >> 3);
    const bool in = px < width && py < height;
    const size_t o = ((size_t)py * width + px) * 3, q = ((size_t)k * 64u + lane) * 3u;
    packed[q] = in ? film[o] : 0.0f; packed[q + 1] = in ? film[o + 1] : 0.0f; packed[q + 2] = in ? film[o + 2] : 0.0f;
}
__global__ void film_unpack_kernel(float* __restrict__ film, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t shard_index, uint32_t shard_count,
                                   uint32_t n_tiles, const float* __restrict__ packed) {
    const uint32_t k = blockIdx.x, lane = threadIdx.x;
    if (k >= n_tiles) return;
    const uint32_t tile = shard_index + k * shard_count;
    const uint32_t px = (tile % tiles_x) * 8u + (lane & 7u), py = (tile / tiles_x) * 8u + (lane >> 3);
    if (px >= width || py >= height) return;
    const size_t o = ((size_t)py * width + px) * 3, q = ((size_t)k * 64u + lane) * 3u;
    film[o] = packed[q]; film[o + 1] = packed[q + 1]; film[o + 2] = packed[q + 2];
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
hipError_t launch_pt(const DevScene& sc, const DevCamera& cam, const DevParams& prm, const uint64_t* d_hash, float* d_accum, float* d_partial,
                     unsigned* d_counter, DevStats* d_stats, bool stats, uint32_t feat, int grid, hipStream_t stream, const PathOut& pout, float* d_defer) {
    const PtLaunchArgs a{sc, cam, prm, d_hash, d_accum, d_partial, d_counter, d_stats, grid, stream, pout, (float4*)d_defer};
    if (stats) {
        // two instrumented variants: scenes without the clearcoat code get the one whose traversal has the production form (merged, 4 waves
        // per SIMD), so that the lane-use diagnostics describe what the benchmarked kernels do
        if ((feat & (FEAT_CC | FEAT_EMTEX)) == 0u)
            hipLaunchKernelGGL((pt_kernel<true, FEAT_STD & ~FEAT_CC, MODE_GENERIC>), dim3(grid), dim3(64), 0, stream, sc, cam, prm, d_hash, d_accum, d_partial, d_counter, d_stats, pout, a.d_defer);
        else
            hipLaunchKernelGGL((pt_kernel<true, FEAT_ALL>), dim3(grid), dim3(64), 0, stream, sc, cam, prm, d_hash, d_accum, d_partial, d_counter, d_stats, pout, a.d_defer);
    } else if (prm.sampler == 1u && prm.strategy == 2u) launch_pt_mis_sobol(a, feat);
    else if (prm.sampler == 1u && prm.strategy == 1u) launch_pt_nee_sobol(a, feat);
    else if (prm.strategy == 0u) launch_pt_strategy_pt(a, feat);
    else launch_pt_mode<MODE_GENERIC>(a, feat);
    if (prm.chunks > 1) {
        const uint32_t n_tiles = (prm.n_work / prm.chunks) >> (6u - 2u * prm.block_log2);   // n_work = tiles * blocks per tile * chunks
        hipLaunchKernelGGL(combine_kernel, dim3(n_tiles), dim3(64), 0, stream, cam, prm, (const float*)d_partial, d_accum, n_tiles);
    }
    return hipGetLastError();
}
hipError_t launch_film_add(float* dst, const float* src, size_t n_floats, hipStream_t stream) {   // n_floats % 4 == 0 (padded by the caller)
    const size_t n4 = n_floats / 4;
    int grid = (int)std::min<size_t>((n4 + 255) / 256, 4096);
    hipLaunchKernelGGL(film_add_kernel, dim3(grid), dim3(256), 0, stream, dst, src, n4);
    return hipGetLastError();
}
hipError_t launch_film_pack(const float* film, uint32_t w, uint32_t h, uint32_t shard_index, uint32_t shard_count, uint32_t n_tiles, float* packed, hipStream_t stream) {
    if (n_tiles) hipLaunchKernelGGL(film_pack_kernel, dim3(n_tiles), dim3(64), 0, stream, film, w, h, (w + 7u) / 8u, shard_index, shard_count, n_tiles, packed);
    return hipGetLastError();
}
hipError_t launch_film_unpack(float* film, uint32_t w, uint32_t h, uint32_t shard_index, uint32_t shard_count, uint32_t n_tiles, const float* packed, hipStream_t stream) {
    if (n_tiles) hipLaunchKernelGGL(film_unpack_kernel, dim3(n_tiles), dim3(64), 0, stream, film, w, h, (w + 7u) / 8u, shard_index, shard_count, n_tiles, packed);
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
#ifdef PT_EXPERIMENTS
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
#endif
    hipLaunchKernelGGL(probe_intersect_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, sc, o, d, n, t, inst, tri, nrm);
    return hipGetLastError();
}
hipError_t launch_probe_sincos(uint32_t first, uint32_t stride, uint32_t n, float* out_s, float* out_c, hipStream_t stream) {
    hipLaunchKernelGGL(probe_sincos_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, first, stride, n, out_s, out_c);
    return hipGetLastError();
}
hipError_t launch_probe_occluded(const DevScene& sc, const float* o, const float* d, const float* tmax, uint32_t n, uint8_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(probe_occluded_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, sc, o, d, tmax, n, out);
    return hipGetLastError();
}

}  // namespace pt

namespace pt {
uint64_t host_murmur_dim_seed(uint32_t dimension, uint32_t seed) { return murmur_dim_seed(dimension, seed); }
size_t query_defer_bytes_per_wave() { return defer_bytes_per_wave(); }     // the deferral queues of pt_kernel.hpp (0: compiled out)
// resident 64-thread blocks (= waves) on the current device of the EXACT kernel instantiation launch_pt takes for (stats, feat, sampler,
// strategy): the persistent grid size.  The MODE specialisations are separate translation units with their own backend flags, so their
// register counts — and with them the occupancy — need not be those of the generic variant.  Cached per scene and device (api.cpp LaunchCtx).
int query_resident_waves(bool stats, uint32_t feat, uint32_t sampler, uint32_t strategy) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 2048;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2048;
    int per_cu = 0;
    if (stats) {
        hipError_t e;
        if ((feat & (FEAT_CC | FEAT_EMTEX)) == 0u) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<true, FEAT_STD & ~FEAT_CC, MODE_GENERIC>, 64, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pt_kernel<true, FEAT_ALL, MODE_GENERIC>, 64, 0);
        if (e != hipSuccess || per_cu <= 0) per_cu = 8;
    } else if (sampler == 1u && strategy == 2u) per_cu = occupancy_pt_mis_sobol(feat);
    else if (sampler == 1u && strategy == 1u) per_cu = occupancy_pt_nee_sobol(feat);
    else if (strategy == 0u) per_cu = occupancy_pt_strategy_pt(feat);
    else per_cu = occupancy_pt_mode<MODE_GENERIC>(feat);
    return prop.multiProcessorCount * per_cu;
}
}  // namespace pt
