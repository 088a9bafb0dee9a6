/* mi355pt_debug.h — test and diagnosis surface of libmi355pt.so.
 *
 * Everything here is used by the parity tests (tests/), the tools under tools/ and bench.py's instrumented launch; nothing here is needed
 * to drop the library in behind renderer::render() (that is include/mi355pt.h).  The probes run the SAME device functions as the render
 * path, and the per-sample log is written by the production kernel itself, which is why they live in the product library
 * rather than in a test build of it. */
#ifndef MI355PT_DEBUG_H
#define MI355PT_DEBUG_H
#include "mi355pt.h"
#ifdef __cplusplus
extern "C" {
#endif

/* mi355pt_params.rr_gate_slack (and nothing else) changes what a render call computes for diagnostic purposes; it is honoured only after
 * mi355pt_debug_unlock(1) and refused otherwise.  Process-wide; mi355pt_debug_unlock(0) locks again.  Returns the previous state. */
int mi355pt_debug_unlock(int on);

/* How the next mi355pt_scene_build lowers the instances (A/B runs and tests; the three paths must give the reference's hit alike):
 * 0 (default) the triangle array holds LOCAL vertices when every instance is the same pure translation - every triangle test of every
 *   traversal is then the reference's (primitive/impls/triangle_mesh.rs:89-119) - else render-space triangles and the triangle found is
 *   intersected again in its mesh's local space; 1 never the local array; 2 in addition every instance through the full matrix path. */
int mi355pt_scene_debug_set_lowering(mi355pt_scene* s, int mode);

/* ---------------- probes (parity tests; same device code as the render path) ---------------- */
/* The built acceleration structure as the device holds it (no reference counterpart; the reference's is scene/src/bvh.rs:300-343):
 * node records of 64 B {bx[4] = lo0.x lo1.x hi0.x hi1.x, by[4], bz[4], int32 child[2] (>= 0 node index, < 0 leaf:
 * first = (c & 0x7fffffff) >> 3, count = (c & 7) + 1), pad[2]} and leaf-ordered render-space triangle records of 48 B
 * {p0 p1 p2 as 9 floats, pad[3]}.  Call with NULL buffers for the counts; *n_nodes / *n_tris hold the buffer capacities on
 * entry.  The parity tests hand the tree to the oracle, which walks it to check the instrumented kernel's step counts. */
int mi355pt_scene_export_bvh(const mi355pt_scene* s, void* out_nodes, uint32_t* n_nodes, void* out_tris, uint32_t* n_tris,
                             int32_t* root);
/* Host-only check of the acceleration structure the render path walks (no reference counterpart): builds the sweep-SAH BVH2 over the
 * n_tris triangles (9 floats each), collapses it to the 4-wide tree of the cooperative traversals and walks BOTH on the CPU for n_rays rays
 * (6 floats each: origin, direction): out_mismatch = rays for which the two trees reach different sets of leaves (must be 0);
 * out_info[4] = {BVH2 nodes, BVH4 nodes, BVH2 depth, worst-case stack entries of the BVH4 (< 24 by construction)}. */
int mi355pt_probe_bvh_collapse(const float* tri_pos, uint32_t n_tris, const float* rays_od, uint32_t n_rays, uint32_t* out_info,
                               uint32_t* out_mismatch);
/* The same collapse + validation on a caller-supplied BVH2 (n_nodes 64-byte records in the layout mi355pt_scene_export_bvh writes; leaves
 * are links < 0 holding first << 3 | count - 1): the guard SceneImpl::build runs before any tree reaches the device, reachable without a
 * device.  out_info = {nodes2, nodes4, 1 if the cost-optimal collapse ran (0: greedy), worst-case per-lane stack entries}; a tree the
 * 24-entry LDS stack cannot serve (too deep, broken links, triangles lost) is refused with MI355PT_E_INVALID.  Host only. */
int mi355pt_probe_bvh_collapse_nodes(const void* bvh2_nodes, uint32_t n_nodes, int32_t root, uint32_t n_tris, uint32_t* out_info /* 4 */);
/* ZSobolSampler: for each query (x, y, sample_index) emit n_dims raw 32-bit Sobol outputs following the draw
 * pattern string `pattern` of '1' (get_1d) and '2' (get_2d) characters.  z_sobol_sampler.rs:198-230 */
int mi355pt_probe_sobol(uint32_t width, uint32_t height, uint32_t spp, uint32_t seed, const uint32_t* xys /* n*3 */,
                        uint32_t n, const char* pattern, uint32_t* out_bits /* n * n_values(pattern) */);
/* sin / cos as the render kernels compute them (csrc/pt_libm.hpp: the host libm's algorithm restated; the reference calls f32::sin_cos,
 * e.g. scene/src/material/bsdf/dielectric.rs:77-112 via common.rs) for the n floats with bit patterns first_bits + i * stride, compared on
 * the host with sinf / cosf of THIS machine's libm: out_counts[3] = {compared, sin results that differ in any bit, cos results}. */
int mi355pt_probe_sincos(uint32_t first_bits, uint32_t stride, uint32_t n, uint64_t* out_counts /* 3 */);
/* Scene::intersect for n rays (render space).  out_t < 0 means miss.  scene.rs:80-90 */
int mi355pt_probe_intersect(const mi355pt_scene* s, const float* origins, const float* dirs, uint32_t n, float* out_t,
                            uint32_t* out_instance, uint32_t* out_triangle, float* out_normal /* n*3 geometric, NULL ok */);
/* Scene::intersect_p for n rays.  scene.rs:93-103 */
int mi355pt_probe_occluded(const mi355pt_scene* s, const float* origins, const float* dirs, const float* t_max, uint32_t n,
                           uint8_t* out_hit);
/* BaseSrgbRenderer::render's per-sample result before the sensor (base_renderer.rs:160-276: the L handed to
 * Sensor::add_sample with its SampledWavelengths): L[4], lambda[4], pdf[4] for every finished path of a render call,
 * written by the PRODUCTION kernel in the launch shape mi355pt_render_accum_device takes for the same arguments
 * (a wave-uniform branch at path end; nothing else differs).  Record r = (k * 64 + (y & 7) * 8 + (x & 7)) *
 * (sample_end - sample_begin) + (sample - sample_begin), k = position of the pixel's 8x8 tile among the tiles of the
 * shard (tile t = shard_index + k * shard_count, row-major tiles); records of pixels outside the frame stay zero.
 * Needs a power-of-two spp.  out_accum (host, W*H*3, NULL ok) receives the linear film sums of the same launch. */
int mi355pt_sample_log_records(const mi355pt_camera* cam, const mi355pt_params* p, uint32_t sample_begin, uint32_t sample_end,
                               size_t* out_records);
int mi355pt_render_sample_log(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p, uint32_t sample_begin,
                              uint32_t sample_end, float* out_L, float* out_lambda, float* out_pdf, size_t n_records,
                              float* out_accum);
/* The same records picked for n (x, y, sample) queries of the whole-frame, whole-job launch (small frames only). */
int mi355pt_probe_radiance(const mi355pt_scene* s, const mi355pt_camera* cam, const mi355pt_params* p, const uint32_t* xys,
                           uint32_t n, float* out_L, float* out_lambda, float* out_pdf);


#ifdef __cplusplus
}
#endif
#endif
