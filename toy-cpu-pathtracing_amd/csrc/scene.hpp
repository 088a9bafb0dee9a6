// Host-side scene: stores what the C ABI hands over, lowers it to the flat HBM layout of layout.hpp
// (single render-space BVH over all instances, leaf-ordered triangles, per-triangle shading records,
// light tables) and owns the device buffers.  Replaces Scene::build (scene/src/scene.rs:64-76).
#pragma once
#include <string>
#include <vector>

#include "../../include/mi355pt.h"
#include "layout.hpp"

namespace pt {

struct HostMesh {
    std::vector<float> pos, nrm, uv, tangent;   // nrm normalised twice like Normal::new (normal.rs:18-20,93-100)
    std::vector<uint32_t> idx;
    uint32_t n_vert = 0, n_tri = 0;
};
struct HostInstance { uint32_t geom, mat; float l2w[16]; };
struct HostDeltaLight { mi355pt_light_desc d; uint32_t material; uint32_t after_instances; uint32_t env_index = 0; };   // hidden emissive material holds the spectrum

struct BuildTri { float lo[3], hi[3], c[3]; };
struct BvhOut {
    std::vector<DevNode> nodes;
    std::vector<uint32_t> order;   // leaf-ordered triangle permutation
    int32_t root = 0;
    int max_depth = 0;
};
// Sweep-SAH BVH2 over triangle bounds; children boxes stored in the parent (layout.hpp DevNode).
void build_bvh(const std::vector<BuildTri>& tris, BvhOut* out);
// BVH2 -> BVH4 for the cooperative traversals (layout.hpp DevNode4), with the stack-need guarantee validated; host only (bvh_builder.cpp)
bool collapse_bvh4(const std::vector<DevNode>& nodes2, int32_t root2, size_t n_tris, std::vector<DevNode4>* nodes4, int32_t* root4, int* max_stack,
                   std::string* err, const char** method = nullptr);   // *method: "dp" (cost-optimal) or "greedy" (above 1.2 M nodes, or no memory for the tables)
// SAH constants shared by both builders (env overrides MI355PT_BVH_COST_TRI / MI355PT_BVH_LEAF are for sweeps only)
void bvh_build_config(float* cost_traverse, float* cost_tri, int* leaf_max);
// The same contract built on the current HIP device (bvh_gpu.hip): breadth-first binned SAH, one round of launches
// per level.  Returns false with *err set when it cannot build (n < 8, out of memory, HIP error).
bool build_bvh_gpu(const std::vector<BuildTri>& tris, BvhOut* out, double* device_ms, std::string* err);

constexpr size_t BVH_GPU_AUTO_TRIS = 1u << 17;   // "auto": scenes from 131 072 triangles on are built on the GPU

struct DeviceBuffers {
    void* ptrs[16] = {nullptr};
    int n = 0;
};

struct SceneImpl {
    // ---- description ----
    std::vector<float> table;                 // reference layout [64][3][64][64][64][3]
    std::vector<std::vector<float>> luts;
    struct Tex { std::vector<uint8_t> rgb; uint32_t w, h; };
    std::vector<Tex> textures;
    std::vector<HostMesh> meshes;
    std::vector<mi355pt_material_desc> mat_descs;
    std::vector<DevMaterial> materials;
    std::vector<HostInstance> instances;
    struct HostEnv { float intensity = 1.0f; uint32_t w = 0, h = 0, illuminant_lut = 0; std::vector<float> rgb; float l2w[16]; };
    std::vector<HostEnv> envs;                  // environment lights in creation order (HostDeltaLight::d.angle_inner carries the index)
    std::vector<HostDeltaLight> delta_lights;   // creation order; after_instances = instances.size() at creation (light_sampler.rs:163-180)
    // ---- lowered ----
    bool built = false;
    int device = -1;              // the HIP device build() uploaded to: render calls must run with it current
    float build_cam_pos[3] = {0, 0, 0};   // camera position baked into the render-space records (world -> render translation)
    DevScene dev{};
    std::vector<void*> allocs;
    uint32_t cmf_lut[3] = {0, 0, 0};
    int bvh_depth = 0;
    size_t bvh4_nodes = 0;        // nodes of the collapsed tree (DevNode4)
    int bvh_builder = 0;          // MI355PT_BVH_AUTO / _HOST / _GPU (mi355pt_scene_set_bvh_builder)
    int bvh_builder_used = 1;     // what build() took
    size_t n_degenerate = 0;      // triangles with an exactly zero cross product: never hit (ray.rs:49-56), left out of the tree
    int lowering = 0;             // mi355pt_scene_debug_set_lowering: 0 auto, 1 never the local triangle array, 2 also every instance through the full matrix path
    double collapse_ms = 0.0;     // host time of the 2-wide -> 4-wide collapse
    const char* collapse_method = "";   // "dp" or "greedy" (scene_info)
    int bvh4_stack_need = 0;      // worst-case per-lane stack entries the collapsed tree can need (< STACK_DEPTH, validated)
    double bvh_build_ms = 0.0;    // wall time of the BVH build inside build(); bvh_device_ms: device part of a GPU build
    double bvh_device_ms = 0.0;
    uint32_t features = FEAT_ALL;   // FEAT_* bits the scene's materials need (kernel specialisation)
    std::string info;

    ~SceneImpl();
    void release();
    // RgbSigmoidPolynomial::from(ColorSrgb) on the host (rgb_sigmoid_polynomial.rs:87-155)
    bool table_lookup_srgb(const float rgb_encoded[3], float c[3], bool linear = false) const;
    int lower_spectrum(const mi355pt_spectrum& in, DevSpectrum* out, int allow_texture /* 0 no, 1 Albedo type, 2 every SpectrumType */, std::string* err) const;
    int build(const mi355pt_camera* cam, const float* cmf_xyz /*3*470*/, std::string* err);
};

// mi355pt_coat_albedo_table (scene.cpp): E(cos theta_o) of the clearcoat's directional-albedo estimator, 64 entries
void coat_albedo_table(float alpha, float r0, float out[64]);

// baked CIE 1931 colour matching functions shipped with the library (data/presets470.bin rows cie_x/y/z)
const float* builtin_cmf_xyz();   // 3*470 floats or nullptr if not loaded
bool load_builtin_cmf(std::string* err);

}  // namespace pt
