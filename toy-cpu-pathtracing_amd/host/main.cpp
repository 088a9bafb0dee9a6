// mi355pt — command line mirroring renderer/src/main.rs:20-140 (clap flags, defaults and flow), with
// RendererImage::render running on the MI355X through libmi355pt.so instead of the rayon pixel loop.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "scenes.hpp"

using namespace renderer;

struct Args {   // main.rs:20-53
    uint32_t scene = 0, spp = 64, seed = 0, width = 800, height = 600, max_depth = 16;
    std::string filter = "box", sampler = "random", renderer = "normal", output = "output.png";
    // not in the reference's CLI: the coat-albedo table option (mi355pt_params.albedo_lut) and the number of GPUs of this node to shard the frame over
    bool albedo_lut = false; int gpus = 1;
};

static void usage() {
    std::puts("Usage: mi355pt [--scene N] [-s|--spp N] [--seed N] [--filter box] [--sampler random|sobol]\n"
              "               [--renderer normal|albedo|pt|nee|mis] [--width N] [--height N] [-d|--max-depth N] [-o|--output FILE]\n"
              "       extensions: [--albedo-lut] (clearcoat albedo from its table instead of the 64-sample estimate)  [--gpus N]");
}

int main(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        if (k == "--scene") a.scene = (uint32_t)std::stoul(val());
        else if (k == "-s" || k == "--spp") a.spp = (uint32_t)std::stoul(val());
        else if (k == "--seed") a.seed = (uint32_t)std::stoul(val());
        else if (k == "--filter") a.filter = val();
        else if (k == "--sampler") a.sampler = val();
        else if (k == "--renderer") a.renderer = val();
        else if (k == "--width") a.width = (uint32_t)std::stoul(val());
        else if (k == "--height") a.height = (uint32_t)std::stoul(val());
        else if (k == "-d" || k == "--max-depth") a.max_depth = (uint32_t)std::stoul(val());
        else if (k == "-o" || k == "--output") a.output = val();
        else if (k == "--albedo-lut") a.albedo_lut = true;
        else if (k == "--gpus") a.gpus = std::stoi(val());
        else if (k == "-h" || k == "--help") { usage(); return 0; }
        else { std::fprintf(stderr, "error: unexpected argument '%s'\n", k.c_str()); usage(); return 2; }
    }
    if (a.filter != "box") { std::fprintf(stderr, "error: invalid value '%s' for '--filter' (main.rs:33-37 offers only box)\n", a.filter.c_str()); return 2; }
    if (a.sampler != "random" && a.sampler != "sobol") { std::fprintf(stderr, "error: invalid value '%s' for '--sampler'\n", a.sampler.c_str()); return 2; }
    try {
        Camera camera(45.0f, a.width, a.height);                                        // main.rs:59-68
        Scene scene;
        switch (a.scene) {                                                              // main.rs:70-92
            case 0: load_scene_0(scene, camera); break;
            case 1: load_scene_1(scene, camera); break;
            case 2: load_scene_2(scene, camera); break;
            case 3: load_scene_3(scene, camera); break;
            case 4: load_scene_4(scene, camera); break;
            case 5: load_scene_5(scene, camera); break;
            case 9: load_scene_9(scene, camera); break;
            case 12: load_scene_12(scene, camera); break;
            case 13: load_scene_13(scene, camera); break;
            case 14: load_scene_14(scene, camera); break;
            case 15: load_scene_15(scene, camera); break;
            case 16: load_scene_16_18(scene, camera, false); break;
            case 18: load_scene_16_18(scene, camera, true); break;
            case 6: load_scene_6(scene, camera); break;
            case 7: load_scene_7(scene, camera); break;
            case 11: load_scene_11(scene, camera); break;
            case 8: load_scene_8(scene, camera); break;
            case 10: load_scene_10(scene, camera); break;
            case 17: load_scene_17(scene, camera); break;
            case 19: load_scene_19(scene, camera); break;
            default: std::fprintf(stderr, "scene %u is outside the MI355X hot-path scope (scenes 0-19)\n", a.scene); return 2;
        }
        std::puts("Start build scene.");                                                // main.rs:103-109
        auto t0 = std::chrono::steady_clock::now();
        if (a.gpus > 1) scene.build_multi(camera, a.gpus); else scene.build(camera);
        std::printf("Finish build scene: %.3f seconds.\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());

        RendererArgs args{a.width, a.height, a.spp, a.seed, &scene, &camera};
        SrgbRenderer r;
        if (a.renderer == "pt") r = SrgbRendererPt(args, 1.0f, a.max_depth);            // main.rs:188-233: exposure 1.0, Reinhard
        else if (a.renderer == "nee") r = SrgbRendererNee(args, 1.0f, a.max_depth);
        else if (a.renderer == "mis") r = SrgbRendererMis(args, 1.0f, a.max_depth);
        else { std::fprintf(stderr, "renderer '%s' (AOV) is outside the MI355X hot-path scope: use pt, nee or mis\n", a.renderer.c_str()); return 2; }
        RendererImage image(a.width, a.height, r);
        std::puts("Start rendering...");                                                // main.rs:166-172
        t0 = std::chrono::steady_clock::now();
        double kernel_s = image.render(a.sampler == "sobol" ? SamplerKind::ZSobol : SamplerKind::Random, a.albedo_lut);
        double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("Finish rendering: %.3f seconds.\n", wall);
        if (kernel_s > 0.0) std::printf("(device %.3f s, %.1f Msamples/s)\n", kernel_s, (double)a.width * a.height * a.spp / kernel_s / 1e6);
        image.save(a.output);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "mi355pt: %s\n", e.what());
        return 1;
    }
    return 0;
}
