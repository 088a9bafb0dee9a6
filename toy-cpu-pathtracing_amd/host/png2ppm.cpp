// png2ppm in.png out.ppm — decodes with the host mirror's own PNG reader (renderer.hpp load_png); used by tests/test_host_png.py
// and handy for converting the reference's PNG assets (texture/loader.rs) once the LFS objects are available.
#include "renderer.hpp"

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: png2ppm in.png out.ppm\n"); return 2; }
    try {
        renderer::ImageRgb8 im = renderer::load_png(argv[1]);
        std::ofstream f(argv[2], std::ios::binary);
        f << "P6\n" << im.w << " " << im.h << "\n255\n";
        f.write((const char*)im.rgb.data(), (std::streamsize)im.rgb.size());
        return f ? 0 : 1;
    } catch (const std::exception& e) { std::fprintf(stderr, "png2ppm: %s\n", e.what()); return 1; }
}
