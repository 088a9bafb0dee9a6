// exr2pfm in.exr out.pfm — decodes with the host mirror's own OpenEXR reader (renderer.hpp load_exr: scanline, NONE / ZIPS / ZIP, HALF / FLOAT
// R G B) and writes a little-endian colour PFM; used by tests/test_host_png.py and to look at the reference's sky (scene_19.rs) once the LFS
// object is available.
#include "renderer.hpp"

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: exr2pfm in.exr out.pfm\n"); return 2; }
    try {
        uint32_t w = 0, h = 0;
        std::vector<float> rgb = renderer::load_exr(argv[1], &w, &h);
        std::ofstream f(argv[2], std::ios::binary);
        f << "PF\n" << w << " " << h << "\n-1.0\n";
        for (uint32_t y = h; y-- > 0;) f.write((const char*)&rgb[(size_t)y * w * 3], (std::streamsize)((size_t)w * 3 * sizeof(float)));   // PFM rows run bottom to top
        return f ? 0 : 1;
    } catch (const std::exception& e) { std::fprintf(stderr, "exr2pfm: %s\n", e.what()); return 1; }
}
