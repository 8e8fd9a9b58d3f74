// Sanitizer driver for the scene-loader stand-in (image and glTF files are untrusted input).
// Built by tests/test_gltf_malformed.py with -fsanitize=address,undefined from host/gltf_loader.cpp itself (the two renderer
// entry points arctic_gltf_upload needs are stubbed: no HIP library, CPU only).  Every argument is a file: *.gltf / *.glb go
// through arctic_gltf_load, everything else through arctic_png_decode (PNG or JPEG by signature).  A malformed file must
// come back as an error string, never as a sanitizer report or a crash; prints one line per file.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/arctic_gltf.h"

extern "C" {
int arctic_create_material(ArcticRenderer *, const void *, uint32_t, uint32_t, const void *, uint32_t, uint32_t, const void *, uint32_t, uint32_t) { return -1; }
int arctic_create_mesh(ArcticRenderer *, const ArcticVertex *, uint64_t, const uint32_t *, uint64_t, uint64_t) { return -1; }
}

int main(int argc, char **argv) {
    int loaded = 0, refused = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string path = argv[i];
        char err[512] = "";
        const bool scene = path.size() > 5 && (path.rfind(".gltf") == path.size() - 5 || path.rfind(".glb") == path.size() - 4);
        bool ok;
        if (scene) {
            ArcticGltf *g = arctic_gltf_load(path.c_str(), err, sizeof err);
            ok = g != nullptr;
            arctic_gltf_free(g);
        } else {
            std::ifstream f(path, std::ios::binary);
            std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            uint32_t w = 0, h = 0;
            uint8_t *px = arctic_png_decode(data.data(), data.size(), &w, &h, err, sizeof err);
            ok = px != nullptr;
            arctic_png_free(px);
        }
        (ok ? loaded : refused)++;
        std::printf("%s %s %s\n", ok ? "ok     " : "refused", path.c_str(), err);
    }
    std::printf("loaded %d refused %d\n", loaded, refused);
    return 0;
}
