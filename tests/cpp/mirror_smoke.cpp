// mirror_smoke.cpp -- drives the C++ Renderer mirror the way src/app.cpp drives the reference class:
// create_material / create_mesh / update_lights at load time, render_frame per frame.  Prints a checksum of the frame.
// Built and run by tests/test_cpp_mirror.py (g++ -std=c++20, links libarctic_hip.so).
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../arctic-renderer_amd/host/renderer.hpp"

using namespace ArcticAMD::Renderer;

int main(int argc, char **argv) {
    const uint32_t W = 160, H = 96;
    Renderer renderer(W, H, 256, 16);
    if (!renderer.init()) { std::printf("init failed: %s\n", renderer.last_error().c_str()); return argc > 1 ? 0 : 2; }
    std::vector<uint8_t> white(16 * 16 * 4, 255), flat(16 * 16 * 4);
    for (size_t i = 0; i < flat.size(); i += 4) { flat[i] = 128; flat[i + 1] = 128; flat[i + 2] = 255; flat[i + 3] = 255; }
    std::vector<uint8_t> grey(16 * 16 * 4);
    for (size_t i = 0; i < grey.size(); i += 4) { grey[i] = 255; grey[i + 1] = 140; grey[i + 2] = 0; grey[i + 3] = 255; }   // rough 0.55, metal 0
    if (!renderer.create_material(white.data(), 16, 16, flat.data(), 16, 16, grey.data(), 16, 16)) return 3;
    // a floor quad and a raised quad casting a shadow on it, both facing +y
    auto quad = [](float x0, float z0, float x1, float z1, float y) {
        std::vector<Vertex> v(4);
        const float p[4][3] = {{x0, y, z1}, {x1, y, z1}, {x1, y, z0}, {x0, y, z0}};
        for (int i = 0; i < 4; ++i) {
            v[i] = Vertex{{p[i][0], p[i][1], p[i][2]}, {0, 1, 0}, {1, 0, 0}, {0, 0, -1}, {p[i][0], p[i][2]}};
        }
        return v;
    };
    std::vector<uint32_t> idx = {0, 1, 2, 0, 2, 3};
    auto floor = quad(-6, -6, 6, 6, 0.0f), plate = quad(-1, -1, 1, 1, 1.0f);
    if (!renderer.create_mesh(floor, idx, 0) || !renderer.create_mesh(plate, idx, 0)) return 4;
    std::vector<PointLight> lights = {PointLight{{0.0f, 1.0f, 0.0f}, 0, {10.0f, 0.0f, 0.0f}, 0}};   // src/app.hpp:57-60
    renderer.update_lights(lights);
    Scene scene{};
    scene.camera = Camera{{0.0f, 5.0f, 8.0f}, {-30.0f, -90.0f}, float(W) / float(H), 45.0f, {0.1f, 1000.0f}};
    scene.ambient = 0.1f;
    scene.sun = DirectionalLight{{-10.0f, 32.0f, -2.48f}, {-70.0f, 12.0f}, {8.0f, 8.0f, 8.0f}};       // src/app.hpp:51-55
    Object o{};
    for (int i = 0; i < 4; ++i) o.trs[i * 5] = 1.0f;
    o.mesh_idx = 0; scene.objects.push_back(o);
    o.mesh_idx = 1; scene.objects.push_back(o);
    std::vector<uint8_t> frame(size_t(W) * H * 4);
    if (!renderer.render_frame(scene, Settings{}, frame.data())) { std::printf("render_frame failed: %s\n", renderer.last_error().c_str()); return 5; }
    if (!renderer.flush()) return 6;
    unsigned long long sum = 0, lit = 0, alpha_ok = 1;
    for (size_t i = 0; i < frame.size(); i += 4) { sum += frame[i] + frame[i + 1] + frame[i + 2]; lit += frame[i] > 100; alpha_ok &= frame[i + 3] == 255; }
    std::printf("frame %ux%u checksum %llu lit %llu alpha %llu\n", W, H, sum, lit, alpha_ok);
    return (sum > 0 && lit > 0 && alpha_ok) ? 0 : 7;
}
