// renderer.hpp -- C++ host-side mirror of the reference's Arctic::Renderer::Renderer over the C-ABI.
//
// Same public surface as reference src/renderer/renderer.hpp:94-125 (init, cleanup, resize, render_frame,
// create_mesh, create_material, create_hdri, update_lights, flush) and the same POD scene types as
// src/renderer/scene.hpp:20-110, so src/app.cpp's calls compile against it unchanged in shape:
//   - bool-returning, [[nodiscard]], errors logged (here: kept in last_error()) -- src/renderer/dxerr.hpp:5-10
//   - glm types replaced by plain float arrays of the same bytes (glm is not a dependency of this library)
//   - no SDL window, no ImGui callback; render_frame hands back the RGBA8 frame instead of presenting it.
// Header only; link with -larctic_hip (arctic-renderer_amd/csrc/libarctic_hip.so).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <span>
#include <string>
#include <vector>

#include "../../include/arctic_hip.h"

namespace ArcticAMD::Renderer {

using MeshIdx = size_t;
using MaterialIdx = size_t;

// scene.hpp:20-110, same member names; vec3/vec2/mat4 as float arrays (glm memory order)
struct Camera { float eye[3]; float rotation[2]; float aspect; float fov_y; std::array<float, 2> z_near_far; };
using Vertex = ArcticVertex;               // position, normal, tangent, bitangent, tex_coords
using Object = ArcticObject;               // trs (column-major mat4), mesh_idx
struct DirectionalLight { float position[3]; float rotation[2]; float color[3]; };
using PointLight = ArcticPointLight;       // position, padding0, color, padding1
struct Scene {
    Camera camera;
    float ambient;
    DirectionalLight sun;
    std::vector<PointLight> point_lights;
    std::vector<Object> objects;
};
struct Settings { int tm_method{0}; float gamma{2.2f}; float exposure{1.0f}; };

class Renderer {
  public:
    static constexpr size_t MAX_NUM_POINT_LIGHTS = 16;   // renderer.hpp:22 (a create-time parameter here)
    static constexpr uint32_t SHADOW_MAP_SIZE = 4000;    // ShadowMapPass::SIZE, shadow_map_pass.hpp:23

    Renderer(uint32_t initial_width, uint32_t initial_height, uint32_t shadow_size = SHADOW_MAP_SIZE,
             uint32_t max_lights = MAX_NUM_POINT_LIGHTS, int device = 0)
        : m_info{initial_width, initial_height, shadow_size, max_lights, device, 0, 0} {}
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;
    ~Renderer() { cleanup(); }

    [[nodiscard]] bool init() {
        char err[512] = {0};
        m_handle = arctic_create(&m_info, err, sizeof err);
        if (!m_handle) m_error = err;
        return m_handle != nullptr;
    }
    void cleanup() { if (m_handle) { arctic_destroy(m_handle); m_handle = nullptr; } }

    [[nodiscard]] bool resize(uint32_t &out_width, uint32_t &out_height) {
        if (!ok(arctic_resize(m_handle, out_width, out_height))) return false;
        m_info.width = out_width; m_info.height = out_height;
        return true;
    }

    // out_rgba8: width*height*4 bytes, row-major, top-left origin (may be nullptr: the frame stays on the device)
    [[nodiscard]] bool render_frame(const Scene &scene, const Settings &settings, uint8_t *out_rgba8) {
        ArcticScene s{};
        std::memcpy(s.camera.eye, scene.camera.eye, sizeof s.camera.eye);
        std::memcpy(s.camera.rotation, scene.camera.rotation, sizeof s.camera.rotation);
        s.camera.aspect = scene.camera.aspect; s.camera.fov_y = scene.camera.fov_y;
        s.camera.z_near_far[0] = scene.camera.z_near_far[0]; s.camera.z_near_far[1] = scene.camera.z_near_far[1];
        s.ambient = scene.ambient;
        std::memcpy(&s.sun, &scene.sun, sizeof s.sun);
        s.point_lights = scene.point_lights.data(); s.n_point_lights = scene.point_lights.size();
        s.objects = scene.objects.data(); s.n_objects = scene.objects.size();
        ArcticSettings st{settings.tm_method, settings.gamma, settings.exposure};
        return ok(arctic_render_frame(m_handle, &s, &st, out_rgba8));
    }

    [[nodiscard]] bool create_mesh(std::span<Vertex> vertices, std::span<uint32_t> indices, MaterialIdx material_idx) {
        return ok(arctic_create_mesh(m_handle, vertices.data(), vertices.size(), indices.data(), indices.size(), material_idx));
    }
    [[nodiscard]] bool create_material(void *diffuse_data, uint32_t diffuse_width, uint32_t diffuse_height, void *normal_data,
                                       uint32_t normal_width, uint32_t normal_height, void *metalness_roughness_data,
                                       uint32_t metalness_roughness_width, uint32_t metalness_roughness_height) {
        return ok(arctic_create_material(m_handle, diffuse_data, diffuse_width, diffuse_height, normal_data, normal_width, normal_height,
                                         metalness_roughness_data, metalness_roughness_width, metalness_roughness_height));
    }
    [[nodiscard]] bool create_hdri(float *data, uint32_t width, uint32_t height) { return ok(arctic_create_hdri(m_handle, data, width, height)); }
    void update_lights(std::span<PointLight> point_lights) { (void)ok(arctic_update_lights(m_handle, point_lights.data(), point_lights.size())); }
    [[nodiscard]] bool flush() { return ok(arctic_flush(m_handle)); }

    const std::string &last_error() const { return m_error; }
    ArcticRenderer *handle() const { return m_handle; }

  private:
    bool ok(int rc) {
        if (rc >= 0) return true;
        m_error = m_handle ? arctic_last_error(m_handle) : "renderer not initialised";
        return false;
    }
    ArcticCreateInfo m_info;
    ArcticRenderer *m_handle = nullptr;
    std::string m_error;
};

}  // namespace ArcticAMD::Renderer
