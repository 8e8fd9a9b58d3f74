// host_math.cpp -- the per-frame constants the host builds before the kernels run.
//
// Replaces the glm 1.0.1 calls in the reference's src/renderer/scene.cpp:9-19 (dir_from_rot),
// :41-54 (Camera::proj_view_matrix = perspectiveRH_ZO * lookAtRH) and :56-70
// (DirectionalLight::direction / proj_view_matrix = orthoRH_ZO(-16,16,-16,16,0.1,50) * lookAtRH),
// which ForwardPass::run evaluates each frame (forward_pass.cpp:166-177).  glm is not vendored
// (reference CMakeLists.txt:99-106); the formulas are glm's published ones, evaluated in fp32 in
// glm's operation order (compiled with -ffp-contract=off so each operation rounds once).
// Matrices are 16 floats, m[col*4 + row] (glm memory order).
#include <cmath>

#include "common.h"

namespace arctic {

namespace {

inline float deg2rad(float d) { return d * 0.01745329251994329576923690768489f; }  // glm::radians

inline float dot3(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

inline void cross3(const float *x, const float *y, float *o) {
    o[0] = x[1] * y[2] - y[1] * x[2];
    o[1] = x[2] * y[0] - y[2] * x[0];
    o[2] = x[0] * y[1] - y[0] * x[1];
}

// glm::normalize: v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
inline void normalize3(float *v) {
    float inv = 1.0f / std::sqrt(dot3(v, v));
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

// glm::lookAtRH(eye, center, up)
void look_at_rh(const float *eye, const float *center, const float *up, float *m) {
    float f[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
    normalize3(f);
    float s[3];
    cross3(f, up, s);
    normalize3(s);
    float u[3];
    cross3(s, f, u);
    m[0] = s[0];  m[4] = s[1];  m[8] = s[2];   m[12] = -dot3(s, eye);
    m[1] = u[0];  m[5] = u[1];  m[9] = u[2];   m[13] = -dot3(u, eye);
    m[2] = -f[0]; m[6] = -f[1]; m[10] = -f[2]; m[14] = dot3(f, eye);
    m[3] = 0.0f;  m[7] = 0.0f;  m[11] = 0.0f;  m[15] = 1.0f;
}

// glm mat4 * mat4: column j of the result = a.col0*b[j][0] + a.col1*b[j][1] + a.col2*b[j][2] + a.col3*b[j][3]
void mat_mul(const float *a, const float *b, float *o) {
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            o[j * 4 + i] = ((a[i] * b[j * 4] + a[4 + i] * b[j * 4 + 1]) + a[8 + i] * b[j * 4 + 2]) + a[12 + i] * b[j * 4 + 3];
}

}  // namespace

// scene.cpp:9-19
void dir_from_rot(const float rot_deg[2], float out[3]) {
    float xr = deg2rad(rot_deg[0]), yr = deg2rad(rot_deg[1]);
    out[0] = std::cos(xr) * std::cos(yr);
    out[1] = std::sin(xr);
    out[2] = std::cos(xr) * std::sin(yr);
}

// scene.cpp:41-54: perspectiveRH with GLM_FORCE_DEPTH_ZERO_TO_ONE (CMakeLists.txt:150) times lookAtRH(eye, eye+forward, +Y)
void camera_proj_view(const float eye[3], const float rot_deg[2], float aspect, float fov_y_deg, float zn, float zf, float out[16]) {
    float fwd[3];
    dir_from_rot(rot_deg, fwd);
    float center[3] = {eye[0] + fwd[0], eye[1] + fwd[1], eye[2] + fwd[2]};
    const float up[3] = {0.0f, 1.0f, 0.0f};
    float view[16];
    look_at_rh(eye, center, up, view);
    float t = std::tan(deg2rad(fov_y_deg) / 2.0f);
    float proj[16] = {0};
    proj[0] = 1.0f / (aspect * t);
    proj[5] = 1.0f / t;
    proj[10] = zf / (zn - zf);
    proj[11] = -1.0f;
    proj[14] = -(zf * zn) / (zf - zn);
    mat_mul(proj, view, out);
}

// scene.cpp:61-70: orthoRH_ZO(-16, 16, -16, 16, 0.1, 50) * lookAtRH(pos, pos+dir, +Y)
void sun_proj_view(const float pos[3], const float rot_deg[2], float out[16]) {
    float fwd[3];
    dir_from_rot(rot_deg, fwd);
    float center[3] = {pos[0] + fwd[0], pos[1] + fwd[1], pos[2] + fwd[2]};
    const float up[3] = {0.0f, 1.0f, 0.0f};
    float view[16];
    look_at_rh(pos, center, up, view);
    const float l = -16.0f, r = 16.0f, b = -16.0f, tp = 16.0f, zn = 0.1f, zf = 50.0f;
    float proj[16] = {0};
    proj[0] = 2.0f / (r - l);
    proj[5] = 2.0f / (tp - b);
    proj[10] = -1.0f / (zf - zn);
    proj[12] = -(r + l) / (r - l);
    proj[13] = -(tp + b) / (tp - b);
    proj[14] = -zn / (zf - zn);
    proj[15] = 1.0f;
    mat_mul(proj, view, out);
}

// SkyboxPass (skybox_pass.cpp:104-138, skybox.hlsl:61-70): the cube is drawn with proj * mat3(lookAtRH) and its
// interpolated object-space position is the lookup direction, i.e. the world-space ray through the pixel.  That ray for
// ndc (x, y) is fwd + x * right + y * up with the camera basis of lookAtRH scaled by the frustum half-extents.
void camera_sky_basis(const float rot_deg[2], float aspect, float fov_y_deg, float fwd[3], float right[3], float up[3]) {
    dir_from_rot(rot_deg, fwd);
    const float wup[3] = {0.0f, 1.0f, 0.0f};
    float f[3] = {fwd[0], fwd[1], fwd[2]};
    normalize3(f);
    float s[3], u[3];
    cross3(f, wup, s);
    normalize3(s);
    cross3(s, f, u);
    const float t = std::tan(deg2rad(fov_y_deg) / 2.0f), tx = aspect * t;
    for (int i = 0; i < 3; ++i) { fwd[i] = f[i]; right[i] = s[i] * tx; up[i] = u[i] * t; }
}

// R8G8B8A8_UNORM_SRGB decode of one channel (renderer.cpp:483-505), IEC 61966-2-1
float srgb8_to_linear(int c) {
    float x = (float)c / 255.0f;
    return x <= 0.04045f ? x / 12.92f : std::pow((x + 0.055f) / 1.055f, 2.4f);
}

}  // namespace arctic
