// arctic_oracle.cpp -- CPU ORACLE for the forward PBR shading path.
//
// *** TEST INFRASTRUCTURE, NOT PRODUCT. ***  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library.  The product
// (arctic-renderer_amd/csrc -> libarctic_hip.so) never links, loads or calls it.
//
// What it is: a scalar FP32 C++17 restatement (g++ -O2 -ffp-contract=off, no
// fast-math) of the per-pixel work of the reference's shaders/forward.hlsl and
// shaders/post_process.hlsl, plus the fixed-function steps D3D12 performs
// around them that the reference never spells out in code (vertex fetch,
// clipping, rasterisation with the D3D top-left rule, perspective-correct
// interpolation, bilinear/WRAP/sRGB texture sampling, float->UNORM8 store),
// plus the glm 1.0.1 matrix builders the host side feeds it
// (src/renderer/scene.cpp).  Every function cites the reference lines it follows.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
// (SURVEY.md section 4 / 8c), cannot be built or run here (Windows + D3D12 + DXC,
// every dependency fetched at configure time), and glm is not on this machine.
// The oracle is therefore pinned only by (a) hand-derived known-answer tests
// K1..K10 of SURVEY.md 8(c), re-derived independently in float64 numpy in
// tests/test_oracle_kat.py, and (b) its own committed golden renders under
// tests/golden/.  Where D3D leaves behaviour to the hardware (filter weight
// precision, interpolation arithmetic, pow/rsqrt accuracy) the choice made
// here is stated next to the code.
//
// Operation order matters: the HIP prepass is required to be BIT-EXACT against
// this file (integer coverage + identically ordered IEEE fp32 ops), so every
// expression below is written as an explicit sequence of single roundings;
// fmaf() appears only where named.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

// ----------------------------------------------------------------------------
// POD types: same bytes as include/arctic_hip.h / reference src/renderer/scene.hpp
// ----------------------------------------------------------------------------
struct Camera { float eye[3]; float rotation[2]; float aspect; float fov_y; float z_near_far[2]; };
struct Vertex { float position[3], normal[3], tangent[3], bitangent[3], tex_coords[2]; };
struct Object { float trs[16]; uint64_t mesh_idx; };
struct DirectionalLight { float position[3]; float rotation[2]; float color[3]; };
struct PointLight { float position[3]; uint32_t pad0; float color[3]; uint32_t pad1; };
struct Scene {
    Camera camera; float ambient; DirectionalLight sun;
    const PointLight *point_lights; uint64_t n_point_lights;
    const Object *objects; uint64_t n_objects;
};
struct Settings { int32_t tm_method; float gamma; float exposure; };

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct M4 { float c[4][4]; };  // c[col][row], glm memory order

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 scale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 divs(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
// dot: products summed left to right, one rounding each (glm compute_dot<vec3>:
// tmp = a*b; tmp.x + tmp.y + tmp.z; HLSL dot3 is the same sum up to contraction)
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 x, V3 y) {
    return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
// normalize = v * (1/sqrt(dot(v,v))): glm::normalize is v * inversesqrt(dot);
// HLSL normalize lowers to dot -> rsqrt -> mul.  IEEE sqrt and divide here.
inline V3 normalize(V3 v) { float inv = 1.0f / std::sqrt(dot(v, v)); return scale(v, inv); }
inline float lerp1(float a, float b, float t) { return std::fmaf(t, b - a, a); }  // HLSL lerp: a + t*(b-a), as one mad
inline float clamp01(float x) { return std::fmin(std::fmax(x, 0.0f), 1.0f); }

// ----------------------------------------------------------------------------
// glm 1.0.1 restatement (third-party, not vendored: reference CMakeLists.txt:99-106)
// call sites: src/renderer/scene.cpp:9-19, 41-54, 56-70
// ----------------------------------------------------------------------------
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// scene.cpp:9-19 dir_from_rot
V3 dir_from_rot(const float rot_deg[2]) {
    float xr = radians(rot_deg[0]), yr = radians(rot_deg[1]);
    return v3(std::cos(xr) * std::cos(yr), std::sin(xr), std::cos(xr) * std::sin(yr));
}

// glm::lookAtRH
M4 look_at_rh(V3 eye, V3 center, V3 up) {
    V3 f = normalize(sub(center, eye));
    V3 s = normalize(cross(f, up));
    V3 u = cross(s, f);
    M4 m{};
    m.c[0][0] = s.x; m.c[1][0] = s.y; m.c[2][0] = s.z;
    m.c[0][1] = u.x; m.c[1][1] = u.y; m.c[2][1] = u.z;
    m.c[0][2] = -f.x; m.c[1][2] = -f.y; m.c[2][2] = -f.z;
    m.c[3][0] = -dot(s, eye); m.c[3][1] = -dot(u, eye); m.c[3][2] = dot(f, eye);
    m.c[0][3] = 0.0f; m.c[1][3] = 0.0f; m.c[2][3] = 0.0f; m.c[3][3] = 1.0f;
    return m;
}

// glm::perspectiveRH_ZO (GLM_FORCE_DEPTH_ZERO_TO_ONE, CMakeLists.txt:150)
M4 perspective_rh_zo(float fovy, float aspect, float zn, float zf) {
    float t = std::tan(fovy / 2.0f);
    M4 m{};
    m.c[0][0] = 1.0f / (aspect * t);
    m.c[1][1] = 1.0f / t;
    m.c[2][2] = zf / (zn - zf);
    m.c[2][3] = -1.0f;
    m.c[3][2] = -(zf * zn) / (zf - zn);
    return m;
}

// glm::orthoRH_ZO
M4 ortho_rh_zo(float l, float r, float b, float t, float zn, float zf) {
    M4 m{};
    m.c[0][0] = 2.0f / (r - l);
    m.c[1][1] = 2.0f / (t - b);
    m.c[2][2] = -1.0f / (zf - zn);
    m.c[3][0] = -(r + l) / (r - l);
    m.c[3][1] = -(t + b) / (t - b);
    m.c[3][2] = -zn / (zf - zn);
    m.c[3][3] = 1.0f;
    return m;
}

// glm mat4*mat4: Result[j] = A[0]*B[j][0] + A[1]*B[j][1] + A[2]*B[j][2] + A[3]*B[j][3]
M4 mat_mul(const M4 &a, const M4 &b) {
    M4 r{};
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.c[j][i] = ((a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1]) + a.c[2][i] * b.c[j][2]) + a.c[3][i] * b.c[j][3];
    return r;
}

// HLSL mul(M, v), M packed column-major (DXC default, compiler.cpp:40-49):
// row i of the result = sum_j M[j][i]*v[j], accumulated left to right
inline V4 mat_vec(const M4 &m, V4 v) {
    V4 r;
    r.x = ((m.c[0][0] * v.x + m.c[1][0] * v.y) + m.c[2][0] * v.z) + m.c[3][0] * v.w;
    r.y = ((m.c[0][1] * v.x + m.c[1][1] * v.y) + m.c[2][1] * v.z) + m.c[3][1] * v.w;
    r.z = ((m.c[0][2] * v.x + m.c[1][2] * v.y) + m.c[2][2] * v.z) + m.c[3][2] * v.w;
    r.w = ((m.c[0][3] * v.x + m.c[1][3] * v.y) + m.c[2][3] * v.z) + m.c[3][3] * v.w;
    return r;
}

// scene.cpp:41-54 Camera::proj_view_matrix
M4 camera_proj_view(const Camera &c) {
    V3 eye = v3(c.eye[0], c.eye[1], c.eye[2]);
    V3 fwd = dir_from_rot(c.rotation);
    M4 view = look_at_rh(eye, add(eye, fwd), v3(0.0f, 1.0f, 0.0f));
    M4 proj = perspective_rh_zo(radians(c.fov_y), c.aspect, c.z_near_far[0], c.z_near_far[1]);
    return mat_mul(proj, view);
}

// scene.cpp:61-70 DirectionalLight::proj_view_matrix
M4 sun_proj_view(const DirectionalLight &s) {
    V3 pos = v3(s.position[0], s.position[1], s.position[2]);
    V3 fwd = dir_from_rot(s.rotation);
    M4 view = look_at_rh(pos, add(pos, fwd), v3(0.0f, 1.0f, 0.0f));
    M4 proj = ortho_rh_zo(-16.0f, 16.0f, -16.0f, 16.0f, 0.1f, 50.0f);
    return mat_mul(proj, view);
}

// ----------------------------------------------------------------------------
// vertex stage: forward.hlsl:50-66 vs_main (depth.hlsl:7-10 for the shadow pass)
// ----------------------------------------------------------------------------
struct VSOut {
    V4 clip;        // SV_POSITION
    float attr[18]; // uv2, tbn9 (t,b,n = the columns of transpose(float3x3(t,b,n))), world3, light4
};

VSOut vs_main(const Vertex &v, const M4 &model, const M4 &proj_view, const M4 &light_proj_view) {
    V4 world = mat_vec(model, V4{v.position[0], v.position[1], v.position[2], 1.0f});
    V3 t = normalize(v3(v.tangent[0], v.tangent[1], v.tangent[2]));
    V3 n = normalize(v3(v.normal[0], v.normal[1], v.normal[2]));
    V3 b = normalize(v3(v.bitangent[0], v.bitangent[1], v.bitangent[2]));
    VSOut o;
    o.clip = mat_vec(proj_view, world);
    V4 ls = mat_vec(light_proj_view, world);
    o.attr[0] = v.tex_coords[0]; o.attr[1] = v.tex_coords[1];
    o.attr[2] = t.x; o.attr[3] = t.y; o.attr[4] = t.z;
    o.attr[5] = b.x; o.attr[6] = b.y; o.attr[7] = b.z;
    o.attr[8] = n.x; o.attr[9] = n.y; o.attr[10] = n.z;
    o.attr[11] = world.x; o.attr[12] = world.y; o.attr[13] = world.z;
    o.attr[14] = ls.x; o.attr[15] = ls.y; o.attr[16] = ls.z; o.attr[17] = ls.w;
    return o;
}

// ----------------------------------------------------------------------------
// clipping + triangle setup + rasterisation.  D3D12 does this in fixed function
// for forward_pass.cpp:137-151 (back-face cull, CCW front, depth LESS, depth
// clip on) and shadow_map_pass.cpp:96-97 (front-face cull).  Rules restated
// from the D3D11.3 functional spec: pixel centres at +0.5, 8 sub-pixel bits,
// top-left fill rule, z/w linear in screen space, attributes perspective-correct.
// ----------------------------------------------------------------------------
constexpr float GUARD = 64.0f;      // guard band: |x|,|y| <= GUARD*w (keeps snapped coordinates in int32)
constexpr int MAX_POLY = 10;

struct ClipVert { V4 p; float bary[3]; };

inline float plane_dist(const ClipVert &v, int plane) {
    switch (plane) {
    case 0: return v.p.z;                    // near: z >= 0
    case 1: return v.p.w - v.p.z;            // far:  z <= w
    case 2: return v.p.x + GUARD * v.p.w;    // x >= -G w
    case 3: return GUARD * v.p.w - v.p.x;    // x <=  G w
    case 4: return v.p.y + GUARD * v.p.w;
    default: return GUARD * v.p.w - v.p.y;
    }
}

// intersection is always computed from the inside vertex towards the outside one
inline ClipVert clip_lerp(const ClipVert &in, const ClipVert &out, float din, float dout) {
    float t = din / (din - dout);
    ClipVert r;
    r.p.x = std::fmaf(t, out.p.x - in.p.x, in.p.x);
    r.p.y = std::fmaf(t, out.p.y - in.p.y, in.p.y);
    r.p.z = std::fmaf(t, out.p.z - in.p.z, in.p.z);
    r.p.w = std::fmaf(t, out.p.w - in.p.w, in.p.w);
    for (int k = 0; k < 3; ++k) r.bary[k] = std::fmaf(t, out.bary[k] - in.bary[k], in.bary[k]);
    return r;
}

// Sutherland-Hodgman against the 6 planes; returns vertex count (0 = fully clipped)
int clip_polygon(ClipVert *poly, int n) {
    ClipVert tmp[MAX_POLY];
    for (int plane = 0; plane < 6; ++plane) {
        float d[MAX_POLY];
        bool all_in = true, any_in = false;
        for (int i = 0; i < n; ++i) {
            d[i] = plane_dist(poly[i], plane);
            if (d[i] >= 0.0f) any_in = true; else all_in = false;
        }
        if (all_in) continue;
        if (!any_in) return 0;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            int j = (i + 1 == n) ? 0 : i + 1;
            bool in_i = d[i] >= 0.0f, in_j = d[j] >= 0.0f;
            if (in_i) tmp[m++] = poly[i];
            if (in_i != in_j) {
                tmp[m++] = in_i ? clip_lerp(poly[i], poly[j], d[i], d[j]) : clip_lerp(poly[j], poly[i], d[j], d[i]);
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) poly[i] = tmp[i];
        if (n < 3) return 0;
    }
    return n;
}

struct SetupTri {
    int32_t X[3], Y[3];   // 24.8 fixed-point screen position, oriented so area2 > 0
    float z[3], iw[3];    // z/w and 1/w per vertex
    float bary[3][3];     // each vertex as a combination of the source triangle's vertices
    int64_t area2;
    int32_t px0, py0, px1, py1;  // inclusive pixel bounds, already scissored
    uint32_t src_tri;     // draw-order id of the source triangle
    uint32_t object;
};

enum CullMode { CULL_BACK = 0, CULL_FRONT = 1 };

inline int32_t snap(float s) { return (int32_t)std::floor(s * 256.0f + 0.5f); }
inline int32_t ceil_div256(int32_t a) { return (a + 255) >> 8; }  // arithmetic shift = floor
inline int32_t floor_div256(int32_t a) { return a >> 8; }

// returns false if culled
bool setup_triangle(const ClipVert v[3], float vw, float vh, int sc_x0, int sc_y0, int sc_x1, int sc_y1,
                    CullMode cull, SetupTri &t) {
    float hx = 0.5f * vw, hy = 0.5f * vh;
    int32_t X[3], Y[3]; float z[3], iw[3];
    for (int i = 0; i < 3; ++i) {
        if (!(v[i].p.w > 0.0f)) return false;   // degenerate (w = 0 survives clipping only when z = w = 0)
        iw[i] = 1.0f / v[i].p.w;
        float nx = v[i].p.x * iw[i], ny = v[i].p.y * iw[i];
        z[i] = v[i].p.z * iw[i];
        float sx = (nx + 1.0f) * hx;    // D3D viewport: X = (x+1) * W/2
        float sy = (1.0f - ny) * hy;    //               Y = (1-y) * H/2
        X[i] = snap(sx); Y[i] = snap(sy);
    }
    int64_t area2 = (int64_t)(X[1] - X[0]) * (int64_t)(Y[2] - Y[0]) - (int64_t)(X[2] - X[0]) * (int64_t)(Y[1] - Y[0]);
    // y-down screen: area2 > 0 <=> clockwise as seen; front faces are counter-clockwise
    // (FrontCounterClockwise = TRUE, forward_pass.cpp:143-144) <=> area2 < 0.
    if (area2 == 0) return false;
    bool front = area2 < 0;
    if (cull == CULL_BACK && !front) return false;
    if (cull == CULL_FRONT && front) return false;
    int o[3] = {0, 1, 2};
    if (area2 < 0) { o[1] = 2; o[2] = 1; area2 = -area2; }
    for (int i = 0; i < 3; ++i) {
        t.X[i] = X[o[i]]; t.Y[i] = Y[o[i]]; t.z[i] = z[o[i]]; t.iw[i] = iw[o[i]];
        for (int k = 0; k < 3; ++k) t.bary[i][k] = v[o[i]].bary[k];
    }
    t.area2 = area2;
    int32_t xmin = std::min(t.X[0], std::min(t.X[1], t.X[2])), xmax = std::max(t.X[0], std::max(t.X[1], t.X[2]));
    int32_t ymin = std::min(t.Y[0], std::min(t.Y[1], t.Y[2])), ymax = std::max(t.Y[0], std::max(t.Y[1], t.Y[2]));
    // pixel px is a candidate when its centre px*256+128 lies in [min,max]
    t.px0 = std::max(ceil_div256(xmin - 128), sc_x0);
    t.px1 = std::min(floor_div256(xmax - 128), sc_x1 - 1);
    t.py0 = std::max(ceil_div256(ymin - 128), sc_y0);
    t.py1 = std::min(floor_div256(ymax - 128), sc_y1 - 1);
    return t.px0 <= t.px1 && t.py0 <= t.py1;
}

// edge i runs from vertex i to vertex (i+1)%3; inside-positive for area2 > 0
struct EdgeEq { int64_t dx, dy; int32_t x0, y0; int64_t bias; };
inline EdgeEq make_edge(const SetupTri &t, int i) {
    int j = (i + 1) % 3;
    EdgeEq e;
    e.dx = (int64_t)t.X[j] - t.X[i];
    e.dy = (int64_t)t.Y[j] - t.Y[i];
    e.x0 = t.X[i]; e.y0 = t.Y[i];
    // top-left rule for a clockwise (as seen, y down) triangle: top = horizontal
    // edge going right, left = edge going up
    bool top_left = (e.dy == 0 && e.dx > 0) || (e.dy < 0);
    e.bias = top_left ? 0 : -1;
    return e;
}
inline int64_t edge_eval(const EdgeEq &e, int32_t px, int32_t py) {
    int64_t Px = (int64_t)px * 256 + 128, Py = (int64_t)py * 256 + 128;
    return e.dx * (Py - e.y0) - e.dy * (Px - e.x0);
}

// coverage + depth of one pixel; returns false when not covered / depth-clipped
inline bool fragment(const SetupTri &t, const EdgeEq e[3], float inv_area, int32_t px, int32_t py, float &l1, float &l2, float &z) {
    int64_t e0 = edge_eval(e[0], px, py), e1 = edge_eval(e[1], px, py), e2 = edge_eval(e[2], px, py);
    if ((e0 + e[0].bias) < 0 || (e1 + e[1].bias) < 0 || (e2 + e[2].bias) < 0) return false;
    // barycentric weight of vertex k is the edge function of the opposite edge
    l1 = (float)e2 * inv_area;   // vertex 1 <-> edge 2 (2->0)
    l2 = (float)e0 * inv_area;   // vertex 2 <-> edge 0 (0->1)
    z = std::fmaf(l2, t.z[2] - t.z[0], std::fmaf(l1, t.z[1] - t.z[0], t.z[0]));
    z = std::fmin(std::fmax(z, 0.0f), 1.0f);
    return true;
}

struct MeshData { std::vector<Vertex> verts; std::vector<uint32_t> indices; uint64_t material; };
struct Texture { std::vector<uint8_t> px; uint32_t w, h; };
struct MaterialData { Texture diffuse, normal, mr; };

struct Oracle {
    uint32_t width = 0, height = 0, shadow_size = 0, max_lights = 0;
    uint32_t row_begin = 0, row_end = 0;
    std::vector<MeshData> meshes;
    std::vector<MaterialData> materials;
    std::vector<PointLight> lights;
    // frame state (row-major, rows = row_end-row_begin)
    std::vector<float> depth;       // rows*width
    std::vector<uint32_t> tri_id;   // rows*width  (setup record index)
    std::vector<uint32_t> src_tri;  // rows*width  (draw-order id of the source triangle)
    std::vector<float> attrs;       // rows*width*18
    std::vector<uint32_t> matid;    // rows*width
    std::vector<float> shadow;      // S*S
    std::vector<float> hdr, ldr;    // rows*width*3
    std::vector<uint8_t> rgba8;     // rows*width*4
    float srgb_lut[256];
    uint64_t stats[6] = {0, 0, 0, 0, 0, 0};
    std::vector<float> env; uint32_t env_w = 0, env_h = 0;   // skybox environment map (RGBA32F); empty = black
    int hdr16 = 0;        // 1: ps_main's colour passes through binary16 (the reference's RGBA16F target) before post_process
    int precision = 64;   // arithmetic of the BRDF/tonemap: 64 = float64 (parity arbiter), 32 = literal fp32 (CPU baseline)
    int sampler_mode = 0; // SAMPLER_* bits: variants of what the reference leaves to the D3D12 sampler hardware (default: none)
    std::string err;
    uint32_t rows() const { return row_end - row_begin; }
};

// one pass of vertex + clip + setup over the scene; calls emit(setup, vs[3]) per kept sub-triangle in draw order
template <class Emit>
void process_geometry(const Oracle &o, const Scene &sc, const M4 &pv, const M4 &lpv,
                      float vw, float vh, int sx0, int sy0, int sx1, int sy1, CullMode cull, Emit emit) {
    uint32_t src = 0;
    for (uint64_t oi = 0; oi < sc.n_objects; ++oi) {
        const Object &ob = sc.objects[oi];
        if (ob.mesh_idx >= o.meshes.size()) continue;
        const MeshData &mesh = o.meshes[ob.mesh_idx];
        M4 model; std::memcpy(&model, ob.trs, sizeof(M4));
        std::vector<VSOut> vs(mesh.verts.size());
        for (size_t i = 0; i < mesh.verts.size(); ++i) vs[i] = vs_main(mesh.verts[i], model, pv, lpv);
        size_t ntri = mesh.indices.size() / 3;
        for (size_t ti = 0; ti < ntri; ++ti, ++src) {
            uint32_t i0 = mesh.indices[3 * ti], i1 = mesh.indices[3 * ti + 1], i2 = mesh.indices[3 * ti + 2];
            if (i0 >= vs.size() || i1 >= vs.size() || i2 >= vs.size()) continue;
            const VSOut *tv[3] = {&vs[i0], &vs[i1], &vs[i2]};
            ClipVert poly[MAX_POLY];
            for (int k = 0; k < 3; ++k) {
                poly[k].p = tv[k]->clip;
                poly[k].bary[0] = k == 0 ? 1.0f : 0.0f; poly[k].bary[1] = k == 1 ? 1.0f : 0.0f; poly[k].bary[2] = k == 2 ? 1.0f : 0.0f;
            }
            int n = clip_polygon(poly, 3);
            for (int f = 1; f + 1 < n; ++f) {
                ClipVert tri[3] = {poly[0], poly[f], poly[f + 1]};
                SetupTri st;
                if (!setup_triangle(tri, vw, vh, sx0, sy0, sx1, sy1, cull, st)) continue;
                st.src_tri = src; st.object = (uint32_t)oi;
                emit(st, tv);
            }
        }
    }
}

// ShadowMapPass::run (shadow_map_pass.cpp:113-169) + depth.hlsl: clip = proj_view * model * pos
void pass_shadow_map(Oracle &o, const Scene &sc) {
    uint32_t S = o.shadow_size;
    if (S == 0) return;
    o.shadow.assign((size_t)S * S, 1.0f);  // ClearDepthStencilView(1.0), shadow_map_pass.cpp:124-131
    M4 lpv = sun_proj_view(sc.sun);
    uint64_t nsetup = 0;
    // depth.hlsl computes mul(proj_view, mul(model, pos)) = the same two mat_vec as vs_main's light_space_position
    process_geometry(o, sc, lpv, lpv, (float)S, (float)S, 0, 0, (int)S, (int)S, CULL_FRONT,
        [&](const SetupTri &t, const VSOut *const *) {
            ++nsetup;
            EdgeEq e[3] = {make_edge(t, 0), make_edge(t, 1), make_edge(t, 2)};
            float inv_area = 1.0f / (float)t.area2;
            for (int32_t py = t.py0; py <= t.py1; ++py)
                for (int32_t px = t.px0; px <= t.px1; ++px) {
                    float l1, l2, z;
                    if (!fragment(t, e, inv_area, px, py, l1, l2, z)) continue;
                    float &d = o.shadow[(size_t)py * S + px];
                    if (z < d) d = z;   // depth func LESS (CD3DX12_DEPTH_STENCIL_DESC default)
                }
        });
    o.stats[2] = nsetup;
}

// ForwardPass::run's fixed-function part (forward_pass.cpp:161-226): visibility, then the
// interpolated VSOut per visible pixel = the G-buffer.
void pass_gbuffer(Oracle &o, const Scene &sc) {
    uint32_t W = o.width, rows = o.rows();
    size_t npx = (size_t)W * rows;
    o.depth.assign(npx, 1.0f);   // ClearDepthStencilView 1.0, forward_pass.cpp:179-186
    o.tri_id.assign(npx, 0xFFFFFFFFu);
    o.src_tri.assign(npx, 0xFFFFFFFFu);
    o.matid.assign(npx, 0xFFFFFFFFu);
    o.attrs.assign(npx * 18, 0.0f);
    M4 pv = camera_proj_view(sc.camera), lpv = sun_proj_view(sc.sun);
    uint32_t rec = 0;
    process_geometry(o, sc, pv, lpv, (float)o.width, (float)o.height, 0, (int)o.row_begin, (int)W, (int)o.row_end, CULL_BACK,
        [&](const SetupTri &t, const VSOut *const *tv) {
            uint32_t my = rec++;
            EdgeEq e[3] = {make_edge(t, 0), make_edge(t, 1), make_edge(t, 2)};
            float inv_area = 1.0f / (float)t.area2;
            uint32_t mat = (uint32_t)o.meshes[sc.objects[t.object].mesh_idx].material;
            for (int32_t py = t.py0; py <= t.py1; ++py)
                for (int32_t px = t.px0; px <= t.px1; ++px) {
                    float l1, l2, z;
                    if (!fragment(t, e, inv_area, px, py, l1, l2, z)) continue;
                    size_t p = (size_t)(py - (int32_t)o.row_begin) * W + px;
                    if (!(z < o.depth[p])) continue;   // LESS; ties keep the first drawn
                    o.depth[p] = z; o.tri_id[p] = my; o.src_tri[p] = t.src_tri; o.matid[p] = mat;
                    // perspective-correct interpolation through the source triangle's barycentrics
                    float l0 = (1.0f - l1) - l2;
                    float pw0 = l0 * t.iw[0], pw1 = l1 * t.iw[1], pw2 = l2 * t.iw[2];
                    float r = 1.0f / ((pw0 + pw1) + pw2);
                    float b0 = pw0 * r, b1 = pw1 * r, b2 = pw2 * r;
                    float B[3];
                    for (int k = 0; k < 3; ++k) B[k] = (b0 * t.bary[0][k] + b1 * t.bary[1][k]) + b2 * t.bary[2][k];
                    float *a = &o.attrs[p * 18];
                    for (int k = 0; k < 18; ++k) a[k] = (B[0] * tv[0]->attr[k] + B[1] * tv[1]->attr[k]) + B[2] * tv[2]->attr[k];
                }
        });
    o.stats[0] = rec;
}

// ----------------------------------------------------------------------------
// texture sampling: static sampler MIN_MAG_MIP_LINEAR + WRAP (forward_pass.cpp:38-51),
// one mip (rhi.cpp:550), formats R8G8B8A8_UNORM_SRGB / R8G8B8A8_UNORM
// (renderer.cpp:483-548), shadow map R32_FLOAT (renderer.cpp:95-109).
// Semantics fixed here (D3D leaves weight precision to the hardware): texel
// centres at integer+0.5, full fp32 weights, sRGB decoded per texel before
// filtering by the IEC 61966-2-1 curve, WRAP via u - floor(u).
// ----------------------------------------------------------------------------
struct Footprint { int x0, x1, y0, y1; float fx, fy; };

// SAMPLER VARIANTS (oracle_set_sampler_mode; default 0 = the semantics above, what the HIP kernels implement).  They exist to BOUND
// what the reference leaves to the D3D12 hardware (forward_pass.cpp:38-51 only names the filter), not as alternatives to ship:
//   bit 0  material textures: texel coordinates in D3D's fixed point -- the scaled coordinate u W - 0.5 snapped to 1/256 texel
//          (round to nearest) before it is split into texel index and weight: 8-bit filter weights, what the D3D11.3 functional
//          specification (3.2.4.1 / 7.18.8) allows a sampler to do and most hardware does
//   bit 1  sRGB textures: the four texels filtered as stored (non-linear) and the RESULT decoded, instead of decoding each texel
//          first (D3D10+ asks for decode-before-filter; the variant bounds what a non-conformant path would change)
//   bit 2  shadow map (R32_FLOAT through the same sampler, forward.hlsl:68-96): the same 8-bit weights for the 25 PCF taps
constexpr int SAMPLER_Q8_MATERIAL = 1, SAMPLER_SRGB_AFTER_FILTER = 2, SAMPLER_Q8_SHADOW = 4;

inline void wrap_axis(float u, uint32_t n, int &i0, int &i1, float &f, bool q8 = false) {
    float uw = u - std::floor(u);
    float x = uw * (float)n - 0.5f;
    if (q8) x = std::floor(x * 256.0f + 0.5f) / 256.0f;   // 24.8 fixed point, round to nearest
    float xf = std::floor(x);
    f = x - xf;
    i0 = (int)xf; i1 = i0 + 1;
    if (i0 < 0) i0 += (int)n;
    if (i1 >= (int)n) i1 -= (int)n;
}
inline Footprint footprint(float u, float v, uint32_t w, uint32_t h, bool q8 = false) {
    Footprint f;
    wrap_axis(u, w, f.x0, f.x1, f.fx, q8);
    wrap_axis(v, h, f.y0, f.y1, f.fy, q8);
    return f;
}
inline float bilerp(float t00, float t10, float t01, float t11, float fx, float fy) {
    float top = lerp1(t00, t10, fx), bot = lerp1(t01, t11, fx);
    return lerp1(top, bot, fy);
}
float srgb_to_linear(uint8_t c) {
    float x = (float)c / 255.0f;
    return x <= 0.04045f ? x / 12.92f : std::pow((x + 0.055f) / 1.055f, 2.4f);
}
float sample_r32(const float *map, uint32_t S, float u, float v, bool q8 = false) {
    Footprint f = footprint(u, v, S, S, q8);
    return bilerp(map[(size_t)f.y0 * S + f.x0], map[(size_t)f.y0 * S + f.x1],
                  map[(size_t)f.y1 * S + f.x0], map[(size_t)f.y1 * S + f.x1], f.fx, f.fy);
}

// forward.hlsl:68-96 calculate_shadow (the `normal` argument is unused there: bias = 0)
float calculate_shadow(const float *map, uint32_t S, V4 ls, bool q8 = false) {
    if (map == nullptr || S == 0) return 0.0f;   // config 1: no shadow map == map cleared to 1.0
    float px = ls.x / ls.w, py = ls.y / ls.w, pz = ls.z / ls.w;
    px = px * 0.5f + 0.5f;
    py = py * 0.5f + 0.5f;
    py = 1.0f - py;
    if (pz > 1.0f || px < 0.0f || py < 0.0f || px > 1.0f || py > 1.0f) return 0.0f;
    float bias = 0.0f;
    float current = pz;
    float shadow = 0.0f;
    for (int i = -2; i <= 2; ++i)
        for (int j = -2; j <= 2; ++j) {
            float ox = (float)i * 0.0001f, oy = (float)j * 0.0001f;
            float closest = sample_r32(map, S, px + ox, py + oy, q8);
            shadow += (current - bias) > closest ? 1.0f : 0.0f;
        }
    shadow /= 25.0f;
    return shadow;
}

// ----------------------------------------------------------------------------
// BRDF + tonemap, templated on the real type R.
//   R = float : the literal fp32 restatement of the HLSL (every operation rounds to fp32, in
//               source order).  Timed as the CPU baseline; its output carries fp32 rounding noise.
//   R = double: the same formulas evaluated in float64 on the same fp32 inputs = the value the
//               HLSL text defines, free of implementation-specific rounding.  THIS is what the
//               parity tests compare the HIP kernel against.
// Why both: forward.hlsl:137 computes denom = n_dot_h^2 * (a2 - 1) + 1, which cancels to ~a2 at a
// highlight; with roughness 0.05 (a2 = 6e-6) one fp32 ulp of n_dot_h moves the NDF by > 1 %, so two
// equally valid fp32 evaluation orders (this file's, DXC's on some GPU, the HIP kernel's) differ by
// up to ~1e-3 after tonemapping on a handful of highlight pixels.  Against the float64 value a
// well-conditioned fp32 kernel can be held to 1e-4 everywhere; tests/test_oracle_kat.py also
// bounds |fp32 oracle - float64 oracle| to document that noise floor.
// Discrete decisions (texel addresses and weights, the 25 shadow compares, coverage) are always
// taken in fp32 exactly as above, for both R.
// PI as written at forward.hlsl:1.
// ----------------------------------------------------------------------------
template <class R> struct Vec3 { R x, y, z; };
template <class R> inline Vec3<R> vec3(R x, R y, R z) { return Vec3<R>{x, y, z}; }
template <class R> inline Vec3<R> operator+(Vec3<R> a, Vec3<R> b) { return vec3<R>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class R> inline Vec3<R> operator-(Vec3<R> a, Vec3<R> b) { return vec3<R>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class R> inline Vec3<R> operator*(Vec3<R> a, Vec3<R> b) { return vec3<R>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <class R> inline Vec3<R> operator*(Vec3<R> a, R s) { return vec3<R>(a.x * s, a.y * s, a.z * s); }
template <class R> inline Vec3<R> operator/(Vec3<R> a, R s) { return vec3<R>(a.x / s, a.y / s, a.z / s); }
template <class R> inline R dot(Vec3<R> a, Vec3<R> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
template <class R> inline Vec3<R> normalize(Vec3<R> v) { R inv = R(1) / std::sqrt(dot(v, v)); return v * inv; }
template <class R> inline R rmax(R a, R b) { return a > b ? a : b; }   // max(x, 0) of finite values
template <class R> inline R rclamp01(R x) { return x < R(0) ? R(0) : (x > R(1) ? R(1) : x); }
inline float rlerp(float a, float b, float t) { return std::fmaf(t, b - a, a); }   // HLSL lerp as one mad
inline double rlerp(double a, double b, double t) { return a + t * (b - a); }
template <class R> constexpr R PI_R = R(3.14159265);   // the literal of forward.hlsl:1 (as double, then as float)

// forward.hlsl:126-129 (cos_theta is a scalar broadcast to float3)
template <class R> inline Vec3<R> fresnel_schlick(R cos_theta, Vec3<R> F0) {
    R p = std::pow(rclamp01<R>(R(1) - cos_theta), R(5));
    return vec3<R>(F0.x + (R(1) - F0.x) * p, F0.y + (R(1) - F0.y) * p, F0.z + (R(1) - F0.z) * p);
}
// forward.hlsl:131-143
template <class R> inline R distribution_ggx(Vec3<R> n, Vec3<R> h, R roughness) {
    R a = roughness * roughness;
    R a2 = a * a;
    R ndh = rmax<R>(dot(n, h), R(0));
    R ndh2 = ndh * ndh;
    R denom = ndh2 * (a2 - R(1)) + R(1);
    denom = PI_R<R> * denom * denom;
    return a2 / denom;
}
// forward.hlsl:145-154
template <class R> inline R geometry_schlick_ggx(R ndwo, R roughness) {
    R r = roughness + R(1);
    R k = (r * r) / R(8);
    return ndwo / (ndwo * (R(1) - k) + k);
}
// forward.hlsl:156-163
template <class R> inline R geometry_smith(Vec3<R> n, Vec3<R> wo, Vec3<R> wi, R roughness) {
    R ndwo = rmax<R>(dot(n, wo), R(0)), ndwi = rmax<R>(dot(n, wi), R(0));
    return geometry_schlick_ggx(ndwo, roughness) * geometry_schlick_ggx(ndwi, roughness);
}
// forward.hlsl:165-175
template <class R> inline Vec3<R> brdf_cook_torrance(Vec3<R> n, Vec3<R> h, Vec3<R> wo, Vec3<R> wi, R roughness, Vec3<R> F) {
    R NDF = distribution_ggx(n, h, roughness);
    R G = geometry_smith(n, wo, wi, roughness);
    Vec3<R> num = F * (NDF * G);
    R denom = R(4) * rmax<R>(dot(n, wo), R(0)) * rmax<R>(dot(n, wi), R(0)) + R(0.0001f);
    return num / denom;
}
// forward.hlsl:177-193
template <class R> Vec3<R> calculate_outgoing_radiance(Vec3<R> n, Vec3<R> wo, Vec3<R> wi, Vec3<R> Li, Vec3<R> base, R metal, R rough) {
    Vec3<R> h = normalize(wo + wi);
    Vec3<R> F0 = vec3<R>(rlerp(R(0.04f), base.x, metal), rlerp(R(0.04f), base.y, metal), rlerp(R(0.04f), base.z, metal));
    Vec3<R> F = fresnel_schlick(rmax<R>(dot(h, wo), R(0)), F0);
    Vec3<R> spec = brdf_cook_torrance(n, h, wo, wi, rough, F);
    Vec3<R> kD = vec3<R>(R(1) - F.x, R(1) - F.y, R(1) - F.z);
    kD = kD * (R(1) - metal);
    R ndwi = rmax<R>(dot(n, wi), R(0));
    Vec3<R> diff = (kD * base) / PI_R<R>;
    return ((diff + spec) * Li) * ndwi;
}

// filtered RGBA of one material texture; texel selection and weights in fp32 (shared with the HIP kernel), the
// filter arithmetic in R
template <class R> void sample_rgba8_r(const Oracle &o, const Texture &t, float u, float v, bool srgb, R out[4]) {
    Footprint f = footprint(u, v, t.w, t.h, (o.sampler_mode & SAMPLER_Q8_MATERIAL) != 0);
    const bool after = srgb && (o.sampler_mode & SAMPLER_SRGB_AFTER_FILTER) != 0;
    const uint8_t *p00 = &t.px[((size_t)f.y0 * t.w + f.x0) * 4], *p10 = &t.px[((size_t)f.y0 * t.w + f.x1) * 4];
    const uint8_t *p01 = &t.px[((size_t)f.y1 * t.w + f.x0) * 4], *p11 = &t.px[((size_t)f.y1 * t.w + f.x1) * 4];
    for (int c = 0; c < 4; ++c) {
        R a, b, cc, d;
        if (srgb && c < 3 && !after) { a = o.srgb_lut[p00[c]]; b = o.srgb_lut[p10[c]]; cc = o.srgb_lut[p01[c]]; d = o.srgb_lut[p11[c]]; }
        else { a = (R)p00[c] / R(255); b = (R)p10[c] / R(255); cc = (R)p01[c] / R(255); d = (R)p11[c] / R(255); }
        R top = rlerp(a, b, (R)f.fx), bot = rlerp(cc, d, (R)f.fx);
        out[c] = rlerp(top, bot, (R)f.fy);
        if (after && c < 3) {   // the IEC 61966-2-1 curve on the filtered value
            const double x = (double)out[c];
            out[c] = (R)(x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4));
        }
    }
}

// forward.hlsl:98-124 material fetch
template <class R> struct SurfaceR { Vec3<R> base, n; R metal, rough; };
template <class R> SurfaceR<R> fetch_surface(const Oracle &o, const MaterialData &m, const float *attr) {
    float u = attr[0], v = attr[1];
    R d[4], nm[4], mr[4];
    sample_rgba8_r<R>(o, m.diffuse, u, v, true, d);
    sample_rgba8_r<R>(o, m.normal, u, v, false, nm);
    sample_rgba8_r<R>(o, m.mr, u, v, false, mr);
    SurfaceR<R> s;
    s.base = vec3<R>(d[0], d[1], d[2]);
    Vec3<R> tn = vec3<R>(nm[0], R(1) - nm[1], nm[2]);                      // normal.g = 1 - normal.g
    tn = vec3<R>(tn.x * R(2) - R(1), tn.y * R(2) - R(1), tn.z * R(2) - R(1));
    // mul(tbn, v) with tbn columns t,b,n: t*v.x + b*v.y + n*v.z per component
    Vec3<R> T = vec3<R>(attr[2], attr[3], attr[4]), B = vec3<R>(attr[5], attr[6], attr[7]), N = vec3<R>(attr[8], attr[9], attr[10]);
    Vec3<R> w = vec3<R>((T.x * tn.x + B.x * tn.y) + N.x * tn.z, (T.y * tn.x + B.y * tn.y) + N.y * tn.z, (T.z * tn.x + B.z * tn.y) + N.z * tn.z);
    s.n = normalize(w);
    s.metal = mr[2];   // .b  (forward.hlsl:117)
    s.rough = mr[1];   // .g  (forward.hlsl:123)
    return s;
}

// forward.hlsl:208-235 ps_main; returns HDR rgb
template <class R> Vec3<R> ps_main(const Oracle &o, const float *attr, uint32_t mat, V3 eye_f, V3 sun_dir_f, V3 sun_color_f, float ambient,
                                   const float *shadow_map) {
    const MaterialData &m = o.materials[mat];
    SurfaceR<R> s = fetch_surface<R>(o, m, attr);
    Vec3<R> eye = vec3<R>(eye_f.x, eye_f.y, eye_f.z), sun_color = vec3<R>(sun_color_f.x, sun_color_f.y, sun_color_f.z);
    Vec3<R> world = vec3<R>(attr[11], attr[12], attr[13]);
    Vec3<R> wo = normalize(eye - world);
    Vec3<R> Lo = vec3<R>(0, 0, 0);
    R shadow = calculate_shadow(shadow_map, o.shadow_size, V4{attr[14], attr[15], attr[16], attr[17]}, (o.sampler_mode & SAMPLER_Q8_SHADOW) != 0);   // always fp32: k/25
    R lit = R(1) - shadow;
    Lo = Lo + calculate_outgoing_radiance<R>(s.n, wo, vec3<R>(-sun_dir_f.x, -sun_dir_f.y, -sun_dir_f.z), sun_color, s.base, s.metal, s.rough) * lit;
    for (size_t i = 0; i < o.lights.size(); ++i) {
        const PointLight &L = o.lights[i];
        Vec3<R> d = vec3<R>(L.position[0], L.position[1], L.position[2]) - world;
        R dist = std::sqrt(dot(d, d));
        Vec3<R> wi = d / dist;
        Vec3<R> radiance = vec3<R>(L.color[0], L.color[1], L.color[2]) / (dist * dist);
        Lo = Lo + calculate_outgoing_radiance<R>(s.n, wo, wi, radiance, s.base, s.metal, s.rough) * lit;
    }
    return Lo + s.base * (R)ambient;
}

// ----------------------------------------------------------------------------
// post_process.hlsl
// ----------------------------------------------------------------------------
template <class R> inline Vec3<R> tm_reinhard(Vec3<R> c) { return vec3<R>(c.x / (c.x + R(1)), c.y / (c.y + R(1)), c.z / (c.z + R(1))); }   // :39-42
template <class R> inline Vec3<R> tm_exposure(Vec3<R> c, R e) { return vec3<R>(R(1) - std::exp(-c.x * e), R(1) - std::exp(-c.y * e), R(1) - std::exp(-c.z * e)); }  // :44-47
template <class R> inline R rrt_odt(R c) {                                                                    // :27-32
    R a = c * (c + R(0.0245786f)) - R(0.000090537f);
    R b = c * (R(0.983729f) * c + R(0.4329510f)) + R(0.238081f);
    return a / b;
}
template <class R> inline Vec3<R> tm_aces(Vec3<R> c) {                                                        // :15-25, :50-57
    Vec3<R> i = vec3<R>((R(0.59719f) * c.x + R(0.35458f) * c.y) + R(0.04823f) * c.z,
                        (R(0.07600f) * c.x + R(0.90834f) * c.y) + R(0.01566f) * c.z,
                        (R(0.02840f) * c.x + R(0.13383f) * c.y) + R(0.837f) * c.z);
    i = vec3<R>(rrt_odt(i.x), rrt_odt(i.y), rrt_odt(i.z));
    Vec3<R> r = vec3<R>((R(1.60475f) * i.x + R(-0.53108f) * i.y) + R(-0.07367f) * i.z,
                        (R(-0.10208f) * i.x + R(1.10813f) * i.y) + R(-0.00605f) * i.z,
                        (R(-0.00327f) * i.x + R(-0.07276f) * i.y) + R(1.07f) * i.z);
    return vec3<R>(rclamp01(r.x), rclamp01(r.y), rclamp01(r.z));
}
template <class R> inline Vec3<R> tonemap_only(Vec3<R> c, const Settings &st) {
    switch (st.tm_method) {
    case 1: return tm_exposure<R>(c, (R)st.exposure);
    case 2: return tm_aces<R>(c);
    default: return tm_reinhard<R>(c);
    }
}
template <class R> inline Vec3<R> post_process(Vec3<R> c, const Settings &st) {                               // :59-93
    c = tonemap_only<R>(c, st);
    R ig = R(1) / (R)st.gamma;                                                                                // :34-37
    return vec3<R>(std::pow(std::fabs(c.x), ig), std::pow(std::fabs(c.y), ig), std::pow(std::fabs(c.z), ig));
}
// fp32 -> binary16 -> fp32, round to nearest even: the R16G16B16A16_FLOAT colour target of the reference
// (forward_pass.cpp:149, renderer.cpp:128-144) that sits between ps_main and post_process
inline float through_half(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    uint32_t sign = x & 0x80000000u, a = x & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return f;                               // inf / nan
    if (a >= 0x477FF000u) { uint32_t inf = sign | 0x7F800000u; float r; std::memcpy(&r, &inf, 4); return r; }   // rounds past 65504
    if (a < 0x33000001u) { float r; std::memcpy(&r, &sign, 4); return r; }                                      // below half the smallest subnormal
    uint32_t r;
    if (a < 0x38800000u) {                                        // binary16 subnormal: quantum 2^-24
        float q = std::nearbyint(std::fabs(f) * 16777216.0f) / 16777216.0f;   // default rounding mode = nearest even
        std::memcpy(&r, &q, 4);
        r |= sign;
    } else {
        uint32_t lsb = (a >> 13) & 1u;
        r = sign | ((a + 0x0FFFu + lsb) & 0xFFFFE000u);
    }
    float out; std::memcpy(&out, &r, 4);
    return out;
}

// float -> UNORM8 store of the RGBA8 target (renderer.cpp:161-175): D3D rule = saturate (NaN -> 0), *255, +0.5, truncate
inline uint8_t to_unorm8(float x) {
    if (!(x > 0.0f)) return 0;
    if (x > 1.0f) x = 1.0f;
    return (uint8_t)(x * 255.0f + 0.5f);
}

// ---- skybox (skybox.hlsl:61-90, skybox_pass.cpp:104-138) ---------------------------------------------------------
// The reference draws a unit cube with proj * mat3(lookAtRH) at depth 1 (clip z = w) after the forward pass; it survives the
// depth test exactly where no geometry was drawn, and its interpolated object-space position is the world-space view ray of
// the pixel.  Restated without the cube: the ray through ndc (x, y) is fwd + x*right + y*up with lookAtRH's basis scaled by
// the frustum half-extents (same direction as the interpolated cube position, which ps_main normalises anyway).
struct SkyBasis { V3 fwd, right, up; };
SkyBasis sky_basis(const Camera &c) {
    V3 f = normalize(dir_from_rot(c.rotation));
    V3 s = normalize(cross(f, v3(0.0f, 1.0f, 0.0f)));
    V3 u = cross(s, f);
    float t = std::tan(radians(c.fov_y) / 2.0f), tx = c.aspect * t;
    return SkyBasis{f, scale(s, tx), scale(u, t)};
}
inline V3 sky_ray(const SkyBasis &b, uint32_t x, uint32_t y, uint32_t W, uint32_t H) {
    float sx = 2.0f / (float)W, sy = 2.0f / (float)H;
    float nx = std::fmaf((float)x + 0.5f, sx, -1.0f), ny = std::fmaf(-((float)y + 0.5f), sy, 1.0f);
    return v3(std::fmaf(b.up.x, ny, std::fmaf(b.right.x, nx, b.fwd.x)), std::fmaf(b.up.y, ny, std::fmaf(b.right.y, nx, b.fwd.y)),
              std::fmaf(b.up.z, ny, std::fmaf(b.right.z, nx, b.fwd.z)));
}
inline void wrap_axis64(double u, uint32_t n, int &i0, int &i1, float &f) {
    double uw = u - std::floor(u);
    double x = uw * (double)n - 0.5, xf = std::floor(x);
    f = (float)(x - xf);
    i0 = (int)xf; i1 = i0 + 1;
    if (i0 < 0) i0 += (int)n;
    if (i1 >= (int)n) i1 -= (int)n;
}
// sample_environment(dir): equirect lookup, LINEAR + WRAP sampler (skybox_pass.cpp:34-47), lookup coordinates in float64
template <class R>
Vec3<R> sample_environment(const Oracle &o, V3 dir, double *uv_out = nullptr) {
    double x = dir.x, y = dir.y, z = dir.z;
    double inv = 1.0 / std::sqrt(x * x + y * y + z * z);
    x *= inv; y *= inv; z *= inv;
    double u = std::atan2(z, x) * (double)0.1591f + 0.5;
    double v = -(std::asin(std::fmin(std::fmax(y, -1.0), 1.0)) * (double)0.3183f + 0.5);
    if (uv_out) { uv_out[0] = u; uv_out[1] = v; }
    if (!o.env_w) return vec3<R>(0, 0, 0);
    int x0, x1, y0, y1; float fx, fy;
    wrap_axis64(u, o.env_w, x0, x1, fx);
    wrap_axis64(v, o.env_h, y0, y1, fy);
    const float *a = &o.env[((size_t)y0 * o.env_w + x0) * 4];
    const float *b = &o.env[((size_t)y0 * o.env_w + x1) * 4];
    const float *c = &o.env[((size_t)y1 * o.env_w + x0) * 4];
    const float *d = &o.env[((size_t)y1 * o.env_w + x1) * 4];
    R gx = (R)1 - (R)fx, gy = (R)1 - (R)fy;
    R w00 = gx * gy, w10 = (R)fx * gy, w01 = gx * (R)fy, w11 = (R)fx * (R)fy;
    return vec3<R>(w00 * a[0] + w10 * b[0] + w01 * c[0] + w11 * d[0], w00 * a[1] + w10 * b[1] + w01 * c[1] + w11 * d[1],
                   w00 * a[2] + w10 * b[2] + w01 * c[2] + w11 * d[2]);
}

template <class R>
void shade_rows(Oracle &o, const Scene &sc, const Settings &st, const float *attrs, const uint32_t *matid,
                uint32_t r0, uint32_t r1, uint32_t y_origin, float *hdr, float *ldr, uint8_t *rgba8, std::atomic<uint64_t> *shaded) {
    uint32_t W = o.width;
    SkyBasis sky = sky_basis(sc.camera);
    V3 eye = v3(sc.camera.eye[0], sc.camera.eye[1], sc.camera.eye[2]);
    V3 sun_dir = dir_from_rot(sc.sun.rotation);
    V3 sun_color = v3(sc.sun.color[0], sc.sun.color[1], sc.sun.color[2]);
    const float *smap = o.shadow_size ? o.shadow.data() : nullptr;
    uint64_t count = 0;
    for (uint32_t y = r0; y < r1; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            size_t p = (size_t)y * W + x;
            Vec3<R> c = vec3<R>(0, 0, 0);
            uint32_t m = matid[p];
            if (m == 0xFFFFFFFFu) {   // no geometry: the skybox (black without an environment map)
                if (o.env_w) c = sample_environment<R>(o, sky_ray(sky, x, y + y_origin, W, o.height));
            } else if (m < o.materials.size()) {
                c = ps_main<R>(o, attrs + p * 18, m, eye, sun_dir, sun_color, sc.ambient, smap);
                ++count;
            }
            if (o.hdr16) c = vec3<R>((R)through_half((float)c.x), (R)through_half((float)c.y), (R)through_half((float)c.z));
            Vec3<R> l = post_process<R>(c, st);
            if (hdr) { hdr[p * 3] = (float)c.x; hdr[p * 3 + 1] = (float)c.y; hdr[p * 3 + 2] = (float)c.z; }
            if (ldr) { ldr[p * 3] = (float)l.x; ldr[p * 3 + 1] = (float)l.y; ldr[p * 3 + 2] = (float)l.z; }
            if (rgba8) { rgba8[p * 4] = to_unorm8((float)l.x); rgba8[p * 4 + 1] = to_unorm8((float)l.y); rgba8[p * 4 + 2] = to_unorm8((float)l.z); rgba8[p * 4 + 3] = 255; }
        }
    if (shaded) shaded->fetch_add(count);
}

void shade_parallel(Oracle &o, const Scene &sc, const Settings &st, const float *attrs, const uint32_t *matid,
                    uint32_t rows, uint32_t y_origin, float *hdr, float *ldr, uint8_t *rgba8, int threads) {
    if (threads < 1) threads = 1;
    std::atomic<uint64_t> shaded{0};
    auto run = [&](uint32_t a, uint32_t b) {
        if (o.precision == 32) shade_rows<float>(o, sc, st, attrs, matid, a, b, y_origin, hdr, ldr, rgba8, &shaded);
        else shade_rows<double>(o, sc, st, attrs, matid, a, b, y_origin, hdr, ldr, rgba8, &shaded);
    };
    if (threads == 1) run(0, rows);
    else {
        // rows dealt in bands of 4 for balance; each thread takes bands t, t+T, ...
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t]() {
                for (uint32_t b = (uint32_t)t * 4; b < rows; b += (uint32_t)threads * 4) run(b, std::min(rows, b + 4));
            });
        for (auto &th : pool) th.join();
    }
    o.stats[4] = shaded.load();
    o.stats[5] = shaded.load() * o.lights.size();
}

}  // namespace

// ============================================================================
// C ABI (ctypes)
// ============================================================================
extern "C" {

void *oracle_create(uint32_t width, uint32_t height, uint32_t shadow_size, uint32_t max_lights,
                    uint32_t row_begin, uint32_t row_end) {
    if (width == 0 || height == 0) return nullptr;
    Oracle *o = new Oracle();
    o->width = width; o->height = height; o->shadow_size = shadow_size; o->max_lights = max_lights;
    if (row_begin == 0 && row_end == 0) row_end = height;
    if (row_end > height || row_begin >= row_end) { delete o; return nullptr; }
    o->row_begin = row_begin; o->row_end = row_end;
    for (int i = 0; i < 256; ++i) o->srgb_lut[i] = srgb_to_linear((uint8_t)i);
    if (shadow_size) o->shadow.assign((size_t)shadow_size * shadow_size, 1.0f);
    return o;
}
void oracle_destroy(void *h) { delete static_cast<Oracle *>(h); }

int oracle_create_material(void *h, const void *d, uint32_t dw, uint32_t dh, const void *n, uint32_t nw, uint32_t nh,
                           const void *m, uint32_t mw, uint32_t mh) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || !d || !n || !m || !dw || !dh || !nw || !nh || !mw || !mh) return -1;
    MaterialData md;
    auto fill = [](Texture &t, const void *p, uint32_t w, uint32_t hh) {
        t.w = w; t.h = hh; t.px.assign((const uint8_t *)p, (const uint8_t *)p + (size_t)w * hh * 4);
    };
    fill(md.diffuse, d, dw, dh); fill(md.normal, n, nw, nh); fill(md.mr, m, mw, mh);
    o->materials.push_back(std::move(md));
    return (int)o->materials.size() - 1;
}

int oracle_create_mesh(void *h, const Vertex *v, uint64_t nv, const uint32_t *idx, uint64_t ni, uint64_t material) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || !v || !idx || nv == 0 || ni == 0 || ni % 3 != 0 || material >= o->materials.size()) return -1;
    MeshData m;
    m.verts.assign(v, v + nv); m.indices.assign(idx, idx + ni); m.material = material;
    o->meshes.push_back(std::move(m));
    return (int)o->meshes.size() - 1;
}

// renderer.cpp:585-603 (clamp to the cap)
int oracle_update_lights(void *h, const PointLight *l, uint64_t n) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || (n && !l)) return -1;
    uint64_t k = std::min<uint64_t>(n, o->max_lights);
    o->lights.assign(l, l + k);
    return 0;
}

int oracle_pass_shadow_map(void *h, const Scene *sc) { Oracle *o = static_cast<Oracle *>(h); if (!o || !sc) return -1; pass_shadow_map(*o, *sc); return 0; }
int oracle_pass_gbuffer(void *h, const Scene *sc) { Oracle *o = static_cast<Oracle *>(h); if (!o || !sc) return -1; pass_gbuffer(*o, *sc); return 0; }

int oracle_pass_shade(void *h, const Scene *sc, const Settings *st, int threads) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || !sc || !st) return -1;
    size_t npx = (size_t)o->width * o->rows();
    if (o->attrs.size() != npx * 18) return -4;
    o->hdr.resize(npx * 3); o->ldr.resize(npx * 3); o->rgba8.resize(npx * 4);
    shade_parallel(*o, *sc, *st, o->attrs.data(), o->matid.data(), o->rows(), o->row_begin, o->hdr.data(), o->ldr.data(), o->rgba8.data(), threads);
    return 0;
}

// shade a caller-provided G-buffer stripe of `rows` rows (row-major attrs[rows*W*18], matid[rows*W]);
// outputs may be NULL.  Used for the timed cpu_baseline and for full-size shading parity.
int oracle_shade_gbuffer(void *h, const Scene *sc, const Settings *st, const float *attrs, const uint32_t *matid,
                         uint32_t rows, float *hdr, float *ldr, uint8_t *rgba8, int threads) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || !sc || !st || !attrs || !matid) return -1;
    shade_parallel(*o, *sc, *st, attrs, matid, rows, 0, hdr, ldr, rgba8, threads);   // stripe starts at frame row 0
    return 0;
}

int oracle_render_frame(void *h, const Scene *sc, const Settings *st, uint8_t *out_rgba8, int threads) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || !sc || !st) return -1;
    pass_shadow_map(*o, *sc);
    pass_gbuffer(*o, *sc);
    int rc = oracle_pass_shade(h, sc, st, threads);
    if (rc) return rc;
    if (out_rgba8) std::memcpy(out_rgba8, o->rgba8.data(), o->rgba8.size());
    return 0;
}

int oracle_read_gbuffer(void *h, float *attrs, uint32_t *material, float *depth, uint32_t *tri) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o) return -1;
    size_t npx = (size_t)o->width * o->rows();
    if (o->attrs.size() != npx * 18) return -4;
    if (attrs) std::memcpy(attrs, o->attrs.data(), npx * 18 * 4);
    if (material) std::memcpy(material, o->matid.data(), npx * 4);
    if (depth) std::memcpy(depth, o->depth.data(), npx * 4);
    if (tri) std::memcpy(tri, o->src_tri.data(), npx * 4);
    return 0;
}
int oracle_write_gbuffer(void *h, const float *attrs, const uint32_t *material) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || !attrs || !material) return -1;
    size_t npx = (size_t)o->width * o->rows();
    o->attrs.assign(attrs, attrs + npx * 18); o->matid.assign(material, material + npx);
    o->depth.assign(npx, 0.0f); o->src_tri.assign(npx, 0); o->tri_id.assign(npx, 0);
    return 0;
}
int oracle_read_shadow_map(void *h, float *d) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || !d || !o->shadow_size) return -1;
    std::memcpy(d, o->shadow.data(), o->shadow.size() * 4); return 0;
}
int oracle_write_shadow_map(void *h, const float *d) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || !d || !o->shadow_size) return -1;
    o->shadow.assign(d, d + (size_t)o->shadow_size * o->shadow_size); return 0;
}
int oracle_read_output(void *h, float *ldr, float *hdr, uint8_t *rgba8) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o) return -1;
    size_t npx = (size_t)o->width * o->rows();
    if (o->ldr.size() != npx * 3) return -4;
    if (ldr) std::memcpy(ldr, o->ldr.data(), npx * 12);
    if (hdr) std::memcpy(hdr, o->hdr.data(), npx * 12);
    if (rgba8) std::memcpy(rgba8, o->rgba8.data(), npx * 4);
    return 0;
}
int oracle_stats(void *h, uint64_t *out, uint32_t n) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || !out) return -1;
    for (uint32_t i = 0; i < n && i < 6; ++i) out[i] = o->stats[i];
    return 0;
}
int oracle_frame_constants(const Scene *sc, float *pv, float *lpv, float *sun_dir) {
    if (!sc) return -1;
    M4 a = camera_proj_view(sc->camera), b = sun_proj_view(sc->sun);
    V3 d = dir_from_rot(sc->sun.rotation);
    if (pv) std::memcpy(pv, &a, 64);
    if (lpv) std::memcpy(lpv, &b, 64);
    if (sun_dir) { sun_dir[0] = d.x; sun_dir[1] = d.y; sun_dir[2] = d.z; }
    return 0;
}

// ---- unit entry points for the known-answer tests ---------------------------
void oracle_dir_from_rot(const float rot[2], float out[3]) { V3 d = dir_from_rot(rot); out[0] = d.x; out[1] = d.y; out[2] = d.z; }
void oracle_outgoing_radiance(const float n[3], const float wo[3], const float wi[3], const float Li[3], const float base[3],
                              float metal, float rough, float out[3]) {
    Vec3<float> r = calculate_outgoing_radiance<float>(vec3<float>(n[0], n[1], n[2]), vec3<float>(wo[0], wo[1], wo[2]), vec3<float>(wi[0], wi[1], wi[2]),
                                                       vec3<float>(Li[0], Li[1], Li[2]), vec3<float>(base[0], base[1], base[2]), metal, rough);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_tonemap(int32_t method, float gamma, float exposure, const float in[3], float tm[3], float out[3]) {
    Settings st{method, gamma, exposure};
    Vec3<float> c = vec3<float>(in[0], in[1], in[2]);
    Vec3<float> t = tonemap_only<float>(c, st);
    Vec3<float> r = post_process<float>(c, st);
    if (tm) { tm[0] = t.x; tm[1] = t.y; tm[2] = t.z; }
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float oracle_calculate_shadow(const float *map, uint32_t S, const float ls[4]) { return calculate_shadow(map, S, V4{ls[0], ls[1], ls[2], ls[3]}); }
// fetch_surface on material `mat` with identity TBN: out = base3, tangent-space normal before tbn/normalise 3, n3, metal, rough
int oracle_fetch_surface(void *h, uint32_t mat, float u, float v, const float tbn[9], float out[11]) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || mat >= o->materials.size()) return -1;
    float attr[18] = {0};
    attr[0] = u; attr[1] = v;
    for (int i = 0; i < 9; ++i) attr[2 + i] = tbn[i];
    SurfaceR<float> s = fetch_surface<float>(*o, o->materials[mat], attr);
    float nm[4]; sample_rgba8_r<float>(*o, o->materials[mat].normal, u, v, false, nm);
    out[0] = s.base.x; out[1] = s.base.y; out[2] = s.base.z;
    out[3] = nm[0] * 2.0f - 1.0f; out[4] = (1.0f - nm[1]) * 2.0f - 1.0f; out[5] = nm[2] * 2.0f - 1.0f;
    out[6] = s.n.x; out[7] = s.n.y; out[8] = s.n.z; out[9] = s.metal; out[10] = s.rough;
    return 0;
}
uint8_t oracle_to_unorm8(float x) { return to_unorm8(x); }
// renderer.cpp:555-583 create_hdri: RGBA32F environment map
int oracle_create_hdri(void *h, const float *rgba, uint32_t w, uint32_t hh) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || !rgba || !w || !hh) return -1;
    o->env.assign(rgba, rgba + (size_t)w * hh * 4); o->env_w = w; o->env_h = hh;
    return 0;
}
// unit entry points: the view ray of pixel (x, y) and the environment lookup along a direction (uv_out: the float64 uv)
void oracle_sky_ray(void *h, const Camera *c, uint32_t x, uint32_t y, float out[3]) {
    Oracle *o = static_cast<Oracle *>(h);
    V3 d = sky_ray(sky_basis(*c), x, y, o->width, o->height);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
void oracle_sample_environment(void *h, const float dir[3], double out_rgb[3], double uv_out[2]) {
    Oracle *o = static_cast<Oracle *>(h);
    Vec3<double> c = sample_environment<double>(*o, v3(dir[0], dir[1], dir[2]), uv_out);
    out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
}
int oracle_set_hdr16(void *h, int on) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o) return -1;
    o->hdr16 = on != 0;
    return 0;
}
float oracle_through_half(float x) { return through_half(x); }
// 64 (default): BRDF + tonemap in float64, the parity arbiter; 32: the literal fp32 restatement (timed CPU baseline)
int oracle_set_precision(void *h, int bits) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || (bits != 32 && bits != 64)) return -1;
    o->precision = bits;
    return 0;
}
// variants of the sampler the reference leaves to the hardware (SAMPLER_* bits above); 0 = the semantics the build implements
int oracle_set_sampler_mode(void *h, int mode) {
    Oracle *o = static_cast<Oracle *>(h);
    if (!o || mode < 0 || mode > 7) return -1;
    o->sampler_mode = mode;
    return 0;
}
int oracle_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
