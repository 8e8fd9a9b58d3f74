/*
 * arctic_hip.h -- C-ABI of the MI355X-native forward PBR shading path.
 *
 * This is the drop-in boundary for the per-pixel work of arctic-renderer's
 * shaders/forward.hlsl + shaders/post_process.hlsl.  The reference has no FFI;
 * the seam is the public surface of class Arctic::Renderer::Renderer
 * (reference src/renderer/renderer.hpp:94-125), the only thing src/app.cpp
 * calls.  Every entry point below names the reference call it replaces.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no C++/torch types.
 *   - every fallible call returns int: >= 0 ok (an index where one is
 *     documented), < 0 an ARCTIC_E_* code.  The reference returns
 *     [[nodiscard]] bool and logs (src/renderer/dxerr.hpp:5-10); here the
 *     message is kept per handle and read with arctic_last_error().
 *   - inputs are borrowed for the duration of the call and copied to device
 *     memory synchronously (reference: blocking fence inside every upload,
 *     src/renderer/rhi.cpp:480-519); the handle owns all device memory.
 *   - a handle is not thread-safe (reference is single threaded).
 *   - matrices are 16 floats in glm memory order (column major), the same
 *     bytes the reference pushes as root constants
 *     (src/renderer/forward_pass.hpp:16-34).
 *   - there is NO CPU fallback: without a HIP device arctic_create() fails.
 */
#ifndef ARCTIC_HIP_H
#define ARCTIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes --------------------------------------------------------- */
#define ARCTIC_OK            0
#define ARCTIC_E_INVALID    -1 /* bad argument (null, zero size, index out of range) */
#define ARCTIC_E_DEVICE     -2 /* a HIP runtime call failed (message has hipGetErrorString) */
#define ARCTIC_E_NO_DEVICE  -3 /* no HIP device / device ordinal out of range */
#define ARCTIC_E_STATE      -4 /* call order (e.g. shade before any G-buffer exists) */
#define ARCTIC_E_CAPACITY   -5 /* more lights than max_lights, etc. */

/* ---- POD scene types: byte-compatible with src/renderer/scene.hpp -------- */

/* scene.hpp:20-38  Camera{vec3 eye; vec2 rotation; float aspect; float fov_y; array<float,2> z_near_far} */
typedef struct ArcticCamera {
    float eye[3];
    float rotation[2];   /* degrees: x = pitch, y = yaw (scene.cpp:9-19) */
    float aspect;
    float fov_y;         /* degrees */
    float z_near_far[2];
} ArcticCamera;

/* scene.hpp:40-47  Vertex, 14 floats = 56 B; input layout forward_pass.cpp:89-135 */
typedef struct ArcticVertex {
    float position[3];
    float normal[3];
    float tangent[3];
    float bitangent[3];
    float tex_coords[2];
} ArcticVertex;

/* scene.hpp:69-73  Object{mat4 trs; size_t mesh_idx} */
typedef struct ArcticObject {
    float    trs[16];    /* glm column-major */
    uint64_t mesh_idx;
} ArcticObject;

/* scene.hpp:75-84  DirectionalLight{vec3 position; vec2 rotation; vec3 color} */
typedef struct ArcticDirectionalLight {
    float position[3];
    float rotation[2];   /* degrees */
    float color[3];
} ArcticDirectionalLight;

/* scene.hpp:88-94  PointLight, 32 B with pads (== HLSL cbuffer packing, forward.hlsl:20-24) */
typedef struct ArcticPointLight {
    float    position[3];
    uint32_t padding0;
    float    color[3];
    uint32_t padding1;
} ArcticPointLight;

/* scene.hpp:96-103 Scene; std::vector members flattened to pointer + count.
 * point_lights here is ignored by render_frame exactly as in the reference
 * (renderer.cpp:285-407 uses the buffer last written by update_lights). */
typedef struct ArcticScene {
    ArcticCamera            camera;
    float                   ambient;
    ArcticDirectionalLight  sun;
    const ArcticPointLight *point_lights;
    uint64_t                n_point_lights;
    const ArcticObject     *objects;
    uint64_t                n_objects;
} ArcticScene;

/* scene.hpp:105-110 Settings{int tm_method=0; float gamma=2.2; float exposure=1} */
typedef struct ArcticSettings {
    int32_t tm_method;   /* 0 Reinhard (also any other value), 1 Exposure, 2 ACES: post_process.hlsl:76-89 */
    float   gamma;
    float   exposure;
} ArcticSettings;

#define ARCTIC_TM_REINHARD 0
#define ARCTIC_TM_EXPOSURE 1
#define ARCTIC_TM_ACES     2

/* create-time parameters: every compile-time constant of the reference that
 * BASELINE.json's configs vary becomes a field here. */
typedef struct ArcticCreateInfo {
    uint32_t width, height;  /* render-target size (renderer.hpp:94, App::WINDOW_WIDTH/HEIGHT app.hpp:20-21) */
    uint32_t shadow_size;    /* ShadowMapPass::SIZE = 4000 (shadow_map_pass.hpp:23); 0 = no shadow map, shadow term 0 */
    uint32_t max_lights;     /* Renderer::MAX_NUM_POINT_LIGHTS = 16 (renderer.hpp:22) */
    int32_t  device;         /* HIP device ordinal */
    uint32_t row_begin;      /* screen-space shard: this handle renders rows [row_begin,row_end) ... */
    uint32_t row_end;        /* ... of the width x height frame; 0,0 = whole frame */
    /* or an INTERLEAVED shard (load balance: lit regions are spatially clustered): with band_rows > 0 (a multiple of 8;
     * row_begin = row_end = 0) the handle owns the rows y with (y / band_rows) % shard_count == shard_index, and its
     * output holds those rows packed in ascending order. */
    uint32_t band_rows;
    uint32_t shard_index;
    uint32_t shard_count;
} ArcticCreateInfo;

typedef struct ArcticRenderer ArcticRenderer; /* opaque */

/* ---- life cycle ---------------------------------------------------------- */

/* replaces Renderer::Renderer(window,w,h) + bool Renderer::init()
 * (renderer.hpp:94-100, renderer.cpp:22-231); no window.  On failure returns
 * NULL and, if err/err_len given, writes the message there. */
ArcticRenderer *arctic_create(const ArcticCreateInfo *info, char *err, uint64_t err_len);

/* replaces Renderer::cleanup() + destructor (renderer.hpp:102) */
void arctic_destroy(ArcticRenderer *r);

/* last error message of this handle ("" if none); valid until the next call */
const char *arctic_last_error(const ArcticRenderer *r);

/* replaces bool Renderer::resize(uint32_t&,uint32_t&) (renderer.hpp:104).  Unlike
 * the reference (which only resizes the swapchain, renderer.cpp:241-272) this
 * really reallocates the targets; the row shard is reset to the whole frame. */
int arctic_resize(ArcticRenderer *r, uint32_t width, uint32_t height);

/* replaces bool Renderer::flush() (renderer.hpp:122-125): device idle */
int arctic_flush(ArcticRenderer *r);

/* run everything on a HIP stream the caller owns (a hipStream_t passed as void*; NULL is HIP's default stream, which is
 * what torch.cuda.current_stream() usually is), e.g. the stream a following RCCL gather is enqueued on, so that no host
 * synchronisation is needed between the frame and the collective.  arctic_use_own_stream() returns to the handle's
 * private stream.  The stream in use before the switch is drained first. */
int arctic_set_stream(ArcticRenderer *r, void *hip_stream);
int arctic_use_own_stream(ArcticRenderer *r);

/* ---- scene upload -------------------------------------------------------- */

/* replaces bool Renderer::create_material(void*,w,h, void*,w,h, void*,w,h)
 * (renderer.hpp:112-116, renderer.cpp:475-553): three tightly packed RGBA8
 * images; diffuse is sRGB, the other two linear.  Returns the material index
 * (= call order, like m_materials.emplace_back). */
int arctic_create_material(ArcticRenderer *r,
                           const void *diffuse, uint32_t diffuse_w, uint32_t diffuse_h,
                           const void *normal, uint32_t normal_w, uint32_t normal_h,
                           const void *metal_rough, uint32_t mr_w, uint32_t mr_h);

/* replaces bool Renderer::create_mesh(span<Vertex>, span<uint32_t>, MaterialIdx)
 * (renderer.hpp:109-110, renderer.cpp:417-473).  Returns the mesh index. */
int arctic_create_mesh(ArcticRenderer *r,
                       const ArcticVertex *vertices, uint64_t n_vertices,
                       const uint32_t *indices, uint64_t n_indices,
                       uint64_t material_idx);

/* replaces void Renderer::update_lights(span<PointLight>) (renderer.hpp:120,
 * renderer.cpp:585-603): clamps to max_lights like the reference clamps to 16. */
int arctic_update_lights(ArcticRenderer *r, const ArcticPointLight *lights, uint64_t n);

/* replaces bool Renderer::create_hdri(float*,w,h) (renderer.hpp:118, renderer.cpp:555-583): RGBA32F equirect
 * environment map.  Pixels without geometry then take it along their view ray (skybox.hlsl:61-90);
 * without a map they are black. */
int arctic_create_hdri(ArcticRenderer *r, const float *rgba32f, uint32_t w, uint32_t h);

/* ---- frames -------------------------------------------------------------- */

/* replaces bool Renderer::render_frame(const Scene&, const Settings&, build_ui)
 * (renderer.hpp:106-107, renderer.cpp:274-415) minus ImGui/present:
 * shadow-map raster -> visibility/G-buffer prepass -> shading + skybox fill
 * (+fused tonemap) -> RGBA8.  out_rgba8 is a HOST buffer of (row_end-row_begin)*width*4 bytes,
 * row-major, top-left origin; NULL = leave the frame on the device. */
int arctic_render_frame(ArcticRenderer *r, const ArcticScene *scene,
                        const ArcticSettings *settings, uint8_t *out_rgba8);

/* same, but the RGBA8 shard is written to DEVICE memory the caller owns
 * (e.g. a torch tensor that RCCL then gathers); stream-ordered on the
 * handle's stream, call arctic_flush() before another stream reads it. */
int arctic_render_frame_device(ArcticRenderer *r, const ArcticScene *scene,
                               const ArcticSettings *settings, void *d_out_rgba8);

/* ---- the passes one by one (bench + parity tests) ------------------------ */

/* ShadowMapPass::run (shadow_map_pass.cpp:113-169, depth.hlsl): light-view
 * depth-only raster, front faces culled, into the handle's shadow map. */
int arctic_pass_shadow_map(ArcticRenderer *r, const ArcticScene *scene);

/* ForwardPass::run's vertex + raster work (forward_pass.cpp:161-226,
 * forward.hlsl:50-66): visibility + G-buffer (the interpolated VSOut). */
int arctic_pass_gbuffer(ArcticRenderer *r, const ArcticScene *scene);

/* ps_main (forward.hlsl:208-235) + post_process main (post_process.hlsl:59-93)
 * over the resident G-buffer.  d_out_rgba8 may be NULL (handle's own buffer). */
int arctic_pass_shade(ArcticRenderer *r, const ArcticScene *scene,
                      const ArcticSettings *settings, void *d_out_rgba8);

/* PostProcessPass::run alone (post_process_pass.cpp:73-95): tonemap + gamma of
 * a float RGBA HDR image (host, w*h*4 floats) to RGBA8 (host). */
int arctic_post_process(ArcticRenderer *r, const float *hdr_rgba32f, uint32_t w, uint32_t h,
                        const ArcticSettings *settings, uint8_t *out_rgba8, float *out_ldr_rgb);

/* time `iters` back-to-back arctic_pass_shade launches with HIP events on the
 * handle's stream (after `warmup` untimed ones); ms_each gets iters floats. */
int arctic_time_shade(ArcticRenderer *r, const ArcticScene *scene, const ArcticSettings *settings,
                      uint32_t warmup, uint32_t iters, float *ms_each);

/* ---- read-back / injection for tests ------------------------------------- */

/* G-buffer of this shard, de-tiled to row-major: attrs = rows*width*18 floats
 * in VSOut order (uv2, tbn9 = t,b,n, world3, light_space4; forward.hlsl:41-48),
 * material = rows*width uint32 (0xFFFFFFFF = no geometry), depth = rows*width
 * floats (1.0 = clear), tri = rows*width uint32 draw-order id.  Any may be NULL. */
int arctic_read_gbuffer(ArcticRenderer *r, float *attrs, uint32_t *material, float *depth, uint32_t *tri);

/* inject a G-buffer (same row-major layout) instead of rasterising one */
int arctic_write_gbuffer(ArcticRenderer *r, const float *attrs, const uint32_t *material);

/* shadow map: shadow_size^2 floats, row-major */
int arctic_read_shadow_map(ArcticRenderer *r, float *depth);
int arctic_write_shadow_map(ArcticRenderer *r, const float *depth);

/* outputs of the last shade: float LDR (rows*width*3, after tonemap+gamma,
 * before the UNORM8 quantisation), float HDR (rows*width*3, ps_main's colour),
 * RGBA8 (rows*width*4).  Any may be NULL. */
int arctic_read_output(ArcticRenderer *r, float *ldr_rgb, float *hdr_rgb, uint8_t *rgba8);

/* the per-frame constants the host builds (forward_pass.cpp:166-177 via
 * scene.cpp:41-70): proj_view[16], light_proj_view[16], sun_dir[3] */
int arctic_frame_constants(const ArcticScene *scene, float *proj_view, float *light_proj_view, float *sun_dir);

/* counters of the last frame: [0] setup triangles (forward), [1] raster work
 * items (forward), [2] setup triangles (shadow), [3] raster work items
 * (shadow), [4] reserved; with ARCTIC_OPT_COUNT_LIGHT_EVALS: [5] point-light
 * evaluations summed over lit pixels, [6] lit pixels (1 - shadow != 0),
 * [7] evaluations with n.wi > 0 (the others contribute exactly 0,
 * forward.hlsl:191-192), [8] (tile, light) pairs whose n.wi <= 0 in every lit
 * pixel of the 8x8 tile (what a per-tile light list could skip), [9] tiles
 * with a lit pixel; [10], [11] work items of the forward / shadow pass that
 * went through the atomicMin rasteriser (all of them with
 * ARCTIC_OPT_RASTER_OWNER = 0; with block ownership those that found their
 * block's bin full or whose record takes the integer path); with
 * ARCTIC_OPT_COUNT_LIGHT_EVALS also [12] tiles of the fast path with a pixel
 * the shadow map's min/max table left undecided, [13] such pixels, [14] tiles
 * that ran the 25 PCF compares, [15] pixels that did.  n <= 16. */
int arctic_stats(ArcticRenderer *r, uint64_t *out, uint32_t n);

/* tuning / debug switches. */
#define ARCTIC_OPT_KEEP_FLOAT_OUTPUT 1 /* 1 = shade also stores float LDR+HDR planes (tests); 0 = RGBA8 only (bench) */
#define ARCTIC_OPT_COUNT_LIGHT_EVALS 2 /* 1 = the shading pass runs its counting variant: stats[5..9] (a few atomics per lit tile; slower) */
#define ARCTIC_OPT_CULLING           3 /* 0 = run the light loop for every covered pixel; 1 (default) = exact culling: fully shadowed pixels skip it
                                          (every term of ps_main carries 1 - shadow, forward.hlsl:222,230) */
#define ARCTIC_OPT_DEBUG             4 /* timing experiments only: bit 0 skip material textures, bit 1 skip shadow test, bit 2 skip tonemap (wrong images);
                                          bit 3 shadow test without the min/max table (same image); bit 4 the 25-tap PCF path of tiles on a shadow
                                          edge reads a per-wave LDS tile instead of a per-lane register window (same image; measured A/B);
                                          bit 5 every triangle through the 64-bit integer rasteriser instead of the binary64 planes (same image: the path
                                          of triangles with snapped coordinates of 2^24 and more); bit 6 whole frames gather records and vertices through 64-bit
                                          pointers (the path of tables of 4 GiB and more) instead of 32-bit offsets (same image); bit 7 arctic_render_frame draws the
                                          shadow map on the main stream before the visibility prepass instead of beside it on a second stream (same image);
                                          bit 8 every tile through the general tile code, none through the fast tile (same image; A/B and tests);
                                          bit 9 the bins of the block owners count every work item offered, full or not (arctic_read_bin_counts as a histogram; same image);
                                          bit 10 the prepasses count the workgroups ARCTIC_OPT_CLUSTER_CULL skipped (arctic_read_cull_counts; same image) */
#define ARCTIC_OPT_HDR16             6 /* 1 = round ps_main's colour through binary16 before post_process, like the reference's
                                        R16G16B16A16_FLOAT colour target (forward_pass.cpp:149, renderer.cpp:128-144); default 0 = fp32 */
#define ARCTIC_OPT_SHADOW_CACHE      9 /* 1 (default) = arctic_render_frame redraws the shadow map only when the sun, the objects or the mesh list changed
                                          (byte-compared); 0 = every frame like the reference (renderer.cpp:300-337).  Same image either way. */
#define ARCTIC_OPT_FRAMES_IN_FLIGHT  15 /* 2 = arctic_render_frame runs the visibility prepass of a frame on a second stream, into a second set of
                                          tables (and, when the shadow map is redrawn, into a second map), beside the shading of the frame before it; 3 (what the
                                          reference keeps, rhi.hpp:25) = three sets, consecutive prepasses on two streams, so that two prepasses overlap as well -- for
                                          small targets, whose prepass is the longer chain; 0 (default) = the library's choice: 3 below 3 Mpx, 2 above; every call
                                          still enqueues one whole frame and the output is complete after arctic_flush / in stream order on the main
                                          stream.  1 = one frame at a time on one stream.  Same images. */
#define ARCTIC_OPT_VISBUFFER        10 /* 1 (default) = arctic_render_frame shades straight from the visibility plane, no 76 B/px G-buffer round trip
                                          (bit-identical image; the G-buffer is materialised later if arctic_read_gbuffer / arctic_pass_shade ask); 0 = via the G-buffer */
#define ARCTIC_OPT_ITEM_TABLE_FLOOR 11 /* smallest size (entries) of the rasteriser's work-item table, default 4 Mi; the table grows to 4x the largest
                                          count seen.  A frame that overflows it returns ARCTIC_E_CAPACITY from the next synchronising call. */
#define ARCTIC_OPT_LIGHT_PATH       12 /* the light loop of the shading kernel: 0 = automatic (default: scalar up to 12 point lights, packed pairs above -- the
                                          measured crossover; the reference's MAX_NUM_POINT_LIGHTS is 16), 1 = scalar fp32, 2 = two lights at a time in packed
                                          fp32; both read the lights through the scalar cache.  Same formulas (images agree to fp32 rounding, ~1e-7). */
#define ARCTIC_OPT_TILES_PER_WAVE    16 /* tiles a wave of the shading pass (arctic_pass_shade) shades one after the other, 1 / n-th of the frame's height apart:
                                         lit (ALU-bound) and shadowed (latency-bound) regions are spatially clustered, and a wave that visits n distant parts of the
                                         frame carries a mix of both, so that every SIMD holds both kinds all the time, wherever in the frame the light falls
                                         (and a wave is launched once for n tiles).  0 (default) = the library's choice: 2, and 1 for targets below ~3 Mpx, where a frame is
                                         only a few rounds of the chip's wave slots.  Placement only: same image */
#define ARCTIC_OPT_TILE_TRACE        17 /* 1 = the shading pass records per 8x8 tile when its wave started and ended and where it ran (a measuring aid, default 0:
                                         the kernels then pay one wave-uniform branch at either end of a tile); read with arctic_read_tile_trace */
#define ARCTIC_OPT_RASTER_OWNER      18 /* -1 (default) = the library's choice: the forward prepass of a handle that owns 4 Mpx of the frame or more; otherwise
                                         bit 0: the forward prepass, bit 1: the shadow pass -- the rasteriser gives every 16x16 block of
                                         its target ONE owner wave: work items are handed to per-block bins, the owner merges its bin in registers and writes the
                                         block once (no clear, no early depth read, no per-pixel atomic); items that find their bin full, and records too large
                                         for the exact binary64 planes, go through the merging atomicMin rasteriser afterwards.  Bit clear = atomicMin rasteriser
                                         only (round 2; the shadow pass is instruction bound, not atomic bound: measured slower with owners).  Same visibility /
                                         shadow map, bit for bit (D3D12's fixed-function raster of forward_pass.cpp:137-151,212-224 / shadow_map_pass.cpp:96-97,157-167) */
#define ARCTIC_OPT_TILE_ORDER        19 /* 1 = arctic_pass_gbuffer leaves, next to the G-buffer, a cost class per 8x8 tile (can a pixel of it be lit at all,
                                          by the shadow map's min/max table -- the shading kernel's own first test) and from the classes the ORDER in which
                                          arctic_pass_shade hands out its work: lit tiles dealt evenly over the dispatch, shadowed ones in between, the end of
                                          the list shadowed ones only.  A hint: images never depend on it.  0 (default since round 5) = the geometric, XCD-aware
                                          order (takes effect at the next G-buffer pass; a G-buffer written by arctic_write_gbuffer has no order).  Off by default
                                          because it is a net loss as built: the order kernel is ONE workgroup (112 us per G-buffer pass at 4K) for <= 2 us of
                                          the shading pass, and strips dealt by cost no longer share an XCD's L2 with their neighbours (+83 MB of fabric reads
                                          per 4K pass: profiles/r5_a_traffic_tile_order.json) */
#define ARCTIC_OPT_ORDER_TAIL        20 /* per mille of the dispatch order, at its end, that holds cheap tiles only (default 60) */
#define ARCTIC_OPT_SAMPLER           21 /* which arithmetic stands in for the reference's D3D12 sampler (MIN_MAG_MIP_LINEAR + WRAP, forward_pass.cpp:38-51; the
                                          shadow map goes through the same sampler, forward.hlsl:84-92).  0 (default) = texel coordinates and bilinear weights in
                                          full fp32.  Bit 0: material textures with the scaled coordinate u W - 0.5 snapped to 1/256 texel (round to nearest)
                                          before it is split into texel index and weight -- 8-bit filter weights, what the D3D11.3 functional specification
                                          (3.2.4.1, 7.18.8) lets a sampler do and hardware does.  Bit 2: the same for the 25 PCF taps.  Same bits as the oracle's
                                          oracle_set_sampler_mode (its bit 1, sRGB decode after filtering, is an oracle-only bound: D3D10+ decodes first).
                                          A DX12 host comparing against captures of the real reference should pick 5; INTEGRATION.md section 7 */
#define ARCTIC_OPT_TEXTURE_TILING    22 /* how arctic_create_material stores the packed image of a material from now on: -1 (default) = the library's choice, tiles of
                                          4 x 4 texels (one 128-byte line each) for images of 2048 texels a side and more, row-major below; 0 = row-major; 1 = tiles.
                                          The reference creates one mip level (rhi.cpp:550), so large textures are minified at mip 0 and every pixel's footprint is its
                                          own cache lines: 2 in a row-major image, 1.56 on average in tiles.  A layout only: same texels, same image */
#define ARCTIC_OPT_CLUSTER_CULL      23 /* the prepasses skip whole workgroups of triangles / vertices whose object-space box (made per mesh by arctic_create_mesh) lies
                                          beyond a side of the pass's scissor rectangle, the near or the far plane -- the shard's rows for a row-range shard, the
                                          rank's slice of a sharded shadow map: 3 (default) = k_setup skips clusters of 256 triangles and k_vertex blocks of 256
                                          vertices all of whose triangles are skipped, 1 = k_setup only, 0 = off.  Conservative (geometry.hip: box_outside): same
                                          visibility plane, shadow map and counts, bit for bit.  No counterpart in the reference, which leaves culling to the
                                          hardware's clipper (forward_pass.cpp:212-224 draws every object) */
#define ARCTIC_OPT_SMALL_TRIANGLES   24 /* 1 (default) = the shadow pass draws triangles whose bounding box holds at most 64 pixels in its set-up kernel: the lanes
                                          that set a wave's triangles up share out the pixels of their boxes, the edge functions as 32-bit integers relative to the
                                          box, one atomicMin per covered pixel -- such a triangle (two thirds of what an orthographic sun sees of a tessellated
                                          scene) never becomes a record or a 16x16 work item.  0 = every triangle through the work-item rasteriser.  Same map, bit
                                          for bit (the hardware rasteriser of shadow_map_pass.cpp:96-97,157-167 / depth.hlsl:7-10 does not care either).
                                          Where the shadow pass runs with block owners (ARCTIC_OPT_RASTER_OWNER bit 1) the owners merge their bins into the map
                                          this path has drawn into, and blocks with an empty bin have no owner (measured: no faster than the atomic rasteriser) */
#define ARCTIC_OPT_MARKERS          13 /* 1 = roctx ranges around each pass, named like the reference's Tracy zones (process-wide; libroctx64 is loaded on demand) */
int arctic_set_option(ArcticRenderer *r, uint32_t option, int64_t value);

/* The trace of the latest shading pass under ARCTIC_OPT_TILE_TRACE: 4 x uint64 per tile, tile-row major over the handle's tile
   grid (tiles_x x tiles_y, returned too): the 100 MHz reference clock (s_memrealtime) when the tile's wave started, when it ended,
   HW_ID | XCC_ID << 32 (XCD / SE / CU / SIMD / wave slot it ran on), and 1 = shaded by the fast tile code | shader-clock ticks between
   start and end << 8.  out == NULL only reports the grid.  Synchronises.  No counterpart in
   the reference (its GPU timing is Tracy zones per pass, renderer.cpp:285-357); tools/experiments/tile_trace.py reads it. */
int arctic_read_tile_trace(ArcticRenderer *r, uint64_t *out, uint64_t capacity_tiles, uint32_t *tiles_x, uint32_t *tiles_y);

/* The dispatch order arctic_pass_gbuffer left for arctic_pass_shade (ARCTIC_OPT_TILE_ORDER) and the cost classes it was built from:
   order = ceil(tiles_x / 4) * tiles_y entries, one per strip of 4 horizontally adjacent tiles, ty << 16 | strip column, in the order
   the shading pass hands them out; tile_class = one byte per tile, row-major over the handle's tile grid, 1 = a pixel of the tile
   can be lit (or takes the environment lookup).  Either pointer may be NULL; both NULL only reports the grid; capacity_tiles >=
   tiles_x * tiles_y.  Synchronises.  A hint for the pass's scheduling, never part of a result; no counterpart in the reference
   (its pixel shader is scheduled by the GPU's fixed function, forward_pass.cpp:212-224). */
int arctic_read_tile_order(ArcticRenderer *r, uint32_t *order, uint8_t *tile_class, uint64_t capacity_tiles, uint32_t *tiles_x, uint32_t *tiles_y);

/* Work items per 16x16 block of the latest forward (shadow_pass = 0) or shadow (1) prepass drawn with block owners
   (ARCTIC_OPT_RASTER_OWNER): blocks_x x blocks_y counters, row-major over the whole target; a block's owner drew the first 32, the
   atomicMin rasteriser the rest.  out == NULL only reports the grid.  Synchronises.  A measuring aid (load balance of the owner
   waves, choice of the bin size); no counterpart in the reference, whose rasteriser is the GPU's fixed function
   (forward_pass.cpp:212-224). */
int arctic_read_bin_counts(ArcticRenderer *r, int shadow_pass, uint32_t *out, uint64_t capacity_blocks, uint32_t *blocks_x, uint32_t *blocks_y);

/* What ARCTIC_OPT_CLUSTER_CULL skipped in the latest forward (shadow_pass = 0) or shadow (1) prepass that ran under ARCTIC_OPT_DEBUG bit 10:
   out[0] = clusters of 256 triangles in the scene, out[1] = of them skipped by k_setup, out[2] = blocks of 256 vertices, out[3] = of them skipped by
   k_vertex.  Synchronises.  A measuring aid; no counterpart in the reference (forward_pass.cpp:212-224 draws every object). */
int arctic_read_cull_counts(ArcticRenderer *r, int shadow_pass, uint32_t *out);

/* The owner grid of a forward prepass as plain numbers -- pure host functions, no device, no handle (the library's kernels use the
   same definitions).  A handle created with these sizes (row range [row_begin, row_end), or -- band_rows > 0 -- the interleaved
   shard shard_index of shard_count) launches grid[0] x grid[1] owner waves, one per 16x16 block; arctic_owner_visit: grid row
   `grid_row` visits block row *block_row of the frame and stores its upper / lower 8-pixel tile row as the shard's tile row
   local_tile_rows[0] / [1] (-1: that tile row is not this shard's).  Every tile row a shard stores must be visited exactly once:
   tests/test_owner_grid.py checks that for worlds 1..8, bands of odd and even numbers of tile rows and row ranges that cut blocks. */
int arctic_owner_grid(uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t shard_index, uint32_t shard_count, uint32_t *grid);
int arctic_owner_visit(uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t shard_index, uint32_t shard_count,
                       uint32_t grid_row, uint32_t *block_row, int32_t *local_tile_rows);

/* library/ABI version: major*10000 + minor*100 + patch */
int arctic_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ARCTIC_HIP_H */
