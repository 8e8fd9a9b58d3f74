// renderer.cpp -- the C-ABI of include/arctic_hip.h: resource ownership + frame graph on one HIP stream.
//
// Plays the role of Arctic::Renderer::Renderer (reference src/renderer/renderer.{hpp,cpp}) for
// the hot path only: create_material/create_mesh/update_lights upload resources
// (renderer.cpp:417-603), render_frame runs shadow-map raster -> visibility/G-buffer prepass ->
// shading+tonemap (renderer.cpp:285-357 minus skybox, ImGui and present).  Everything D3D12
// (RHI, descriptor heaps, barriers, swapchain) is replaced by plain device allocations and
// stream order.  No CPU fallback: every entry point needs a HIP device.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <cmath>
#include <limits>
#include <string>
#include <vector>

#include "../../include/arctic_hip.h"
#include "../../include/arctic_dist.h"
#include "common.h"

using namespace arctic;

namespace {

// Optional profiler ranges named like the reference's Tracy zones ("Shadow Map Pass", "Forward Pass": shadow_map_pass.cpp:116,
// forward_pass.cpp:164): roctx is loaded at run time only when ARCTIC_OPT_MARKERS is set, so the library has no link-time
// dependency on it.  Host-side ranges around the enqueue of each pass (rocprofv3 --marker-trace shows them).
struct Markers {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false, on = false;
    void enable(bool want) {
        on = want;
        if (want && !tried) {
            tried = true;
            void *h = dlopen("libroctx64.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_LOCAL);
            if (h) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            }
        }
    }
    bool available() const { return push && pop; }
};
Markers g_markers;
struct Range {
    bool live;
    explicit Range(const char *name) : live(g_markers.on && g_markers.available()) { if (live) g_markers.push(name); }
    ~Range() { if (live) g_markers.pop(); }
};


// RCCL, loaded at run time like roctx (arctic_dist.h): the few entry points the exchange steps use.  Types are declared here
// the way rccl.h declares them (opaque communicator, 128-byte id by value, int enums) so the library needs no RCCL headers.
struct Rccl {
    struct UniqueId { char internal[ARCTIC_COMM_ID_BYTES]; };
    typedef void *Comm;
    static constexpr int Uint8 = 1, Uint32 = 3, Float32 = 7;   // ncclDataType_t (rccl.h:459-466)
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool tried = false;
    bool load() {
        if (!tried) {
            tried = true;
            // an RCCL the process has loaded already (a host that also runs torch.distributed: PyTorch-ROCm brings its own librccl.so.1)
            // is the one to use -- ONE library instance with two communicators, not two RCCL builds side by side on the same devices
            // ARCTIC_RCCL_LIB (honoured only when set): the library that provides the nccl* entry points instead -- the test suite's
            // loopback communicator (tests/cpp/loopback_rccl.cpp: ranks = threads of one process on one GPU, where RCCL refuses two
            // ranks per device), so that the R > 1 branch of the exchange runs where no second GPU exists.  No fallback when it is set.
            const char *override_lib = std::getenv("ARCTIC_RCCL_LIB");
            const bool overridden = override_lib && *override_lib;
            void *h = overridden ? dlopen(override_lib, RTLD_NOW | RTLD_LOCAL) : dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (overridden && !h) return false;
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
            if (h) {
                GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
                CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(h, "ncclCommInitRank"));
                CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
                AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(h, "ncclAllGather"));
                Send = reinterpret_cast<decltype(Send)>(dlsym(h, "ncclSend"));
                Recv = reinterpret_cast<decltype(Recv)>(dlsym(h, "ncclRecv"));
                GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(h, "ncclGroupStart"));
                GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(h, "ncclGroupEnd"));
                GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
            }
        }
        return GetUniqueId && CommInitRank && CommDestroy && AllGather && Send && Recv && GroupStart && GroupEnd;
    }
    const char *why(int rc) const { return GetErrorString ? GetErrorString(rc) : "RCCL error"; }
};
Rccl g_rccl;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// per-pass host->device tables (frame constants, object records, block tables): written into pinned memory and sent with
// ONE asynchronous copy; the pinned buffer is reused only after the copy that last read it has completed (event), so
// uploading never waits for the kernels of the previous pass or frame.
struct PassTables {
    void *h = nullptr;          // pinned staging
    size_t h_cap = 0;
    DevBuf d;                   // device arena
    hipEvent_t copied = nullptr;
    bool pending = false;
    std::vector<char> last;     // what the device arena holds: an unchanged object list is not uploaded again
    GeomParams gp;              // this pass's frame constants (kernel argument, by value)
    // device views into the arena (valid after upload)
    const ObjectRec *objs = nullptr;
    const uint32_t *vblock_obj = nullptr, *vblock_first = nullptr, *tblock_obj = nullptr, *tblock_first = nullptr;
    const float *vblock_bounds = nullptr, *tblock_bounds = nullptr;   // 6 floats per block (cluster_bounds)
};

struct Mesh {
    float *d_vertices = nullptr;
    uint32_t *d_indices = nullptr;
    uint32_t n_vertices = 0, n_indices = 0;
    uint64_t material = 0;
    std::vector<float> tbounds, vbounds;   // cluster_bounds: 6 floats per block of SETUP_THREADS triangles / of 256 vertices
};

// Object-space boxes for geometry.hip's box_outside, made once per mesh on the host.  tbounds[c] = {min xyz, max xyz} of the positions
// the triangles [c * SETUP_THREADS, (c + 1) * SETUP_THREADS) use -- one workgroup of k_setup --, vbounds[b] = the union of the boxes
// of every TRIANGLE that uses one of the vertices [256 b, 256 b + 256) -- one workgroup of k_vertex: a vertex may be skipped only when
// all its triangles are.  Triangles with an index out of range draw nothing (load_indices) and bound nothing; a block that holds a
// position that is not finite, or no triangle at all, gets an infinite box (never culled).
void cluster_bounds(const ArcticVertex *v, uint32_t n_vertices, const uint32_t *ind, uint32_t n_indices, std::vector<float> &tb, std::vector<float> &vb) {
    const uint32_t n_tri = n_indices / 3, n_tb = (n_tri + SETUP_THREADS - 1) / SETUP_THREADS, n_vb = (n_vertices + 255) / 256;
    const float inf = std::numeric_limits<float>::infinity();
    tb.assign((size_t)n_tb * 6, 0.0f); vb.assign((size_t)n_vb * 6, 0.0f);
    for (uint32_t b = 0; b < n_tb; ++b) for (int k = 0; k < 3; ++k) { tb[6 * b + k] = inf; tb[6 * b + 3 + k] = -inf; }
    for (uint32_t b = 0; b < n_vb; ++b) for (int k = 0; k < 3; ++k) { vb[6 * b + k] = inf; vb[6 * b + 3 + k] = -inf; }
    std::vector<char> bad_t(n_tb, 0), bad_v(n_vb, 0);
    for (uint32_t t = 0; t < n_tri; ++t) {
        const uint32_t i[3] = {ind[3 * t], ind[3 * t + 1], ind[3 * t + 2]};
        if (!(i[0] < n_vertices && i[1] < n_vertices && i[2] < n_vertices)) continue;
        float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        bool finite = true;
        for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
            const float p = v[i[j]].position[k];
            finite = finite && std::isfinite(p);
            lo[k] = std::min(lo[k], p); hi[k] = std::max(hi[k], p);
        }
        const auto grow = [&](std::vector<float> &box, std::vector<char> &bad, uint32_t b) {
            if (!finite) bad[b] = 1;
            for (int k = 0; k < 3; ++k) { box[6 * b + k] = std::min(box[6 * b + k], lo[k]); box[6 * b + 3 + k] = std::max(box[6 * b + 3 + k], hi[k]); }
        };
        grow(tb, bad_t, t / SETUP_THREADS);
        for (int j = 0; j < 3; ++j) if (j == 0 || (i[j] / 256 != i[0] / 256 && (j == 1 || i[j] / 256 != i[1] / 256))) grow(vb, bad_v, i[j] / 256);
    }
    const auto finish = [&](std::vector<float> &box, const std::vector<char> &bad, uint32_t n) {
        for (uint32_t b = 0; b < n; ++b)
            if (bad[b] || !(box[6 * b] <= box[6 * b + 3])) for (int k = 0; k < 3; ++k) { box[6 * b + k] = -inf; box[6 * b + 3 + k] = inf; }
    };
    finish(tb, bad_t, n_tb); finish(vb, bad_v, n_vb);
}

}  // namespace

struct ArcticRenderer {
    int device = 0;
    hipStream_t stream = nullptr;       // the stream every pass is enqueued on (own_stream unless the caller set one)
    hipStream_t own_stream = nullptr;
    uint32_t width = 0, height = 0, shadow_size = 0, max_lights = 0, row_begin = 0, row_end = 0;
    uint32_t tiles_x = 0, tiles_y = 0, tile_y0 = 0, row0_in_tile = 0;
    uint32_t band_rows = 0, shard_index = 0, shard_count = 1, owned_rows = 0;   // interleaved shard (band_rows > 0)
    std::vector<Mesh> meshes;
    std::vector<TexDesc> tex;        // 3 per material (device pointers)
    std::vector<void *> tex_allocs;
    DevBuf d_tex, d_lut, d_lights, d_light_pairs, d_env;
    // the shadow map, and the min/max of it per 4x4 texel block / per 4x4-aligned 8x8 block (k_shadow_bounds).  Two sets: a frame in
    // flight that redraws the map draws into the other one while the previous frame's shading still reads this one (the second set
    // is allocated when that first happens); `scur` names the map of the latest frame, the one every other call sees
    DevBuf d_shadow_set[2], d_shadow_blocks_set[2], d_shadow_bounds_set[2];
    bool bounds_valid_set[2] = {false, false};   // false whenever the map has been written since its table was built
    int scur = 0;
    hipEvent_t ev_shadow_released[2] = {nullptr, nullptr};   // like ev_released, for the shadow-map sets
    bool shadow_released_valid[2] = {false, false};
    DevBuf &d_shadow() { return d_shadow_set[scur]; }
    DevBuf &d_shadow_blocks() { return d_shadow_blocks_set[scur]; }
    DevBuf &d_shadow_bounds() { return d_shadow_bounds_set[scur]; }
    bool &bounds_valid() { return bounds_valid_set[scur]; }
    uint32_t env_w = 0, env_h = 0;
    uint32_t n_lights = 0;
    // frame targets
    DevBuf d_vis_set[3], d_p0, d_p1, d_p2, d_p3, d_p4, d_rgba8, d_ldr, d_hdr, d_counter;
    bool have_gbuffer = false, have_output = false, have_vis = false;   // have_vis: d_vis holds the visibility of the current G-buffer
    int light_path = 0;             // ARCTIC_OPT_LIGHT_PATH: 0 automatic, 1 scalar light loop, 2 packed pairs
    bool visbuffer = true;          // arctic_render_frame shades straight from the visibility plane (no G-buffer)
    // per-frame geometry scratch
    PassTables tables[4];   // [0], [2], [3] forward pass (one per frame in flight), [1] shadow pass
    // transformed vertices, records, work items: one set per pass ([0] forward, [1] shadow), so that the two prepasses of a frame
    // can run side by side (arctic_render_frame) and the forward pass's records outlive a shadow pass (k_resolve, k_material_vis)
    struct GeoSet {
        DevBuf d_xverts, d_recs, d_rrecs, d_clip_list, d_rec_of, d_items;
        DevBuf d_left;                     // ... and the work items the bins did not take (same capacity as d_items), for the atomic rasteriser
        DevBuf d_bin_count, d_bin_slots;   // block ownership (common.h: BinTables): a counter and BIN_SLOTS record indices per 16x16 block of the target
        uint32_t item_cap = 0;      // entries of d_items (work-item table of the rasteriser)
        uint32_t bins_x = 0, bins_y = 0;   // blocks of the latest owned pass (arctic_read_bin_counts)
        uint32_t n_tblocks = 0, n_vblocks = 0; bool cull_counted = false;   // the latest pass's workgroups (arctic_read_cull_counts)
    } geo[4];   // indexed like tables
    DevBuf d_geo_counters, d_stage;
    hipStream_t shadow_stream = nullptr;            // arctic_render_frame draws the shadow map here while the main stream runs the visibility prepass
    hipEvent_t ev_fork = nullptr, ev_shadow = nullptr;
    // The shadow pass has ONE set of scratch (geo[1], tables[1], its four counters) whichever stream it runs on -- the main stream
    // (arctic_pass_shadow_map, a sharded map, debug bit 7) or shadow_stream (whole frames).  ev_shadow_scratch marks the end of the
    // last shadow pass on shadow_scratch_stream; a pass on another stream waits for it first, so two never share the scratch.
    hipEvent_t ev_shadow_scratch = nullptr;
    hipStream_t shadow_scratch_stream = nullptr;
    bool shadow_scratch_valid = false;
    // Frames in flight (arctic_render_frame, ARCTIC_OPT_FRAMES_IN_FLIGHT = 2): the visibility prepass of frame k + 1 runs on
    // prepass_stream while the main stream still shades frame k, so there are two sets of what the prepass writes and the
    // shading reads -- visibility plane, vertex / record / item tables, object tables -- and `cur` names the set of the latest frame
    // (the one the pass-level calls and arctic_read_gbuffer see).  ev_released[s]: everything enqueued on the main stream up to
    // the moment the handle moved on from set s; the next prepass into s waits for it.
    // Up to three sets (the reference keeps 3 frames in flight, rhi.hpp:25) and two prepass streams used alternately: with three
    // sets the prepasses of two consecutive frames are independent of each other as well, and a small frame -- whose prepass is a
    // chain of launches longer than its shading -- is bound by neither chain alone.
    int frames_in_flight = 2, frames_in_flight_opt = 0 /* ARCTIC_OPT_FRAMES_IN_FLIGHT; 0: by the target's size (alloc_targets) */, cur = 0, prepass_turn = 0;
    hipStream_t prepass_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_prepass[2] = {nullptr, nullptr}, ev_released[3] = {nullptr, nullptr, nullptr};
    bool released_valid[3] = {false, false, false};
    DevBuf &d_vis() { return d_vis_set[cur]; }
    int fwd() const { return cur ? cur + 1 : 0; }      // index of the current forward set in tables / geo
    bool recs_worst_case = false;   // record table at 7 per source triangle (after an overflow of the 2-per-triangle table)
    uint32_t item_cap_floor = 1u << 22;   // its smallest size (ARCTIC_OPT_ITEM_TABLE_FLOOR; tests shrink it to reach the overflow path)
    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t edge_stats[4] = {0, 0, 0, 0};   // stats[12..15]: fast tiles the shadow table left undecided, their undecided pixels, tiles / pixels that ran the 25 compares
    uint64_t light_stats[2] = {0, 0};   // stats[8], [9]: (tile, light) pairs with n.wi <= 0 in every lit lane; tiles with a lit pixel
    int keep_float = 0, count_evals = 0, culling = 1, debug = 0, hdr16 = 0, tile_trace = 0;
    uint32_t tiles_per_wave = 0;     // ARCTIC_OPT_TILES_PER_WAVE (0 = the library's default)
    // The shading pass's dispatch order (round 4): k_resolve leaves a cost class per tile next to the G-buffer (can a pixel of the tile be
    // lit at all?), k_tile_order turns the classes into the order in which arctic_pass_shade hands out its strips of 4 tiles.  A hint:
    // a stale or missing order changes the pass's time, never its image.
    DevBuf d_tile_class, d_order_lists, d_tile_order;
    bool have_order = false;         // d_tile_order belongs to the G-buffer in place
    uint32_t order_group = 0, order_slots = 0;   // ... built for this many tiles per wave, this many slots (common.h order_slot)
    int texture_tiling = -1;         // ARCTIC_OPT_TEXTURE_TILING: materials created from now on: -1 = 4 x 4-texel tiles for images of 2048 texels a side and more, 0 = never, 1 = always
    int sampler = 0;                 // ARCTIC_OPT_SAMPLER: bit 0 material footprints, bit 2 PCF taps with coordinates snapped to 1/256 texel (D3D-style 8-bit filter weights)
    int tile_order = 0;              // ARCTIC_OPT_TILE_ORDER: 0 (default since round 5) = the geometric, XCD-aware order of round 3; 1 = the cost-class order of round 4.
                                     // Measured (profiles/r5_a_*): the order gains <= 2 us of the pass, its one-workgroup kernel costs the G-buffer pass 112 us, and handing strips
                                     // out by cost instead of by XCD row costs the pass 83 MB of fabric reads per launch (L2 hits 2.33 M -> 1.73 M: neighbouring strips no longer meet in one L2)
    uint32_t order_tail = 60;        // ARCTIC_OPT_ORDER_TAIL: the last part of the order (per mille) that holds cheap strips only
    DevBuf d_tile_trace;             // ARCTIC_OPT_TILE_TRACE: 4 x u64 per tile of the latest shading pass
    int cluster_cull = 3;            // ARCTIC_OPT_CLUSTER_CULL: bit 0 k_setup skips clusters, bit 1 k_vertex skips vertex blocks, that cannot touch the pass's pixels
    int small_triangles = 1;         // ARCTIC_OPT_SMALL_TRIANGLES: 1 = the shadow pass's k_setup draws triangles with a small bounding box itself (geometry.hip: draw_small)
    int raster_owner = -1;           // ARCTIC_OPT_RASTER_OWNER: bit 0 forward pass, bit 1 shadow pass: the blocks of the target are written once by owner waves
                                     // (k_bin + k_raster_owned) instead of per-pixel atomics; -1: the library's choice
    uint32_t raster_blocks[2] = {2048, 2048};  // persistent grid of k_raster: [0] forward pass, [1] shadow pass
    // render_frame re-renders the shadow map only when its inputs changed (sun, objects, meshes): the reference redraws it
    // every frame (renderer.cpp:300-337), but a depth map of unchanged geometry from an unchanged light is the same map
    std::vector<uint8_t> shadow_key; bool shadow_cache = true;
    uint32_t cu_count = 256;
    // multi-GPU exchange (arctic_dist.h): an RCCL communicator owned by the handle, a communication stream ordered against the
    // main stream with events, the shard layout of every rank (all-gathered once at arctic_comm_init)
    Rccl::Comm comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    hipStream_t comm_stream = nullptr;
    bool comm_stream_borrowed = false;   // the communication stream is the handle's own stream, idle because the caller brought its stream (one stream -- one hardware queue -- fewer)
    // ONE role per stream: own_stream is the main stream, or -- the caller brought its own -- EITHER the exchange's stream (borrowed by
    // arctic_comm_init) OR the second prepass stream of three frames in flight, never both: a gather queued in front of a prepass would
    // stall it behind the other ranks, and synchronising a prepass stream must never wait for an unmatched collective.
    bool own_stream_is_prepass() const { return stream != own_stream && !comm_stream_borrowed; }
    hipEvent_t ev_main = nullptr;
    struct InFlight { const void *ptr = nullptr; hipEvent_t done = nullptr; } inflight[4];   // gathers that still read a shard buffer
    std::vector<uint32_t> peer_rows, peer_ranges;    // rows of every rank's shard; [begin, end) of every rank (row-range shards)
    std::vector<uint64_t> peer_offset;               // byte offset of every rank's shard in the root's staging buffer (arctic_exchange_plan)
    uint64_t staging_bytes = 0;
    DevBuf d_staging, d_layout;                       // root: all shards back to back; ranges (2 u32 per rank) + byte offsets (u64 per rank)
    uint32_t layout_world = 0; bool layout_from_comm = false;
    bool shadow_sharded = false;                      // ARCTIC_OPT_SHADOW_SHARDED
    uint32_t *dh_counts = nullptr;  // the device's address of h_counts
    uint32_t *h_counts = nullptr;   // pinned, mapped: [0] records, [1] work items (forward), [2], [3] the same for the shadow pass, [4], [5] item-table overflow flags, [6], [7] work items drawn by the atomic rasteriser (forward, shadow)
    std::string err;

    int fail(int code, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
    uint32_t rows() const { return band_rows ? owned_rows : row_end - row_begin; }
    size_t n_tiles() const { return (size_t)tiles_x * tiles_y; }
    GBuffer gbuffer() const { return GBuffer{d_p0.as<float4>(), d_p4.as<float>(), d_p1.as<float4>(), d_p2.as<float4>(), d_p3.as<float4>()}; }
};

#define HIPCHECK(r, expr)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return (r)->fail(ARCTIC_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

// bytes of a shadow-map buffer: S rows, or -- with a communicator -- world * ceil(S / world) rows, the room the in-place all-gather
// of a sharded map writes (ARCTIC_OPT_SHADOW_SHARDED); + 8: the map is cleared in 8-byte words.  ONE definition for every place
// that allocates a map (arctic_create, the second map of frames in flight, arctic_comm_init).
size_t shadow_alloc_bytes(const ArcticRenderer *r) {
    const uint32_t S = r->shadow_size, w = (uint32_t)std::max(1, r->comm_world);
    const uint32_t rows = std::max(S, w * ((S + w - 1) / w));
    return (size_t)rows * S * 4 + 8;
}

int select_device(ArcticRenderer *r) {
    HIPCHECK(r, hipSetDevice(r->device));
    return ARCTIC_OK;
}

int alloc_targets(ArcticRenderer *r) {
    r->shadow_key.clear();   // buffers may move: render_frame redraws the shadow map
    shard_tile_rows(r->height, r->row_begin, r->row_end, r->band_rows, r->shard_index, r->shard_count, r->tile_y0, r->tiles_y);
    r->row0_in_tile = r->row_begin - r->tile_y0 * TILE;
    r->tiles_x = (r->width + TILE - 1) / TILE;
    if (r->band_rows) {   // interleaved shard: count the pixel rows this shard owns
        const int bt = (int)(r->band_rows / TILE), all = (int)((r->height + TILE - 1) / TILE);
        uint32_t own_r = 0;
        for (int ty = 0; ty < all; ++ty)
            if (row_owned(ty, bt, (int)r->shard_count, (int)r->shard_index)) own_r += std::min<uint32_t>(TILE, r->height - (uint32_t)ty * TILE);
        r->owned_rows = own_r;
    }
    size_t px = r->n_tiles() * TILE_PIXELS, out_px = (size_t)r->rows() * r->width;
    for (DevBuf &v : r->d_vis_set) HIPCHECK(r, v.ensure(px * 8));
    // the library's choice of frames in flight: 2, and 3 for targets below 3 Mpx, whose prepass -- a chain of launches -- is longer than
    // their shading (tools/experiments/in_flight.py, 2 / 3 in flight: 1080p config 2 0.077 / 0.065 ms, config 3 at 1080p 0.094 / 0.088,
    // config 1 0.030 / 0.026; 4K 0.278 / 0.284)
    if (r->frames_in_flight_opt == 0) { r->frames_in_flight = (uint64_t)r->rows() * r->width < 3000000ull ? 3 : 2; r->cur = 0; }
    r->released_valid[0] = r->released_valid[1] = r->released_valid[2] = false;   // (callers synchronise before they resize)
    HIPCHECK(r, r->d_p0.ensure(px * 16));
    HIPCHECK(r, r->d_p1.ensure(px * 16));
    HIPCHECK(r, r->d_p2.ensure(px * 16));
    HIPCHECK(r, r->d_p3.ensure(px * 16));
    HIPCHECK(r, r->d_p4.ensure(px * 12));
    HIPCHECK(r, r->d_rgba8.ensure(out_px * 4));
    HIPCHECK(r, r->d_counter.ensure(8 * N_SHADE_STATS));
    HIPCHECK(r, r->d_geo_counters.ensure(4 * N_GEO_COUNTERS * 4));   // N_GEO_COUNTERS words per table set
    r->have_gbuffer = r->have_output = r->have_vis = r->have_order = false;
    return ARCTIC_OK;
}

// Scene.objects -> ObjectRec[] + block tables in one asynchronous upload, skipped while the list stays byte-identical (a
// static scene under a moving camera: the frame constants travel as a kernel argument, so such frames copy nothing)
int upload_pass_tables(ArcticRenderer *r, PassTables &T, DevBuf &d_xverts, hipStream_t stream, const GeomParams &gp, const ArcticScene *sc, uint32_t &n_objs,
                       uint32_t &n_xverts, uint32_t &n_src_tris, uint32_t &n_vblocks, uint32_t &n_tblocks) {
    std::vector<ObjectRec> objs;
    std::vector<uint32_t> vb_obj, vb_first, tb_obj, tb_first;
    std::vector<float> vb_box, tb_box;
    uint64_t xv = 0, tri = 0;
    for (uint64_t i = 0; i < sc->n_objects; ++i) {
        const ArcticObject &o = sc->objects[i];
        if (o.mesh_idx >= r->meshes.size()) continue;   // the reference would index out of bounds; skipped here
        const Mesh &m = r->meshes[o.mesh_idx];
        ObjectRec rec;
        std::memcpy(rec.trs, o.trs, sizeof rec.trs);
        rec.vertices = m.d_vertices;
        rec.indices = m.d_indices;
        rec.n_vertices = m.n_vertices;
        rec.n_triangles = m.n_indices / 3;
        rec.first_xvert = (uint32_t)xv;
        rec.first_triangle = (uint32_t)tri;
        rec.material = (uint32_t)m.material;
        rec.pad = 0;
        uint32_t oi = (uint32_t)objs.size();
        for (uint32_t b = 0; b < rec.n_vertices; b += 256) { vb_obj.push_back(oi); vb_first.push_back(b); }
        for (uint32_t b = 0; b < rec.n_triangles; b += SETUP_THREADS) { tb_obj.push_back(oi); tb_first.push_back(b); }
        vb_box.insert(vb_box.end(), m.vbounds.begin(), m.vbounds.end());
        tb_box.insert(tb_box.end(), m.tbounds.begin(), m.tbounds.end());
        xv += rec.n_vertices;
        tri += rec.n_triangles;
        objs.push_back(rec);
    }
    if (xv > 0xFFFFFFF0ull || tri > 0xFFFFFFF0ull) return r->fail(ARCTIC_E_CAPACITY, "scene too large: %llu vertices, %llu triangles", (unsigned long long)xv, (unsigned long long)tri);
    n_objs = (uint32_t)objs.size(); n_xverts = (uint32_t)xv; n_src_tris = (uint32_t)tri;
    n_vblocks = (uint32_t)vb_obj.size(); n_tblocks = (uint32_t)tb_obj.size();
    // arena layout (16-byte aligned sections)
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_objs = 0, o_vo = align16(o_objs + objs.size() * sizeof(ObjectRec)),
                 o_vf = align16(o_vo + vb_obj.size() * 4), o_to = align16(o_vf + vb_first.size() * 4),
                 o_tf = align16(o_to + tb_obj.size() * 4), o_vb = align16(o_tf + tb_first.size() * 4), o_tb = align16(o_vb + vb_box.size() * 4),
                 total = align16(o_tb + tb_box.size() * 4) + 16;
    if (vb_box.size() != vb_obj.size() * 6 || tb_box.size() != tb_obj.size() * 6) return r->fail(ARCTIC_E_INVALID, "internal: cluster bounds out of step with the block tables");
    T.gp = gp;
    std::vector<char> stage(total, 0);
    char *h = stage.data();
    if (!objs.empty()) std::memcpy(h + o_objs, objs.data(), objs.size() * sizeof(ObjectRec));
    if (!vb_obj.empty()) { std::memcpy(h + o_vo, vb_obj.data(), vb_obj.size() * 4); std::memcpy(h + o_vf, vb_first.data(), vb_first.size() * 4); }
    if (!tb_obj.empty()) { std::memcpy(h + o_to, tb_obj.data(), tb_obj.size() * 4); std::memcpy(h + o_tf, tb_first.data(), tb_first.size() * 4); }
    if (!vb_box.empty()) std::memcpy(h + o_vb, vb_box.data(), vb_box.size() * 4);
    if (!tb_box.empty()) std::memcpy(h + o_tb, tb_box.data(), tb_box.size() * 4);
    if (!(T.d.p && stage == T.last)) {
        if (T.pending) { HIPCHECK(r, hipEventSynchronize(T.copied)); T.pending = false; }
        if (total > T.h_cap) {
            if (T.h) HIPCHECK(r, hipHostFree(T.h));
            T.h = nullptr; T.h_cap = 0;
            HIPCHECK(r, hipHostMalloc(&T.h, total * 2));
            T.h_cap = total * 2;
        }
        if (!T.copied) HIPCHECK(r, hipEventCreateWithFlags(&T.copied, hipEventDisableTiming));
        T.last.clear();
        HIPCHECK(r, T.d.ensure(total));
        std::memcpy(T.h, h, total);
        HIPCHECK(r, hipMemcpyAsync(T.d.p, T.h, total, hipMemcpyHostToDevice, stream));
        HIPCHECK(r, hipEventRecord(T.copied, stream));
        T.pending = true;
        T.last.swap(stage);
    }
    const char *d = T.d.as<char>();
    T.objs = reinterpret_cast<const ObjectRec *>(d + o_objs);
    T.vblock_obj = reinterpret_cast<const uint32_t *>(d + o_vo); T.vblock_first = reinterpret_cast<const uint32_t *>(d + o_vf);
    T.tblock_obj = reinterpret_cast<const uint32_t *>(d + o_to); T.tblock_first = reinterpret_cast<const uint32_t *>(d + o_tf);
    T.vblock_bounds = reinterpret_cast<const float *>(d + o_vb); T.tblock_bounds = reinterpret_cast<const float *>(d + o_tb);
    if (n_objs) {
        HIPCHECK(r, d_xverts.ensure((size_t)n_xverts * sizeof(XVert)));
    }
    return ARCTIC_OK;
}

// vertex -> clip/setup -> raster, shared by the forward prepass and the shadow pass
int run_geometry(ArcticRenderer *r, const ArcticScene *sc, bool shadow_pass, hipStream_t stream) {
    const int set = shadow_pass ? 1 : r->fwd();
    ArcticRenderer::GeoSet &G = r->geo[set];
    // the pass's target is cleared by the vertex kernel's launch (or by a fill when there is nothing to draw)
    unsigned long long *clear = shadow_pass ? r->d_shadow().as<unsigned long long>() : r->d_vis().as<unsigned long long>();
    const unsigned long long clear_value = shadow_pass ? 0x3F8000003F800000ull : ~0ull;   // depth 1.0 (shadow_map_pass.cpp:124-131) / no triangle
    const size_t clear_count = shadow_pass ? ((size_t)r->shadow_size * r->shadow_size + 1) / 2 : r->n_tiles() * TILE_PIXELS;
    GeomParams gp;
    std::memset(&gp, 0, sizeof gp);
    sun_proj_view(sc->sun.position, sc->sun.rotation, gp.light_from_world);
    if (shadow_pass) {
        std::memcpy(gp.clip_from_world, gp.light_from_world, sizeof gp.clip_from_world);
        gp.vp_w = gp.vp_h = (float)r->shadow_size;
        gp.sc_x0 = gp.sc_y0 = 0; gp.sc_x1 = gp.sc_y1 = (int32_t)r->shadow_size;
        if (r->shadow_sharded && r->comm && r->comm_world > 1) {   // this rank's slice of the light-space rows; the all-gather follows the raster
            const uint32_t per = (r->shadow_size + (uint32_t)r->comm_world - 1) / (uint32_t)r->comm_world;
            gp.sc_y0 = (int32_t)std::min(r->shadow_size, per * (uint32_t)r->comm_rank);
            gp.sc_y1 = (int32_t)std::min(r->shadow_size, per * ((uint32_t)r->comm_rank + 1));
        }
        gp.cull_front = 1;                                       // shadow_map_pass.cpp:96-97
        gp.tiles_x = (int32_t)((r->shadow_size + 7) / 8); gp.tile_y0 = 0; gp.pitch = (int32_t)r->shadow_size;
    } else {
        const ArcticCamera &c = sc->camera;
        camera_proj_view(c.eye, c.rotation, c.aspect, c.fov_y, c.z_near_far[0], c.z_near_far[1], gp.clip_from_world);
        gp.vp_w = (float)r->width; gp.vp_h = (float)r->height;
        gp.sc_x0 = 0; gp.sc_x1 = (int32_t)r->width; gp.sc_y0 = (int32_t)r->row_begin; gp.sc_y1 = (int32_t)r->row_end;
        gp.cull_front = 0;                                       // forward_pass.cpp:143-144
        gp.tiles_x = (int32_t)r->tiles_x; gp.tile_y0 = (int32_t)r->tile_y0; gp.pitch = 0;
        gp.band_tiles = (int32_t)(r->band_rows / TILE); gp.shard_index = (int32_t)r->shard_index; gp.shard_count = (int32_t)r->shard_count;
    }
    gp.raster_flags = ((r->debug & 32) ? 1 : 0) | ((r->debug & 1024) ? 2 : 0);
    gp.tiles_y = shadow_pass ? (int32_t)((r->shadow_size + 7) / 8) : (int32_t)r->tiles_y;
    // block ownership (ARCTIC_OPT_RASTER_OWNER; default -1 = the choice below): the blocks are written once by their owners, so nothing is cleared
    // the library's choice: the forward pass of a handle that owns 4 Mpx or more -- three launches instead of one cost ~12 us of fixed
    // time, what the atomics cost grows with the pixels.  Whole frames with / without owners: 4K 0.283 / 0.304 ms, one rank of R = 2 at 4K
    // (4.1 Mpx) 0.178 / 0.184, of R = 4 0.136 / 0.130, of R = 8 0.106 / 0.102; config 3 at 1080p 0.113 / 0.114, config 2 (1080p, dense
    // meshes: 14 % of the items beyond their bins) 0.106 / 0.093, config 1 (512^2) 0.064 / 0.038;
    // never the shadow pass (instruction bound, two thirds of its blocks empty: 0.124 -> 0.163 ms)
    const bool owned = r->raster_owner < 0 ? (!shadow_pass && (uint64_t)r->rows() * r->width >= 4000000ull) : (r->raster_owner & (shadow_pass ? 2 : 1)) != 0;
    // owners beside the small-triangle path: the map is cleared as without owners, k_setup draws the small triangles into it, the owners MERGE their bins into it
    const bool merge = owned && shadow_pass && r->small_triangles && !(r->debug & 32);
    BinTables B{};
    if (owned) {
        const uint32_t tw = shadow_pass ? r->shadow_size : r->width, th = shadow_pass ? r->shadow_size : r->height;
        B.blocks_x = (tw + 15) / 16;
        B.n_blocks = B.blocks_x * ((th + 15) / 16);
        if (shadow_pass) { B.grid_x = B.blocks_x; B.by0 = 0; B.grid_y = (th + 15) / 16; B.local_rows = 0; }
        else owner_grid(tw, th, r->tile_y0, r->tiles_y, (uint32_t)gp.band_tiles, B.grid_x, B.grid_y, B.by0, B.local_rows);
        HIPCHECK(r, G.d_bin_count.ensure((size_t)B.n_blocks * 4));
        HIPCHECK(r, G.d_bin_slots.ensure((size_t)B.n_blocks * BIN_SLOTS * 4));
        B.count = G.d_bin_count.as<uint32_t>(); B.slots = G.d_bin_slots.as<uint32_t>();
        B.count_all = (r->debug & 512) ? 1u : 0u;
        G.bins_x = B.blocks_x; G.bins_y = B.n_blocks / B.blocks_x;
    } else G.bins_x = G.bins_y = 0;
    PassTables &T = r->tables[set];
    uint32_t n_objs, n_xverts, n_src, n_vblocks, n_tblocks;
    int rc = upload_pass_tables(r, T, G.d_xverts, stream, gp, sc, n_objs, n_xverts, n_src, n_vblocks, n_tblocks);
    if (rc != ARCTIC_OK) return rc;
    const GeomParams &d_gp = T.gp;
    if (n_objs == 0 || n_src == 0 || n_vblocks == 0) {
        r->h_counts[shadow_pass ? 2 : 0] = r->h_counts[shadow_pass ? 3 : 1] = 0;
        HIPCHECK(r, launch_fill_u64(clear, clear_value, clear_count, stream));
        return ARCTIC_OK;
    }
    const ObjectRec *objs = T.objs;
    G.n_tblocks = n_tblocks; G.n_vblocks = n_vblocks; G.cull_counted = (r->debug & 1024) != 0;
    if (G.cull_counted) HIPCHECK(r, hipMemsetAsync(r->d_geo_counters.as<uint32_t>() + N_GEO_COUNTERS * set + 6, 0, 4, stream));   // (k_vertex counts into it from its first workgroup on)
    HIPCHECK(r, launch_vertex(objs, T.vblock_obj, T.vblock_first, n_vblocks, d_gp, G.d_xverts.as<XVert>(), shadow_pass ? 1 : 0,
                              r->d_geo_counters.as<uint32_t>() + N_GEO_COUNTERS * set, clear, clear_value, (owned && !merge) ? 0 : clear_count, B.count, owned ? B.n_blocks : 0,
                              (r->cluster_cull & 2) ? T.vblock_bounds : nullptr, stream));
    // Record slots: a triangle clipped against 6 planes yields at most 7 triangles, so 7 * n_src slots can never overflow.
    // Records and work items are allocated on the device from two counters (k_setup): no count pass, no scan, and neither
    // count has to come back to the host -- the frame stays asynchronous.
    const uint64_t slots64 = 7ull * n_src;
    if (slots64 > 0x0FFFFFF0ull) return r->fail(ARCTIC_E_CAPACITY, "scene too large: %u triangles", n_src);
    // ... but 7 x 128 B per source triangle is a lot of memory for what only heavy clipping can produce: the table starts at
    // 2 records per source triangle and goes to the worst case only after a frame has overflowed it (flagged like the item table)
    const uint32_t n_slots = r->recs_worst_case ? (uint32_t)slots64 : (uint32_t)std::min<uint64_t>(slots64, 2ull * n_src + 4096);
    // work-item table: explicit (record, 16x16 block) pairs.  Its size follows the largest count seen so far (pinned
    // h_counts, refreshed asynchronously each pass) with 4x headroom; an overflow drops work, is flagged by the kernel and
    // reported by the next call that synchronises (arctic_flush / read-backs) -- and the table has grown by then.
    const uint64_t want = std::max<uint64_t>(r->item_cap_floor, 4ull * r->h_counts[shadow_pass ? 3 : 1]);
    if (want > G.item_cap) {
        HIPCHECK(r, G.d_items.ensure((size_t)want * 8));
        G.item_cap = (uint32_t)std::min<uint64_t>(want, 0x7FFFFFF0ull);
    }
    if (owned) HIPCHECK(r, G.d_left.ensure((size_t)G.item_cap * 8));
    HIPCHECK(r, G.d_recs.ensure((size_t)n_slots * sizeof(SetupRec)));
    HIPCHECK(r, G.d_rrecs.ensure((size_t)n_slots * sizeof(RasterRec)));
    HIPCHECK(r, G.d_rec_of.ensure((size_t)n_src * 8 * 4));
    HIPCHECK(r, G.d_clip_list.ensure((size_t)n_src * 8));
    uint32_t *counters = r->d_geo_counters.as<uint32_t>() + N_GEO_COUNTERS * set;   // zeroed by k_vertex
    HIPCHECK(r, launch_setup(objs, T.tblock_obj, T.tblock_first, n_tblocks, d_gp, G.d_xverts.as<XVert>(), G.d_recs.as<SetupRec>(),
                             G.d_rrecs.as<RasterRec>(), G.d_rec_of.as<uint32_t>(), G.d_items.as<uint2>(), G.item_cap, n_slots, counters,
                             G.d_clip_list.as<uint2>(), (r->cluster_cull & 1) ? T.tblock_bounds : nullptr,
                             // (block owners STORE their blocks -- that store is the clear --, so nothing may be drawn before them unless they merge)
                             shadow_pass && (!owned || merge) && r->small_triangles ? r->d_shadow().as<uint32_t>() : nullptr, stream));
    // counts for arctic_stats() and the overflow flag: k_raster stores them into pinned, mapped memory (no copy launches between
    // the kernels); looked at only when the stream has been synchronised
    uint32_t *dh = r->dh_counts + (shadow_pass ? 2 : 0), *dh_overflow = r->dh_counts + 4 + (shadow_pass ? 1 : 0);
    uint32_t grid = r->raster_blocks[shadow_pass ? 1 : 0];
    const uint2 *draw = G.d_items.as<uint2>();
    if (owned) {
        HIPCHECK(r, launch_raster_owned(shadow_pass, G.d_rrecs.as<RasterRec>(), G.d_items.as<uint2>(), G.d_left.as<uint2>(), G.item_cap, counters, B, d_gp,
                                        shadow_pass ? nullptr : r->d_vis().as<unsigned long long>(), shadow_pass ? r->d_shadow().as<uint32_t>() : nullptr, merge, stream));
        // what the bins left goes through the atomic rasteriser: when the previous pass left next to nothing (pinned h_counts, a frame
        // late: it only sizes the grid) a launch of 64 workgroups, else the whole persistent grid -- the kernel shortens its chunks
        // so that every resident wave gets one (a grid cut to the item count measured slower: config 1, half of whose items are left)
        draw = G.d_left.as<uint2>();
        if (r->h_counts[6 + (shadow_pass ? 1 : 0)] <= 64) grid = std::min<uint32_t>(grid, 64);
    }
    if (shadow_pass)
        HIPCHECK(r, launch_raster_depth(G.d_recs.as<SetupRec>(), G.d_rrecs.as<RasterRec>(), draw, G.item_cap, counters, grid, d_gp,
                                        r->d_shadow().as<uint32_t>(), dh, dh_overflow, owned, stream));
    else
        HIPCHECK(r, launch_raster_vis(G.d_recs.as<SetupRec>(), G.d_rrecs.as<RasterRec>(), draw, G.item_cap, counters, grid, d_gp,
                                      r->d_vis().as<unsigned long long>(), dh, dh_overflow, owned, stream));
    return ARCTIC_OK;
}

// the min/max table of the shadow map (shade.hip: the test that replaces the 25 PCF taps for most pixels): rebuilt, two small
// launches, after the map was written
int build_shadow_bounds(ArcticRenderer *r, hipStream_t stream) {
    const uint32_t nb = shadow_bounds_pitch(r->shadow_size);
    if (!nb || r->bounds_valid()) return ARCTIC_OK;
    HIPCHECK(r, r->d_shadow_blocks().ensure((size_t)nb * nb * 8));
    HIPCHECK(r, r->d_shadow_bounds().ensure((size_t)nb * nb * 8));
    HIPCHECK(r, launch_shadow_bounds(r->d_shadow().as<float>(), r->shadow_size, r->d_shadow_blocks().as<float2>(), r->d_shadow_bounds().as<float2>(), stream));
    r->bounds_valid() = true;
    return ARCTIC_OK;
}

int pass_shadow_map(ArcticRenderer *r, const ArcticScene *sc, hipStream_t stream) {
    if (r->shadow_size == 0) return ARCTIC_OK;
    Range zone("Shadow Map Pass");
    r->bounds_valid() = false;
    if (r->shadow_scratch_valid && r->shadow_scratch_stream != stream)   // the previous shadow pass ran on another stream: it owns the scratch until it ends
        HIPCHECK(r, hipStreamWaitEvent(stream, r->ev_shadow_scratch, 0));
    int rc = run_geometry(r, sc, true, stream);
    HIPCHECK(r, hipEventRecord(r->ev_shadow_scratch, stream));
    r->shadow_scratch_stream = stream; r->shadow_scratch_valid = true;
    if (rc != ARCTIC_OK) return rc;
    if (r->shadow_sharded && r->comm && r->comm_world > 1) {
        // every rank has drawn ceil(S / world) rows of the map: one in-place all-gather (the send buffer is this rank's slice of the
        // receive buffer) completes it everywhere, on the pass's stream -- the shading pass needs it next
        const size_t per = (size_t)((r->shadow_size + (uint32_t)r->comm_world - 1) / (uint32_t)r->comm_world) * r->shadow_size;
        float *base = r->d_shadow().as<float>();
        if (r->d_shadow().cap < per * (size_t)r->comm_world * 4)   // (cannot happen through this API: every map is sized by shadow_alloc_bytes)
            return r->fail(ARCTIC_E_STATE, "sharded shadow map: the map buffer has no room for %d x %zu texels", r->comm_world, per);
        const int nrc = g_rccl.AllGather(base + per * (size_t)r->comm_rank, base, per, Rccl::Float32, r->comm, stream);
        if (nrc != 0) return r->fail(ARCTIC_E_DEVICE, "ncclAllGather(shadow map): %s", g_rccl.why(nrc));
    }
    return ARCTIC_OK;
}

// visibility only: vertex -> setup -> raster of the camera view
int pass_visibility(ArcticRenderer *r, const ArcticScene *sc, hipStream_t stream) {
    Range zone("Forward Pass: visibility");
    r->have_gbuffer = false; r->have_order = false;
    int rc = run_geometry(r, sc, false, stream);
    if (rc != ARCTIC_OK) return rc;
    r->have_vis = true;
    return ARCTIC_OK;
}

// visibility -> the 76 B/pixel G-buffer.  Valid while the records of the forward pass are still in place.
int resolve_gbuffer(ArcticRenderer *r) {
    Range zone("Forward Pass: G-buffer");
    if (!r->have_vis)
        return r->fail(ARCTIC_E_STATE, "no G-buffer: no visibility plane to resolve it from (run arctic_pass_gbuffer)");
    // the cost classes need the shadow map's min/max table as the shading pass will read it (rebuilt here if the map was written since)
    TileHint hint = {nullptr, nullptr, 0, 0, 0};
    const uint32_t bpr = (r->tiles_x + 3) / 4, n_jobs = bpr * r->tiles_y;
    r->have_order = false;
    // (the order's slots come in groups of the tiles a wave of the pass will shade: launch_shade's rule)
    r->order_group = r->tiles_per_wave ? r->tiles_per_wave : ((uint32_t)r->n_tiles() < SMALL_FRAME_TILES ? 1u : DEFAULT_TILES_PER_WAVE);
    r->order_slots = order_slots(r->tiles_x, r->tiles_y, r->order_group);
    if (r->tile_order && n_jobs) {
        HIPCHECK(r, r->d_tile_class.ensure(r->n_tiles()));
        HIPCHECK(r, r->d_order_lists.ensure((size_t)n_jobs * 8));
        HIPCHECK(r, r->d_tile_order.ensure((size_t)r->order_slots * 4));
        hint.tile_class = r->d_tile_class.as<uint8_t>();
        hint.sky = r->env_w ? 1u : 0u;
        const uint32_t nb = shadow_bounds_pitch(r->shadow_size);
        if (nb && !(r->debug & 8)) {
            int rc = build_shadow_bounds(r, r->stream);
            if (rc != ARCTIC_OK) return rc;
            hint.bounds = r->d_shadow_bounds().as<float2>(); hint.S = r->shadow_size; hint.pitch = nb;
        }
    }
    HIPCHECK(r, launch_resolve(r->d_vis().as<unsigned long long>(), r->geo[r->fwd()].d_recs.as<SetupRec>(), r->geo[r->fwd()].d_rrecs.as<RasterRec>(), r->geo[r->fwd()].d_rec_of.as<uint32_t>(), r->tables[r->fwd()].objs,
                               r->geo[r->fwd()].d_xverts.as<XVert>(), r->tables[r->fwd()].gp, (uint32_t)r->n_tiles(), r->gbuffer(), hint,
                               r->stream));
    if (hint.tile_class) {
        HIPCHECK(r, launch_tile_order(hint.tile_class, r->tiles_x, r->tiles_y, r->order_tail, r->order_group, r->d_order_lists.as<uint32_t>(), r->d_tile_order.as<uint32_t>(), r->stream));
        r->have_order = true;
    }
    r->have_gbuffer = true;
    return ARCTIC_OK;
}

int pass_gbuffer(ArcticRenderer *r, const ArcticScene *sc) {
    int rc = pass_visibility(r, sc, r->stream);
    return rc != ARCTIC_OK ? rc : resolve_gbuffer(r);
}

int fill_shade_params(ArcticRenderer *r, const ArcticScene *sc, const ArcticSettings *st, void *d_out, ShadeParams &sp, bool from_vis = false) {
    if (!from_vis && !r->have_gbuffer) {
        if (r->have_vis) { int rc = resolve_gbuffer(r); if (rc != ARCTIC_OK) return rc; }   // frame came from arctic_render_frame
        else return r->fail(ARCTIC_E_STATE, "shade: no G-buffer (run arctic_pass_gbuffer or arctic_write_gbuffer first)");
    }
    std::memset(&sp, 0, sizeof sp);
    sp.g = r->gbuffer();
    const ArcticRenderer::GeoSet &G = r->geo[r->fwd()];
    sp.compact_tables = (G.d_recs.cap < (1ull << 32) && G.d_rrecs.cap < (1ull << 32) && G.d_xverts.cap < (1ull << 32) && G.d_rec_of.cap < (1ull << 32) &&
                         r->tables[r->fwd()].d.cap < (1ull << 32) && !(r->debug & 64)) ? 1 : 0;
    sp.vis = r->d_vis().as<unsigned long long>(); sp.recs = G.d_recs.as<SetupRec>(); sp.rrecs = G.d_rrecs.as<RasterRec>(); sp.rec_of = G.d_rec_of.as<uint32_t>();
    sp.objs = r->tables[r->fwd()].objs; sp.xv = G.d_xverts.as<XVert>();
    sp.tex = r->d_tex.as<TexDesc>();
    sp.n_materials = (uint32_t)(r->tex.size() / 3);
    sp.srgb_lut = r->d_lut.as<float>();
    sp.shadow_map = r->shadow_size ? r->d_shadow().as<float>() : nullptr;
    sp.shadow_size = r->shadow_size;
    sp.lights = r->d_lights.as<float4>(); sp.light_pairs = r->d_light_pairs.as<float4>();
    sp.n_lights = r->n_lights;
    std::memcpy(sp.eye, sc->camera.eye, 12);
    dir_from_rot(sc->sun.rotation, sp.sun_dir);     // DirectionalLight::direction(), scene.cpp:56-59
    std::memcpy(sp.sun_color, sc->sun.color, 12);
    sp.ambient = sc->ambient;
    sp.tm_method = st->tm_method;
    sp.inv_gamma = 1.0f / st->gamma;
    sp.exposure = st->exposure;
    sp.width = r->width; sp.rows = r->rows(); sp.row0_in_tile = r->row0_in_tile;
    sp.tiles_x = r->tiles_x; sp.tiles_y = r->tiles_y;
    sp.out_rgba8 = static_cast<uint8_t *>(d_out ? d_out : r->d_rgba8.p);
    size_t out_px = (size_t)r->rows() * r->width;
    if (r->keep_float) {
        HIPCHECK(r, r->d_ldr.ensure(out_px * 12));
        HIPCHECK(r, r->d_hdr.ensure(out_px * 12));
        sp.out_ldr = r->d_ldr.as<float>(); sp.out_hdr = r->d_hdr.as<float>();
    }
    sp.stats = nullptr;
    sp.trace = nullptr;
    if (r->tile_trace) {
        const size_t bytes = (size_t)r->tiles_x * r->tiles_y * 32;
        HIPCHECK(r, r->d_tile_trace.ensure(bytes));
        HIPCHECK(r, hipMemsetAsync(r->d_tile_trace.p, 0, bytes, r->stream));
        sp.trace = r->d_tile_trace.as<unsigned long long>();
    }
    // (the order was built for groups of order_group tiles per wave: a pass that shades another number per wave takes the geometric order)
    if (!from_vis && r->have_order && r->tile_order && (r->tiles_per_wave == 0 || r->tiles_per_wave == r->order_group)) {
        sp.tile_order = r->d_tile_order.as<uint32_t>(); sp.n_jobs = r->order_slots; sp.tiles_per_wave = r->order_group;
    }
    sp.culling = r->culling;
    sp.debug = r->debug | (sp.trace ? (1 << 30) : 0) | ((r->sampler & 5) << 20);   // bit 30: the kernels learn of the trace from the first block of their arguments; bits 20..22: ARCTIC_OPT_SAMPLER (shade.hip SAMPLER_SHIFT)
    sp.hdr16 = r->hdr16;
    sp.env = r->env_w ? r->d_env.as<float4>() : nullptr; sp.env_w = r->env_w; sp.env_h = r->env_h;
    camera_sky_basis(sc->camera.rotation, sc->camera.aspect, sc->camera.fov_y, sp.sky_fwd, sp.sky_right, sp.sky_up);
    sp.ndc_sx = 2.0f / (float)r->width; sp.ndc_sy = 2.0f / (float)r->height;
    sp.band_tiles = (int32_t)(r->band_rows / TILE); sp.shard_index = (int32_t)r->shard_index; sp.shard_count = (int32_t)r->shard_count;
    sp.tile_y0 = (int32_t)r->tile_y0;
    const uint32_t nb = shadow_bounds_pitch(r->shadow_size);
    if (nb && !(r->debug & 8)) {   // the min/max table of the shadow map: rebuilt whenever the map was written
        int rc = build_shadow_bounds(r, r->stream);
        if (rc != ARCTIC_OK) return rc;
        sp.shadow_bounds = r->d_shadow_bounds().as<float2>(); sp.bounds_pitch = nb;
    }
    return ARCTIC_OK;
}

// one shading pass = one launch.  The light loop: scalar up to 12 point lights, packed pairs above (both read the lights through the scalar
// cache; measured crossover, profiles/r3_light_paths.txt: 8 lights 0.1062 / 0.1072 ms, 12: 0.1125 / 0.1138, 16: 0.123 / 0.1186;
// ARCTIC_OPT_LIGHT_PATH overrides).
hipError_t shade_once(ArcticRenderer *r, const ShadeParams &sp, bool from_vis, bool stats) {
    ShadeLaunch L;
    L.stream = r->stream;
    L.loop = r->light_path == 0 ? (sp.n_lights <= 12 ? 1u : 2u) : (uint32_t)r->light_path;
    L.from_vis = from_vis ? 1u : 0u;
    L.stats = stats ? 1u : 0u;
    L.tiles_per_wave = r->tiles_per_wave;
    return launch_shade(sp, L);
}

int pass_shade(ArcticRenderer *r, const ArcticScene *sc, const ArcticSettings *st, void *d_out, bool from_vis = false) {
    Range zone("Forward Pass: shading + Skybox Pass + Post Process Pass");
    ShadeParams sp;
    int rc = fill_shade_params(r, sc, st, d_out, sp, from_vis);
    if (rc != ARCTIC_OK) return rc;
    for (auto &f : r->inflight)   // a gather of an earlier frame may still be reading this output buffer
        if (f.ptr && f.ptr == static_cast<const void *>(sp.out_rgba8)) { HIPCHECK(r, hipStreamWaitEvent(r->stream, f.done, 0)); f.ptr = nullptr; }
    if (r->count_evals) {
        sp.stats = r->d_counter.as<unsigned long long>();
        HIPCHECK(r, hipMemsetAsync(r->d_counter.p, 0, 8 * N_SHADE_STATS, r->stream));
    }
    HIPCHECK(r, shade_once(r, sp, from_vis, r->count_evals != 0));
    if (r->count_evals) {
        unsigned long long n[N_SHADE_STATS] = {};
        HIPCHECK(r, hipMemcpyAsync(n, r->d_counter.p, 8 * N_SHADE_STATS, hipMemcpyDeviceToHost, r->stream));
        HIPCHECK(r, hipStreamSynchronize(r->stream));
        r->stats[5] = n[0]; r->stats[6] = n[1]; r->stats[7] = n[2];
        r->light_stats[0] = n[3]; r->light_stats[1] = n[4];
        for (int i = 0; i < 4; ++i) r->edge_stats[i] = n[5 + i];
    }
    r->have_output = (d_out == nullptr);
    return ARCTIC_OK;
}

// everything the shadow map is a function of, as bytes (compared exactly, not hashed)
std::vector<uint8_t> shadow_inputs(const ArcticRenderer *r, const ArcticScene *sc) {
    std::vector<uint8_t> k;
    if (!r->shadow_size) return k;
    auto put = [&](const void *p, size_t n) { const uint8_t *b = static_cast<const uint8_t *>(p); k.insert(k.end(), b, b + n); };
    const uint64_t head[3] = {r->shadow_size, (uint64_t)r->meshes.size(), sc->n_objects};
    put(head, sizeof head);
    put(sc->sun.position, sizeof sc->sun.position);
    put(sc->sun.rotation, sizeof sc->sun.rotation);
    for (uint64_t i = 0; i < sc->n_objects; ++i) { put(sc->objects[i].trs, sizeof sc->objects[i].trs); put(&sc->objects[i].mesh_idx, sizeof sc->objects[i].mesh_idx); }
    return k;
}

// after a stream synchronisation: did a rasteriser pass run out of work-item slots?  (k_setup flags it; the table is re-sized
// from the counts of that very pass, so rendering the frame again succeeds)
int check_item_overflow(ArcticRenderer *r) {
    if (!r->h_counts || !(r->h_counts[4] | r->h_counts[5])) return ARCTIC_OK;
    const uint32_t need = std::max(r->h_counts[1], r->h_counts[3]);
    r->h_counts[4] = r->h_counts[5] = 0;
    r->recs_worst_case = true;   // whichever table it was: the record table takes its worst-case size from now on
    r->have_gbuffer = false; r->have_output = false; r->have_vis = false; r->have_order = false; r->shadow_key.clear();
    return r->fail(ARCTIC_E_CAPACITY, "a rasteriser table overflowed (%u work items needed, %u slots; or more than 2 records per source triangle): the last frame "
                   "is incomplete; the tables grow on the next pass -- render the frame again", need, std::max(r->geo[0].item_cap, r->geo[1].item_cap));
}

bool valid_scene(const ArcticScene *sc) { return sc && (sc->n_objects == 0 || sc->objects); }

}  // namespace

// =================================================================================================
extern "C" {

int arctic_version(void) { return 100; }

ArcticRenderer *arctic_create(const ArcticCreateInfo *info, char *err, uint64_t err_len) {
    auto say = [&](const char *m) { if (err && err_len) { std::snprintf(err, (size_t)err_len, "%s", m); } };
    if (!info || info->width == 0 || info->height == 0) { say("arctic_create: width/height must be > 0"); return nullptr; }
    if (info->width > MAX_TARGET || info->height > MAX_TARGET || info->shadow_size > MAX_TARGET) { say("arctic_create: targets above 16384 pixels a side are not supported"); return nullptr; }
    uint32_t rb = info->row_begin, re = info->row_end;
    if (rb == 0 && re == 0) re = info->height;
    if (re > info->height || rb >= re) { say("arctic_create: bad row shard"); return nullptr; }
    if (info->band_rows) {
        if (info->band_rows % TILE != 0 || info->shard_count == 0 || info->shard_index >= info->shard_count || info->row_begin || info->row_end) {
            say("arctic_create: bad interleaved shard (band_rows must be a multiple of 8, shard_index < shard_count, row_begin = row_end = 0)");
            return nullptr;
        }
        if ((uint64_t)info->shard_index * info->band_rows >= info->height) { say("arctic_create: interleaved shard owns no rows"); return nullptr; }
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) { say("arctic_create: no HIP device (this library has no CPU path)"); return nullptr; }
    if (info->device < 0 || info->device >= n_dev) { say("arctic_create: device ordinal out of range"); return nullptr; }
    ArcticRenderer *r = new ArcticRenderer();
    r->device = info->device;
    r->width = info->width; r->height = info->height; r->shadow_size = info->shadow_size; r->max_lights = info->max_lights;
    r->row_begin = rb; r->row_end = re;
    r->band_rows = info->band_rows; r->shard_index = info->shard_index; r->shard_count = info->band_rows ? info->shard_count : 1;
    auto bail = [&](const char *what, hipError_t e) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "arctic_create: %s: %s", what, hipGetErrorString(e));
        say(buf);
        arctic_destroy(r);
        return (ArcticRenderer *)nullptr;
    };
    hipError_t e;
    // The events below order streams of THIS device among each other (prepass / shadow pass / shading of frames in flight); nothing waits for them on the
    // host or on another device.  Without the system-scope fence HIP otherwise puts around every record and wait -- a cache write-back and invalidate between
    // two shading kernels -- a 4K frame takes 0.256 instead of 0.261 ms, one rank of 8's 0.065 instead of 0.069 (round 5, tools/experiments/cull_ab.py's loop).
    // (ev_main and the gather's events, which RCCL's transfers to other devices follow, keep the default.)
    const unsigned in_device_event = hipEventDisableTiming | hipEventDisableSystemFence;
    if ((e = hipSetDevice(r->device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&r->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    r->stream = r->own_stream;
    if ((e = hipStreamCreateWithFlags(&r->shadow_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreateWithFlags(&r->ev_fork, in_device_event)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&r->ev_shadow, in_device_event)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&r->ev_shadow_scratch, in_device_event)) != hipSuccess) return bail("hipEventCreate", e);
    // The HIP runtime deals its hardware queues (4 by default, GPU_MAX_HW_QUEUES) to streams in the order they are created, and two
    // streams on one queue do not overlap: the handle's own stream, the shadow stream and the first prepass stream are created here, in
    // that order (with a caller's stream that makes four); the second prepass stream, which only three frames in flight use, is the
    // handle's own stream when the caller brought one (it is idle then), else created when first needed (second_prepass_stream)
    if ((e = hipStreamCreateWithFlags(&r->prepass_stream[0], hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    for (int k = 0; k < 2; ++k)
        if ((e = hipEventCreateWithFlags(&r->ev_prepass[k], in_device_event)) != hipSuccess) return bail("hipEventCreate", e);
    for (hipEvent_t &ev : r->ev_released)
        if ((e = hipEventCreateWithFlags(&ev, in_device_event)) != hipSuccess) return bail("hipEventCreate", e);
    for (hipEvent_t &ev : r->ev_shadow_released)
        if ((e = hipEventCreateWithFlags(&ev, in_device_event)) != hipSuccess) return bail("hipEventCreate", e);
    {
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, r->device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
        r->cu_count = (uint32_t)std::max(1, prop.multiProcessorCount);
        r->raster_blocks[0] = raster_grid_blocks(false, r->cu_count);
        r->raster_blocks[1] = raster_grid_blocks(true, r->cu_count);
        if ((e = hipHostMalloc((void **)&r->h_counts, 64, hipHostMallocMapped)) != hipSuccess) return bail("hipHostMalloc", e);
        std::memset(r->h_counts, 0, 64);
        if ((e = hipHostGetDevicePointer((void **)&r->dh_counts, r->h_counts, 0)) != hipSuccess) return bail("hipHostGetDevicePointer", e);
    }
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = srgb8_to_linear(i);
    if ((e = r->d_lut.ensure(sizeof lut)) != hipSuccess) return bail("hipMalloc lut", e);
    if ((e = hipMemcpy(r->d_lut.p, lut, sizeof lut, hipMemcpyHostToDevice)) != hipSuccess) return bail("upload lut", e);
    if ((e = r->d_lights.ensure(std::max<size_t>(32, (size_t)r->max_lights * 32))) != hipSuccess) return bail("hipMalloc lights", e);
    if ((e = r->d_light_pairs.ensure(std::max<size_t>(48, (size_t)((r->max_lights + 1) / 2) * 48))) != hipSuccess) return bail("hipMalloc light pairs", e);
    if ((e = r->d_tex.ensure(48)) != hipSuccess) return bail("hipMalloc tex table", e);
    if (r->shadow_size) {
        size_t n = (size_t)r->shadow_size * r->shadow_size;
        if ((e = r->d_shadow().ensure(shadow_alloc_bytes(r))) != hipSuccess) return bail("hipMalloc shadow map", e);
        if ((e = launch_fill_u32(r->d_shadow().as<uint32_t>(), 0x3F800000u, n, r->stream)) != hipSuccess) return bail("clear shadow map", e);
    }
    if (alloc_targets(r) != ARCTIC_OK) { say(r->err.c_str()); arctic_destroy(r); return nullptr; }
    if ((e = hipStreamSynchronize(r->stream)) != hipSuccess) return bail("sync", e);
    return r;
}

void arctic_destroy(ArcticRenderer *r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->h_counts) (void)hipHostFree(r->h_counts);
    (void)hipStreamSynchronize(r->stream);
    (void)arctic_comm_destroy(r);
    if (r->shadow_stream) { (void)hipStreamSynchronize(r->shadow_stream); (void)hipStreamDestroy(r->shadow_stream); }
    for (hipStream_t ps : r->prepass_stream) if (ps) { (void)hipStreamSynchronize(ps); (void)hipStreamDestroy(ps); }
    for (hipEvent_t ev : r->ev_prepass) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : r->ev_released) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : r->ev_shadow_released) if (ev) (void)hipEventDestroy(ev);
    if (r->ev_fork) (void)hipEventDestroy(r->ev_fork);
    if (r->ev_shadow) (void)hipEventDestroy(r->ev_shadow);
    if (r->ev_shadow_scratch) (void)hipEventDestroy(r->ev_shadow_scratch);
    if (r->own_stream) { (void)hipStreamSynchronize(r->own_stream); (void)hipStreamDestroy(r->own_stream); }
    for (Mesh &m : r->meshes) { if (m.d_vertices) (void)hipFree(m.d_vertices); if (m.d_indices) (void)hipFree(m.d_indices); }
    for (void *p : r->tex_allocs) (void)hipFree(p);
    DevBuf *bufs[] = {&r->d_tex, &r->d_lut, &r->d_lights, &r->d_light_pairs, &r->d_shadow_set[0], &r->d_shadow_set[1], &r->d_env, &r->d_vis_set[0], &r->d_vis_set[1], &r->d_vis_set[2], &r->d_p0, &r->d_p1, &r->d_p2, &r->d_p3, &r->d_p4,
                      &r->d_rgba8, &r->d_ldr, &r->d_hdr, &r->d_counter, &r->d_shadow_blocks_set[0], &r->d_shadow_blocks_set[1], &r->d_shadow_bounds_set[0], &r->d_shadow_bounds_set[1], &r->d_staging, &r->d_layout, &r->geo[0].d_xverts, &r->geo[1].d_xverts, &r->geo[2].d_xverts,
                      &r->geo[2].d_recs, &r->geo[2].d_rrecs, &r->geo[2].d_clip_list, &r->geo[2].d_rec_of, &r->geo[2].d_items, &r->tables[2].d,
                      &r->geo[3].d_xverts, &r->geo[3].d_recs, &r->geo[3].d_rrecs, &r->geo[3].d_clip_list, &r->geo[3].d_rec_of, &r->geo[3].d_items, &r->geo[3].d_left, &r->geo[3].d_bin_count, &r->geo[3].d_bin_slots, &r->tables[3].d,
                      &r->geo[0].d_recs, &r->geo[0].d_rrecs, &r->geo[0].d_clip_list, &r->geo[0].d_rec_of, &r->geo[0].d_items,
                      &r->geo[1].d_recs, &r->geo[1].d_rrecs, &r->geo[1].d_clip_list, &r->geo[1].d_rec_of, &r->geo[1].d_items, &r->geo[0].d_left, &r->geo[1].d_left, &r->geo[2].d_left, &r->geo[0].d_bin_count, &r->geo[0].d_bin_slots, &r->geo[1].d_bin_count, &r->geo[1].d_bin_slots, &r->geo[2].d_bin_count, &r->geo[2].d_bin_slots, &r->d_geo_counters, &r->d_stage, &r->tables[0].d, &r->tables[1].d};
    for (PassTables &T : r->tables) { if (T.h) (void)hipHostFree(T.h); if (T.copied) (void)hipEventDestroy(T.copied); }
    for (DevBuf *b : bufs) b->release();
    delete r;
}

const char *arctic_last_error(const ArcticRenderer *r) { return r ? r->err.c_str() : "null handle"; }

int arctic_resize(ArcticRenderer *r, uint32_t width, uint32_t height) {
    if (!r) return ARCTIC_E_INVALID;
    if (width == 0 || height == 0) return r->fail(ARCTIC_E_INVALID, "resize: zero size");
    if (width > MAX_TARGET || height > MAX_TARGET) return r->fail(ARCTIC_E_CAPACITY, "resize: targets above 16384 pixels a side are not supported");
    int rc = select_device(r);
    if (rc) return rc;
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    r->width = width; r->height = height; r->row_begin = 0; r->row_end = height;
    r->band_rows = 0; r->shard_index = 0; r->shard_count = 1;
    return alloc_targets(r);
}

int arctic_flush(ArcticRenderer *r) {
    if (!r) return ARCTIC_E_INVALID;
    int rc = select_device(r);
    if (rc) return rc;
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    for (hipStream_t ps : r->prepass_stream) if (ps) HIPCHECK(r, hipStreamSynchronize(ps));   // (joined into the main stream by every frame; after a failed frame they may not be)
    if (r->own_stream_is_prepass()) HIPCHECK(r, hipStreamSynchronize(r->own_stream));         // (the second prepass stream of a handle on a caller's stream)
    HIPCHECK(r, hipStreamSynchronize(r->shadow_stream));
    if (r->comm_stream) HIPCHECK(r, hipStreamSynchronize(r->comm_stream));
    if (int ov = check_item_overflow(r)) return ov;
    return ARCTIC_OK;
}

int arctic_set_stream(ArcticRenderer *r, void *hip_stream) {
    if (!r) return ARCTIC_E_INVALID;
    int rc = select_device(r);
    if (rc) return rc;
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    // (own_stream may be the second prepass stream of the frames in flight; when it carries the exchange instead -- borrowed as the
    //  communication stream -- it is NOT waited for here: it may hold a collective the other ranks have not reached yet)
    if (r->own_stream_is_prepass()) HIPCHECK(r, hipStreamSynchronize(r->own_stream));
    r->stream = static_cast<hipStream_t>(hip_stream);   // NULL = the default stream
    return ARCTIC_OK;
}

int arctic_use_own_stream(ArcticRenderer *r) {
    if (!r) return ARCTIC_E_INVALID;
    int rc = select_device(r);
    if (rc) return rc;
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    if (r->own_stream_is_prepass()) HIPCHECK(r, hipStreamSynchronize(r->own_stream));
    if (r->comm_stream_borrowed) {
        // One role per stream.  The exchange KEEPS the stream it has been running on -- it may hold gathers the other ranks have not matched yet, and
        // passes queued behind those would stall until they do -- and becomes its owner; the passes get a fresh stream (ordered against earlier
        // gathers the way they always are: by the events of the shard buffers they write, pass_shade).  The new stream goes into a local first: a
        // failed creation leaves the handle exactly as it was.
        hipStream_t fresh = nullptr;
        HIPCHECK(r, hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
        r->comm_stream_borrowed = false;   // comm_stream == the old own_stream: owned by the exchange from here on (destroyed with the communicator)
        r->own_stream = fresh;
    }
    r->stream = r->own_stream;
    return ARCTIC_OK;
}

int arctic_create_material(ArcticRenderer *r, const void *diffuse, uint32_t dw, uint32_t dh, const void *normal, uint32_t nw,
                           uint32_t nh, const void *mr, uint32_t mw, uint32_t mh) {
    if (!r) return ARCTIC_E_INVALID;
    if (!diffuse || !normal || !mr || !dw || !dh || !nw || !nh || !mw || !mh) return r->fail(ARCTIC_E_INVALID, "create_material: null image or zero size");
    if ((dw | dh | nw | nh | mw | mh) & 0xFFFF0000u) return r->fail(ARCTIC_E_CAPACITY, "create_material: image side above 65535");
    int rc = select_device(r);
    if (rc) return rc;
    const void *src[3] = {diffuse, normal, mr};
    const uint32_t w[3] = {dw, nw, mw}, h[3] = {dh, nh, mh};
    TexDesc td[3];
    // byte offsets into an image are 32-bit in the kernels (shade.hip: saddr loads)
    if ((uint64_t)(dw + 2) * (dh + 2) > (1ull << 29) || (uint64_t)nw * nh > (1ull << 30) || (uint64_t)mw * mh > (1ull << 30))
        return r->fail(ARCTIC_E_CAPACITY, "create_material: image above 2^29 texels");
    if (dw == nw && dw == mw && dh == nh && dh == mh) {
        // equal sizes (the usual glTF case): pack the eight channels ps_main reads into 8-byte texels, with a one-texel WRAP
        // border around the image (layout: common.h TexDesc, shade.hip)
        const uint32_t pw = dw + 2, ph = dh + 2;
        // 4 x 4-texel tiles for textures that will be minified at their only mip level (common.h TexDesc::tile_row_bytes): the library's choice is by
        // size -- a 2048^2 image on a model that covers a part of a 1080p or 4K frame is sampled several texels apart, a 1024^2 wall texture about 1:1,
        // where the row-major strips of neighbouring pixels share their lines and the tiled address (15 instructions against 3) would only cost
        const bool tiled = r->texture_tiling < 0 ? std::min(dw, dh) >= 2048u : r->texture_tiling != 0;
        const uint32_t tpr = (pw + 3) / 4, trows = (ph + 3) / 4;
        size_t n = tiled ? (size_t)tpr * trows * 16 : (size_t)pw * ph;
        if (n > (1ull << 29)) return r->fail(ARCTIC_E_CAPACITY, "create_material: image above 2^29 texels");
        std::vector<uint32_t> packed(n * 2);
        const uint8_t *a = static_cast<const uint8_t *>(diffuse), *b = static_cast<const uint8_t *>(normal), *c = static_cast<const uint8_t *>(mr);
        for (uint32_t Y = 0; Y < ph; ++Y) {
            const size_t sy = (size_t)((Y + dh - 1) % dh) * dw;
            for (uint32_t X = 0; X < pw; ++X) {
                const size_t i = sy + (X + dw - 1) % dw;
                const size_t o = tiled ? ((size_t)(Y >> 2) * tpr + (X >> 2)) * 16 + (Y & 3) * 4 + (X & 3) : (size_t)Y * pw + X;
                packed[2 * o] = (uint32_t)a[4 * i] | ((uint32_t)a[4 * i + 1] << 8) | ((uint32_t)a[4 * i + 2] << 16) | ((uint32_t)b[4 * i] << 24);
                packed[2 * o + 1] = (uint32_t)b[4 * i + 1] | ((uint32_t)b[4 * i + 2] << 8) | ((uint32_t)c[4 * i + 1] << 16) | ((uint32_t)c[4 * i + 2] << 24);
            }
        }
        void *p = nullptr;
        HIPCHECK(r, hipMalloc(&p, n * 8));
        r->tex_allocs.push_back(p);
        HIPCHECK(r, hipMemcpy(p, packed.data(), n * 8, hipMemcpyHostToDevice));   // synchronous like rhi.cpp:480-519
        for (int i = 0; i < 3; ++i) { td[i] = TexDesc{static_cast<const uint32_t *>(p), dw | TEX_INTERLEAVED, dh, (float)dw, (float)dh, pw, tiled ? tpr * 128u : 0u}; }
    } else {
        for (int i = 0; i < 3; ++i) {
            void *p = nullptr;
            size_t bytes = (size_t)w[i] * h[i] * 4;
            HIPCHECK(r, hipMalloc(&p, bytes));
            r->tex_allocs.push_back(p);
            HIPCHECK(r, hipMemcpy(p, src[i], bytes, hipMemcpyHostToDevice));
            td[i] = TexDesc{static_cast<const uint32_t *>(p), w[i], h[i], (float)w[i], (float)h[i], w[i], 0u};
        }
    }
    r->tex.insert(r->tex.end(), td, td + 3);
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    HIPCHECK(r, r->d_tex.ensure(r->tex.size() * sizeof(TexDesc)));
    HIPCHECK(r, hipMemcpy(r->d_tex.p, r->tex.data(), r->tex.size() * sizeof(TexDesc), hipMemcpyHostToDevice));
    return (int)(r->tex.size() / 3) - 1;
}

int arctic_create_mesh(ArcticRenderer *r, const ArcticVertex *vertices, uint64_t n_vertices, const uint32_t *indices,
                       uint64_t n_indices, uint64_t material_idx) {
    if (!r) return ARCTIC_E_INVALID;
    if (!vertices || !indices || n_vertices == 0 || n_indices == 0) return r->fail(ARCTIC_E_INVALID, "create_mesh: empty mesh");
    if (n_indices % 3 != 0) return r->fail(ARCTIC_E_INVALID, "create_mesh: index count %llu is not a multiple of 3 (triangle list)", (unsigned long long)n_indices);
    if (material_idx >= r->tex.size() / 3) return r->fail(ARCTIC_E_INVALID, "create_mesh: material %llu does not exist", (unsigned long long)material_idx);
    if (n_vertices > 0x7FFFFFFFull || n_indices > 0xFFFFFFF0ull) return r->fail(ARCTIC_E_CAPACITY, "create_mesh: mesh too large");
    int rc = select_device(r);
    if (rc) return rc;
    Mesh m;
    HIPCHECK(r, hipMalloc((void **)&m.d_vertices, n_vertices * sizeof(ArcticVertex)));
    if (hipMalloc((void **)&m.d_indices, n_indices * 4) != hipSuccess) { (void)hipFree(m.d_vertices); return r->fail(ARCTIC_E_DEVICE, "create_mesh: hipMalloc indices"); }
    m.n_vertices = (uint32_t)n_vertices; m.n_indices = (uint32_t)n_indices; m.material = material_idx;
    cluster_bounds(vertices, m.n_vertices, indices, m.n_indices, m.tbounds, m.vbounds);
    r->meshes.push_back(m);
    HIPCHECK(r, hipMemcpy(m.d_vertices, vertices, n_vertices * sizeof(ArcticVertex), hipMemcpyHostToDevice));
    HIPCHECK(r, hipMemcpy(m.d_indices, indices, n_indices * 4, hipMemcpyHostToDevice));
    return (int)r->meshes.size() - 1;
}

int arctic_update_lights(ArcticRenderer *r, const ArcticPointLight *lights, uint64_t n) {
    if (!r) return ARCTIC_E_INVALID;
    if (n && !lights) return r->fail(ARCTIC_E_INVALID, "update_lights: null");
    int rc = select_device(r);
    if (rc) return rc;
    uint32_t k = (uint32_t)std::min<uint64_t>(n, r->max_lights);   // renderer.cpp:587-588
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    if (k) HIPCHECK(r, hipMemcpy(r->d_lights.p, lights, (size_t)k * sizeof(ArcticPointLight), hipMemcpyHostToDevice));
    {   // the same lights as pairs for the packed loop (common.h: ShadeParams::light_pairs); an odd count gets a black partner
        const uint32_t n_pairs = (k + 1) / 2;
        std::vector<float> pairs((size_t)std::max(1u, n_pairs) * 12, 0.0f);
        for (uint32_t i = 0; i < 2 * n_pairs; ++i) {
            float *dst = pairs.data() + (size_t)(i / 2) * 12 + (i & 1);
            if (i < k) { dst[0] = lights[i].position[0]; dst[2] = lights[i].position[1]; dst[4] = lights[i].position[2];
                         dst[6] = lights[i].color[0]; dst[8] = lights[i].color[1]; dst[10] = lights[i].color[2]; }
            else { dst[0] = 0.0f; dst[2] = 1.0e6f; dst[4] = 0.0f; dst[6] = dst[8] = dst[10] = 0.0f; }
        }
        HIPCHECK(r, r->d_light_pairs.ensure(pairs.size() * 4));
        HIPCHECK(r, hipMemcpy(r->d_light_pairs.p, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
    }
    r->n_lights = k;
    return ARCTIC_OK;
}

int arctic_create_hdri(ArcticRenderer *r, const float *rgba32f, uint32_t w, uint32_t h) {
    if (!r) return ARCTIC_E_INVALID;
    if (!rgba32f || !w || !h) return r->fail(ARCTIC_E_INVALID, "create_hdri: null image or zero size");
    if ((w | h) & 0xFFFF0000u) return r->fail(ARCTIC_E_CAPACITY, "create_hdri: image side above 65535");
    int rc = select_device(r);
    if (rc) return rc;
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    HIPCHECK(r, r->d_env.ensure((size_t)w * h * 16));
    HIPCHECK(r, hipMemcpy(r->d_env.p, rgba32f, (size_t)w * h * 16, hipMemcpyHostToDevice));
    r->env_w = w; r->env_h = h;
    return ARCTIC_OK;
}

int arctic_pass_shadow_map(ArcticRenderer *r, const ArcticScene *scene) {
    if (!r) return ARCTIC_E_INVALID;
    if (!valid_scene(scene)) return r->fail(ARCTIC_E_INVALID, "pass_shadow_map: bad scene");
    int rc = select_device(r);
    if (rc) return rc;
    r->shadow_key.clear();   // an explicit pass always renders; render_frame's cache starts over
    return pass_shadow_map(r, scene, r->stream);
}

int arctic_pass_gbuffer(ArcticRenderer *r, const ArcticScene *scene) {
    if (!r) return ARCTIC_E_INVALID;
    if (!valid_scene(scene)) return r->fail(ARCTIC_E_INVALID, "pass_gbuffer: bad scene");
    int rc = select_device(r);
    return rc ? rc : pass_gbuffer(r, scene);
}

int arctic_pass_shade(ArcticRenderer *r, const ArcticScene *scene, const ArcticSettings *settings, void *d_out) {
    if (!r) return ARCTIC_E_INVALID;
    if (!scene || !settings) return r->fail(ARCTIC_E_INVALID, "pass_shade: null scene/settings");
    int rc = select_device(r);
    return rc ? rc : pass_shade(r, scene, settings, d_out);
}

int arctic_render_frame_device(ArcticRenderer *r, const ArcticScene *scene, const ArcticSettings *settings, void *d_out) {
    if (!r) return ARCTIC_E_INVALID;
    if (!valid_scene(scene) || !settings) return r->fail(ARCTIC_E_INVALID, "render_frame: bad scene/settings");
    int rc = select_device(r);
    if (rc) return rc;
    std::vector<uint8_t> key = shadow_inputs(r, scene);
    // The shadow pass and the visibility prepass of a frame are independent until the shading pass, and both are bound by latency
    // and by the memory side's atomic rate, not by the ALUs: when the map has to be redrawn it is drawn on
    // a second stream, forked after everything enqueued so far (the previous frame still reads the map) and joined before the
    // shading pass.  (Not with a sharded map: its all-gather stays on the main stream.  ARCTIC_OPT_DEBUG bit 7: one after the other.)
    const bool redraw = r->shadow_size != 0 && (!r->shadow_cache || key != r->shadow_key);
    const bool beside = redraw && !(r->debug & 128) && !(r->shadow_sharded && r->comm && r->comm_world > 1);
    // ... and with frames in flight into the OTHER shadow map, ordered only after what had been enqueued when that one was last
    // left: the pass then runs beside the previous frame's shading too
    const bool shadow_in_flight = beside && r->frames_in_flight > 1;
    if (redraw) {
        r->shadow_key.clear();
        if (beside) {
            if (shadow_in_flight) {
                HIPCHECK(r, hipEventRecord(r->ev_shadow_released[r->scur], r->stream));
                r->shadow_released_valid[r->scur] = true;
                r->scur ^= 1;
                HIPCHECK(r, r->d_shadow().ensure(shadow_alloc_bytes(r)));   // first use of the second map (cleared by the pass itself)
                if (r->shadow_released_valid[r->scur]) HIPCHECK(r, hipStreamWaitEvent(r->shadow_stream, r->ev_shadow_released[r->scur], 0));
            } else {
                HIPCHECK(r, hipEventRecord(r->ev_fork, r->stream));
                HIPCHECK(r, hipStreamWaitEvent(r->shadow_stream, r->ev_fork, 0));
            }
            rc = pass_shadow_map(r, scene, r->shadow_stream);
            if (rc == ARCTIC_OK && !(r->debug & 8)) rc = build_shadow_bounds(r, r->shadow_stream);
            HIPCHECK(r, hipEventRecord(r->ev_shadow, r->shadow_stream));
        } else rc = pass_shadow_map(r, scene, r->stream);
        if (rc != ARCTIC_OK) { if (beside) (void)hipStreamWaitEvent(r->stream, r->ev_shadow, 0); return rc; }
        r->shadow_key.swap(key);
    }
    // whole frames skip the G-buffer: the shading pass interpolates from the visibility plane (k_material_vis), bit-identical
    // to visibility -> G-buffer -> shading; arctic_read_gbuffer / arctic_pass_shade materialise the G-buffer afterwards if asked
    const bool vis_path = r->visbuffer;
    // frames in flight: this frame's visibility prepass goes to the other table set, on prepass_stream, ordered only after what
    // the main stream had enqueued when that set was last left -- so it runs beside the shading of the previous frame
    // (not in a frame that redraws the shadow map on the main stream -- sharded map, debug bit 7 --: that pass has to follow the previous
    // frame's shading and precede this one's, a prepass running ahead would only compete with it)
    const bool in_flight = vis_path && r->frames_in_flight > 1 && (!redraw || shadow_in_flight);
    if (in_flight) {
        HIPCHECK(r, hipEventRecord(r->ev_released[r->cur], r->stream));   // everything that reads or writes the set being left is enqueued by now
        r->released_valid[r->cur] = true;
        r->cur = (r->cur + 1) % r->frames_in_flight;
        r->have_vis = r->have_gbuffer = r->have_order = false;            // of the set entered: overwritten now
        // consecutive prepasses on alternate streams when there are three sets (they overlap then); with two, one stream as before
        const int turn = r->frames_in_flight >= 3 ? (r->prepass_turn ^= 1) : 0;
        hipStream_t ps = r->prepass_stream[0];
        if (turn) {
            if (r->own_stream_is_prepass()) ps = r->own_stream;
            else { if (!r->prepass_stream[1]) HIPCHECK(r, hipStreamCreateWithFlags(&r->prepass_stream[1], hipStreamNonBlocking)); ps = r->prepass_stream[1]; }
        }
        if (r->released_valid[r->cur]) HIPCHECK(r, hipStreamWaitEvent(ps, r->ev_released[r->cur], 0));
        rc = pass_visibility(r, scene, ps);
        HIPCHECK(r, hipEventRecord(r->ev_prepass[turn], ps));
        HIPCHECK(r, hipStreamWaitEvent(r->stream, r->ev_prepass[turn], 0));
    } else rc = vis_path ? pass_visibility(r, scene, r->stream) : pass_gbuffer(r, scene);
    if (beside) HIPCHECK(r, hipStreamWaitEvent(r->stream, r->ev_shadow, 0));   // the map and its table are complete
    if (rc != ARCTIC_OK) return rc;
    return pass_shade(r, scene, settings, d_out, vis_path);
}

int arctic_render_frame(ArcticRenderer *r, const ArcticScene *scene, const ArcticSettings *settings, uint8_t *out_rgba8) {
    int rc = arctic_render_frame_device(r, scene, settings, nullptr);
    if (rc != ARCTIC_OK) return rc;
    if (out_rgba8) HIPCHECK(r, hipMemcpyAsync(out_rgba8, r->d_rgba8.p, (size_t)r->rows() * r->width * 4, hipMemcpyDeviceToHost, r->stream));
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    if (int ov = check_item_overflow(r)) return ov;
    return ARCTIC_OK;
}

int arctic_post_process(ArcticRenderer *r, const float *hdr, uint32_t w, uint32_t h, const ArcticSettings *st, uint8_t *out_rgba8,
                        float *out_ldr) {
    if (!r) return ARCTIC_E_INVALID;
    if (!hdr || !w || !h || !st || !out_rgba8) return r->fail(ARCTIC_E_INVALID, "post_process: null argument or zero size");
    int rc = select_device(r);
    if (rc) return rc;
    size_t n = (size_t)w * h;
    HIPCHECK(r, r->d_stage.ensure(n * 16 + n * 4 + n * 12));
    char *base = r->d_stage.as<char>();
    float4 *d_in = reinterpret_cast<float4 *>(base);
    uint8_t *d_out = reinterpret_cast<uint8_t *>(base + n * 16);
    float *d_ldr = reinterpret_cast<float *>(base + n * 20);
    HIPCHECK(r, hipMemcpyAsync(d_in, hdr, n * 16, hipMemcpyHostToDevice, r->stream));
    HIPCHECK(r, launch_post_process(d_in, w, h, st->tm_method, 1.0f / st->gamma, st->exposure, d_out, out_ldr ? d_ldr : nullptr, r->stream));
    HIPCHECK(r, hipMemcpyAsync(out_rgba8, d_out, n * 4, hipMemcpyDeviceToHost, r->stream));
    if (out_ldr) HIPCHECK(r, hipMemcpyAsync(out_ldr, d_ldr, n * 12, hipMemcpyDeviceToHost, r->stream));
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    return ARCTIC_OK;
}

int arctic_time_shade(ArcticRenderer *r, const ArcticScene *scene, const ArcticSettings *settings, uint32_t warmup, uint32_t iters,
                      float *ms_each) {
    if (!r) return ARCTIC_E_INVALID;
    if (!scene || !settings || !ms_each || iters == 0) return r->fail(ARCTIC_E_INVALID, "time_shade: bad arguments");
    int rc = select_device(r);
    if (rc) return rc;
    ShadeParams sp;
    if ((rc = fill_shade_params(r, scene, settings, nullptr, sp)) != ARCTIC_OK) return rc;
    for (uint32_t i = 0; i < warmup; ++i) HIPCHECK(r, shade_once(r, sp, false, false));
    std::vector<hipEvent_t> ev(2 * (size_t)iters);
    for (auto &e : ev) HIPCHECK(r, hipEventCreate(&e));
    for (uint32_t i = 0; i < iters; ++i) {
        HIPCHECK(r, hipEventRecord(ev[2 * i], r->stream));
        HIPCHECK(r, shade_once(r, sp, false, false));
        HIPCHECK(r, hipEventRecord(ev[2 * i + 1], r->stream));
    }
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    for (uint32_t i = 0; i < iters; ++i) HIPCHECK(r, hipEventElapsedTime(&ms_each[i], ev[2 * i], ev[2 * i + 1]));
    for (auto &e : ev) (void)hipEventDestroy(e);
    r->have_output = true;
    return ARCTIC_OK;
}

int arctic_read_gbuffer(ArcticRenderer *r, float *attrs, uint32_t *material, float *depth, uint32_t *tri) {
    if (!r) return ARCTIC_E_INVALID;
    if (!r->have_gbuffer) {
        if (!r->have_vis) return r->fail(ARCTIC_E_STATE, "read_gbuffer: no G-buffer");
        int rs = select_device(r);
        if (rs) return rs;
        if ((rs = resolve_gbuffer(r)) != ARCTIC_OK) return rs;   // the frame was shaded from the visibility plane
    }
    int rc = select_device(r);
    if (rc) return rc;
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    if (int ov = check_item_overflow(r)) return ov;
    size_t px = (size_t)r->rows() * r->width;
    if (attrs || material) {
        HIPCHECK(r, r->d_stage.ensure(px * 76));
        float *d_attrs = r->d_stage.as<float>();
        uint32_t *d_mat = reinterpret_cast<uint32_t *>(r->d_stage.as<char>() + px * 72);
        HIPCHECK(r, launch_gbuffer_tile(r->gbuffer(), d_attrs, d_mat, r->width, r->rows(), r->row0_in_tile, r->tiles_x, r->tiles_y, 0, r->stream));
        if (attrs) HIPCHECK(r, hipMemcpyAsync(attrs, d_attrs, px * 72, hipMemcpyDeviceToHost, r->stream));
        if (material) HIPCHECK(r, hipMemcpyAsync(material, d_mat, px * 4, hipMemcpyDeviceToHost, r->stream));
        HIPCHECK(r, hipStreamSynchronize(r->stream));
    }
    if (depth || tri) {
        // both come straight from the visibility plane the prepass resolved: key = depth bits << 32 | order id, and
        // order id = 8 * (draw-order index of the source triangle) + sub-triangle (k_setup); ~0 = nothing drawn
        if (!r->have_vis) return r->fail(ARCTIC_E_STATE, "read_gbuffer: depth / triangle ids exist only after arctic_pass_gbuffer (not after arctic_write_gbuffer)");
        size_t tp = r->n_tiles() * TILE_PIXELS;
        std::vector<unsigned long long> hv(tp);
        HIPCHECK(r, hipStreamSynchronize(r->stream));
        HIPCHECK(r, hipMemcpy(hv.data(), r->d_vis().p, tp * 8, hipMemcpyDeviceToHost));
        for (uint32_t y = 0; y < r->rows(); ++y)
            for (uint32_t x = 0; x < r->width; ++x) {
                uint32_t yy = y + r->row0_in_tile;
                size_t idx = ((size_t)(yy / 8) * r->tiles_x + x / 8) * 64 + (yy % 8) * 8 + (x % 8);
                const unsigned long long key = hv[idx];
                const uint32_t zb = key == ~0ull ? 0x3F800000u : (uint32_t)(key >> 32);
                if (depth) std::memcpy(&depth[(size_t)y * r->width + x], &zb, 4);
                if (tri) tri[(size_t)y * r->width + x] = key == ~0ull ? 0xFFFFFFFFu : (uint32_t)key >> 3;
            }
    }
    return ARCTIC_OK;
}

int arctic_write_gbuffer(ArcticRenderer *r, const float *attrs, const uint32_t *material) {
    if (!r) return ARCTIC_E_INVALID;
    if (!attrs || !material) return r->fail(ARCTIC_E_INVALID, "write_gbuffer: null");
    int rc = select_device(r);
    if (rc) return rc;
    size_t px = (size_t)r->rows() * r->width;
    HIPCHECK(r, r->d_stage.ensure(px * 76));
    float *d_attrs = r->d_stage.as<float>();
    uint32_t *d_mat = reinterpret_cast<uint32_t *>(r->d_stage.as<char>() + px * 72);
    HIPCHECK(r, hipMemcpyAsync(d_attrs, attrs, px * 72, hipMemcpyHostToDevice, r->stream));
    HIPCHECK(r, hipMemcpyAsync(d_mat, material, px * 4, hipMemcpyHostToDevice, r->stream));
    HIPCHECK(r, launch_gbuffer_tile(r->gbuffer(), d_attrs, d_mat, r->width, r->rows(), r->row0_in_tile, r->tiles_x, r->tiles_y, 1, r->stream));
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    r->have_vis = false;
    r->have_gbuffer = true;
    r->have_order = false;   // (a caller's G-buffer comes without cost classes: the geometric order)
    return ARCTIC_OK;
}

int arctic_read_shadow_map(ArcticRenderer *r, float *depth) {
    if (!r) return ARCTIC_E_INVALID;
    if (!depth || !r->shadow_size) return r->fail(ARCTIC_E_INVALID, "read_shadow_map: null or no shadow map");
    int rc = select_device(r);
    if (rc) return rc;
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    if (int ov = check_item_overflow(r)) return ov;
    HIPCHECK(r, hipMemcpy(depth, r->d_shadow().p, (size_t)r->shadow_size * r->shadow_size * 4, hipMemcpyDeviceToHost));
    return ARCTIC_OK;
}

int arctic_write_shadow_map(ArcticRenderer *r, const float *depth) {
    if (!r) return ARCTIC_E_INVALID;
    if (!depth || !r->shadow_size) return r->fail(ARCTIC_E_INVALID, "write_shadow_map: null or no shadow map");
    int rc = select_device(r);
    if (rc) return rc;
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    HIPCHECK(r, hipMemcpy(r->d_shadow().p, depth, (size_t)r->shadow_size * r->shadow_size * 4, hipMemcpyHostToDevice));
    r->shadow_key.clear(); r->bounds_valid() = false;
    return ARCTIC_OK;
}

int arctic_read_output(ArcticRenderer *r, float *ldr, float *hdr, uint8_t *rgba8) {
    if (!r) return ARCTIC_E_INVALID;
    if (!r->have_output) return r->fail(ARCTIC_E_STATE, "read_output: nothing shaded into the handle's own buffers yet");
    if ((ldr || hdr) && !r->keep_float) return r->fail(ARCTIC_E_STATE, "read_output: float planes need ARCTIC_OPT_KEEP_FLOAT_OUTPUT");
    int rc = select_device(r);
    if (rc) return rc;
    size_t px = (size_t)r->rows() * r->width;
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    if (int ov = check_item_overflow(r)) return ov;
    if (ldr) HIPCHECK(r, hipMemcpy(ldr, r->d_ldr.p, px * 12, hipMemcpyDeviceToHost));
    if (hdr) HIPCHECK(r, hipMemcpy(hdr, r->d_hdr.p, px * 12, hipMemcpyDeviceToHost));
    if (rgba8) HIPCHECK(r, hipMemcpy(rgba8, r->d_rgba8.p, px * 4, hipMemcpyDeviceToHost));
    return ARCTIC_OK;
}

int arctic_frame_constants(const ArcticScene *scene, float *proj_view, float *light_proj_view, float *sun_dir) {
    if (!scene) return ARCTIC_E_INVALID;
    const ArcticCamera &c = scene->camera;
    if (proj_view) camera_proj_view(c.eye, c.rotation, c.aspect, c.fov_y, c.z_near_far[0], c.z_near_far[1], proj_view);
    if (light_proj_view) sun_proj_view(scene->sun.position, scene->sun.rotation, light_proj_view);
    if (sun_dir) dir_from_rot(scene->sun.rotation, sun_dir);
    return ARCTIC_OK;
}

int arctic_stats(ArcticRenderer *r, uint64_t *out, uint32_t n) {
    if (!r || !out) return ARCTIC_E_INVALID;
    if (select_device(r) == ARCTIC_OK) (void)hipStreamSynchronize(r->stream);
    if (r->h_counts) for (int i = 0; i < 4; ++i) r->stats[i] = r->h_counts[i];
    for (uint32_t i = 0; i < n && i < 16; ++i) out[i] = i < 8 ? r->stats[i] : i < 10 ? r->light_stats[i - 8] : i < 12 ? (r->h_counts ? r->h_counts[6 + (i - 10)] : 0) : r->edge_stats[i - 12];
    return ARCTIC_OK;
}

// the owner grid as plain numbers (pure host functions: tests/test_owner_grid.py checks that every tile row a shard stores is visited
// exactly once, for layouts no GPU test has time for)
int arctic_owner_grid(uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t shard_index, uint32_t shard_count,
                      uint32_t *grid /* grid_x, grid_y, first block row, 1 = grid rows are the shard's own pairs of tile rows */) {
    if (!grid || !width || !height || (band_rows ? (band_rows % TILE || !shard_count || shard_index >= shard_count) : (row_begin >= row_end || row_end > height))) return ARCTIC_E_INVALID;
    if (band_rows) { row_begin = 0; row_end = height; }
    uint32_t tile_y0, tiles_y;
    shard_tile_rows(height, row_begin, row_end, band_rows, shard_index, band_rows ? shard_count : 1u, tile_y0, tiles_y);
    owner_grid(width, height, tile_y0, tiles_y, band_rows / TILE, grid[0], grid[1], grid[2], grid[3]);
    return ARCTIC_OK;
}
int arctic_owner_visit(uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t shard_index, uint32_t shard_count,
                       uint32_t grid_row, uint32_t *block_row, int32_t *local_tile_rows /* [2]: the shard's tile row stored for the block's upper / lower tile row, -1: none */) {
    uint32_t grid[4];
    const int rc = arctic_owner_grid(width, height, row_begin, row_end, band_rows, shard_index, shard_count, grid);
    if (rc != ARCTIC_OK) return rc;
    if (!block_row || !local_tile_rows || grid_row >= grid[1]) return ARCTIC_E_INVALID;
    if (band_rows) { row_begin = 0; row_end = height; } else shard_count = 1, shard_index = 0;
    uint32_t tile_y0, tiles_y;
    shard_tile_rows(height, row_begin, row_end, band_rows, shard_index, shard_count, tile_y0, tiles_y);
    const int bt = (int)(band_rows / TILE);
    *block_row = owner_block_row(grid_row, grid[2], grid[3], bt, (int)shard_count, (int)shard_index, (int)tile_y0);
    for (int j = 0; j < 2; ++j) {
        int lrow;
        local_tile_rows[j] = owner_tile_row(*block_row, j, (int)tile_y0, (int)tiles_y, bt, (int)shard_count, (int)shard_index, lrow) ? lrow : -1;
    }
    return ARCTIC_OK;
}

int arctic_read_bin_counts(ArcticRenderer *r, int shadow_pass, uint32_t *out, uint64_t capacity_blocks, uint32_t *blocks_x, uint32_t *blocks_y) {
    if (!r) return ARCTIC_E_INVALID;
    const ArcticRenderer::GeoSet &G = r->geo[shadow_pass ? 1 : r->fwd()];
    if (blocks_x) *blocks_x = G.bins_x;
    if (blocks_y) *blocks_y = G.bins_y;
    const uint64_t n = (uint64_t)G.bins_x * G.bins_y;
    if (!out) return ARCTIC_OK;   // size query
    if (!n) return r->fail(ARCTIC_E_STATE, "read_bin_counts: the latest %s pass had no block owners (ARCTIC_OPT_RASTER_OWNER)", shadow_pass ? "shadow" : "forward");
    if (capacity_blocks < n) return r->fail(ARCTIC_E_CAPACITY, "read_bin_counts: %llu blocks, room for %llu", (unsigned long long)n, (unsigned long long)capacity_blocks);
    int rc = arctic_flush(r);
    if (rc) return rc;
    HIPCHECK(r, hipMemcpy(out, G.d_bin_count.p, n * 4, hipMemcpyDeviceToHost));
    return ARCTIC_OK;
}

int arctic_read_cull_counts(ArcticRenderer *r, int shadow_pass, uint32_t *out) {
    if (!r || !out) return ARCTIC_E_INVALID;
    const int set = shadow_pass ? 1 : r->fwd();
    const ArcticRenderer::GeoSet &G = r->geo[set];
    if (!G.cull_counted) return r->fail(ARCTIC_E_STATE, "read_cull_counts: the latest %s pass did not count (ARCTIC_OPT_DEBUG bit 10)", shadow_pass ? "shadow" : "forward");
    int rc = arctic_flush(r);
    if (rc) return rc;
    uint32_t c[N_GEO_COUNTERS];
    HIPCHECK(r, hipMemcpy(c, r->d_geo_counters.as<uint32_t>() + N_GEO_COUNTERS * set, sizeof c, hipMemcpyDeviceToHost));
    out[0] = G.n_tblocks; out[1] = c[5]; out[2] = G.n_vblocks; out[3] = c[6];
    return ARCTIC_OK;
}

int arctic_read_tile_trace(ArcticRenderer *r, uint64_t *out, uint64_t capacity_tiles, uint32_t *tiles_x, uint32_t *tiles_y) {
    if (!r) return ARCTIC_E_INVALID;
    if (tiles_x) *tiles_x = r->tiles_x;
    if (tiles_y) *tiles_y = r->tiles_y;
    const uint64_t n = (uint64_t)r->tiles_x * r->tiles_y;
    if (!out) return ARCTIC_OK;   // size query
    if (!r->tile_trace || r->d_tile_trace.cap < n * 32) return r->fail(ARCTIC_E_STATE, "read_tile_trace: no trace (set ARCTIC_OPT_TILE_TRACE, then shade)");
    if (capacity_tiles < n) return r->fail(ARCTIC_E_CAPACITY, "read_tile_trace: %llu tiles, room for %llu", (unsigned long long)n, (unsigned long long)capacity_tiles);
    int rc = select_device(r);
    if (rc) return rc;
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    HIPCHECK(r, hipMemcpy(out, r->d_tile_trace.p, n * 32, hipMemcpyDeviceToHost));
    return ARCTIC_OK;
}

int arctic_read_tile_order(ArcticRenderer *r, uint32_t *order, uint8_t *tile_class, uint64_t capacity_tiles, uint32_t *tiles_x, uint32_t *tiles_y) {
    if (!r) return ARCTIC_E_INVALID;
    if (tiles_x) *tiles_x = r->tiles_x;
    if (tiles_y) *tiles_y = r->tiles_y;
    const uint64_t n = (uint64_t)r->tiles_x * r->tiles_y, n_jobs = (uint64_t)((r->tiles_x + 3) / 4) * r->tiles_y;
    if (!order && !tile_class) return ARCTIC_OK;   // size query
    if (!r->have_order) return r->fail(ARCTIC_E_STATE, "read_tile_order: the G-buffer in place has no dispatch order (ARCTIC_OPT_TILE_ORDER, arctic_pass_gbuffer)");
    if (capacity_tiles < n) return r->fail(ARCTIC_E_CAPACITY, "read_tile_order: %llu tiles, room for %llu", (unsigned long long)n, (unsigned long long)capacity_tiles);
    int rc = select_device(r);
    if (rc) return rc;
    HIPCHECK(r, hipStreamSynchronize(r->stream));
    if (order) {   // the slots in dispatch order, the empty ones left out: n_jobs entries
        std::vector<uint32_t> slots(r->order_slots);
        HIPCHECK(r, hipMemcpy(slots.data(), r->d_tile_order.p, (size_t)r->order_slots * 4, hipMemcpyDeviceToHost));
        uint64_t k = 0;
        for (uint32_t e : slots) if (e != ORDER_NONE && k < n_jobs) order[k++] = e;
        if (k != n_jobs) return r->fail(ARCTIC_E_STATE, "read_tile_order: %llu strips in the order, %llu in the frame", (unsigned long long)k, (unsigned long long)n_jobs);
    }
    if (tile_class) HIPCHECK(r, hipMemcpy(tile_class, r->d_tile_class.p, n, hipMemcpyDeviceToHost));
    return ARCTIC_OK;
}

int arctic_set_option(ArcticRenderer *r, uint32_t option, int64_t value) {
    if (!r) return ARCTIC_E_INVALID;
    switch (option) {
    case ARCTIC_OPT_KEEP_FLOAT_OUTPUT: r->keep_float = value != 0; break;
    case ARCTIC_OPT_COUNT_LIGHT_EVALS: r->count_evals = value != 0; break;
    case ARCTIC_OPT_CULLING: r->culling = value != 0; break;
    case ARCTIC_OPT_DEBUG: r->debug = (int)value; break;
    case ARCTIC_OPT_HDR16: r->hdr16 = value != 0; break;
    case ARCTIC_OPT_VISBUFFER: r->visbuffer = value != 0; break;
    case ARCTIC_OPT_FRAMES_IN_FLIGHT:
        if (value < 0 || value > 3) return r->fail(ARCTIC_E_INVALID, "set_option: frames in flight must be 0 (the library's choice), 1, 2 or 3");
        r->frames_in_flight_opt = (int)value;
        if (value == 0) value = (uint64_t)r->rows() * r->width < 3000000ull ? 3 : 2;
        if ((int)value != r->frames_in_flight) {   // the sets change roles: nothing may be in flight, and `cur` must name a set that exists
            HIPCHECK(r, hipStreamSynchronize(r->stream));
            for (hipStream_t ps : r->prepass_stream) if (ps) HIPCHECK(r, hipStreamSynchronize(ps));
            if (r->own_stream_is_prepass()) HIPCHECK(r, hipStreamSynchronize(r->own_stream));
            if (r->cur >= (int)value) { r->cur = 0; r->have_vis = r->have_gbuffer = r->have_order = false; }
            r->released_valid[0] = r->released_valid[1] = r->released_valid[2] = false;
        }
        r->frames_in_flight = (int)value;
        break;
    case ARCTIC_OPT_MARKERS:
        g_markers.enable(value != 0);
        if (value && !g_markers.available()) return r->fail(ARCTIC_E_STATE, "set_option: libroctx64.so could not be loaded");
        break;
    case ARCTIC_OPT_LIGHT_PATH:
        if (value < 0 || value > 2) return r->fail(ARCTIC_E_INVALID, "set_option: light path must be 0..2");
        r->light_path = (int)value;
        break;
    case ARCTIC_OPT_ITEM_TABLE_FLOOR:
        if (value < 64 || value > 0x7FFFFFF0ll) return r->fail(ARCTIC_E_INVALID, "set_option: item table floor out of range");
        r->item_cap_floor = (uint32_t)value; r->geo[0].item_cap = r->geo[1].item_cap = r->geo[2].item_cap = r->geo[3].item_cap = 0;
        break;
    case ARCTIC_OPT_TILES_PER_WAVE:
        if (value < 0 || value > 64) return r->fail(ARCTIC_E_INVALID, "set_option: tiles per wave must be 0 (default) .. 64");
        r->tiles_per_wave = (uint32_t)value;
        break;
    case ARCTIC_OPT_TILE_TRACE: r->tile_trace = value != 0; break;
    case ARCTIC_OPT_TEXTURE_TILING: r->texture_tiling = value < 0 ? -1 : (value != 0); break;
    case ARCTIC_OPT_SAMPLER:
        if (value < 0 || (value & ~5ll)) return r->fail(ARCTIC_E_INVALID, "set_option: sampler is a mask of bit 0 (material footprints) and bit 2 (PCF taps); bit 1 (sRGB decode after filtering) exists in the oracle only");
        r->sampler = (int)value;
        break;
    case ARCTIC_OPT_TILE_ORDER: r->tile_order = value != 0; if (!r->tile_order) r->have_order = false; break;
    case ARCTIC_OPT_ORDER_TAIL:
        if (value < 0 || value > 1000) return r->fail(ARCTIC_E_INVALID, "set_option: order tail is per mille, 0..1000");
        r->order_tail = (uint32_t)value;
        break;
    case ARCTIC_OPT_SMALL_TRIANGLES:
        if (value < 0 || value > 1) return r->fail(ARCTIC_E_INVALID, "ARCTIC_OPT_SMALL_TRIANGLES: 0 or 1");
        r->small_triangles = (int)value; r->shadow_key.clear(); break;
    case ARCTIC_OPT_CLUSTER_CULL:
        if (value < 0 || value > 3 || value == 2) return r->fail(ARCTIC_E_INVALID, "ARCTIC_OPT_CLUSTER_CULL: 0, 1 or 3 (vertex blocks are skipped only where their clusters are)");
        r->cluster_cull = (int)value; r->shadow_key.clear(); break;
    case ARCTIC_OPT_RASTER_OWNER: r->raster_owner = value < 0 ? -1 : (int)(value & 3); r->shadow_key.clear(); break;
    case ARCTIC_OPT_SHADOW_CACHE: r->shadow_cache = value != 0; r->shadow_key.clear(); break;
    case ARCTIC_OPT_SHADOW_SHARDED: r->shadow_sharded = value != 0; r->shadow_key.clear(); break;
    default: return r->fail(ARCTIC_E_INVALID, "set_option: unknown option %u", option);
    }
    return ARCTIC_OK;
}

// ---- multi-GPU exchange steps (include/arctic_dist.h) ------------------------------------------------------------------------
// the plan, as pure host functions (no device, no handle): everything below is built on them
int arctic_exchange_plan(uint32_t width, uint32_t height, uint32_t band_rows, uint32_t world, const uint32_t *row_ranges,
                         uint32_t *rows, uint64_t *offset, uint64_t *total_bytes) {
    if (!rows || !offset || world == 0 || width == 0 || height == 0) return ARCTIC_E_INVALID;
    if (row_ranges) {
        if (band_rows) return ARCTIC_E_INVALID;
        for (uint32_t k = 0; k < world; ++k) {
            if (row_ranges[2 * k] > row_ranges[2 * k + 1] || row_ranges[2 * k + 1] > height) return ARCTIC_E_INVALID;
            rows[k] = row_ranges[2 * k + 1] - row_ranges[2 * k];
        }
    } else {
        if (!band_rows) return ARCTIC_E_INVALID;
        // band b -> rank b % world: full bands each, and the frame's last (possibly short) band to whoever it falls to
        const uint32_t n_bands = (height + band_rows - 1) / band_rows, last_rows = height - (n_bands - 1) * band_rows;
        for (uint32_t k = 0; k < world; ++k) {
            const uint32_t mine = n_bands / world + (k < n_bands % world ? 1u : 0u);
            rows[k] = mine * band_rows;
            if (mine && (n_bands - 1) % world == k) rows[k] -= band_rows - last_rows;
        }
    }
    uint64_t off = 0;
    for (uint32_t k = 0; k < world; ++k) { offset[k] = off; off += (uint64_t)rows[k] * width * 4; }
    if (total_bytes) *total_bytes = off;
    return ARCTIC_OK;
}

int arctic_exchange_row_source(uint32_t y, uint32_t height, uint32_t band_rows, uint32_t world, const uint32_t *row_ranges,
                               uint32_t *owner, uint32_t *local_row) {
    if (!owner || !local_row || world == 0 || y >= height || (!row_ranges && !band_rows) || (row_ranges && band_rows)) return ARCTIC_E_INVALID;
    return shard_row_source(y, band_rows, world, row_ranges, *owner, *local_row) ? 0 : 1;
}

int arctic_exchange_transfers(uint32_t width, uint32_t world, int32_t rank, int32_t root, const uint32_t *rows, const uint64_t *offset,
                              ArcticTransfer *out, uint32_t cap) {
    if (!rows || !offset || !out || world == 0 || rank < 0 || root < 0 || (uint32_t)rank >= world || (uint32_t)root >= world) return ARCTIC_E_INVALID;
    uint32_t n = 0;
    const uint64_t row_bytes = (uint64_t)width * 4;
    if (rank != root) {
        if (rows[rank]) { if (n >= cap) return ARCTIC_E_CAPACITY; out[n++] = ArcticTransfer{root, 1, 0, rows[rank] * row_bytes}; }
        return (int)n;
    }
    for (uint32_t k = 0; k < world; ++k) {
        if ((int32_t)k == root || !rows[k]) continue;
        if (n >= cap) return ARCTIC_E_CAPACITY;
        out[n++] = ArcticTransfer{(int32_t)k, 0, offset[k], rows[k] * row_bytes};
    }
    return (int)n;
}

namespace {

// rows of every rank's shard and where each shard starts in a staging buffer that holds them back to back; uploaded to d_layout
int upload_layout(ArcticRenderer *r, uint32_t world, const uint32_t *ranges /* 2 * world, or null: interleaved bands */, bool from_comm) {
    std::vector<uint32_t> rows(world, 0), rg(2 * (size_t)world, 0);
    std::vector<uint64_t> off(world, 0);
    uint64_t total = 0;
    if (ranges) std::memcpy(rg.data(), ranges, (size_t)world * 8);
    else if (!r->band_rows || r->shard_count != world) return r->fail(ARCTIC_E_INVALID, "frame layout: interleaved bands need band_rows > 0 and shard_count == world");
    if (arctic_exchange_plan(r->width, r->height, ranges ? 0u : r->band_rows, world, ranges, rows.data(), off.data(), &total) != ARCTIC_OK)
        return r->fail(ARCTIC_E_INVALID, "frame layout: bad row ranges");
    const size_t ranges_bytes = (((size_t)world * 8) + 15) / 16 * 16;
    HIPCHECK(r, r->d_layout.ensure(ranges_bytes + (size_t)world * 8));
    HIPCHECK(r, hipMemcpyAsync(r->d_layout.p, rg.data(), (size_t)world * 8, hipMemcpyHostToDevice, r->stream));
    HIPCHECK(r, hipMemcpyAsync(r->d_layout.as<char>() + ranges_bytes, off.data(), (size_t)world * 8, hipMemcpyHostToDevice, r->stream));
    HIPCHECK(r, hipStreamSynchronize(r->stream));   // the host vectors go out of scope
    r->peer_rows = rows; r->peer_offset = off; r->staging_bytes = total; r->peer_ranges = ranges ? rg : std::vector<uint32_t>();
    r->layout_world = world; r->layout_from_comm = from_comm;
    return ARCTIC_OK;
}
const uint32_t *layout_ranges(const ArcticRenderer *r) { return r->d_layout.as<uint32_t>(); }
const unsigned long long *layout_offsets(const ArcticRenderer *r) {
    return reinterpret_cast<const unsigned long long *>(r->d_layout.as<char>() + (((size_t)r->layout_world * 8) + 15) / 16 * 16);
}

// the second half of arctic_comm_init, free of RCCL calls (unit-testable through arctic_comm_init's error paths): `all` holds every
// rank's {row_begin, row_end, rows, band_rows}; checks them against this handle's sharding, uploads the layout, grows the shadow maps
int comm_adopt_layout(ArcticRenderer *r, const uint32_t *all, int world) {
    std::vector<uint32_t> ranges(2 * (size_t)world);
    for (int k = 0; k < world; ++k) {
        if (all[4 * k + 3] != r->band_rows) return r->fail(ARCTIC_E_INVALID, "comm_init: rank %d shards the frame differently (band_rows %u vs %u)", k, all[4 * k + 3], r->band_rows);
        ranges[2 * k] = all[4 * k]; ranges[2 * k + 1] = all[4 * k + 1];
    }
    int rc = upload_layout(r, (uint32_t)world, r->band_rows ? nullptr : ranges.data(), true);
    if (rc != ARCTIC_OK) return rc;
    for (int k = 0; k < world; ++k)
        if (r->peer_rows[(size_t)k] != all[4 * k + 2]) return r->fail(ARCTIC_E_INVALID, "comm_init: rank %d reports %u rows, the layout gives it %u", k, all[4 * k + 2], r->peer_rows[(size_t)k]);
    if (r->shadow_size) {   // room for the in-place all-gather of a sharded shadow map, in BOTH map sets (frames in flight flip between them)
        const size_t need = shadow_alloc_bytes(r);
        for (int s = 0; s < 2; ++s) {
            DevBuf &b = r->d_shadow_set[s];
            if (!b.p || b.cap >= need) continue;   // (the second set is allocated, at this size, when first used)
            HIPCHECK(r, hipStreamSynchronize(r->stream));
            HIPCHECK(r, b.ensure(need));
            HIPCHECK(r, launch_fill_u32(b.as<uint32_t>(), 0x3F800000u, (need - 8) / 4, r->stream));
            r->shadow_key.clear(); r->bounds_valid_set[s] = false;
        }
    }
    return ARCTIC_OK;
}

}  // namespace

int arctic_comm_unique_id(void *id_out, char *err, uint64_t err_len) {
    auto say = [&](const char *m) { if (err && err_len) std::snprintf(err, (size_t)err_len, "%s", m); return ARCTIC_E_DEVICE; };
    if (!id_out) return ARCTIC_E_INVALID;
    if (!g_rccl.load()) return say("arctic_comm_unique_id: librccl.so could not be loaded");
    Rccl::UniqueId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return say(g_rccl.why(rc));
    std::memcpy(id_out, id.internal, ARCTIC_COMM_ID_BYTES);
    return ARCTIC_OK;
}

int arctic_comm_init(ArcticRenderer *r, const void *id_bytes, int rank, int world) {
    if (!r) return ARCTIC_E_INVALID;
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) return r->fail(ARCTIC_E_INVALID, "comm_init: bad id / rank / world");
    if (r->comm) return r->fail(ARCTIC_E_STATE, "comm_init: the handle already has a communicator");
    if (r->band_rows && ((uint32_t)world != r->shard_count || (uint32_t)rank != r->shard_index))
        return r->fail(ARCTIC_E_INVALID, "comm_init: rank %d of %d does not match the handle's shard %u of %u", rank, world, r->shard_index, r->shard_count);
    int rc = select_device(r);
    if (rc) return rc;
    if (!g_rccl.load()) return r->fail(ARCTIC_E_DEVICE, "comm_init: librccl.so could not be loaded");
    Rccl::UniqueId id;
    std::memcpy(id.internal, id_bytes, ARCTIC_COMM_ID_BYTES);
    int nrc = g_rccl.CommInitRank(&r->comm, world, id, rank);
    if (nrc != 0) { r->comm = nullptr; return r->fail(ARCTIC_E_DEVICE, "ncclCommInitRank: %s", g_rccl.why(nrc)); }
    r->comm_rank = rank; r->comm_world = world;
    // From here on the handle owns a communicator: any failure below gives it back (arctic_comm_destroy) before returning, so the
    // handle is never left half initialised (comm set, no layout) and arctic_comm_init can be called again.
    DevBuf tmp;
    const auto finish = [&]() -> int {
        if (!r->comm_stream) {
            // (the runtime deals few hardware queues to many streams, and two streams on one queue do not overlap: DESIGN.md 4.3)
            if (r->stream != r->own_stream) { r->comm_stream = r->own_stream; r->comm_stream_borrowed = true; }
            else HIPCHECK(r, hipStreamCreateWithFlags(&r->comm_stream, hipStreamNonBlocking));
        }
        if (!r->ev_main) HIPCHECK(r, hipEventCreateWithFlags(&r->ev_main, hipEventDisableTiming));
        for (auto &f : r->inflight) if (!f.done) HIPCHECK(r, hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
        // every rank's shard layout: {row_begin, row_end, rows, band_rows} all-gathered once (the root places shards of unequal size)
        HIPCHECK(r, tmp.ensure((size_t)world * 16));
        const uint32_t mine[4] = {r->band_rows ? 0u : r->row_begin, r->band_rows ? 0u : r->row_end, r->rows(), r->band_rows};
        HIPCHECK(r, hipMemcpyAsync(tmp.as<char>() + (size_t)rank * 16, mine, 16, hipMemcpyHostToDevice, r->comm_stream));
        const int arc = g_rccl.AllGather(tmp.as<char>() + (size_t)rank * 16, tmp.p, 4, Rccl::Uint32, r->comm, r->comm_stream);
        if (arc != 0) return r->fail(ARCTIC_E_DEVICE, "ncclAllGather(layout): %s", g_rccl.why(arc));
        std::vector<uint32_t> all((size_t)world * 4);
        HIPCHECK(r, hipMemcpyAsync(all.data(), tmp.p, (size_t)world * 16, hipMemcpyDeviceToHost, r->comm_stream));
        HIPCHECK(r, hipStreamSynchronize(r->comm_stream));
        return comm_adopt_layout(r, all.data(), world);
    };
    rc = finish();
    tmp.release();
    if (rc != ARCTIC_OK) {
        const std::string why = r->err;
        (void)arctic_comm_destroy(r);
        r->err = why;
        return rc;
    }
    return ARCTIC_OK;
}

int arctic_comm_destroy(ArcticRenderer *r) {
    if (!r) return ARCTIC_E_INVALID;
    if (r->comm_stream) (void)hipStreamSynchronize(r->comm_stream);
    if (r->comm) { (void)g_rccl.CommDestroy(r->comm); r->comm = nullptr; }
    r->comm_rank = 0; r->comm_world = 1;
    if (r->layout_from_comm) { r->layout_world = 0; r->layout_from_comm = false; }
    for (auto &f : r->inflight) { if (f.done) (void)hipEventDestroy(f.done); f.done = nullptr; f.ptr = nullptr; }
    if (r->ev_main) { (void)hipEventDestroy(r->ev_main); r->ev_main = nullptr; }
    if (r->comm_stream) { if (!r->comm_stream_borrowed) (void)hipStreamDestroy(r->comm_stream); r->comm_stream = nullptr; r->comm_stream_borrowed = false; }
    return ARCTIC_OK;
}

int arctic_gather_frame(ArcticRenderer *r, const void *d_shard, void *d_frame, int root) {
    if (!r) return ARCTIC_E_INVALID;
    if (!r->comm || !r->layout_from_comm) return r->fail(ARCTIC_E_STATE, "gather_frame: no communicator (arctic_comm_init)");
    if (root < 0 || root >= r->comm_world) return r->fail(ARCTIC_E_INVALID, "gather_frame: bad root");
    const bool is_root = r->comm_rank == root;
    if (is_root && !d_frame) return r->fail(ARCTIC_E_INVALID, "gather_frame: the root needs a frame buffer");
    if (!d_shard) {
        if (!r->have_output) return r->fail(ARCTIC_E_STATE, "gather_frame: no shaded output to gather");
        d_shard = r->d_rgba8.p;
    }
    int rc = select_device(r);
    if (rc) return rc;
    const size_t row_bytes = (size_t)r->width * 4;
    HIPCHECK(r, hipEventRecord(r->ev_main, r->stream));               // after the shading that produced the shard ...
    HIPCHECK(r, hipStreamWaitEvent(r->comm_stream, r->ev_main, 0));   // ... on the communication stream, beside the next frame
    const uint8_t *src = static_cast<const uint8_t *>(d_shard);
    if (r->comm_world > 1) {
        // the transfers of this rank, straight from the plan (arctic_exchange_transfers: what the CPU tests check for worlds 2..8)
        std::vector<ArcticTransfer> xfers((size_t)r->comm_world);
        const int n_x = arctic_exchange_transfers(r->width, (uint32_t)r->comm_world, r->comm_rank, root, r->peer_rows.data(), r->peer_offset.data(),
                                                  xfers.data(), (uint32_t)xfers.size());
        if (n_x < 0) return r->fail(ARCTIC_E_STATE, "gather_frame: no transfer plan");
        if (is_root) HIPCHECK(r, r->d_staging.ensure((size_t)r->staging_bytes));
        int nrc = g_rccl.GroupStart();
        if (nrc != 0) return r->fail(ARCTIC_E_DEVICE, "ncclGroupStart: %s", g_rccl.why(nrc));
        for (int i = 0; i < n_x && nrc == 0; ++i) {
            const ArcticTransfer &t = xfers[(size_t)i];
            nrc = t.is_send ? g_rccl.Send(src, (size_t)t.bytes, Rccl::Uint8, t.peer, r->comm, r->comm_stream)
                            : g_rccl.Recv(r->d_staging.as<char>() + t.staging_offset, (size_t)t.bytes, Rccl::Uint8, t.peer, r->comm, r->comm_stream);
        }
        const int erc = g_rccl.GroupEnd();
        if (nrc != 0 || erc != 0) return r->fail(ARCTIC_E_DEVICE, "ncclSend/ncclRecv(frame shards): %s", g_rccl.why(nrc ? nrc : erc));
        if (is_root) {
            HIPCHECK(r, hipMemcpyAsync(r->d_staging.as<char>() + r->peer_offset[(size_t)root], src, (size_t)r->rows() * row_bytes, hipMemcpyDeviceToDevice, r->comm_stream));
            src = r->d_staging.as<uint8_t>();
        }
    }
    if (is_root)
        HIPCHECK(r, launch_place_rows(src, static_cast<uint8_t *>(d_frame), r->width, r->height, r->band_rows, (uint32_t)r->comm_world,
                                      layout_ranges(r), layout_offsets(r), r->comm_stream));
    // whoever writes this shard buffer next waits for the transfers that still read it
    ArcticRenderer::InFlight *slot = nullptr;
    for (auto &f : r->inflight) if (f.ptr == d_shard) slot = &f;
    if (!slot) for (auto &f : r->inflight) if (!f.ptr) { slot = &f; break; }
    if (!slot) { HIPCHECK(r, hipStreamSynchronize(r->comm_stream)); for (auto &f : r->inflight) f.ptr = nullptr; slot = &r->inflight[0]; }
    slot->ptr = d_shard;
    HIPCHECK(r, hipEventRecord(slot->done, r->comm_stream));
    return ARCTIC_OK;
}

int arctic_assemble_frame(ArcticRenderer *r, const void *d_staging, void *d_frame, uint32_t world, const uint32_t *row_ranges) {
    if (!r) return ARCTIC_E_INVALID;
    if (!d_staging || !d_frame || world == 0) return r->fail(ARCTIC_E_INVALID, "assemble_frame: null buffer or world == 0");
    int rc = select_device(r);
    if (rc) return rc;
    if (r->layout_from_comm && (uint32_t)r->comm_world != world) return r->fail(ARCTIC_E_STATE, "assemble_frame: the handle's communicator has another world size");
    if (!r->layout_from_comm && (rc = upload_layout(r, world, row_ranges, false)) != ARCTIC_OK) return rc;
    HIPCHECK(r, launch_place_rows(static_cast<const uint8_t *>(d_staging), static_cast<uint8_t *>(d_frame), r->width, r->height,
                                  row_ranges ? 0u : r->band_rows, world, layout_ranges(r), layout_offsets(r), r->stream));
    return ARCTIC_OK;
}

}  // extern "C"
