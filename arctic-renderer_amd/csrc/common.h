// common.h -- internal types shared by the host side (renderer.cpp) and the HIP kernels.
// Not part of the C-ABI (that is include/arctic_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace arctic {

// ---- data layout in HBM ----------------------------------------------------------------------
// Screen: tiles of 8x8 pixels, one tile = one wavefront (64 lanes); lane l <-> pixel (l & 7, l >> 3).
// Every per-pixel plane (visibility, G-buffer) is TILE-MAJOR: element (tile, lane) at tile*64 + lane,
// tile = ty * tiles_x + tx (ty counted from the first tile row of the row shard), so a wave-wide
// load of a float4 plane is one contiguous 1 KiB run.  The RGBA8 frame handed to the caller is
// row-major.
constexpr int TILE = 8;
constexpr int TILE_PIXELS = 64;
constexpr uint32_t NO_MATERIAL = 0xFFFFFFFFu;
constexpr uint32_t MAX_TARGET = 16384;   // largest render target / shadow map side: keeps the 24.8 edge arithmetic of edges.h in 32-bit factors

// G-buffer = the interpolated VSOut (forward.hlsl:41-48) minus SV_POSITION, 76 B / pixel, split by WHEN it is needed:
//   first wave of loads, every pixel (28 B): texture coordinates, shadow lookup, material
//     a float4  uv.xy, light_space_position.xy
//     b float3  light_space_position.zw, material id (bits)          (12-byte stride)
//   second wave of loads, only tiles with a lit pixel (48 B): position and tangent frame
//     c float4  world.xyz, t.x
//     d float4  t.yz, b.xy
//     e float4  b.z, n.xyz
// (a fully shadowed pixel is ambient * base: it never needs its normal or position, see k_material)
struct GBuffer {
    float4 *a;
    float *b;
    float4 *c, *d, *e;
};
__host__ __device__ inline void gbuffer_pack(const float *v /*18 attrs in VSOut order*/, uint32_t mat, float4 &A, float *B3, float4 &C, float4 &D, float4 &E) {
    A = make_float4(v[0], v[1], v[14], v[15]);
    B3[0] = v[16]; B3[1] = v[17]; B3[2] = __builtin_bit_cast(float, mat);
    C = make_float4(v[11], v[12], v[13], v[2]);
    D = make_float4(v[3], v[4], v[5], v[6]);
    E = make_float4(v[7], v[8], v[9], v[10]);
}

// what k_resolve needs to leave a COST CLASS per tile next to the G-buffer (a dispatch hint for the shading pass, never a result):
// the shadow map's min/max table as the shading kernel's first shadow test reads it
struct TileHint {
    uint8_t *tile_class;     // one byte per tile of the shard: 1 = a pixel of the tile can be lit / takes the environment lookup; null: no hint wanted
    const float2 *bounds;    // ShadeParams::shadow_bounds, or null (no map, no table): every covered pixel counts as costly
    uint32_t S, pitch;       // shadow-map size, entries per table row
    uint32_t sky;            // 1: an environment map is set, pixels without geometry are costly too
};

// transformed vertex (the reference's VSOut), 96 B
struct XVert {
    float clip[4];
    float attr[18];  // uv2, t3, b3, n3, world3, light4
    float pad[2];
};

// per-object record uploaded every frame (Scene.objects flattened, scene.hpp:69-73 + the mesh it points to)
struct ObjectRec {
    float trs[16];            // model matrix, glm column-major
    const float *vertices;    // ArcticVertex[] as 14 floats each
    const uint32_t *indices;
    uint32_t n_vertices, n_triangles;
    uint32_t first_xvert;     // offset of this object's vertices in the transformed-vertex buffer
    uint32_t first_triangle;  // draw-order id of this object's first triangle
    uint32_t material;
    uint32_t pad;
};

// one rasterisable (clipped, culled, oriented) triangle, 128 B
struct SetupRec {
    int32_t X[3], Y[3];   // 24.8 fixed point, oriented so area2 > 0
    float z[3], iw[3];    // z/w, 1/w
    float bary[3][3];     // vertex k of this triangle as a combination of the source triangle's vertices
    int64_t area2;
    int32_t px0, py0, px1, py1;  // inclusive pixel bounds (scissored)
    uint32_t src_tri, object;
    uint32_t order_id;    // 8 * src_tri + index of this sub-triangle: the tie-break of equal depths (first drawn wins)
    uint32_t pad;
};
static_assert(sizeof(SetupRec) == 128, "SetupRec layout");

// what the rasteriser reads per work item, written by k_setup next to the SetupRec, 128 B (two scalar loads).
// Edge function i of the record at pixel (px, py), i.e. edges.h's edge_eval, as the plane C + A*px + B*py held in binary64:
// with every snapped coordinate below 2^24 in magnitude all three terms and every partial sum are integers below 2^53, so
// the binary64 evaluation IS the 64-bit integer one (flag RASTER_EXACT_F64; other records take the integer path).
struct RasterRec {
    double A[3], B[3], C[3];   // C[1] has the fill-rule threshold folded in (covered <=> value >= 0); edges 0 and 2 also feed the depth
    double t0, t2;             // fill-rule thresholds of edges 0 and 2: covered <=> e >= t, t = 0 (top-left edge) or 1
    float z0, dz1, dz2;        // depth plane: z = fma(l2, dz2, fma(l1, dz1, z0))
    float inv_area;            // 1 / (float)area2
    uint32_t order_id;
    uint32_t flags;
    uint32_t src_vertex[3];    // the SOURCE triangle's transformed vertices (indices into the pass's XVert table) and ...
    uint32_t material;         // ... its material: what k_material_vis needs to interpolate the attributes of a pixel (one 16-byte load)
};
static_assert(sizeof(RasterRec) == 128, "RasterRec layout");
constexpr uint32_t RASTER_EXACT_F64 = 1u;
// the record is a whole source triangle (no plane cut it): the rows of SetupRec::bary are unit vectors, so the source triangle's
// barycentrics ARE the record's perspective-corrected ones -- (b0, b1, b2), or (b0, b2, b1) when set-up exchanged vertices 1 and 2
// to orient the triangle (RASTER_SWAPPED) -- and b * 1 + b' * 0 + b'' * 0 is b bit for bit (the b are never negative zeros)
constexpr uint32_t RASTER_UNIT_BARY = 2u, RASTER_SWAPPED = 4u;
// every edge function of the record stays below 2^31 in magnitude over every 16x16 block its bounding box touches (k_setup: plan_triangle): the shadow
// raster evaluates such a record's pixels in 32-bit integers from the binary64 value at its first pixel (item_pixels)
constexpr uint32_t RASTER_I32 = 8u;
// a work item = {record, block}: block = bx | by << 12 | ITEM_SCISSOR (the 16x16 block is cut by the scissor / the target's edge);
// record ITEM_SKIP: an entry of a large record whose block no edge function reaches (its slot was taken before that was known)
constexpr uint32_t SETUP_THREADS = 256;   // triangles per workgroup of k_setup (one slot atomic per workgroup).  512 is 1 us faster alone, but a 512-thread
                                          // workgroup finds no room beside a shading kernel of 256-thread workgroups (frames in flight): it starved until that ended
constexpr uint32_t ITEM_SCISSOR = 1u << 24;
constexpr uint32_t ITEM_INEXACT = 1u << 25;   // the record takes the 64-bit integer rasteriser (RASTER_EXACT_F64 clear): never binned
constexpr uint32_t ITEM_SKIP = 0xFFFFFFFFu;
// Block ownership (round 3, k_bin / k_raster_owned): every 16x16 block of the target has a bin of BIN_SLOTS record indices; k_bin moves
// the work items into the bins of their blocks (one returning atomic per item on the block's counter), one wave per block then merges
// its bin in registers and WRITES the block once -- no clear, no early read, no per-pixel atomic.  Items that find their bin full
// (a dense mesh in a few blocks) and inexact records stay in the item table for the merging atomic rasteriser (k_raster), which
// runs afterwards and spreads them over the whole chip.
constexpr uint32_t BIN_SLOTS = 32;
constexpr uint32_t N_GEO_COUNTERS = 8;   // device counters per table set: [0] records, [1] work items, [2] a table overflowed, [3] clip-list entries,
                                         // [4] k_bin left items for the atomic rasteriser, [5] clusters k_setup skipped, [6] vertex blocks k_vertex skipped (both
                                         // counted only under GeomParams::raster_flags bit 1); zeroed by k_vertex
struct BinTables {
    uint32_t *count;      // items offered to each block's bin (may exceed BIN_SLOTS: the bin holds the first BIN_SLOTS)
    uint32_t *slots;      // BIN_SLOTS record indices per block
    uint32_t blocks_x;    // blocks per row of the table (whole target, whatever the shard)
    uint32_t n_blocks;    // entries of count
    // the grid of k_raster_owned: grid_x x grid_y blocks starting at block row by0; local_rows: grid row i is the i-th pair of
    // tile rows this interleaved shard owns (bands of an even number of tile rows), else block row by0 + i of the target
    uint32_t grid_x, grid_y, by0, local_rows;
    uint32_t count_all;   // ARCTIC_OPT_DEBUG bit 9: full bins are asked too, so that count is every item offered (arctic_read_bin_counts as a histogram)
};

// frame constants for the geometry kernels
struct GeomParams {
    float clip_from_world[16];   // proj_view (forward) or light_proj_view (shadow)
    float light_from_world[16];  // light_proj_view
    float vp_w, vp_h;            // viewport size in pixels
    int32_t sc_x0, sc_y0, sc_x1, sc_y1;  // scissor (exclusive upper)
    int32_t cull_front;          // 0: cull back faces (forward pass), 1: cull front faces (shadow pass)
    int32_t tiles_x;             // tile columns of the target
    int32_t tile_y0;             // first tile row stored (row shard); 0 for the shadow map
    int32_t pitch;               // shadow pass: row pitch of the row-major depth map (= S)
    int32_t band_tiles;          // interleaved shard: tile rows per band (0 = contiguous shard), else see row_* below
    int32_t shard_index, shard_count;
    int32_t raster_flags;        // 1: every record takes the integer rasteriser (ARCTIC_OPT_DEBUG bit 5: A/B of the two paths)
    int32_t tiles_y;             // tile rows the target stores (this shard's)
};
// tile-row bookkeeping of a shard.  ty_rel = global tile row - tile_y0.  Contiguous shard: local == ty_rel.  Interleaved
// shard: bands of band_tiles tile rows are dealt round-robin, shard r owns bands r, r + n, ...; local rows are packed.
__host__ __device__ inline bool row_owned(int ty_rel, int band_tiles, int n, int r) { return band_tiles == 0 || ((ty_rel / band_tiles) % n) == r; }
__host__ __device__ inline int row_local(int ty_rel, int band_tiles, int n) { return band_tiles == 0 ? ty_rel : ((ty_rel / band_tiles) / n) * band_tiles + ty_rel % band_tiles; }
__host__ __device__ inline int row_global(int lt, int band_tiles, int n, int r) { return band_tiles == 0 ? lt : ((lt / band_tiles) * n + r) * band_tiles + lt % band_tiles; }

// The owner grid of a forward prepass (k_raster_owned; the host-side plan: arctic_owner_grid / arctic_owner_visit).  ONE definition of
// which block row a grid row visits and which of a block's two tile rows the shard stores, for the kernel and for the CPU tests.
__host__ __device__ inline uint32_t owner_block_row(uint32_t gy, uint32_t by0, uint32_t local_rows, int band_tiles, int shard_count, int shard_index, int tile_y0) {
    // of an interleaved shard only the rows it owns are in the grid (bands of an even number of tile rows): grid row gy = its gy-th pair of tile rows
    return local_rows ? (uint32_t)(row_global((int)(2 * gy), band_tiles, shard_count, shard_index) + tile_y0) >> 1 : by0 + gy;
}
// tile row j (0: upper, 1: lower) of block row `by`: does this shard store it, and as which of its tile rows?
__host__ __device__ inline bool owner_tile_row(uint32_t by, int j, int tile_y0, int tiles_y, int band_tiles, int shard_count, int shard_index, int &lrow) {
    const int ty_rel = (int)(2 * by) + j - tile_y0;
    lrow = row_local(ty_rel, band_tiles, shard_count);
    return ty_rel >= 0 && row_owned(ty_rel, band_tiles, shard_count, shard_index) && lrow < tiles_y;
}
// grid_x x grid_y blocks from block row by0 (see BinTables) for a target of tw x th pixels stored as tiles_y tile rows from tile_y0
inline void owner_grid(uint32_t tw, uint32_t th, uint32_t tile_y0, uint32_t tiles_y, uint32_t band_tiles, uint32_t &grid_x, uint32_t &grid_y, uint32_t &by0, uint32_t &local_rows) {
    grid_x = (tw + 15) / 16;
    if (band_tiles && band_tiles % 2 == 0 && tile_y0 == 0) { by0 = 0; grid_y = (tiles_y + 1) / 2; local_rows = 1; }
    else if (band_tiles) { by0 = 0; grid_y = (th + 15) / 16; local_rows = 0; }   // bands of an odd number of tile rows: every block row looks for its own tile rows
    else { by0 = tile_y0 / 2; grid_y = tiles_y ? (tile_y0 + tiles_y + 1) / 2 - by0 : 0; local_rows = 0; }
}
// the tile rows a shard stores: from tile_y0, tiles_y of them (interleaved: the tile rows of its bands, packed)
inline void shard_tile_rows(uint32_t height, uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t shard_index, uint32_t shard_count, uint32_t &tile_y0, uint32_t &tiles_y) {
    tile_y0 = row_begin / TILE;
    tiles_y = (row_end + TILE - 1) / TILE - tile_y0;
    if (band_rows) {
        const int bt = (int)(band_rows / TILE), all = (int)((height + TILE - 1) / TILE);
        tiles_y = 0;
        for (int ty = 0; ty < all; ++ty) if (row_owned(ty, bt, (int)shard_count, (int)shard_index)) ++tiles_y;
    }
}

// Which shard holds frame row y, and as which of its rows (the exchange step, include/arctic_dist.h).  Interleaved sharding
// (band_rows > 0): row y belongs to band y / band_rows, the band to rank band % world, and is that rank's local row
// (band / world) * band_rows + y % band_rows (all its earlier bands are full).  Row ranges (band_rows == 0): ranges[2k],
// ranges[2k+1] = rank k's [begin, end); false for a row outside every range.  One definition for the placement kernel
// (exchange.hip) and the host-side plan (arctic_exchange_plan / arctic_exchange_row_source).
__host__ __device__ inline bool shard_row_source(uint32_t y, uint32_t band_rows, uint32_t world, const uint32_t *ranges, uint32_t &owner, uint32_t &local) {
    if (band_rows) {
        const uint32_t band = y / band_rows;
        owner = band % world;
        local = (band / world) * band_rows + y % band_rows;
        return true;
    }
    for (uint32_t k = 0; k < world; ++k)
        if (y >= ranges[2 * k] && y < ranges[2 * k + 1]) { owner = k; local = y - ranges[2 * k]; return true; }
    return false;
}

// texture descriptor, 32 B (one s_load_dwordx8); three consecutive per material: diffuse (sRGB), normal, metal-rough
// When a material's three images have equal size they are stored PACKED: one image of 8-byte texels holding exactly the
// eight channels ps_main reads (diffuse rgb, normal rgb, metal-rough gb; layout in shade.hip) WITH A ONE-TEXEL WRAP BORDER:
// (w + 2) x (h + 2) texels, padded texel (X, Y) = image texel ((X - 1) mod w, (Y - 1) mod h).  A bilinear footprint under
// WRAP addressing starts at image texel (i0, j0) in [-1, w - 1] x [-1, h - 1] and ends one further, i.e. it is padded texels
// (i0 + 1 .. i0 + 2, j0 + 1 .. j0 + 2): never a wrap test, and the two texels of a row are adjacent -- one footprint = two
// 16-byte loads instead of twelve 4-byte ones.  Bit 31 of w marks it; descriptors 1 and 2 are then unused.
struct TexDesc {
    const uint32_t *texels;  // RGBA8 little endian (r = low byte), row-major, tightly packed -- or the packed, bordered image
    uint32_t w, h;           // image size (without the border); w bit 31: packed
    float wf, hf;            // (float)w, (float)h: exact (sides are below 65536)
    uint32_t pitch;          // packed: texels per row of the bordered image = w + 2
    uint32_t tile_row_bytes; // packed only.  0: the bordered image is row-major.  Otherwise it is stored in TILES of 4 x 4 texels (128 bytes = one cache line, row-major
                             // inside the tile), tile_row_bytes = 128 * tiles per row of tiles: padded texel (X, Y) at (Y >> 2) * tile_row_bytes + (X >> 2) * 128 +
                             // ((Y & 3) * 4 + (X & 3)) * 8.  The reference creates ONE mip level (rhi.cpp:550), so a large texture on a small object is minified at
                             // mip 0: every pixel's 2 x 2 footprint is then its own cache lines -- two of them in a row-major image (the footprint's two rows),
                             // (1 + 1/4)^2 = 1.56 on average in 4 x 4 tiles (round 5; the choice is made per material at creation: ARCTIC_OPT_TEXTURE_TILING)
};
static_assert(sizeof(TexDesc) == 32, "TexDesc layout (tex_desc loads it with one s_load_dwordx8)");
constexpr uint32_t TEX_INTERLEAVED = 0x80000000u;

// point light as uploaded (scene.hpp:88-94): float3 pos, pad, float3 color, pad = 2 x float4
// The kernels' argument block.  The fields are ordered by WHEN a wave of k_material needs them, in 64-byte blocks, so that each
// phase of a tile is ONE batch of scalar loads behind one wait (shade.hip: args_a / args_b / args_c / args_d) instead of a scalar-cache
// round trip per field at its point of use:
//   block A  before a tile's first bytes can be asked for: the G-buffer planes, the tile grid, the dispatch order
//   block B  while those bytes are in flight: what turns them into texel and shadow-table addresses, and the target
//   block C  while the texels are in flight: the epilogue's constants, the lit pixels' constants
//   block D  lit tiles only: sun, light list
// Everything behind block D is read at its point of use (cold paths: general tile, skybox, statistics, visibility plane).
struct ShadeParams {
    // ---- block A, byte 0
    GBuffer g;                   // 5 plane pointers
    uint32_t tiles_x, tiles_y;   // tile grid of the shard
    uint32_t tiles_per_wave;     // tiles a wave of k_material shades one after the other (filled by launch_shade)
    uint32_t group_stride;       // ... 8 * group_stride tile rows apart: ceil(groups of 8 tile rows / tiles_per_wave) (filled by launch_shade)
    int32_t debug;               // timing experiments only: 1 skip material textures, 2 skip shadow test, ... (ARCTIC_OPT_DEBUG)
    uint32_t n_materials;
    // ---- block B, byte 64
    const TexDesc *tex;          // 3 * n_materials
    const float2 *shadow_bounds; // conservative (min, max) of the map per 4x4-aligned 8x8 texel block (k_shadow_bounds), or null
    const float *shadow_map;     // S*S floats row-major, or null
    uint8_t *out_rgba8;          // rows*width*4, row-major
    uint32_t shadow_size;
    uint32_t bounds_pitch;       // entries per row of shadow_bounds = shadow_bounds_pitch(S)
    uint32_t width;              // frame width in pixels
    uint32_t rows;               // rows of this shard
    uint32_t row0_in_tile;       // row_begin - tile_y0*8: offset of the shard's first row inside its first tile row
    int32_t culling;
    int32_t hdr16;               // 1: round ps_main's colour through binary16 like the reference's RGBA16F target
    int32_t tm_method;
    // ---- block C, byte 128
    float ambient;
    float inv_gamma;
    float exposure;
    uint32_t pad_c0;
    float *out_ldr;              // optional rows*width*3
    float *out_hdr;              // optional rows*width*3
    float eye[3];                // (byte 160: from here to the end of block D is what a lit tile loads in one batch)
    uint32_t n_lights;
    const float4 *light_pairs;   // the lights as pairs, 3 float4 per pair {x0,x1,y0,y1} {z0,z1,r0,r1} {g0,g1,b0,b1}; an odd count is padded with a black light
    const float4 *lights;        // 2 float4 per light, as uploaded
    // ---- block D, byte 192
    float sun_dir[3];
    uint32_t pad_d0;
    float sun_color[3];
    uint32_t pad_d1;
    // ---- the rest: read at the point of use
    const float *srgb_lut;       // 256 floats, sRGB8 -> linear
    unsigned long long *stats;   // STATS kernels only: [0] point-light evaluations, [1] lit pixels, [2] evaluations with n.wi > 0,
                                 // [3] (tile, light) pairs with n.wi <= 0 in every lit lane, [4] tiles with a lit pixel
    unsigned long long *trace;   // ARCTIC_OPT_TILE_TRACE: 4 x u64 per tile (shade.hip: trace_end), or null
    const uint32_t *tile_order;  // the pass's dispatch order (k_tile_order: strips of 4 tiles, ty << 16 | strip column), n_jobs entries; null: the geometric order
    uint32_t n_jobs;
    uint32_t pad_o;
    // whole frames without a G-buffer (k_material_vis): the visibility plane and what the prepass left behind
    const unsigned long long *vis; const SetupRec *recs; const RasterRec *rrecs; const uint32_t *rec_of; const ObjectRec *objs; const XVert *xv;
    // skybox (skybox.hlsl:61-90): environment map for pixels without geometry; env == null -> black
    const float4 *env;                  // RGBA32F equirect, row-major
    uint32_t env_w, env_h;
    float sky_fwd[3], sky_right[3], sky_up[3];   // ray through ndc (x,y) = fwd + x*right + y*up (right/up scaled by the frustum)
    float ndc_sx, ndc_sy;               // 2/width, 2/height of the whole frame
    int32_t band_tiles, shard_index, shard_count, tile_y0;   // local tile row -> global row (see row_global)
    int32_t compact_tables;             // 1: the record, vertex and object tables are below 4 GiB each: k_material_vis addresses them with 32-bit byte offsets
};
static_assert(offsetof(ShadeParams, tex) == 64 && offsetof(ShadeParams, ambient) == 128 && offsetof(ShadeParams, sun_dir) == 192 &&
              offsetof(ShadeParams, srgb_lut) == 224, "ShadeParams: the blocks k_material loads in one batch each");
struct ShadeLaunch {
    hipStream_t stream;
    uint32_t loop;       // 1: scalar light loop; 2: two lights at a time in packed fp32 (both read the lights through the scalar cache)
    uint32_t from_vis;   // 1: k_material_vis (attributes interpolated from the visibility plane) instead of k_material
    uint32_t stats;      // 1: the counting variant (ShadeParams::stats)
    uint32_t tiles_per_wave;   // 0: DEFAULT_TILES_PER_WAVE (ARCTIC_OPT_TILES_PER_WAVE)
};
constexpr uint32_t N_SHADE_STATS = 9;   // [0..4] light statistics, [5..8] shadow-edge statistics of the fast tile (shade.hip)
constexpr uint32_t DEFAULT_TILES_PER_WAVE = 2;
// The dispatch order's slots (geometry.hip k_tile_order, shade.hip next_tile): eight lists -- list x = the strips of tile rows ty = x (mod 8) -- of L slots
// each (the longest list in whole groups), interleaved in groups of `group` slots: slot (q * 8 + x) * group + k = entry q * group + k of list x.  Block b of
// the pass takes slots b * group ..., i.e. ONE list: b % 8 = x, the XCD the hardware deals block b to.  ORDER_NONE: no strip in this slot.
constexpr uint32_t ORDER_NONE = 0xFFFFFFFFu;
__host__ __device__ inline uint32_t order_slot(uint32_t list, uint32_t i, uint32_t group) { return ((i / group) * 8u + list) * group + i % group; }
__host__ __device__ inline uint32_t order_slots(uint32_t tiles_x, uint32_t tiles_y, uint32_t group) {
    const uint32_t longest = ((tiles_x + 3) / 4) * ((tiles_y + 7) / 8);
    return 8u * ((longest + group - 1) / group * group);
}
constexpr uint32_t SMALL_FRAME_TILES = 48000;   // fewer 8x8 tiles than this (~3 Mpx): one tile per wave (launch_shade)
// the shadow-bounds table: one entry per 4x4 texel block; only for maps whose 25 PCF taps (4e-4 S apart end to end, in fp32)
// span less than 2 texels, so that a footprint never leaves the 4x4 window behind its first texel
inline uint32_t shadow_bounds_pitch(uint32_t S) { return S >= 4 && S <= 4900 ? (S + 3) / 4 : 0; }

// ---- kernel launchers (geometry.hip, shade.hip) ---------------------------------------------
// every launcher enqueues on `s` and returns the launch error, never synchronises.
hipError_t launch_vertex(const ObjectRec *objs, const uint32_t *block_obj, const uint32_t *block_first,
                         uint32_t n_blocks, const GeomParams &gp, XVert *xv, int clip_only, uint32_t *counters /*zeroed here for k_setup*/,
                         unsigned long long *clear, unsigned long long clear_value, size_t clear_count /*the pass's target, cleared in the same launch*/,
                         uint32_t *zero, size_t zero_count /*the bin counters of an owned raster, zeroed in the same launch*/,
                         const float *block_bounds /*6 floats per block: object-space box of every cluster that uses a vertex of the block, or null (no culling)*/, hipStream_t s);
hipError_t launch_setup(const ObjectRec *objs, const uint32_t *block_obj, const uint32_t *block_first, uint32_t n_blocks,
                        const GeomParams &gp, const XVert *xv, SetupRec *recs, RasterRec *rrecs, uint32_t *rec_of /*8 per source triangle*/,
                        uint2 *items, uint32_t item_cap, uint32_t rec_cap, uint32_t *counters /*records, items, overflow, clip-list length: zeroed by launch_vertex*/,
                        uint2 *clip_list /*one entry per source triangle*/,
                        const float *block_bounds /*6 floats per block of SETUP_THREADS triangles: object-space box of their positions, or null*/,
                        uint32_t *small_depth_bits /*shadow pass: the map, when k_setup is to draw the triangles with a small bounding box itself; else null*/, hipStream_t s);
uint32_t raster_grid_blocks(bool depth_only, uint32_t cu_count);
// after_owned: the blocks were written by k_raster_owned; only what k_bin left in the item table (counters[4] != 0) is drawn
hipError_t launch_raster_vis(const SetupRec *recs, const RasterRec *rrecs, const uint2 *items, uint32_t item_cap, const uint32_t *counters, uint32_t grid_blocks,
                             const GeomParams &gp, unsigned long long *vis, uint32_t *host_counts /*mapped: records, items*/, uint32_t *host_overflow, bool after_owned, hipStream_t s);
hipError_t launch_raster_depth(const SetupRec *recs, const RasterRec *rrecs, const uint2 *items, uint32_t item_cap, const uint32_t *counters, uint32_t grid_blocks,
                               const GeomParams &gp, uint32_t *depth_bits, uint32_t *host_counts, uint32_t *host_overflow, bool after_owned, hipStream_t s);
// block ownership: k_bin (work items -> bins) + k_raster_owned (one wave per block, the block written once); the atomic rasteriser
// launched after them finds what they left (counters[4]) or nothing
hipError_t launch_raster_owned(bool depth_only, const RasterRec *rrecs, const uint2 *items, uint2 *left /*what the bins did not take: counters[4] entries*/, uint32_t item_cap, uint32_t *counters, const BinTables &B,
                               const GeomParams &gp, unsigned long long *vis, uint32_t *depth_bits,
                               bool merge /*shadow pass: the map is cleared and holds k_setup's small triangles; owners merge into it and empty bins have none*/, hipStream_t s);
hipError_t launch_resolve(const unsigned long long *vis, const SetupRec *recs, const RasterRec *rrecs, const uint32_t *rec_of, const ObjectRec *objs, const XVert *xv,
                          const GeomParams &gp, uint32_t n_tiles, GBuffer g, const TileHint &hint, hipStream_t s);
// the shading pass's dispatch order (ShadeParams::tile_order) from the cost classes k_resolve left: lists = scratch of 2 N words, N = ceil(tiles_x / 4) *
// tiles_y strips; order = order_slots() words; tail_permille: the last part of each list that holds cheap strips only; group = tiles per wave of the pass
hipError_t launch_tile_order(const uint8_t *tile_class, uint32_t tiles_x, uint32_t tiles_y, uint32_t tail_permille, uint32_t group, uint32_t *lists, uint32_t *order, hipStream_t s);
hipError_t launch_fill_u64(unsigned long long *p, unsigned long long v, size_t n, hipStream_t s);
hipError_t launch_fill_u32(uint32_t *p, uint32_t v, size_t n, hipStream_t s);
hipError_t launch_shade(const ShadeParams &sp, const ShadeLaunch &L);
hipError_t launch_shadow_bounds(const float *map, uint32_t S, float2 *blocks, float2 *bounds, hipStream_t s);
hipError_t launch_place_rows(const uint8_t *staging, uint8_t *frame, uint32_t width, uint32_t height, uint32_t band_rows, uint32_t world,
                             const uint32_t *d_ranges, const unsigned long long *d_shard_offset, hipStream_t s);
hipError_t launch_post_process(const float4 *hdr, uint32_t w, uint32_t h, int32_t tm, float inv_gamma, float exposure,
                               uint8_t *rgba8, float *ldr, hipStream_t s);
hipError_t launch_gbuffer_tile(GBuffer g, float *attrs, uint32_t *mat, uint32_t width, uint32_t rows,
                               uint32_t row0_in_tile, uint32_t tiles_x, uint32_t tiles_y, int to_tiled, hipStream_t s);

// ---- host math (host_math.cpp): glm-equivalent builders, scene.cpp:9-19,41-70 ----------------
void dir_from_rot(const float rot_deg[2], float out[3]);
void camera_proj_view(const float eye[3], const float rot_deg[2], float aspect, float fov_y_deg, float zn, float zf, float out[16]);
void sun_proj_view(const float pos[3], const float rot_deg[2], float out[16]);
float srgb8_to_linear(int c);
void camera_sky_basis(const float rot_deg[2], float aspect, float fov_y_deg, float fwd[3], float right[3], float up[3]);

}  // namespace arctic
