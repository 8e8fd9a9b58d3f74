/* arctic_dist.h -- the multi-GPU exchange steps of the forward PBR shading path, below any language binding.
 *
 * The reference is single-adapter (src/renderer/rhi.cpp:120-124); BASELINE.json's north_star shards the frame in screen
 * space over the GPUs of one node and gathers the finished tiles.  Model: ONE process = ONE handle (arctic_hip.h,
 * ArcticCreateInfo::row_begin/row_end or band_rows/shard_index/shard_count) = ONE GPU.  This header adds what a C++ host
 * like src/app.cpp needs beyond rendering its shard -- the collectives themselves -- so that Python (sharding.py, bench.py)
 * is a thin caller and not the place where the exchange lives:
 *
 *   arctic_comm_unique_id / arctic_comm_init   an RCCL communicator owned by the handle (ncclCommInitRank; the 128-byte id
 *                                              travels over whatever channel the host already has: MPI, a file, torch.distributed)
 *   arctic_gather_frame                        RGBA8 shards -> the full row-major frame on the root: grouped ncclSend/ncclRecv
 *                                              of each rank's contiguous shard (R - 1 independent transfers into the root over the
 *                                              xGMI mesh) into a staging buffer, then ONE placement kernel (interleaved bands or
 *                                              row ranges -> frame rows).  Runs on a communication stream of the handle, ordered
 *                                              after the shading that produced the shard and overlapping the next frame's.
 *   arctic_assemble_frame                      the placement kernel alone (staging = all shards back to back in rank order)
 *   ARCTIC_OPT_SHADOW_SHARDED                  the shadow map stops being redundant work: each rank rasterises S / R light-space
 *                                              rows and one in-place ncclAllGather completes the map on every rank
 *
 * RCCL (librccl.so) is loaded at run time when arctic_comm_unique_id / arctic_comm_init is first called: libarctic_hip.so has no
 * link-time dependency on it, and single-GPU hosts never touch it.  Errors: the int codes of arctic_hip.h + arctic_last_error.
 */
#ifndef ARCTIC_DIST_H
#define ARCTIC_DIST_H

#include "arctic_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ARCTIC_COMM_ID_BYTES 128 /* NCCL_UNIQUE_ID_BYTES */

/* ncclGetUniqueId: called once (by the rank that will be rank 0, or any one process); the caller distributes the bytes. */
int arctic_comm_unique_id(void *id_out /* ARCTIC_COMM_ID_BYTES */, char *err, uint64_t err_len);

/* ncclCommInitRank(world, id, rank) on the handle's device + an all-gather of every rank's shard layout (so the root can place
 * shards of unequal size).  Collective: every rank of `world` calls it.  world == 1 is valid (a one-rank communicator). */
int arctic_comm_init(ArcticRenderer *r, const void *id /* ARCTIC_COMM_ID_BYTES */, int rank, int world);

/* ncclCommDestroy (also done by arctic_destroy). */
int arctic_comm_destroy(ArcticRenderer *r);

/* Gather the RGBA8 shard `d_shard` (device pointer, rows() * width * 4 bytes; NULL = the handle's own output of the last
 * arctic_pass_shade / arctic_render_frame) on rank `root`, whose `d_frame` (device pointer, height * width * 4 bytes) receives
 * the full row-major frame; d_frame is ignored on the other ranks.  Collective and asynchronous: enqueued on the handle's
 * communication stream after everything already enqueued on its main stream; a later pass that writes `d_shard` again waits for
 * this gather by itself, so alternating two shard buffers overlaps the exchange of frame k with the shading of frame k + 1.
 * arctic_flush waits for it. */
int arctic_gather_frame(ArcticRenderer *r, const void *d_shard, void *d_frame, int root);

/* The placement step alone, on the main stream: `d_staging` holds the shards of all `world` ranks back to back in rank order
 * (rank k's rows_k * width * 4 bytes), laid out like this handle's sharding (same band_rows / shard_count, or -- for row-range
 * shards -- the ranges given in `row_ranges`, 2 * world uint32: begin, end); d_frame receives height * width * 4 bytes.
 * No communicator needed: this is what the root does after the transfers, callable by tests and by hosts with another transport. */
int arctic_assemble_frame(ArcticRenderer *r, const void *d_staging, void *d_frame, uint32_t world, const uint32_t *row_ranges);

/* ---- the plan of the exchange step as plain numbers: pure host functions, no device, no communicator, no handle ------------------
 * arctic_comm_init, arctic_gather_frame, arctic_assemble_frame and the placement kernel are all built on these three, so what they
 * will do on R GPUs can be checked on any machine (tests/test_exchange_plan.py: worlds 2..8, heights that no band count divides).
 *
 * arctic_exchange_plan: a frame of `height` rows sharded over `world` ranks -- interleaved bands of `band_rows` rows dealt
 * round-robin (row_ranges == NULL), or the row ranges row_ranges[2k], row_ranges[2k+1] = rank k's [begin, end) (band_rows == 0).
 * Fills rows[k] = rows of rank k's shard and offset[k] = byte offset of that shard in the root's staging buffer (shards back to
 * back in rank order, rows of width * 4 bytes); *total_bytes = size of the staging buffer. */
int arctic_exchange_plan(uint32_t width, uint32_t height, uint32_t band_rows, uint32_t world, const uint32_t *row_ranges,
                         uint32_t *rows /* world */, uint64_t *offset /* world */, uint64_t *total_bytes);

/* Where frame row y comes from: *owner = the rank whose shard holds it, *local_row = its index inside that shard.  Returns 0, or
 * 1 when no rank owns the row (row ranges with a gap), < 0 for bad arguments.  The placement kernel evaluates the same function. */
int arctic_exchange_row_source(uint32_t y, uint32_t height, uint32_t band_rows, uint32_t world, const uint32_t *row_ranges,
                               uint32_t *owner, uint32_t *local_row);

/* The point-to-point transfers rank `rank` posts inside one ncclGroupStart / ncclGroupEnd of arctic_gather_frame with root `root`:
 * a non-root rank sends its whole shard to the root (one entry, is_send = 1, staging_offset unused = 0); the root receives every
 * other rank's shard at staging_offset (is_send = 0) -- its own shard is a local copy to offset[root].  Empty shards post nothing.
 * Returns the number of entries written (<= world), < 0 for bad arguments. */
typedef struct ArcticTransfer {
    int32_t peer;            /* the other rank */
    int32_t is_send;
    uint64_t staging_offset; /* receives: byte offset in the root's staging buffer */
    uint64_t bytes;
} ArcticTransfer;
int arctic_exchange_transfers(uint32_t width, uint32_t world, int32_t rank, int32_t root, const uint32_t *rows /* world */,
                              const uint64_t *offset /* world */, ArcticTransfer *out, uint32_t cap);

#define ARCTIC_OPT_SHADOW_SHARDED 14 /* 1 = with a communicator of world > 1 attached, arctic_pass_shadow_map / arctic_render_frame rasterise
                                        only this rank's ceil(S / world) light-space rows and complete the map with one in-place ncclAllGather
                                        (same map bit for bit: shadow raster results do not depend on the scissor); 0 (default) = every
                                        rank draws the whole map, no communication */

#ifdef __cplusplus
}
#endif
#endif
