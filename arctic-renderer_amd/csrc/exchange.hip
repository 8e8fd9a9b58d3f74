// exchange.hip -- the one kernel of the multi-GPU exchange step (include/arctic_dist.h): shards, back to back in rank order in a
// staging buffer, are placed into the rows of the full row-major RGBA8 frame.  Pure copy, 16 bytes per lane.
#include "common.h"

namespace arctic {

namespace {

// one workgroup per frame row; shard_row_source (common.h) names the shard and the row inside it -- the same function the host-side
// plan (arctic_exchange_plan) is made of; rows outside every range are left alone.
__global__ __launch_bounds__(256) void k_place_rows(const uint8_t *__restrict__ staging, uint8_t *__restrict__ frame, uint32_t width, uint32_t height,
                                                    uint32_t band_rows, uint32_t world, const uint32_t *__restrict__ ranges,
                                                    const unsigned long long *__restrict__ shard_offset /* world entries: byte offset of rank k's shard */) {
    const uint32_t y = blockIdx.x;
    uint32_t owner = 0, local = 0;
    if (!shard_row_source(y, band_rows, world, ranges, owner, local)) return;
    const size_t row_bytes = (size_t)width * 4;
    const uint8_t *src = staging + shard_offset[owner] + (size_t)local * row_bytes;
    uint8_t *dst = frame + (size_t)y * row_bytes;
    if ((row_bytes & 15) == 0 && ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0) {
        const uint4 *s = reinterpret_cast<const uint4 *>(src);
        uint4 *d = reinterpret_cast<uint4 *>(dst);
        for (uint32_t i = threadIdx.x; i < row_bytes / 16; i += 256) d[i] = s[i];
    } else {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
        uint32_t *d = reinterpret_cast<uint32_t *>(dst);
        for (uint32_t i = threadIdx.x; i < width; i += 256) d[i] = s[i];
    }
}

}  // namespace

hipError_t launch_place_rows(const uint8_t *staging, uint8_t *frame, uint32_t width, uint32_t height, uint32_t band_rows, uint32_t world,
                             const uint32_t *d_ranges, const unsigned long long *d_shard_offset, hipStream_t s) {
    if (height == 0) return hipSuccess;
    k_place_rows<<<height, 256, 0, s>>>(staging, frame, width, height, band_rows, world, d_ranges, d_shard_offset);
    return hipGetLastError();
}

}  // namespace arctic
