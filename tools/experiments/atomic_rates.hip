// atomic_rates.hip -- what a depth-test atomic costs on the box, for the access pattern of the rasteriser (geometry.hip):
// a wave addresses one 512-byte run (an 8x8 tile of 64-bit visibility keys) of a 66 MB plane, a different pseudo-random
// tile every trip; persistent grid of 256 CUs x 7 waves x 4 SIMDs like k_raster.
// Variants: agent-scope atomicMin (what the rasteriser issues), workgroup-scope (stays in the XCD's L2: correct only if a
// tile is touched from one XCD), plain stores, loads; all 64 lanes or every fourth lane active; 64-bit and 32-bit.
// Output: microseconds for 10 M lane operations, lane operations per ns, requests (wave instructions) per ns.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/atomic_rates.hip -o /tmp/atomic_rates && /tmp/atomic_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr uint32_t TILES = 3840 * 2160 / 64;   // 129600 tiles of 512 B
constexpr int TRIPS = 48;

enum Op { ATOMIC_AGENT, ATOMIC_WG, STORE, LOAD, LOAD_THEN_ATOMIC_AGENT, LOAD_THEN_ATOMIC_WG };

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int OP, int LANE_STEP, bool XCD_LOCAL, class T>
__global__ __launch_bounds__(256) void k_rate(T *plane, uint32_t *sink) {
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool active = (lane % LANE_STEP) == 0;
    T acc = 0;
    for (int t = 0; t < TRIPS; ++t) {
        uint32_t tile = hash(wave * 977u + (uint32_t)t * 7919u) % TILES;
        if (XCD_LOCAL) tile = (tile & ~7u) | (blockIdx.x & 7u);   // blocks go round-robin over the 8 XCDs: a tile is only ever touched from one
        if (tile >= TILES) tile -= 8;
        T *p = plane + (size_t)tile * 64 + lane;
        const T key = (T)hash(tile * 64u + lane + (uint32_t)t) | ((T)1 << (sizeof(T) * 8 - 2));
        if (!active) continue;
        if (OP == LOAD || OP == LOAD_THEN_ATOMIC_AGENT || OP == LOAD_THEN_ATOMIC_WG) {
            const T cur = __builtin_nontemporal_load(p);
            acc += cur;
            if (OP == LOAD_THEN_ATOMIC_AGENT && key < cur) __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (OP == LOAD_THEN_ATOMIC_WG && key < cur) __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (OP == ATOMIC_AGENT) __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (OP == ATOMIC_WG) __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == STORE) *p = key;
    }
    if (acc == 12345) sink[0] = 1;
}

template <int OP, int LANE_STEP, bool XCD_LOCAL, class T>
void run(const char *name, T *plane, uint32_t *sink, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(plane, 0xFF, (size_t)TILES * 64 * sizeof(T));
        hipDeviceSynchronize();
        hipEventRecord(a);
        k_rate<OP, LANE_STEP, XCD_LOCAL, T><<<blocks, 256>>>(plane, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double instr = (double)blocks * 4 * TRIPS, lanes = instr * (64 / LANE_STEP);
    printf("%-58s %8.1f us  %7.2f lane-ops/ns  %6.3f wave-instr/ns   (%.1f M lane-ops)\n", name, best * 1e3, lanes / (best * 1e6), instr / (best * 1e6), lanes / 1e6);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 7;
    unsigned long long *plane; uint32_t *sink;
    hipMalloc(&plane, (size_t)TILES * 64 * 8); hipMalloc(&sink, 4);
    printf("%d CUs, grid %d blocks x 256, %d trips per wave, plane %d tiles\n", prop.multiProcessorCount, blocks, TRIPS, TILES);
#define RUN(OP, STEP, LOCAL, T) run<OP, STEP, LOCAL, T>(#OP " lanes/" #STEP " xcd_local=" #LOCAL " " #T, (T *)plane, sink, blocks)
    RUN(ATOMIC_AGENT, 1, false, unsigned long long);
    RUN(ATOMIC_AGENT, 4, false, unsigned long long);
    RUN(ATOMIC_AGENT, 16, false, unsigned long long);
    RUN(ATOMIC_AGENT, 1, true, unsigned long long);
    RUN(ATOMIC_WG, 1, true, unsigned long long);
    RUN(ATOMIC_WG, 4, true, unsigned long long);
    RUN(ATOMIC_WG, 16, true, unsigned long long);
    RUN(STORE, 1, false, unsigned long long);
    RUN(STORE, 4, false, unsigned long long);
    RUN(LOAD, 1, false, unsigned long long);
    RUN(LOAD, 4, false, unsigned long long);
    RUN(LOAD_THEN_ATOMIC_AGENT, 1, false, unsigned long long);
    RUN(LOAD_THEN_ATOMIC_AGENT, 4, false, unsigned long long);
    RUN(LOAD_THEN_ATOMIC_WG, 1, true, unsigned long long);
    RUN(LOAD_THEN_ATOMIC_WG, 4, true, unsigned long long);
    RUN(ATOMIC_AGENT, 1, false, uint32_t);
    RUN(ATOMIC_AGENT, 4, false, uint32_t);
    RUN(ATOMIC_WG, 1, true, uint32_t);
    RUN(LOAD_THEN_ATOMIC_AGENT, 1, false, uint32_t);
    return 0;
}
