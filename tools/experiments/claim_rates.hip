// How fast can the waves of the chip claim work items from atomic counters?  One lane per wave does N returning atomic adds in a row
// (each waits for the one before: the latency of a claim) or keeps D of them in flight; counters: one for the whole chip, one per XCD
// (block b runs on XCD b % 8: every counter is touched by one XCD only), or one per XCD and shader engine.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/claim_rates.hip -o /tmp/claim_rates && /tmp/claim_rates
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_claim(unsigned *ctr, int n, int mode, unsigned *out) {
    const unsigned lane = threadIdx.x & 63;
    unsigned acc = 0;
    unsigned xcd = blockIdx.x & 7u;
    unsigned idx = mode == 0 ? 0u : mode == 1 ? xcd * 32u : (xcd * 8u + ((blockIdx.x >> 3) & 7u)) * 32u;
    if (lane == 0) {
        for (int i = 0; i < n; ++i) acc += __hip_atomic_fetch_add(ctr + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u;
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
int main() {
    unsigned *ctr, *out;
    hipMalloc(&ctr, 64 * 32 * 4 * 4); hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"one counter", "one per XCD", "8 per XCD"};
    for (int mode = 0; mode < 3; ++mode)
        for (int wg_per_cu : {1, 6}) {
            const int blocks = 256 * wg_per_cu, n = 200;
            hipMemset(ctr, 0, 64 * 32 * 4 * 4);
            k_claim<<<blocks, 256>>>(ctr, 10, mode, out);
            hipEventRecord(e0);
            k_claim<<<blocks, 256>>>(ctr, n, mode, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double claims = (double)blocks * 4 * n;
            printf("%-12s %d workgroups per CU (%5d waves): %7.1f claims per us in all, %6.1f per counter; a wave's claim takes %.2f us\n", names[mode], wg_per_cu, blocks * 4,
                   claims / (ms * 1e3), claims / (ms * 1e3) / (mode == 0 ? 1 : mode == 1 ? 8 : 64), ms * 1e3 / n);
        }
    return 0;
}
