// What the memory system of the box delivers to the simplest kernels: read N bytes (16 B per lane, coalesced), optionally write a quarter
// as many; one workgroup per 4 KiB / persistent grid.  The practical roof for the shading pass's traffic (490 MB per 4K pass).
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/stream_rates.hip -o /tmp/stream_rates && /tmp/stream_rates
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ in, size_t n, float *out, int per_thread) {
    float acc = 0.f;
    size_t i = (size_t)blockIdx.x * 256 * per_thread + threadIdx.x;
#pragma unroll 4
    for (int k = 0; k < per_thread; ++k, i += 256) if (i < n) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read_write(const float4 *__restrict__ in, size_t n, unsigned *__restrict__ o, int per_thread) {
    size_t i = (size_t)blockIdx.x * 256 * per_thread + threadIdx.x;
#pragma unroll 4
    for (int k = 0; k < per_thread; ++k, i += 256) if (i < n) { float4 v = in[i]; o[i] = __float_as_uint(v.x + v.y + v.z + v.w); }
}
int main() {
    const size_t bytes = 512ull << 20, n = bytes / 16;
    float4 *in; float *out; unsigned *o;
    hipMalloc(&in, bytes); hipMalloc(&out, 4); hipMalloc(&o, n * 4);
    hipMemset(in, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int per_thread : {1, 4, 16}) {
        const int blocks = (int)((n + 256ull * per_thread - 1) / (256ull * per_thread));
        for (int rw = 0; rw < 2; ++rw) {
            for (int w = 0; w < 20; ++w) { if (rw) k_read_write<<<blocks, 256>>>(in, n, o, per_thread); else k_read<<<blocks, 256>>>(in, n, out, per_thread); }
            hipEventRecord(e0);
            const int reps = 20;
            for (int w = 0; w < reps; ++w) { if (rw) k_read_write<<<blocks, 256>>>(in, n, o, per_thread); else k_read<<<blocks, 256>>>(in, n, out, per_thread); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double b = (double)bytes * (rw ? 1.25 : 1.0) * reps;
            printf("%s, %2d x 16 B per thread (%7d workgroups): %.3f ms per pass, %.2f TB/s\n", rw ? "read 512 MiB + write 128 MiB" : "read 512 MiB               ", per_thread, blocks, ms / reps, b / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
