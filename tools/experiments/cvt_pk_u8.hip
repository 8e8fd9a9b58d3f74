// What v_cvt_pk_u8_f32 does with fractions, negatives, overflow and NaN on gfx950 (the RGBA8 store wants: truncate, saturate, NaN -> 0),
// and v_exp_f32 with the clamp bit.   hipcc --offload-arch=gfx950 -O2 cvt_pk_u8.hip -o /tmp/cvt_pk_u8 && /tmp/cvt_pk_u8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float *in, unsigned *out, float *outf, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1u, 0xFF0000AAu);
    float r;
    asm volatile("v_exp_f32_e64 %0, %1 clamp\n\ts_nop 0" : "=v"(r) : "v"(in[i]));
    outf[i] = r;
}
int main() {
    const float h[] = {0.0f, 0.49f, 0.5f, 0.51f, 0.99f, 1.0f, 1.5f, 2.5f, 2.51f, 3.5f, 254.5f, 254.99f, 255.0f, 255.5f, 255.99f, 256.0f, 300.0f, 1e9f, -0.5f, -1.0f, -300.0f, NAN, INFINITY, -INFINITY};
    const int n = sizeof(h) / sizeof(h[0]);
    float *d; unsigned *o; float *of;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4); hipMalloc(&of, n * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, of, n);
    unsigned r[n]; float rf[n];
    hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rf, of, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%12g -> cvt_pk_u8 byte1 %3u (word %08x)   exp2 clamp %g\n", h[i], (r[i] >> 8) & 0xFF, r[i], rf[i]);
    return 0;
}
