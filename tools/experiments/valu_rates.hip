// valu_rates.hip -- issue cost of the vector instructions the shading kernel is made of, measured on the box.
// Every kernel runs N_IT trips of 16 independent instructions of one kind per wave; grid = 256 CUs x 4 SIMDs x W waves.
// Output: cycles per instruction per SIMD (nominal 2.4 GHz) for W = 1, 2, 4, 8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N_IT = 4096;
typedef float v2 __attribute__((ext_vector_type(2)));
__device__ inline v2 splat2(float x) { v2 r; r.x = x; r.y = x + 1.0f; return r; }

#define KERNEL(name, decl, body, fin)                                                            \
    __global__ __launch_bounds__(256) void name(float *out, float seed) {                       \
        decl;                                                                                    \
        for (int it = 0; it < N_IT; ++it) { body; }                                              \
        fin;                                                                                     \
    }

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// scalar fma
#define D_F(i) float a##i = seed + i;
#define B_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(seed));
#define F_F(i) s += a##i;
KERNEL(k_fma, R16(D_F), R16(B_FMA), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(seed));
KERNEL(k_mul, R16(D_F), R16(B_MUL), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
KERNEL(k_rcp, R16(D_F), R16(B_RCP), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_RSQ(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a##i));
KERNEL(k_rsq, R16(D_F), R16(B_RSQ), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a##i));
KERNEL(k_exp, R16(D_F), R16(B_EXP), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a##i));
KERNEL(k_floor, R16(D_F), R16(B_FLOOR), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_CVTI(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a##i));
KERNEL(k_cvt_i32, R16(D_F), R16(B_CVTI), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_CVTUB(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a##i));
KERNEL(k_cvt_ubyte, R16(D_F), R16(B_CVTUB), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(seed));
KERNEL(k_cndmask, R16(D_F), R16(B_CND), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_CMP(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a##i), "v"(seed) : "vcc");
KERNEL(k_cmp, R16(D_F), R16(B_CMP), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(seed));
KERNEL(k_add_u32, R16(D_F), R16(B_ADDU), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a##i) : "v"(seed));
KERNEL(k_lshl_add, R16(D_F), R16(B_LSHLADD), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_BFE(i) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a##i));
KERNEL(k_bfe, R16(D_F), R16(B_BFE), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(seed));
KERNEL(k_mul_lo_u32, R16(D_F), R16(B_MULLO), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_MADU24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a##i) : "v"(seed));
KERNEL(k_mad_u32_u24, R16(D_F), R16(B_MADU24), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(seed));
KERNEL(k_med3, R16(D_F), R16(B_MED3), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(seed));
KERNEL(k_min3, R16(D_F), R16(B_MIN3), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_FMAK(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a##i) : "v"(seed));
KERNEL(k_fmac, R16(D_F), R16(B_FMAK), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
// DPP move (row_shr) and readlane-style cross lane
#define B_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i));
KERNEL(k_mov_dpp, R16(D_F), R16(B_DPP), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
// fp64 fma
#define D_D(i) double d##i = seed + i;
#define B_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d##i) : "v"((double)seed));
#define F_D(i) sd += d##i;
KERNEL(k_fma_f64, R16(D_D), R16(B_FMA64), double sd = 0; R16(F_D) if (sd == 12345.) out[0] = (float)sd)
// packed
#define D_P(i) v2 p##i = splat2(seed + i);
#define B_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##i) : "v"(ps));
#define F_P(i) s += p##i.x + p##i.y;
KERNEL(k_pk_fma, v2 ps = splat2(seed); R16(D_P), R16(B_PKFMA), float s = 0; R16(F_P) if (s == 12345.f) out[0] = s)
#define B_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##i) : "v"(ps));
KERNEL(k_pk_mul, v2 ps = splat2(seed); R16(D_P), R16(B_PKMUL), float s = 0; R16(F_P) if (s == 12345.f) out[0] = s)
#define B_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##i) : "v"(ps));
KERNEL(k_pk_add, v2 ps = splat2(seed); R16(D_P), R16(B_PKADD), float s = 0; R16(F_P) if (s == 12345.f) out[0] = s)
// packed with a scalar (SGPR pair) operand and op_sel broadcast, as the light loop uses them
#define B_PKFMA_OPSEL(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel_hi:[1,0,0]" : "+v"(p##i) : "v"(ps));
KERNEL(k_pk_fma_opsel, v2 ps = splat2(seed); R16(D_P), R16(B_PKFMA_OPSEL), float s = 0; R16(F_P) if (s == 12345.f) out[0] = s)
// dependent chains: one accumulator, 16 deep -- latency per instruction with W waves to hide it
#define B_FMA_DEP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(seed));
KERNEL(k_fma_dependent, R16(D_F), R16(B_FMA_DEP), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_PKFMA_DEP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(ps));
KERNEL(k_pk_fma_dependent, v2 ps = splat2(seed); R16(D_P), R16(B_PKFMA_DEP), float s = 0; R16(F_P) if (s == 12345.f) out[0] = s)
#define B_RCP_DEP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a0));
KERNEL(k_rcp_dependent, R16(D_F), R16(B_RCP_DEP), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
// mix: 3 fma : 1 rcp (does the transcendental overlap with plain VALU of the same / other waves?)
#define B_MIX(i) asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(a##i), "+v"(b##i) : "v"(seed));
#define D_F2(i) float a##i = seed + i, b##i = seed - i;
#define F_F2(i) s += a##i + b##i;
#define B_MIX_T(i) asm volatile("v_fma_f32 %0, %0, %2, %2\n v_rcp_f32 %1, %1" : "+v"(a##i), "+v"(b##i) : "v"(seed));
KERNEL(k_mix_fma_fma, R16(D_F2), R16(B_MIX), float s = 0; R16(F_F2) if (s == 12345.f) out[0] = s)
KERNEL(k_mix_fma_rcp, R16(D_F2), R16(B_MIX_T), float s = 0; R16(F_F2) if (s == 12345.f) out[0] = s)


// ---- second set: operand-source variants (register-file read ports) and the compare / select idioms of the PCF code ----
#define D_F3(i) float a##i = seed + i, b##i = seed - i, c##i = seed * i;
#define F_F3(i) s += a##i + b##i + c##i;
#define B_FMA3(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##i) : "v"(b##i), "v"(c##i));
KERNEL(k_fma_3vgpr, R16(D_F3), R16(B_FMA3), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_FMA_D3(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a##i) : "v"(b##i), "v"(c##i), "v"(seed));
KERNEL(k_fma_3vgpr_nodep, R16(D_F3), R16(B_FMA_D3), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_FMA_S(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "s"(seed));
KERNEL(k_fma_sgpr, R16(D_F), R16(B_FMA_S), float s = 0; R16(F_F) if (s == 12345.f) out[0] = s)
#define B_FMA_S2(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b##i), "s"(seed));
KERNEL(k_fma_2vgpr_sgpr, R16(D_F3), R16(B_FMA_S2), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_FMA_K(i) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a##i) : "v"(b##i));
KERNEL(k_fma_2vgpr_const, R16(D_F3), R16(B_FMA_K), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_MUL2(i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a##i) : "v"(b##i), "v"(c##i));
KERNEL(k_mul_2vgpr_nodep, R16(D_F3), R16(B_MUL2), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_add_f32, R16(D_F3), R16(B_ADDF), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_SUBF(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_sub_f32, R16(D_F3), R16(B_SUBF), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_max_f32, R16(D_F3), R16(B_MAXF), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_MINF(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_min_f32, R16(D_F3), R16(B_MINF), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_FMAC2(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##i) : "v"(b##i), "v"(c##i));
KERNEL(k_fmac_3vgpr, R16(D_F3), R16(B_FMAC2), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a##i) : "v"(b##i));
KERNEL(k_mov, R16(D_F3), R16(B_MOV), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_and, R16(D_F3), R16(B_AND), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_LSHR(i) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(a##i));
KERNEL(k_lshrrev, R16(D_F3), R16(B_LSHR), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_PKFMA3(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p##i) : "v"(q##i), "v"(ps));
#define D_P2(i) v2 p##i = splat2(seed + i), q##i = splat2(seed - i);
#define F_P2(i) s += p##i.x + p##i.y + q##i.x;
KERNEL(k_pk_fma_3vgpr, v2 ps = splat2(seed); R16(D_P2), R16(B_PKFMA3), float s = 0; R16(F_P2) if (s == 12345.f) out[0] = s)
#define B_PKFMA_S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##i) : "s"(ps));
KERNEL(k_pk_fma_sgpr, v2 ps = splat2(seed); R16(D_P2), R16(B_PKFMA_S), float s = 0; R16(F_P2) if (s == 12345.f) out[0] = s)
// selects
#define B_CND_VCCSET(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b##i) : "vcc");
KERNEL(k_cndmask_vcc_set, R16(D_F3) asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"), R16(B_CND_VCCSET), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_CND_SG(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b##i), "s"(mask));
KERNEL(k_cndmask_sgpr, unsigned long long mask = __ballot(seed > threadIdx.x); R16(D_F3), R16(B_CND_SG), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_CMPCND(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a##i) : "v"(b##i), "v"(c##i) : "vcc");
KERNEL(k_cmp_cndmask_pair, R16(D_F3), R16(B_CMPCND), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_CMPADDC(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_addc_co_u32 %2, vcc, 0, %2, vcc" : : "v"(a##i), "v"(b##i), "v"(c##i) : "vcc");
KERNEL(k_cmp_addc_pair, R16(D_F3), R16(B_CMPADDC), float s = 0; R16(F_F3) if (s == 12345.f) out[0] = s)
#define B_CMPS(i) asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m##i) : "v"(a##i), "v"(b##i));
#define D_M(i) unsigned long long m##i;
#define F_M(i) mm += __popcll(m##i);
KERNEL(k_cmp_to_sgpr, R16(D_F3) R16(D_M), R16(B_CMPS), float s = 0; int mm = 0; R16(F_F3) R16(F_M) if (s == 12345.f + mm) out[0] = s)
// popcount of a compare mask on the scalar unit (s_bcnt1) next to vector work: the wave-uniform tap counter
#define B_CMPBCNT(i) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n s_bcnt1_i32_b64 %0, vcc" : "=s"(k##i) : "v"(a##i), "v"(b##i) : "vcc", "scc");
#define D_K(i) int k##i;
#define F_K(i) mm += k##i;
KERNEL(k_cmp_bcnt, R16(D_F3) R16(D_K), R16(B_CMPBCNT), float s = 0; int mm = 0; R16(F_F3) R16(F_K) if (s == 12345.f + mm) out[0] = s)
// LDS: broadcast b128 read (light pairs), b32 gather (sRGB LUT)
__global__ __launch_bounds__(256) void k_ds_read_b128_bcast(float *out, float seed) {
    __shared__ float4 sm[256];
    sm[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { float4 v = sm[(it + j) & 255]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_ds_read_b32_gather(float *out, float seed) {
    __shared__ float sm[256];
    sm[threadIdx.x] = seed * threadIdx.x;
    __syncthreads();
    float acc = 0; unsigned idx = threadIdx.x * 7u;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { float v = sm[(idx + j * 37u) & 255u]; acc += v; idx += __float_as_uint(v) & 3u; }
    }
    if (acc == 12345.f) out[0] = acc;
}

struct Entry { const char *name; void (*k)(float *, float); int per_trip; };

int main() {
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<Entry> ks = {
        {"v_fma_f32", k_fma, 16}, {"v_fmac_f32", k_fmac, 16}, {"v_mul_f32", k_mul, 16}, {"v_pk_fma_f32", k_pk_fma, 16}, {"v_pk_fma_f32 op_sel", k_pk_fma_opsel, 16},
        {"v_pk_mul_f32", k_pk_mul, 16}, {"v_pk_add_f32", k_pk_add, 16}, {"v_rcp_f32", k_rcp, 16}, {"v_rsq_f32", k_rsq, 16}, {"v_exp_f32", k_exp, 16},
        {"v_floor_f32", k_floor, 16}, {"v_cvt_i32_f32", k_cvt_i32, 16}, {"v_cvt_f32_ubyte1", k_cvt_ubyte, 16}, {"v_cndmask_b32", k_cndmask, 16},
        {"v_cmp_gt_f32", k_cmp, 16}, {"v_add_u32", k_add_u32, 16}, {"v_lshl_add_u32", k_lshl_add, 16}, {"v_bfe_u32", k_bfe, 16},
        {"v_mul_lo_u32", k_mul_lo_u32, 16}, {"v_mad_u32_u24", k_mad_u32_u24, 16}, {"v_med3_f32", k_med3, 16}, {"v_min3_f32", k_min3, 16},
        {"v_mov_b32_dpp", k_mov_dpp, 16}, {"v_fma_f64", k_fma_f64, 16},
        {"v_fma_f32 dependent chain", k_fma_dependent, 16}, {"v_pk_fma_f32 dependent chain", k_pk_fma_dependent, 16}, {"v_rcp_f32 dependent chain", k_rcp_dependent, 16},
        {"2 x v_fma_f32 (per pair)", k_mix_fma_fma, 16}, {"v_fma_f32 + v_rcp_f32 (per pair)", k_mix_fma_rcp, 16},
        {"v_fma_f32 d+=b*c (3 VGPR)", k_fma_3vgpr, 16}, {"v_fma_f32 d=b*c+e (3 VGPR, no dep)", k_fma_3vgpr_nodep, 16}, {"v_fma_f32 a*s+s (SGPR)", k_fma_sgpr, 16},
        {"v_fma_f32 a*b+s (2 VGPR + SGPR)", k_fma_2vgpr_sgpr, 16}, {"v_fma_f32 a*b+1.0 (2 VGPR + const)", k_fma_2vgpr_const, 16}, {"v_mul_f32 d=b*c (no dep)", k_mul_2vgpr_nodep, 16},
        {"v_add_f32", k_add_f32, 16}, {"v_sub_f32", k_sub_f32, 16}, {"v_max_f32", k_max_f32, 16}, {"v_min_f32", k_min_f32, 16}, {"v_fmac_f32 a+=b*c", k_fmac_3vgpr, 16},
        {"v_mov_b32", k_mov, 16}, {"v_and_b32", k_and, 16}, {"v_lshrrev_b32", k_lshrrev, 16}, {"v_pk_fma_f32 p+=q*r (3 VGPR)", k_pk_fma_3vgpr, 16}, {"v_pk_fma_f32 p*s+s (SGPR)", k_pk_fma_sgpr, 16},
        {"v_cndmask_b32 vcc (vcc set once)", k_cndmask_vcc_set, 16}, {"v_cndmask_b32 sgpr mask", k_cndmask_sgpr, 16}, {"v_cmp + v_cndmask (per pair)", k_cmp_cndmask_pair, 16},
        {"v_cmp + v_addc_co (per pair)", k_cmp_addc_pair, 16}, {"v_cmp -> sgpr pair", k_cmp_to_sgpr, 16}, {"v_cmp + s_bcnt1 (per pair)", k_cmp_bcnt, 16},
        {"ds_read_b128 broadcast + 4 add", k_ds_read_b128_bcast, 16}, {"ds_read_b32 gather + add + 2 int", k_ds_read_b32_gather, 16},
    };
    printf("%-36s %8s %8s %8s %8s   (cycles per instruction per SIMD at 2.4 GHz nominal; W = waves per SIMD)\n", "instruction", "W=1", "W=2", "W=4", "W=8");
    for (auto &en : ks) {
        printf("%-36s", en.name);
        for (int W : {1, 2, 4, 8}) {
            const int blocks = 256 * W;   // 256 CUs x W workgroups of 4 waves = W waves per SIMD
            en.k<<<blocks, 256>>>(out, 1.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            en.k<<<blocks, 256>>>(out, 1.0f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)N_IT * en.per_trip * W;
            printf(" %8.2f", ms * 1e-3 * 2.4e9 / instr_per_simd);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
