"""Generates tools/experiments/pk_banks.hip: issue cost of v_pk_fma_f32 / v_pk_mul_f32 as a function of WHICH registers its operands sit in
(VGPR bank = register number mod 4?), of an s_nop between instructions, and of dependent chains -- the questions the light loop's 48 packed
instructions raise (round 5).  Every kernel: N_IT trips of 16 instructions with hard-coded registers inside one asm block.
usage: python tools/experiments/gen_pk_banks.py > build_tmp/pk_banks.hip && hipcc -O3 --offload-arch=gfx950 build_tmp/pk_banks.hip -o build_tmp/pk_banks"""
PAT = {
    # name: (template with {d} = destination pair index, sources fixed)
    "fma a01 b45 c89 (all pairs on banks 0,1)":      "v_pk_fma_f32 v[{d}:{d1}], v[0:1], v[4:5], v[8:9]",
    "fma a01 b23 c45 (a,c share banks)":            "v_pk_fma_f32 v[{d}:{d1}], v[0:1], v[2:3], v[4:5]",
    "fma a01 b23 c67 (b,c share banks)":            "v_pk_fma_f32 v[{d}:{d1}], v[0:1], v[2:3], v[6:7]",
    "fma a01 b01 c23 (square + c)":                 "v_pk_fma_f32 v[{d}:{d1}], v[0:1], v[0:1], v[2:3]",
    "fma a01 b23 c23":                              "v_pk_fma_f32 v[{d}:{d1}], v[0:1], v[2:3], v[2:3]",
    "fma a01 b23 c=d (accumulate)":                 "v_pk_fma_f32 v[{d}:{d1}], v[0:1], v[2:3], v[{d}:{d1}]",
    "fma a01 b45 c=d (accumulate, a,b share banks)": "v_pk_fma_f32 v[{d}:{d1}], v[0:1], v[4:5], v[{d}:{d1}]",
    "fma s[0:1] b23 c=d (SGPR colour, accumulate)": "v_pk_fma_f32 v[{d}:{d1}], s[0:1], v[2:3], v[{d}:{d1}]",
    "fma a01 b23 c45 op_sel broadcast":             "v_pk_fma_f32 v[{d}:{d1}], v[0:1], v[2:3], v[4:5] op_sel:[0,1,0] op_sel_hi:[1,1,1]",
    "fma a01 b23 const":                            "v_pk_fma_f32 v[{d}:{d1}], v[0:1], v[2:3], 1.0 op_sel_hi:[1,1,0]",
    "mul a01 b23":                                  "v_pk_mul_f32 v[{d}:{d1}], v[0:1], v[2:3]",
    "mul a01 b45 (same banks)":                     "v_pk_mul_f32 v[{d}:{d1}], v[0:1], v[4:5]",
    "mul a01 a01 (square)":                         "v_pk_mul_f32 v[{d}:{d1}], v[0:1], v[0:1]",
    "add s[0:1] b23 (light - world)":               "v_pk_add_f32 v[{d}:{d1}], s[0:1], v[2:3] neg_lo:[0,1] neg_hi:[0,1]",
    "plain v_fma_f32 a0 b1 c2":                     "v_fma_f32 v{d}, v0, v1, v2",
    "plain v_fma_f32 a0 b4 c8 (same bank)":         "v_fma_f32 v{d}, v0, v4, v8",
    "plain v_mul_f32 a0 b1":                        "v_mul_f32 v{d}, v0, v1",
    "plain v_mul_f32 a0 b4 (same bank)":            "v_mul_f32 v{d}, v0, v4",
    "fma a01 b23 c45 + s_nop 0 (per pair)":         "v_pk_fma_f32 v[{d}:{d1}], v[0:1], v[2:3], v[4:5]\\n\\ts_nop 0",
    "fma dependent chain d=d*b+c":                  "v_pk_fma_f32 v[16:17], v[16:17], v[2:3], v[4:5]",
    "fma dependent chain + s_nop 0":                "v_pk_fma_f32 v[16:17], v[16:17], v[2:3], v[4:5]\\n\\ts_nop 0",
    "rsq + pk_mul (per pair)":                      "v_rsq_f32 v{d}, v0\\n\\tv_pk_mul_f32 v[{e}:{e1}], v[2:3], v[4:5]",
    "rsq rsq pk_mul (per triple, as the loop)":     "v_rsq_f32 v{d}, v0\\n\\tv_rsq_f32 v{d1}, v1\\n\\tv_pk_mul_f32 v[{e}:{e1}], v[2:3], v[4:5]",
}
print('#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <vector>\nconstexpr int N_IT = 4096;')
names = []
for k, (name, tpl) in enumerate(PAT.items()):
    body = []
    for i in range(16):
        d = 16 + 2 * i
        e = 48 + 2 * (i % 8)
        body.append(tpl.format(d=d, d1=d + 1, e=e, e1=e + 1))
    text = "\\n\\t".join(body)
    clob = ", ".join(f'"v{r}"' for r in range(0, 64))
    print(f'''__global__ __launch_bounds__(256) void k{k}(float *out, float seed) {{
    asm volatile("v_mov_b32 v0, %0\\n\\tv_mov_b32 v1, %0\\n\\tv_mov_b32 v2, %0\\n\\tv_mov_b32 v3, %0\\n\\tv_mov_b32 v4, %0\\n\\tv_mov_b32 v5, %0\\n\\tv_mov_b32 v6, %0\\n\\tv_mov_b32 v7, %0\\n\\tv_mov_b32 v8, %0\\n\\tv_mov_b32 v9, %0\\n\\ts_mov_b32 s0, 1.0\\n\\ts_mov_b32 s1, 1.0" :: "v"(seed) : {clob}, "s0", "s1");
    for (int it = 0; it < N_IT; ++it) asm volatile("{text}" ::: {clob});
    float r; asm volatile("v_mov_b32 %0, v16" : "=v"(r));
    if (r == 12345.f) out[0] = r;
}}''')
    names.append(name)
print('struct E { const char *name; void (*k)(float *, float); int per; };\nint main() {\n    float *out; hipMalloc(&out, 4); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);\n    std::vector<E> ks = {')
for k, name in enumerate(names):
    print(f'        {{"{name}", k{k}, 16}},')
print('''    };
    printf("%-52s %8s %8s %8s %8s   (cycles per instruction (or per group) per SIMD at 2.4 GHz nominal; W = waves per SIMD)\\n", "instruction", "W=1", "W=2", "W=4", "W=8");
    for (auto &en : ks) {
        printf("%-52s", en.name);
        for (int W : {1, 2, 4, 8}) {
            const int blocks = 256 * W;
            en.k<<<blocks, 256>>>(out, 1.0f); hipDeviceSynchronize();
            hipEventRecord(e0); en.k<<<blocks, 256>>>(out, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf(" %8.2f", ms * 1e-3 * 2.4e9 / ((double)N_IT * en.per * W));
        }
        printf("\\n"); fflush(stdout);
    }
    return 0;
}''')
