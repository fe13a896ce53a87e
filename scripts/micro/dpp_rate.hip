// Issue cost of DPP on gfx950: v_fmac_f32 vs v_fmac_f32_dpp (row_newbcast / quad_perm) vs v_mov_b32_dpp + v_fmac_f32,
// 4 independent chains per wavefront, 2 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(64) chains(float* out, int iters, float w)
{
    float acc[4], y[4];
    for (int c = 0; c < 4; ++c) { acc[c] = (float)threadIdx.x + c; y[c] = 1.0f + 0.001f * (threadIdx.x + c); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (MODE == 0) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[c]) : "v"(y[c]), "v"(w));
                if constexpr (MODE == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc[c]) : "v"(y[c]), "v"(w));
                if constexpr (MODE == 2) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc[c]) : "v"(y[c]), "v"(w));
                if constexpr (MODE == 3) { float t; asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t) : "v"(y[c]));
                                           asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[c]) : "v"(t), "v"(w)); }
                if constexpr (MODE == 4) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc[c]) : "v"(y[c]), "v"(w));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE> void run(float* out, const char* name)
{
    const int iters = 4096, blocks = 256 * 4 * 2;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((chains<MODE>), dim3(blocks), dim3(64), 0, 0, out, 16, 0.999f);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((chains<MODE>), dim3(blocks), dim3(64), 0, 0, out, iters, 0.999f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-40s %.2f ms, %.2f cycles per FMA per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * 4 * 2));
}

int main()
{
    float* out; (void)hipMalloc(&out, 256 * 4 * 2 * 64 * sizeof(float));
    run<0>(out, "v_fmac_f32");
    run<1>(out, "v_fmac_f32_dpp row_newbcast");
    run<2>(out, "v_fmac_f32_dpp quad_perm");
    run<3>(out, "v_mov_b32_dpp row_newbcast + v_fmac_f32");
    run<4>(out, "v_fmac_f32_dpp row_shr");
    return 0;
}
