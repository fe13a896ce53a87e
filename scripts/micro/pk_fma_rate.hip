// Issue rate of v_fma_f32 vs v_pk_fma_f32 in dependent chains on gfx950: C independent chains per wavefront,
// W wavefronts per SIMD.  hipcc --offload-arch=gfx950 -O3 -o pk_fma_rate pk_fma_rate.hip && ./pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int C, bool Packed>
__global__ void __launch_bounds__(64) chains(float* out, int iters, float w)
{
    f2 acc[C];
    for (int c = 0; c < C; ++c) acc[c] = f2{(float)threadIdx.x + c, 1.0f + c};
    f2 ww = {w, w * 0.5f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if constexpr (Packed) acc[c] = __builtin_elementwise_fma(acc[c], ww, ww);
                else { acc[c].x = __builtin_fmaf(acc[c].x, ww.x, ww.x); acc[c].y = __builtin_fmaf(acc[c].y, ww.y, ww.y); }
            }
        }
    }
    float s = 0;
    for (int c = 0; c < C; ++c) s += acc[c].x + acc[c].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int C, bool Packed> void run(float* out, int waves_per_simd)
{
    const int iters = 4096, blocks = 256 * 4 * waves_per_simd;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((chains<C, Packed>), dim3(blocks), dim3(64), 0, 0, out, 16, 0.999f);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((chains<C, Packed>), dim3(blocks), dim3(64), 0, 0, out, iters, 0.999f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double fma_pairs = (double)iters * 16 * C;                // per wavefront: pair-FMAs (= 1 pk or 2 scalar instructions)
    const double cycles = ms * 1e-3 * 2.4e9;                        // nominal clock
    printf("%s chains %d waves/SIMD %d: %.2f ms, %.2f cycles per pair-FMA per SIMD (at 2.4 GHz)\n", Packed ? "pk_fma" : "fma   ", C,
           waves_per_simd, ms, cycles / (fma_pairs * waves_per_simd));
}

int main()
{
    float* out; (void)hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run<1, false>(out, w); run<1, true>(out, w);
        run<2, false>(out, w); run<2, true>(out, w);
        run<4, false>(out, w); run<4, true>(out, w);
        run<8, false>(out, w); run<8, true>(out, w);
    }
    return 0;
}
